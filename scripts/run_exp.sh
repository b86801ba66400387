O=gpurun_out/r02i; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; python -c "
import json; d=json.load(open('$O/bench_1gpu.json')); print('config4', d['value'], d['ms_per_step'], d['roofline']['frac'])"
{ for s in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 160 200; do python scripts/kexp.py --step $s --windows 65536 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default/"; done
  python scripts/kexp.py --snps 4000000 --step 200 --windows 18500 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default4M/"
  python scripts/kexp.py --snps 4000000 --step 64 --windows 57812 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default4M/"
  for s in 32 48 64 96; do BMX_DIAG=1 BMX_DENSE_GAP=0 BMX_LIB_NAME=libbmx_diag.so python scripts/kexp.py --step $s --windows 65536 2>&1 | grep -v "^#" | sed "s/^ lib=libbmx_diag.so variant=0/persite/"; done
  for s in 48 64 96; do BMX_DIAG=1 BMX_DENSE_GAP=100000 BMX_FORCE_J=4 BMX_LIB_NAME=libbmx_diag.so python scripts/kexp.py --step $s --windows 65536 2>&1 | grep -v "^#" | sed "s/^ lib=libbmx_diag.so variant=0/J4/"; done
} > $O/stride.txt 2>&1
grep -c "" $O/stride.txt
