O=gpurun_out/r02h; mkdir -p $O
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
python bench.py --config 5 > $O/bench_c5.json 2>/dev/null
python bench.py --config 3 --n-spread 10 --no-cpu-baseline > $O/bench_c3_11.json 2>/dev/null
python bench.py --config 3 --n-spread 40 --no-cpu-baseline > $O/bench_c3_41.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats_bench.json 2> $GRAFT_REPO_ROOT/$O/stats.err )
BMX_DIAG=1 BMX_LIB_NAME=libbmx_prof.so python scripts/kexp.py --reps 1 --windows 200000 2>&1 | grep -v "^#" > $O/sections.txt
BMX_DIAG=1 BMX_LIB_NAME=libbmx_count.so python scripts/kexp.py --reps 1 --windows 200000 2>&1 | grep -v "^#" > $O/counts.txt
BMX_DIAG=1 BMX_LIB_NAME=libbmx_diag.so python scripts/far_accuracy.py 65536 > $O/far_accuracy.txt 2>&1
python scripts/parity_summary.py > $O/parity.txt 2>&1
{ for s in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 160 200; do python scripts/kexp.py --step $s --windows 65536 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default/"; done
  python scripts/kexp.py --snps 4000000 --step 200 --windows 18500 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default4M/"
  python scripts/kexp.py --snps 4000000 --step 64 --windows 57812 2>&1 | grep -v "^#" | sed "s/^ lib=libbmxscan.so variant=0/default4M/"
  for s in 1 2 3 4 6 8 12 16 24 32 48; do for J in 16 8 4; do BMX_DIAG=1 BMX_DENSE_GAP=100000 BMX_FORCE_J=$J BMX_LIB_NAME=libbmx_diag.so python scripts/kexp.py --step $s --windows 65536 2>&1 | grep -v "^#" | sed "s/^ lib=libbmx_diag.so variant=0/J$J/"; done; done
} > $O/stride.txt 2>&1
python scripts/e2e_timing.py > $O/e2e.txt 2>&1
python scripts/config4_cli_pipeline.py > $O/cfg4_pipeline.txt 2>&1
ls $O; tail -c 400 $O/bench_1gpu.json; cat $O/e2e.txt | tail -6; cat $O/cfg4_pipeline.txt | tail -3
