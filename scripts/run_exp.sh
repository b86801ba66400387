mkdir -p gpurun_out
O=gpurun_out/exp13.txt
: > $O
python -m pytest tests/test_gpu_round2.py -x -q -k "config5" 2>&1 | tail -3 >> $O
for s in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 160 200; do
  python scripts/kexp.py --step $s --windows 65536 --reps 2 --tag default >> $O 2>&1
  for j in 16 8 4; do BMX_LIB_NAME=libbmx_diag.so BMX_FORCE_J=$j python scripts/kexp.py --step $s --windows 65536 --reps 2 --tag J$j >> $O 2>&1; done
  python scripts/kexp.py --step $s --windows 65536 --reps 2 --variant 2 --tag persite >> $O 2>&1
done
grep -v "^\[" $O | awk '{print $1, $4, $5, $7, $8, $9, $10}'
