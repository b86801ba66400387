mkdir -p gpurun_out
O=gpurun_out/exp6.txt
: > $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $O
python scripts/kexp.py --tag base >> $O 2>&1
python scripts/kexp.py --windows 1000000 --tag base1M >> $O 2>&1
BMX_LIB_NAME=libbmx_prof.so python scripts/kexp.py --reps 1 --tag prof >> $O 2>&1
python scripts/e2e_timing.py >> $O 2>&1
cat $O
