mkdir -p gpurun_out/r03
O=gpurun_out/r03/rag2.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 10,0 --reps 3
run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 12,0 --reps 3
run timeout 300 python scripts/prep_check.py --windows 262144 --n-spread 40 --variants 12,0 --reps 3
run timeout 300 python scripts/prep_check.py --windows 262144 --n-spread 10 --variants 12,0 --reps 3
BMX_ALLOW_STALE=1 BMX_PROF_PREPARED=1 BMX_LIB_NAME=libbmx_prof.so run timeout 300 python scripts/prep_check.py --windows 262144 --n-spread 40 --variants 0 --reps 1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_rag.txt 2>&1; tail -5 gpurun_out/r03/pytest_rag.txt >> $O
cat $O
