mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab5.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 12,0 --reps 3
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_prev.so run timeout 300 python scripts/prep_check.py --windows 262144 --variants 0 --reps 3
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_xcd.so run timeout 300 python scripts/prep_check.py --windows 262144 --variants 12,0 --reps 3
run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 12,0 --reps 2
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_prev.so run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 0 --reps 2
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_xcd.so run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 0 --reps 2
for S in 4 8 16 32 64; do run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 12,0,2 --reps 2; done
for SP in 10 40; do
  echo "### n-spread $SP" >> $O
  timeout 300 python bench.py --config 3 --n-spread $SP --no-cpu-baseline --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant 0 ', d['roofline']['kernel'], d['value'])" >> $O 2>&1
  timeout 300 python bench.py --config 3 --n-spread $SP --no-cpu-baseline --steps 2 --variant 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant 12', d['roofline']['kernel'], d['value'])" >> $O 2>&1
done
cat $O
