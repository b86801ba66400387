# A/B: 12th-order (shipped) against a 16th-order build of the prepared kernels' far field (GPU box)
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab_p16.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 10,0 --reps 3
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_p16.so run timeout 300 python scripts/prep_check.py --windows 262144 --variants 10,0 --reps 3
run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 0 --reps 3
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_p16.so run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 0 --reps 3
cat $O
