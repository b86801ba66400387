# A/B measurements of round 3 (GPU box): strides through the default plan, the solo pipeline and the round-2 kernels
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab6.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 12,0 --reps 3
run timeout 300 python scripts/prep_check.py --config 5 --windows 65536 --variants 12,0 --reps 2
for S in 2 4 6 8 12 16 24 32 48 64 128 200; do run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 12,0,14,15,16 --reps 2; done
cat $O
