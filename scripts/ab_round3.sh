# Stride table of round 3 on the final build (GPU box): round-2 kernels (variant 12) and the default plan (variant 0)
mkdir -p gpurun_out/r03
O=gpurun_out/r03/stride_final.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 12,0 --reps 3
for S in 2 3 4 6 8 12 13 16 24 32 48 64 96 128 200; do run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 12,0 --reps 2; done
cat $O
