# Stride table (GPU box, from the repo root): the round-2 kernels (variant 12) against the default plan at test-site strides 1 .. 200,
# every run with the comparison of the two (argmax / nSites identical, max |dCLR|).  bash scripts/ab_round3.sh [out name]
mkdir -p gpurun_out/r03
O=gpurun_out/r03/${1:-stride_table_final}.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run timeout 300 python scripts/prep_check.py --windows 262144 --variants 12,0 --reps 3
for s in 2 3 4 6 8 12 13 16 24 32 48 64 96 128 200; do
  run timeout 300 python scripts/prep_check.py --step $s --windows 65536 --variants 12,0 --reps 2
done
grep -E "^variant +(0|12) clr" $O | awk '{print $2, $3, $(NF-7), $(NF-6), $(NF-5)}'
