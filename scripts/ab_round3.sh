# A/B measurements of round 3 (GPU box)
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab11.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
for S in 12 16 64 200; do
run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 2,0,14 --reps 3
done
cat $O
