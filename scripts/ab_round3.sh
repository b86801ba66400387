# A/B measurements of round 3 (GPU box)
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab10.txt
: > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
for S in 16 64 200; do
run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 0 --reps 3
BMX_ALLOW_STALE=1 BMX_LIB_NAME=libbmx_soloxcd0.so run timeout 300 python scripts/prep_check.py --step $S --windows 65536 --variants 0 --reps 3
done
cat $O
