# Round-4 section profile of the prepared scan kernel (diagnostic builds): bash scripts/round4_profile.sh
set -u
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
for c in 3 5; do
  BMX_LIB_NAME=libbmx_prof.so BMX_PROF_PREPARED=1 python scripts/prep_check.py --config $c --windows $((c==3?262144:65536)) --variants 0 --reps 2 > $O/prof_c$c.txt 2>&1
  BMX_LIB_NAME=libbmx_count.so python scripts/prep_check.py --config $c --windows 65536 --variants 0 --reps 1 > $O/count_c$c.txt 2>&1
done
python scripts/time_by_A.py > $O/time_by_A.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/icache -- python3 $R/scripts/prep_check.py --config 3 --windows 262144 --variants 0 --reps 1 > $O/icache.txt 2>&1
cd $R
python scripts/pmc_summary.py $(find $O/icache -name "*counter_collection.csv") > $O/icache_summary.txt 2>&1
tail -30 $O/prof_c3.txt; tail -30 $O/prof_c5.txt; cat $O/time_by_A.txt; cat $O/icache_summary.txt | head -40
