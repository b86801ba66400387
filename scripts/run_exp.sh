bash scripts/pmc_collect.sh c3 --config 3 > /dev/null 2>&1
bash scripts/pmc_collect.sh c4 --config 4 > /dev/null 2>&1
bash scripts/pmc_collect.sh c5 --config 5 > /dev/null 2>&1
for t in c3 c4 c5; do echo "=== $t"; cat gpurun_out/r02/$t/summary.txt; done
