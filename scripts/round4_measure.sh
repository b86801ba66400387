# Round-4 measurement campaign on the GPU box (from the repo root): bash scripts/round4_measure.sh A|B|C
set -u
R=$(pwd)
O=$R/gpurun_out/r04
mkdir -p $O
if [ "$1" = "A" ]; then
  timeout 900 python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
  timeout 600 python bench.py --config 3 --no-cpu-baseline > $O/bench_1gpu_config3.json 2>/dev/null
  timeout 900 python bench.py --config 5 --no-cpu-baseline > $O/bench_1gpu_config5.json 2>/dev/null
  timeout 600 python bench.py --config 3 --n-spread 10 --no-cpu-baseline > $O/bench_1gpu_config3_11_sample_sizes.json 2>/dev/null
  timeout 600 python bench.py --config 3 --n-spread 40 --no-cpu-baseline > $O/bench_1gpu_config3_41_sample_sizes.json 2>/dev/null
  timeout 600 python bench.py --config 3 --snps 4000000 --step 64 --no-cpu-baseline > $O/bench_1gpu_config3_step64.json 2>/dev/null
  for f in $O/bench_1gpu*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],1), d['roofline']['kernel'], d['roofline'].get('frac'), d.get('parity_sample_max_rel'), d['config'].get('end_to_end_windows_per_s'))"; done
elif [ "$1" = "B" ]; then
  bash scripts/pmc_collect.sh c4 --config 4 > $O/pmc_c4.log 2>&1
  bash scripts/pmc_collect.sh c3 --config 3 > $O/pmc_c3.log 2>&1
  bash scripts/pmc_collect.sh c5 --config 5 > $O/pmc_c5.log 2>&1
  PMC_KEY=config3_step64 bash scripts/pmc_collect.sh c3s64 --config 3 --snps 4000000 --step 64 > $O/pmc_c3s64.log 2>&1
  PMC_KEY=config3_nspread40 bash scripts/pmc_collect.sh c3ns40 --config 3 --n-spread 40 > $O/pmc_c3ns40.log 2>&1
  PMC_KEY=config3_nspread10 bash scripts/pmc_collect.sh c3ns10 --config 3 --n-spread 10 > $O/pmc_c3ns10.log 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats_bench.err
  cd $R
  find $O/stats -name "*kernel_stats.csv" | xargs cat | cut -c1-220 | head -12
  for t in c3 c4 c5 c3s64 c3ns40 c3ns10; do head -1 $O/$t/summary.txt | cut -c1-300; grep -E "valu_per_64|cycles_per_valu|hbm_read|hbm_write|l2_hit|fp64_flops_per_eval" $O/$t/summary.txt; done
else
  bash scripts/stride_ab.sh "1 2 3 4 6 8 12 13 16 24 32 64 128 200" libbmxscan.so 2>&1 | grep "clr_scan" | tee $O/stride_table.txt
  timeout 900 python scripts/e2e_timing.py > $O/cli_end_to_end_1M.txt 2>&1
  timeout 1500 python scripts/config4_cli_pipeline.py > $O/config4_file_pipeline_1gpu.txt 2>&1
  tail -3 $O/cli_end_to_end_1M.txt; tail -3 $O/config4_file_pipeline_1gpu.txt
fi
