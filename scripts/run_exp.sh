O=gpurun_out/r02k; mkdir -p $O
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
python bench.py --config 5 > $O/bench_c5.json 2>/dev/null
python bench.py --config 3 --n-spread 10 --no-cpu-baseline > $O/bench_c3_11.json 2>/dev/null
python bench.py --config 3 --n-spread 40 --no-cpu-baseline > $O/bench_c3_41.json 2>/dev/null
python scripts/e2e_timing.py > $O/e2e.txt 2>&1
python scripts/config4_cli_pipeline.py > $O/cfg4_pipeline.txt 2>&1
python - <<'PY'
import json
for f in ('bench_1gpu','bench_c3','bench_c5','bench_c3_11','bench_c3_41'):
    d=json.load(open('gpurun_out/r02k/'+f+'.json')); r=d['roofline']
    print(f, round(d['value']), round(d['ms_per_step'],1), round(r['kernel_ms'],1), round(r['frac'],4), round(r.get('fp64_flops_frac',0),4), r.get('traffic'))
PY
grep "CLI end\|host buffers\|stages" $O/e2e.txt; tail -2 $O/cfg4_pipeline.txt
