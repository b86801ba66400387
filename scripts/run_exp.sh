mkdir -p gpurun_out
python scripts/config4_cli_pipeline.py > gpurun_out/cfg4_pipeline.txt 2>&1; cat gpurun_out/cfg4_pipeline.txt
python scripts/e2e_timing.py > gpurun_out/e2e_r02.txt 2>&1; tail -5 gpurun_out/e2e_r02.txt
python -m pytest tests/test_gpu_round2.py -x -q -k "not config" 2>&1 | tail -3
