mkdir -p gpurun_out/fuzz
timeout 1500 python scripts/fuzz_gpu.py 100 400 > gpurun_out/fuzz/fuzz.txt 2>&1; tail -5 gpurun_out/fuzz/fuzz.txt
