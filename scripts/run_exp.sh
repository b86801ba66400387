mkdir -p gpurun_out
O=gpurun_out/exp8.txt
: > $O
python -m pytest tests/test_gpu_round2.py -x -q -k "not config4 and not config5" 2>&1 | tail -25 >> $O
cat $O
