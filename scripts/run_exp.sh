python -m pytest tests/test_gpu_round2.py -x -q -k "more_test_sites" 2>&1 | tail -4
export BMX_LIB_NAME=libbmx_diag.so
for g in 8 15 23 35 50 80 150; do BMX_KMOM_GAIN=$g python scripts/kexp.py --tag gain$g 2>&1 | cut -c1-110; done
