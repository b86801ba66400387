mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
BMX_LIB_NAME=libbmx_diag.so python scripts/far_accuracy.py 65536 > $O/far_accuracy.txt 2>&1
python scripts/parity_summary.py > $O/parity_summary.txt 2>&1
BMX_LIB_NAME=libbmx_prof.so python scripts/kexp.py --reps 1 --windows 200000 --tag prof > $O/section_shares.txt 2>&1
BMX_LIB_NAME=libbmx_count.so python scripts/kexp.py --reps 1 --windows 200000 --tag count > $O/event_counts.txt 2>&1
python bench.py --config 3 --no-cpu-baseline --n-spread 10 > $O/bench_c3_spread10.json 2> /dev/null
python bench.py --config 3 --no-cpu-baseline --n-spread 40 > $O/bench_c3_spread40.json 2> /dev/null
tail -c 400 $O/bench_1gpu.json; echo; cat $O/far_accuracy.txt; tail -15 $O/parity_summary.txt
