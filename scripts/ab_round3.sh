mkdir -p gpurun_out/r03
O=gpurun_out/r03/hot_rows.txt
: > $O
for SP in 10 40; do
  for V in 0 12; do
    echo "### n-spread $SP variant $V" >> $O
    timeout 300 python bench.py --config 3 --n-spread $SP --no-cpu-baseline --steps 2 --variant $V 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel'], d['value'], d['config']['checksum_clr'])" >> $O 2>&1
  done
done
cat $O
