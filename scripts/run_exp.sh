python -m pytest tests -m gpu -x -q 2>&1 | tail -3
O=gpurun_out/stride_r02b.txt; : > $O
for s in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 160 200; do python scripts/kexp.py --step $s --windows 65536 --reps 2 --tag default >> $O 2>&1; done
python scripts/kexp.py --snps 4000000 --step 200 --windows 65536 --reps 2 --tag default4M >> $O 2>&1
python scripts/kexp.py --snps 4000000 --step 64 --windows 65536 --reps 2 --tag default4M >> $O 2>&1
cut -c1-110 $O
