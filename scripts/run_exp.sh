python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for s in 64 128 200; do python scripts/kexp.py --step $s --windows 65536 --reps 2 --tag default 2>&1 | cut -c1-120; python scripts/kexp.py --step $s --windows 65536 --reps 2 --variant 2 --tag exact 2>&1 | cut -c1-120; done
python scripts/kexp.py --snps 4000000 --step 200 --windows 65536 --reps 2 --tag default4M 2>&1 | cut -c1-120
