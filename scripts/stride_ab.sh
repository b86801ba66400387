#!/bin/bash
# Stride table, same box: bash scripts/stride_ab.sh "2 3 4 6 8 12" libA.so libB.so ...   (config-3 chromosome, 131072 test sites per stride)
set -u
STRIDES=$1; shift
for st in $STRIDES; do
  for lib in "$@"; do
    BMX_LIB_NAME=$lib BMX_ALLOW_STALE=1 python scripts/prep_check.py --config 3 --snps 2000000 --step $st --windows 131072 --variants 12,0 --reps 2 2>&1 | grep -E "^variant  0" | sed "s/^/stride $st  $lib  /" | cut -c1-250
  done
done
