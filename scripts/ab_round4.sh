#!/bin/bash
# Same-box A/B of library builds on the config-3 and config-5 blocks: bash scripts/ab_round4.sh libA.so libB.so ...   (two rounds each)
set -u
for r in 1 2; do
  for lib in "$@"; do
    a=$(BMX_LIB_NAME=$lib BMX_ALLOW_STALE=1 python scripts/prep_check.py --config 3 --windows 262144 --variants 0 --reps 3 2>&1 | grep "^variant  0 clr" | sed 's/.*-> \([0-9.]*\) M.*/\1/')
    b=$(BMX_LIB_NAME=$lib BMX_ALLOW_STALE=1 python scripts/prep_check.py --config 5 --windows 131072 --variants 0 --reps 3 2>&1 | grep "^variant  0 clr" | sed 's/.*-> \([0-9.]*\) M.*/\1/')
    echo "round $r  $lib  config3 $a  config5 $b"
  done
done
