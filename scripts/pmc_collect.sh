#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline (run on the GPU box, from the repo root):
#   [PMC_KEY=config3_step64] bash scripts/pmc_collect.sh <tag> <bench args...>      e.g.  bash scripts/pmc_collect.sh c3 --config 3
# One process per counter group (the SQ block has 8 slots, FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), each with
# --kernel-trace so that the dispatch list comes with it; the program sits directly behind `--`.  Output: gpurun_out/${ROUND:-r04}/<tag>/.
set -u
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/${ROUND:-r04}/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$* --steps 1 --warmup 0 --no-cpu-baseline --no-cold-pass --no-parity-sample"
pass() {   # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
}
pass A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
pass B GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass C SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
pass D FETCH_SIZE
pass E WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
python3 scripts/pmc_calibrate.py $TAG $OUT ${PMC_KEY:-} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
