#!/bin/bash
# Collects SQ counters for the extend kernel on a short C2 render (separate passes, one per run).
# usage: tools/pmc_extend.sh <outdir> [extra bench args]
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py --spp 16 --steps 1 --warmup 0 --cpu-seconds 0 --no-profile $EXTRA > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
run tcc1 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
