#!/bin/bash
# Collects SQ / TCC counters on a short C2 render, one rocprofv3 run per counter group
# (counters are not split over passes automatically).
# usage: [SPP=16] [BENCH_ARGS=...] tools/pmc_render.sh <outdir> [groups...]   groups: sq1 sq2 tcc1 fetch write (default: sq1 sq2)
OUT=$1; shift
GROUPS_WANTED=${@:-sq1 sq2}
export TMPDIR=/tmp
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py --spp ${SPP:-16} --steps 1 --warmup 0 --cpu-seconds 0 --no-profile ${BENCH_ARGS:-} > $OUT/$name.log 2>&1 || exit 1
}
for g in $GROUPS_WANTED; do
  case $g in
    sq1) run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES ;;
    sq2) run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM ;;
    tcc1) run tcc1 TCC_HIT_sum TCC_MISS_sum ;;
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
  esac
done
