#!/bin/bash
# dev: SQ counters of one C5 launch with and without QNodes
set -o pipefail
for q in 1 0; do
  export MIRT_QNODES=$q
  SPP=8 BENCH_ARGS="--config C5" tools/pmc_render.sh gpurun_out/r3/pmc_C5_q$q sq1 sq2 || exit 1
  python tools/pmc_summary.py gpurun_out/r3/pmc_C5_q$q > gpurun_out/r3/pmc_C5_q$q.json || exit 1
  cat gpurun_out/r3/pmc_C5_q$q.json
done
