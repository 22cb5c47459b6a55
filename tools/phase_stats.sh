#!/bin/bash
# Development tool (GPU box): wave-clock shares of the path kernel's pieces per configuration, from libmirt_stats.so
#   usage: tools/phase_stats.sh [configs...]
declare -A SPP=([C2]=69 [C3]=32 [C4]=64 [C5]=8)
for c in ${@:-C2 C3 C4 C5}; do
  echo "== $c (bench.py --config $c --spp ${SPP[$c]} --steps 1 --warmup 1, libmirt_stats.so = -DMIRT_PATHS_STATS, last launch)"
  MIRT_LIBMIRT=$PWD/ray_tracer_amd/csrc/libmirt_stats.so timeout -k 10 120 python bench.py --config $c --spp ${SPP[$c]} --steps 1 --warmup 1 --cpu-seconds 0 --no-profile 2>&1 | grep "k_paths\]" | tail -17
done
