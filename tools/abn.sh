#!/bin/bash
# Development tool: several builds of the library against each other on one GPU box.
#   usage: CONFIGS="C2 C5" SPP=64 REPEATS=2 tools/abn.sh <lib.so>...
for i in $(seq ${REPEATS:-2}); do
  for C in ${CONFIGS:-C2}; do
    for L in "$@"; do
      MIRT_LIBMIRT=$PWD/$L timeout -k 10 120 python bench.py --config $C --spp ${SPP:-64} --steps 3 --warmup 1 --cpu-seconds 0 --no-profile 2>/dev/null |
        python -c "import sys,json; d=json.load(sys.stdin); print('$C', '$L'.split('/')[-1], round(d['value'],1))" || exit 1
    done
  done
done
