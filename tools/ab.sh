#!/bin/bash
# Development tool: alternate two builds of the library on one GPU box, several times each.
#   usage: tools/ab.sh <libA.so> <libB.so> [config] [spp] [repeats]
A=$1; B=$2; C=${3:-C2}; S=${4:-64}; N=${5:-3}
for i in $(seq $N); do
  for L in $A $B; do
    MIRT_LIBMIRT=$PWD/$L timeout -k 10 120 python bench.py --config $C --spp $S --steps 3 --warmup 1 --cpu-seconds 0 --no-profile 2>/dev/null |
      python -c "import sys,json; d=json.load(sys.stdin); print('$C', '$L'.split('/')[-1], round(d['value'],1))" || exit 1
  done
done
