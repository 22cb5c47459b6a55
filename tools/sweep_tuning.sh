#!/bin/bash
# Development tool: renders C2 at 32 spp for a grid of traversal scheduling parameters
# (MIRT_TRAV_TUNING = stay | refill << 8) and prints Msamples/s and the extend kernel time.
for stay in ${STAYS:-0 2 3 4}; do
  for refill in ${REFILLS:-8 16 32}; do
    export MIRT_TRAV_TUNING=$((stay | (refill << 8)))
    timeout -k 10 120 python bench.py --spp 32 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null |
      python -c "import json,sys; d=json.load(sys.stdin); print('stay=$stay refill=$refill', round(d['value'],1), 'Msamples/s  extend', round(d['kernel_ms']['extend'],1), 'ms')" || exit 1
  done
done
