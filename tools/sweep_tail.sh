#!/bin/bash
# Development tool: renders C2 at 32 spp for a range of tail thresholds (MIRT_TAIL_PATHS).
for paths in ${PATHS:-1 16384 65536 262144 1048576 4194304}; do
  export MIRT_TAIL_PATHS=$paths
  timeout -k 10 120 python bench.py --spp ${SPP:-32} --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null |
    python -c "import json,sys; d=json.load(sys.stdin); print('tail_paths=$paths', round(d['value'],1), 'Msamples/s', d['tail'], {k: round(v,1) for k,v in d['kernel_ms'].items()})" || exit 1
done
