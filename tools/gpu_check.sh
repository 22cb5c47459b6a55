#!/bin/bash
# Development tool (run on the GPU box through gpurun): the GPU tests, then one short bench line
# per configuration, then the clock statistics of the path kernel for C2.
#   usage: tools/gpu_check.sh [configs...]      default: C2 C3 C4 C5
set -o pipefail
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
for c in ${@:-C2 C3 C4 C5}; do
  timeout -k 10 120 python bench.py --config $c --spp 64 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null |
    python -c "import sys,json; d=json.load(sys.stdin); print('$c', round(d['value'],1), 'Msamples/s', {k: round(x,1) for k,x in d.get('kernel_ms',{}).items() if x})" || exit 1
done
if [ -f ray_tracer_amd/csrc/libmirt_stats.so ]; then
  MIRT_LIBMIRT=$PWD/ray_tracer_amd/csrc/libmirt_stats.so timeout -k 10 120 python bench.py --config C2 --spp 64 --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | grep "k_paths\]" | tail -16
fi
