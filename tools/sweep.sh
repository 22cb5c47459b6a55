#!/bin/bash
# Development tool: one bench.py line per value of an environment variable.
#   usage: tools/sweep.sh VAR "v1 v2 ..." [bench.py arguments]
#   e.g.   tools/sweep.sh MIRT_PATHS_TUNING "$((2|48<<8|16<<16)) $((2|32<<8|16<<16))" --spp 64
var=$1; vals=$2; shift 2
for v in $vals; do
  export $var=$v
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null |
    python -c "import json,sys; d=json.load(sys.stdin); print('$var=$v', round(d['value'],1), 'Msamples/s', d['handover']['bounce'], {k: round(x,1) for k,x in d.get('kernel_ms',{}).items() if x})" || exit 1
done
