#!/bin/bash
# Run on the GPU box (through gpurun): everything the committed profiles/ come from.
#   usage: tools/measure_round.sh <outdir under gpurun_out/> [tag]
# 1. the GPU tests   2. the default bench line   3. rocprofv3 kernel statistics of the same command
# 4. SQ / TCC counters of ONE k_paths launch of C2 (69 spp = 66,240,000 paths), one run per group
# 5. one short bench line per other configuration
OUT=$1; TAG=${2:-r01}
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { cat $OUT/bench.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python3 bench.py --cpu-seconds 0 > $OUT/prof.log 2>&1 || { tail $OUT/prof.log; exit 1; }
SPP=69 tools/pmc_render.sh $OUT/pmc sq1 sq2 fetch write || exit 1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.json || exit 1
for c in C3 C4 C5; do
  timeout -k 10 200 python bench.py --config $c --spp 128 --steps 2 --warmup 1 --cpu-seconds 8 > $OUT/bench_$c.json 2>> $OUT/bench.err || exit 1
done
cat $OUT/bench.json
