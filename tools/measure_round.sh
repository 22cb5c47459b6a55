#!/bin/bash
# Run on the GPU box (through gpurun): everything the committed profiles/ come from.
#   usage: tools/measure_round.sh <outdir under gpurun_out/> [tag]
# 1. per configuration: SQ / TCC counters of ONE k_paths launch (one rocprofv3 --pmc run per counter group), copied
#    into this tree's profiles/ so that the bench lines below find the counters of the library they run
# 2. the default bench line (C2)   3. rocprofv3 kernel statistics of the same command
# 4. one bench line per other configuration   5. the launch-per-bounce pipeline on C2 (--handover -1)
# 6. wave-clock shares of the path kernel's pieces (development build)
OUT=$1; TAG=${2:-r03}
export TMPDIR=/tmp
mkdir -p $OUT
# samples per pixel that make ONE launch of about 66 M paths
declare -A SPP=([C2]=69 [C3]=32 [C4]=64 [C5]=8)
declare -A PATHS=([C2]=66240000 [C3]=66355200 [C4]=67108864 [C5]=66355200)
for c in C2 C3 C4 C5; do
  SPP=${SPP[$c]} BENCH_ARGS="--config $c" tools/pmc_render.sh $OUT/pmc_$c sq1 sq2 fetch write || exit 1
  python tools/pmc_summary.py $OUT/pmc_$c > $OUT/pmc_${c}_summary.json || exit 1
  python tools/pmc_to_profile.py $OUT/pmc_${c}_summary.json ${PATHS[$c]} "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --config $c --spp ${SPP[$c]} --steps 1 --warmup 0 --cpu-seconds 0 --no-profile (one run per counter group: tools/pmc_render.sh)" > $OUT/${TAG}_pmc_$c.json || exit 1
  cp $OUT/${TAG}_pmc_$c.json profiles/${TAG}_pmc_$c.json
  echo "pmc $c done"
done
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { cat $OUT/bench.err; exit 1; }
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python3 bench.py --cpu-seconds 0 > $OUT/prof.log 2>&1 || { tail $OUT/prof.log; exit 1; }
echo "kernel stats done"
for c in C3 C4 C5; do
  timeout -k 10 200 python bench.py --config $c --spp 128 --steps 2 --warmup 1 --cpu-seconds 8 > $OUT/bench_$c.json 2>> $OUT/bench.err || exit 1
done
timeout -k 10 200 python bench.py --handover -1 --spp 64 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench_wavefront.json 2>> $OUT/bench.err || exit 1
timeout -k 10 200 python bench.py --spp 64 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench_paths_64spp.json 2>> $OUT/bench.err || exit 1
cat $OUT/bench.json
# wave-clock shares of the path kernel's pieces, from the development build of the same sources
if [ -f ray_tracer_amd/csrc/libmirt_stats.so ]; then
  : > $OUT/${TAG}_phase_stats.txt
  for c in C2 C3 C4 C5; do
    echo "== $c (python bench.py --config $c --spp ${SPP[$c]} --steps 1 --warmup 1, libmirt_stats.so = -DMIRT_PATHS_STATS, last launch)" >> $OUT/${TAG}_phase_stats.txt
    MIRT_LIBMIRT=$PWD/ray_tracer_amd/csrc/libmirt_stats.so timeout -k 10 120 python bench.py --config $c --spp ${SPP[$c]} --steps 1 --warmup 1 --cpu-seconds 0 --no-profile 2>&1 |
      grep "k_paths\]" | tail -17 >> $OUT/${TAG}_phase_stats.txt || exit 1
  done
fi
