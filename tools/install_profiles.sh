#!/bin/bash
# Copies what tools/measure_round.sh left under gpurun_out/<dir> into profiles/ under the round's tag.
#   usage: tools/install_profiles.sh gpurun_out/<dir> [tag]
D=$1; TAG=${2:-r03}
cp $D/${TAG}_pmc_C*.json profiles/ && cp $D/${TAG}_phase_stats.txt profiles/ && cp $D/prof/${TAG}_kernel_stats.csv profiles/ &&
cp $D/bench.json profiles/${TAG}_bench.json && for c in C3 C4 C5; do cp $D/bench_$c.json profiles/${TAG}_bench_$c.json; done &&
cp $D/bench_wavefront.json profiles/${TAG}_bench_wavefront.json && cp $D/bench_paths_64spp.json profiles/${TAG}_bench_paths_64spp.json
