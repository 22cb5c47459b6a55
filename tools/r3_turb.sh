#!/bin/bash
set -o pipefail
CONFIGS="C3 C5" SPP=32 REPEATS=2 tools/abn.sh ray_tracer_amd/csrc/libmirt_nowave.so ray_tracer_amd/csrc/libmirt.so ray_tracer_amd/csrc/libmirt_lean.so ray_tracer_amd/csrc/libmirt_call.so || exit 1
MIRT_LIBMIRT=$PWD/ray_tracer_amd/csrc/libmirt_stats.so timeout -k 10 120 python bench.py --config C3 --spp 32 --steps 1 --warmup 0 --cpu-seconds 0 --no-profile 2>&1 | grep "k_paths\]" | head -14
