#!/bin/bash
# dev: parity of the default library on the scenes that matter, then A/B against csrc/libmirt_before.so
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_fuzz.py tests/test_gpu_corners.py tests/test_golden.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
CONFIGS="${CONFIGS:-C2 C3 C4 C5}" SPP=${SPP:-64} REPEATS=2 tools/abn.sh ray_tracer_amd/csrc/libmirt_before.so ray_tracer_amd/csrc/libmirt.so
