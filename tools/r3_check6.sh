#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_multi.py tests/test_gpu_cli.py tests/test_zzz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -15 || exit 1
timeout -k 10 300 python tools/calibrate_image_gate.py || exit 1
