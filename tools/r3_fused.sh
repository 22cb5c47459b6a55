#!/bin/bash
set -o pipefail
# same bytes first (scheduling only)
for L in libmirt.so libmirt_f8.so; do
MIRT_LIBMIRT=$PWD/ray_tracer_amd/csrc/$L python - <<'PY' || exit 1
import hashlib, sys
sys.path.insert(0, '.')
from ray_tracer_amd import api, configs
for name, (w, h, spp) in {"C1": (96, 54, 6), "C3": (64, 36, 4), "C5": (64, 36, 3)}.items():
    cfg = configs.ALL[name]
    img, st = api.render_world(cfg.background, cfg.world(), cfg.camera(w, h), w, h, spp, return_stats=True)
    print(name, hashlib.sha256(img.tobytes()).hexdigest()[:16], st["segments"])
PY
done
CONFIGS="C2 C3 C5" SPP=64 REPEATS=2 tools/abn.sh ray_tracer_amd/csrc/libmirt.so ray_tracer_amd/csrc/libmirt_f8.so ray_tracer_amd/csrc/libmirt_f16.so
