"""Development tool: what one rank of an N-GPU tile split renders, timed on one GPU.
    python tools/rank_share.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import api, configs  # noqa: E402

cfg = configs.C2
world = cfg.world()
scene = api.Scene(world)
cam = cfg.camera()
out = np.zeros((cfg.height, cfg.width, 3), dtype=np.uint8)
base = None
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    times = []
    for rep in range(4):
        t0 = time.perf_counter()
        _, st = api.render_world(cfg.background, scene, cam, cfg.width, cfg.height, cfg.spp, out=out, return_stats=True,
                                 tiles=(cfg.width, n, 0), profile=True)
        times.append(time.perf_counter() - t0)
    t = min(times[1:])
    base = base or t
    print(f"N={n}: rank 0 share {t * 1e3:7.1f} ms (kernels {st['ms_paths'] + st['ms_accumulate']:7.1f} ms), "
          f"ideal {base / n * 1e3:7.1f} ms, efficiency {base / n / t:.3f}")
scene.close()
