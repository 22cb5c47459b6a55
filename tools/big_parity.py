"""Development tool (run on the GPU box): frames larger than the parity tests' against the CPU oracle with the
same sampler -- images byte for byte, and the number of closest-hit queries.
  usage: python tools/big_parity.py"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from ray_tracer_amd import api, configs
for name, (w, h, spp) in {"C1": (400, 225, 10), "C3": (320, 180, 16), "C4": (160, 160, 24), "C5": (240, 135, 8)}.items():
    cfg = configs.ALL[name]
    world = cfg.world()
    cam = cfg.camera(w, h)
    ref, _, ctr, _ = oracle_lib.render(world.desc, cam, cfg.background, w, h, spp)
    gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, return_stats=True)
    d = np.abs(gpu.astype(int) - ref.astype(int))
    print(name, "segments", st["segments"], ctr["segments"], "max diff", d.max(), "bytes differing", np.count_nonzero(d), "of", d.size, flush=True)
    assert st["segments"] == ctr["segments"] and d.max() <= 1
