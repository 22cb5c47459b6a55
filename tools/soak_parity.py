"""Development tool (run on the GPU box): the four configurations at (nearly) their own frame sizes against the CPU
oracle with the same sampler -- images byte for byte, and the number of closest-hit queries.  Tens of millions of
paths per scene; takes about a minute, most of it the oracle.
  usage: python tools/soak_parity.py"""
import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from ray_tracer_amd import api, configs
for name, (w, h, spp) in {"C2": (1200, 800, 16), "C3": (1920, 1080, 8), "C4": (1024, 1024, 32), "C5": (1920, 1080, 6)}.items():
    cfg = configs.ALL[name]
    world = cfg.world()
    cam = cfg.camera(w, h)
    t = time.perf_counter()
    ref, _, ctr, _ = oracle_lib.render(world.desc, cam, cfg.background, w, h, spp)
    t_cpu = time.perf_counter() - t
    gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, return_stats=True)
    d = np.abs(gpu.astype(int) - ref.astype(int))
    print(f"{name} {w}x{h}x{spp}: {w * h * spp / 1e6:.1f} M paths, segments {st['segments']} (oracle {ctr['segments']}), "
          f"max byte difference {d.max()}, bytes differing {np.count_nonzero(d)} of {d.size}, oracle {t_cpu:.1f} s", flush=True)
    assert st["segments"] == ctr["segments"] and d.max() <= 1
