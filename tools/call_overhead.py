"""Development tool (run on the GPU box): the fixed cost of one mirt_render call -- a frame of a few pixels at one
sample per pixel, where the kernels' own work is nothing -- and what a full-size image adds (resolve + copy out).
  usage: python tools/call_overhead.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ray_tracer_amd import api, configs

cfg = configs.C2
world = cfg.world()
scene = api.Scene(world)
for (w, h, spp) in ((8, 8, 1), (1200, 800, 1), (1200, 800, 8)):
    cam = cfg.camera(w, h)
    out = np.zeros((h, w, 3), dtype=np.uint8)
    for _ in range(3):
        api.render_world(cfg.background, scene, cam, w, h, spp, out=out)
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        api.render_world(cfg.background, scene, cam, w, h, spp, out=out)
    print(f"{w}x{h}x{spp}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")
scene.close()
