"""Development tool: the numbers behind the frozen caps of tests/test_gpu_render.py::test_converged_images_of_every_scene:
per scene, RMSE (8-bit, after gamma, per channel) and mean-luminance difference of a device render against the oracle
with another seed, beside the noise floor (two device renders with different seeds)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from ray_tracer_amd import api, configs

FRAMES = {"C3": (160, 90, 2048), "C4": (128, 128, 4096), "C5": (160, 90, 4096)}


def rmse(x, y):
    return np.sqrt(((x - y) ** 2).reshape(-1, 3).mean(axis=0))


def lum(img):
    return (img @ np.array([0.2126, 0.7152, 0.0722])).mean()


for name, (w, h, spp) in FRAMES.items():
    cfg = configs.ALL[name]
    world = cfg.world()
    cam = cfg.camera(w, h)
    scene = api.Scene(world)
    a = api.render_world(cfg.background, scene, cam, w, h, spp, seed=1111).astype(np.float64)
    b = api.render_world(cfg.background, scene, cam, w, h, spp, seed=4444).astype(np.float64)
    scene.close()
    ref, _, _, secs = oracle_lib.render(world.desc, cam, cfg.background, w, h, spp, seed=2222)
    ref = ref.astype(np.float64)
    print(name, (w, h, spp), "floor", rmse(a, b).round(3), "device vs oracle", rmse(a, ref).round(3), "lum diff %.4f %%" % (100 * abs(lum(a) - lum(ref)) / lum(ref)),
          "lum floor %.4f %%" % (100 * abs(lum(a) - lum(b)) / lum(a)), "oracle %.1f s" % secs, flush=True)
