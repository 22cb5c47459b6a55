"""Development tool: scenes of a few shapes (spheres on a ground sphere, or spheres and cubes) at 1200x800,
rendered with and without the tree (MIRT_SCAN_MAX decides; run once per setting).
  usage: MIRT_SCAN_MAX=<0|64> python tools/small_scene_bench.py [spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import scene_util as su
from ray_tracer_amd import _abi as abi, api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H = 1200, 800
cam = api.Camera.new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0, 0.0, 1.0)
textures = [su.constant(0.5, 0.5, 0.5), su.constant(0.7, 0.3, 0.3)]
materials = [su.material(abi.MAT_LAMBERTIAN, 0), su.material(abi.MAT_LAMBERTIAN, 1), su.material(abi.MAT_DIELECTRIC, param=1.5),
             su.material(abi.MAT_METAL, albedo=(0.7, 0.6, 0.5), param=0.0)]
for n, cubes in ((3, False), (7, False), (15, False), (7, True), (10, True)):
    rng = np.random.default_rng(n)
    shapes = [su.sphere(0, -1000, 0, 1000, 0)]
    for k in range(n):
        c = np.array([rng.uniform(-4, 4), 0.5, rng.uniform(-3, 3)])
        if cubes and k % 2:
            shapes.append(su.shape(abi.SHAPE_CUBE, 1 + k % 3, *(c - 0.4), *(c + 0.4)))
        else:
            shapes.append(su.sphere(*c, 0.5, 1 + k % 3))
    built = su.Built(shapes, materials, textures, seed=3)
    scene = api.Scene(built.desc)
    api.render_world(api.SKY, scene, cam, W, H, 2)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        img, st = api.render_world(api.SKY, scene, cam, W, H, spp, return_stats=True)
        best = min(best, time.perf_counter() - t)
    print(f"MIRT_SCAN_MAX={os.environ.get('MIRT_SCAN_MAX', 'default')}: {len(shapes)} shapes ({'spheres and cubes' if cubes else 'spheres'}): "
          f"{W * H * spp / best / 1e6:.1f} Msamples/s, {st['segments'] / st['samples']:.2f} segments per sample", flush=True)
    scene.close()
