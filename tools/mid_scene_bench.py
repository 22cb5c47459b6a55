"""Development tool: a scene between the two regimes of the path kernel -- too large for its records to fit LDS,
small enough for its nodes to (N ground cubes on a grid + spheres; default ~1 900 shapes) -- rendered at 1920x1080.
  usage: python tools/mid_scene_bench.py [grid] [spheres] [spp]"""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import scene_util as su
from ray_tracer_amd import _abi as abi, api

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 36
n_spheres = int(sys.argv[2]) if len(sys.argv) > 2 else 600
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(5)
shapes = []
w = 2000.0 / grid
for i in range(grid):
    for j in range(grid):
        shapes.append(su.shape(abi.SHAPE_CUBE, 0, -1000 + i * w, 0.0, -1000 + j * w, -1000 + (i + 1) * w, 1 + 100 * rng.uniform(), -1000 + (j + 1) * w))
for k in range(n_spheres):
    c = np.array([-100.0, 270.0, 395.0]) + 165 * rng.uniform(0, 1, 3)
    shapes.append(su.sphere(*c, 10.0, 1))
shapes += [su.sphere(260, 150, 45, 50, 2), su.sphere(0, 150, 145, 50, 3), su.shape(abi.SHAPE_XZ_RECT, 4, 123, 423, 147, 412, 554)]
textures = [su.constant(0.48, 0.83, 0.53), su.constant(0.73, 0.73, 0.73), su.constant(7, 7, 7)]
materials = [su.material(abi.MAT_LAMBERTIAN, 0), su.material(abi.MAT_LAMBERTIAN, 1), su.material(abi.MAT_DIELECTRIC, param=1.5),
             su.material(abi.MAT_METAL, albedo=(0.8, 0.8, 0.9), param=1.0), su.material(abi.MAT_DIFFUSE_LIGHT, 2)]
built = su.Built(shapes, materials, textures, seed=3)
W, H = 1920, 1080
cam = api.Camera.new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0)
scene = api.Scene(built.desc)
api.render_world(api.BLACK, scene, cam, W, H, 2)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    img, st = api.render_world(api.BLACK, scene, cam, W, H, spp, return_stats=True)
    best = min(best, time.perf_counter() - t)
print(f"{len(shapes)} shapes: {W * H * spp / best / 1e6:.1f} Msamples/s, {st['segments'] / best / 1e9:.2f} Grays/s, {st['segments'] / st['samples']:.2f} segments per sample")
