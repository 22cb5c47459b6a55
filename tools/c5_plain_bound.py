"""Development tool: what the 3 507-shape scene would run at if its two rarely hit textures (one noise sphere,
one image sphere) did not force the kernel with all texture code on every path: the same scene with those
two textures replaced by constants, which selects the plain kernel (4 waves per SIMD).
  usage: python tools/c5_plain_bound.py [spp]"""
import sys, time
sys.path.insert(0, ".")
from ray_tracer_amd import _abi as abi, api, configs

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = configs.C5
W, H = cfg.width, cfg.height
cam = cfg.camera()


def run(world, label):
    scene = api.Scene(world)
    api.render_world(cfg.background, scene, cam, W, H, 1)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        img, st = api.render_world(cfg.background, scene, cam, W, H, spp, return_stats=True)
        best = min(best, time.perf_counter() - t)
    print(f"{label}: {W * H * spp / best / 1e6:.1f} Msamples/s, {st['segments'] / st['samples']:.3f} segments per sample", flush=True)
    scene.close()


world = cfg.world()
run(world, "as it is")
d = world.desc.contents
n = 0
for i in range(d.n_textures):
    t = d.textures[i]
    if t.kind not in (abi.TEX_CONSTANT, abi.TEX_CHECKER):
        t.kind = abi.TEX_CONSTANT
        t.color[0], t.color[1], t.color[2] = 0.5, 0.5, 0.5
        n += 1
print(n, "textures flattened")
run(world, "noise and image textures as constants")
