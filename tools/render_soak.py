"""Development tool (run on the GPU box): small renders of random worlds (those of tests/test_gpu_fuzz.py) whose
materials, textures and cameras are pushed to the ends of what is finite -- albedos below zero and above one, fuzz
of 50 or negative, refraction indices from 0.001 to 50 and negative, colours up to 9, nested checkers five deep,
a lens as wide as the scene, a field of view of 0.01 and of 179 degrees, a camera inside or on a shape, shutter
spans that are empty or reversed, backgrounds below zero, depth budgets from 1 to 255 -- through the path kernel
and the launch-per-bounce pipeline against the oracle: equal numbers of closest-hit queries, bytes within a level.
(Colours are kept where 255 of them multiplied stay finite: the device multiplies a path's attenuations from the
camera on, the reference from the path's end back (renderer.rs:107-108), and where one of the two orders overflows
and the other does not -- albedos of 1e6 over sixty bounces -- a pixel is NaN in one and a number in the other.)
  usage: python tools/render_soak.py [first seed] [worlds]"""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib
import scene_util as su
import test_gpu_fuzz as fz
from ray_tracer_amd import _abi as abi, api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
failed = 0
t0 = time.perf_counter()
for k in range(worlds):
    seed = first + k
    rng = np.random.default_rng(31337 + seed)
    kind = k % 4
    built0, eye, _, _ = fz._random_world(seed, [(24, 90), (1 + seed % 5, 16), (1 + seed % 4, 11), (200, 500)][kind], planar=kind == 2)
    materials = list(built0.materials)
    textures = list(built0.textures)
    for m in materials:
        if rng.random() < 0.5:
            for c in range(3):
                m.albedo[c] = float(rng.choice([-2.0, 0.0, 0.3, 1.0, 7.0]))
            if m.kind == abi.MAT_METAL:
                m.param = float(rng.choice([0.0, 0.5, 1.0, 50.0, -3.0]))
            if m.kind == abi.MAT_DIELECTRIC:
                m.param = float(rng.choice([1e-3, 0.5, 1.0, 1.5, 50.0, -1.5]))
    for t in textures:
        if t.kind == abi.TEX_CONSTANT and rng.random() < 0.3:
            for c in range(3):
                t.color[c] = float(rng.choice([-1.0, 0.0, 0.5, 1.0, 3.0, 9.0]))
        if t.kind == abi.TEX_NOISE and rng.random() < 0.5:
            t.scale = float(rng.choice([0.0, -4.0, 1e-200, 1e5]))
    # checkers nested five deep on top of the table (children before parents is not required of a caller)
    base = len(textures)
    for d in range(5):
        textures.append(su.checker(int(rng.integers(0, base + d)), int(rng.integers(0, base + d))))
    for m in materials:
        if m.kind in (abi.MAT_LAMBERTIAN, abi.MAT_DIFFUSE_LIGHT) and rng.random() < 0.2:
            m.texture = base + 4
    shapes = list(built0.shapes)
    built = su.Built(shapes, materials, textures, perlins=list(built0.perlins), images=built0.image_arrays, seed=seed)
    w, h = 40, 28
    spp = int(rng.choice([1, 3, 6]))
    depth = int(rng.choice([1, 2, 5, 50, 255]))
    where = rng.random()
    s0 = shapes[int(rng.integers(0, len(shapes)))]
    if where < 0.2:
        look_from = (s0.f[0], s0.f[1], s0.f[2])  # a shape's centre, corner or the start of a rect
    elif where < 0.3:
        look_from = (s0.f[0] + abs(s0.f[3]), s0.f[1], s0.f[2])  # on a sphere's surface, as likely as not
    else:
        look_from = tuple(eye)
    look_at = tuple(rng.uniform(-3.0, 3.0, 3))
    if look_at == look_from:
        continue
    fov = float(rng.choice([0.01, 5.0, 40.0, 120.0, 179.0]))
    aperture = float(rng.choice([0.0, 0.0, 0.3, 30.0]))
    span = [(0.0, 1.0), (0.0, 0.0), (1.0, 0.0), (-5.0, 5.0)][int(rng.integers(0, 4))]
    cam = api.Camera.new(look_from, look_at, (0.0, 1.0, 0.0), fov, w / h, aperture, float(rng.choice([0.1, 10.0, 1e4])), *span)
    background = tuple(float(x) for x in rng.choice([-1.0, 0.0, 0.7, 1.0, 40.0], 3))
    ref, _, ctr, _ = oracle_lib.render(built.desc, cam, background, w, h, spp, max_depth=depth)
    try:
        scene = api.Scene(built.desc)
    except api.MirtError as e:
        print(f"refused seed {seed}: {e}", flush=True)
        continue
    try:
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0), (1, 0)):
            img, st = api.render_world(background, scene, cam, w, h, spp, max_depth=depth, return_stats=True, handover=handover)
            d = np.abs(img.astype(int) - ref.astype(int))
            if st["segments"] != ctr["segments"] or d.max() > 1 or np.count_nonzero(d) > max(2, 0.002 * d.size):
                failed += 1
                print(f"FAILED seed {seed} ({len(shapes)} shapes, depth {depth}, fov {fov}, aperture {aperture}, from {where:.2f}) handover {handover}: "
                      f"segments {st['segments']} vs {ctr['segments']}, max byte difference {d.max()}, {np.count_nonzero(d)} bytes differ", flush=True)
    finally:
        scene.close()
    if k % 50 == 49:
        print(f"{k + 1} worlds, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{worlds} worlds, {failed} renders differed")
sys.exit(1 if failed else 0)
