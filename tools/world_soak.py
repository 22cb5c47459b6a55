"""Development tool (run on the GPU box): Hittable::hit and small renders against worlds nobody would build -- spheres
of radius zero, tiny, huge or negative, cubes with unordered or coinciding corners, rects of no area or with their
ends swapped, shapes on top of each other, moving spheres whose shutter span is empty, negative or not a power of
two, everything scaled by 10^-30 .. 10^30 -- with ordinary and wild rays; all three traversals against the oracle.
A world the library refuses to create (mirt_scene_create's own checks) is counted, not an error.
  usage: python tools/world_soak.py [first seed] [worlds] [rays per world] [big]
`big`: worlds of 900 .. 20 000 shapes instead (the path kernel's other placements of the tree: nodes in LDS with the
records in global memory, 28-byte nodes, everything from global memory, 32-bit stack entries)."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib
import scene_util as su
import test_gpu_intersect as ti
from ray_tracer_amd import _abi as abi, api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
big = len(sys.argv) > 4
failed = refused = total = 0
t0 = time.perf_counter()
for w in range(worlds):
    seed = first + w
    rng = np.random.default_rng(4242 + seed)
    scale = 10.0 ** rng.choice([0, 0, 0, -30, -8, 5, 12, 30])
    count = int(rng.choice([900, 1500, 2500, 3600, 3800, 6000, 20000, 70000] if big else [1, 2, 3, 7, 20, 60, 400]))
    shapes = []
    for k in range(count):
        c = rng.uniform(-6.0, 6.0, 3)
        if rng.random() < 0.2:
            c = np.round(c)  # coinciding centres, faces and planes
        kind = int(rng.integers(0, 6))
        if kind == 0:
            r = rng.choice([0.0, 1e-20, 1e-3, 0.5, 1.0, 3.0, 1e6, -0.7, -1e-9])
            shapes.append(su.sphere(*(c * scale), r * scale))
        elif kind == 1:
            t_a, t_b = rng.choice([(0.0, 1.0), (0.0, 0.0), (1.0, 0.0), (0.25, 0.55), (-3.0, 5.0), (0.0, 1e-300)])
            shapes.append(su.moving_sphere(c * scale, (c + rng.uniform(-2.0, 2.0, 3)) * scale, rng.choice([0.4, 1.0, 0.0, -0.5]) * scale, t_a, t_b))
        elif kind == 2:
            e = rng.uniform(-1.0, 2.5, 3) * rng.choice([1.0, 1.0, 0.0], 3)  # unordered corners, flat cubes
            shapes.append(su.shape(abi.SHAPE_CUBE, 0, *(c * scale), *((c + e) * scale)))
        else:
            a = rng.uniform(-1.0, 5.0, 2) * rng.choice([1.0, 1.0, 0.0], 2)  # swapped ends, no area
            shapes.append(su.shape(abi.SHAPE_XY_RECT + kind - 3, 0, c[0] * scale, (c[0] + a[0]) * scale, c[1] * scale, (c[1] + a[1]) * scale, c[2] * scale))
    if rng.random() < 0.3 and shapes:
        shapes += [shapes[int(rng.integers(0, len(shapes)))] for _ in range(3)]  # the same shape several times
    t_lo, t_hi = rng.choice([(0.0, 1.0), (0.0, 0.0), (0.3, 0.4)])
    built = su.Built(shapes, seed=seed, t0=float(t_lo), t1=float(t_hi))
    # every third world with the reference's boxes altered: shrunk towards a point (they still nest but cut into
    # their shapes), each bound moved at random (they no longer nest), or blown up; the answer is still what the
    # reference's recursion makes of THESE boxes (bvh.rs:61-66)
    boxes = ("as built", "as built", "shrunk", "as built", "scrambled", "as built", "inflated")[seed % 7]
    for i in range(built.desc.n_nodes):
        m = built.nodes[i]
        if m.is_leaf or boxes == "as built":
            continue
        for a in range(3):
            if boxes == "shrunk":
                m.bmin[a] = scale + (m.bmin[a] - scale) * 0.8
                m.bmax[a] = scale + (m.bmax[a] - scale) * 0.8
            elif boxes == "scrambled":
                m.bmin[a] += rng.uniform(-1.0, 2.0) * scale
                m.bmax[a] += rng.uniform(-2.0, 1.0) * scale
            else:
                m.bmin[a] -= rng.uniform(0.0, 3.0) * scale
                m.bmax[a] += rng.uniform(0.0, 3.0) * scale
    rays = ti._rays_nobody_would_send(n, 99 + seed, built.shapes)
    # half of the rays in the world's own scale, aimed at it
    half = n // 2
    rays["origin"][:half] = rng.uniform(-14.0, 14.0, (half, 3)) * scale
    target = rng.uniform(-6.0, 6.0, (half, 3)) * scale
    rays["direction"][:half] = (target - rays["origin"][:half]) * (10.0 ** rng.uniform(-3, 3, (half, 1)))
    rays["time"][:half] = rng.uniform(-0.2, 1.2, half)
    try:
        scene = api.Scene(built.desc)
    except api.MirtError:
        refused += 1
        continue
    try:
        for t_min, t_max in ((0.001, float("inf")), (-5.0 * scale, 40.0 * scale), (0.001 * scale, 1e30)):
            ref, _ = oracle_lib.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            ref_list, _ = oracle_lib.intersect(built.desc, rays, t_min=t_min, t_max=t_max, use_bvh=False)
            for what, got, want in (("fast", scene.hit(rays, t_min, t_max), ref),
                                    ("reference-order", scene.hit(rays, t_min, t_max, traversal=abi.TRAVERSE_REFERENCE), ref),
                                    ("list", scene.hit(rays, t_min, t_max, use_bvh=False), ref_list)):
                total += n
                try:
                    ti._compare(got, want, what)
                except AssertionError as e:
                    failed += 1
                    print(f"FAILED seed {seed} ({len(shapes)} shapes, scale {scale:g}, boxes {boxes}) range ({t_min:g}, {t_max:g}) {what}: {str(e)[:160]}", flush=True)
        # a small render of the same world through the path kernel and the launch-per-bounce pipeline
        cam = api.Camera.new(tuple(np.array([3.0, 4.0, 13.0]) * scale), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0, float(t_lo), float(t_hi))
        refimg, _, ctr, _ = oracle_lib.render(built.desc, cam, api.SKY, 48, 32, 2)
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0)):
            img, st = api.render_world(api.SKY, scene, cam, 48, 32, 2, return_stats=True, handover=handover)
            d = np.abs(img.astype(int) - refimg.astype(int))
            if st["segments"] != ctr["segments"] or d.max() > 1:
                failed += 1
                print(f"FAILED seed {seed} ({len(shapes)} shapes, scale {scale:g}, boxes {boxes}) render handover {handover}: segments {st['segments']} vs {ctr['segments']}, max byte difference {d.max()}", flush=True)
    finally:
        scene.close()
    if w % 20 == 19:
        print(f"{w + 1} worlds ({refused} refused), {total / 1e6:.1f} M ray queries, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{worlds} worlds ({refused} refused), {total / 1e6:.1f} M ray queries, {failed} comparisons failed")
sys.exit(1 if failed else 0)
