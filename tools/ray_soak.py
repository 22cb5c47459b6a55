"""Development tool (run on the GPU box): Hittable::hit on rays nobody would send -- components of any magnitude from
1e-310 to 1e300, zeros of both signs, infinities, NaN, origins on shapes' surfaces and centres, directions along
axes -- against random worlds (those of tests/test_gpu_fuzz.py), finite and infinite (t_min, t_max): the fast and
the reference-order traversal and the linear list against the oracle, hit / miss, shape, side and the bits of t.
  usage: python tools/ray_soak.py [first seed] [worlds] [rays per world]"""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib
import test_gpu_fuzz as fz
import test_gpu_intersect as ti
from ray_tracer_amd import _abi as abi, api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 1e-300, 1e300, -1e300, 1.0, -1.0, 0.001])
failed = 0
total = 0
t0 = time.perf_counter()
for w in range(worlds):
    seed = first + w
    kind = w % 4
    built, eye, _, _ = fz._random_world(seed, [(24, 90), (1 + seed % 5, 16), (1 + seed % 4, 11), (300, 900)][kind], planar=kind == 2)
    rng = np.random.default_rng(77 + seed)
    rays = np.zeros(n, dtype=api.RAY_DTYPE)
    # ordinary rays towards the scene ...
    rays["origin"] = rng.uniform(-14.0, 14.0, (n, 3))
    rays["direction"] = rng.uniform(-1.0, 1.0, (n, 3))
    rays["time"] = rng.uniform(-0.5, 1.5, n)
    # ... a third with magnitudes from all over the range of f64 ...
    wild = rng.random(n) < 0.33
    mag = 10.0 ** rng.uniform(-300, 300, (n, 3))
    rays["direction"][wild] *= mag[wild]
    wild_o = rng.random(n) < 0.15
    rays["origin"][wild_o] *= (10.0 ** rng.uniform(-30, 30, (n, 3)))[wild_o]
    # ... origins exactly on shapes' parameters (centres, corners, planes), directions along axes ...
    shapes = built.shapes
    pick = rng.integers(0, len(shapes), n)
    on = rng.random(n) < 0.2
    for i in np.nonzero(on)[0][:2000]:
        f = shapes[pick[i]].f
        rays["origin"][i] = (f[0], f[1], f[2]) if rng.random() < 0.5 else (f[0], f[2], f[4])
    axis = rng.random(n) < 0.1
    for i in np.nonzero(axis)[0]:
        d = np.zeros(3); d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0, 1e-200, 1e200]); rays["direction"][i] = d
    # ... and special values dropped into single components
    drop = rng.random(n) < 0.25
    for i in np.nonzero(drop)[0]:
        rays[("origin", "direction")[rng.integers(0, 2)]][i, rng.integers(0, 3)] = special[rng.integers(0, len(special))]
    nasty_t = rng.random(n) < 0.05
    rays["time"][nasty_t] = special[rng.integers(0, len(special), np.count_nonzero(nasty_t))]
    scene = api.Scene(built.desc)
    try:
        for t_min, t_max in ((0.001, float("inf")), (-5.0, 40.0), (1e-9, 1e30)):
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
                    print(f"FAILED seed {seed} ({len(shapes)} shapes) range ({t_min}, {t_max}) {what}: {str(e)[:200]}", flush=True)
    finally:
        scene.close()
    if w % 20 == 19:
        print(f"{w + 1} worlds, {total / 1e6:.1f} M ray queries, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{worlds} worlds, {total / 1e6:.1f} M ray queries, {failed} comparisons failed")
sys.exit(1 if failed else 0)
