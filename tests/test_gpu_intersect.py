"""Closest-hit parity on identical rays: the HIP traversals (reference order, fast
order, linear list) against the CPU oracle.  Bar: hit/miss, primitive id and cube
side bit-exact; t bit-exact too (the arithmetic is the same f64 sequence), with
the 1e-5 relative bound of the north star kept as the stated tolerance."""
import numpy as np
import pytest

from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api, configs

pytestmark = pytest.mark.gpu

T_REL_TOL = 1e-5


def _ray_set(oracle, world, cfg, width, height, n_pixels=1500, samples=2):
    """Primary + secondary rays: everything the oracle traces for a spread of pixels."""
    cam = cfg.camera(width, height)
    rng = np.random.default_rng(7)
    pixels = rng.choice(width * height, size=min(n_pixels, width * height), replace=False).astype(np.uint32)
    rays, hits = oracle.trace_rays(world.desc, cam, cfg.background, width, height, pixels, (0, samples))
    return rays, hits


def _compare(gpu, ref, what):
    miss_g, miss_r = gpu["prim"] < 0, ref["prim"] < 0
    assert np.array_equal(miss_g, miss_r), f"{what}: hit/miss differs on {np.count_nonzero(miss_g != miss_r)} rays"
    assert np.array_equal(gpu["prim"], ref["prim"]), f"{what}: primitive id differs on {np.count_nonzero(gpu['prim'] != ref['prim'])} rays"
    assert np.array_equal(gpu["side"], ref["side"]), f"{what}: cube side differs"
    # a hit's t may be NaN or infinite for a degenerate ray (hittable.rs:189-193 lets both through): those must agree as such
    assert np.array_equal(np.isnan(gpu["t"][~miss_r]), np.isnan(ref["t"][~miss_r])), f"{what}: NaN t differs"
    h = ~miss_r & np.isfinite(ref["t"])
    assert np.array_equal(gpu["t"][~miss_r & np.isinf(ref["t"])], ref["t"][~miss_r & np.isinf(ref["t"])]), f"{what}: infinite t differs"
    with np.errstate(invalid="ignore", divide="ignore"):  # a hit at t = 0 exactly: 0 / 0, skipped by nanmax
        rel = np.abs(gpu["t"][h] - ref["t"][h]) / np.abs(ref["t"][h])
    assert rel.size == 0 or np.all(np.isnan(rel)) or np.nanmax(rel) <= T_REL_TOL, f"{what}: t off by {np.nanmax(rel)}"
    # same arithmetic, same bits
    assert np.array_equal(gpu["t"][h], ref["t"][h]), f"{what}: t not bit-identical on {np.count_nonzero(gpu['t'][h] != ref['t'][h])} rays"


@pytest.mark.parametrize("name", ["C1", "C3", "C4", "C5"])
def test_bvh_traversals_match_oracle(name, worlds, oracle):
    cfg = configs.ALL[name]
    world = worlds(name)
    rays, ref = _ray_set(oracle, world, cfg, 160, 90)
    assert rays.shape[0] > 3000
    scene = api.Scene(world)
    try:
        for trav, label in ((abi.TRAVERSE_REFERENCE, "reference-order"), (abi.TRAVERSE_FAST, "fast")):
            gpu = scene.hit(rays, use_bvh=True, traversal=trav)
            _compare(gpu, ref, f"{name} {label}")
    finally:
        scene.close()


@pytest.mark.parametrize("name", ["C1", "C4", "Basic", "SimpleLight"])
def test_linear_list_matches_oracle(name, worlds, oracle):
    cfg = configs.ALL.get(name, configs.C1)
    world = worlds(name)
    cam = cfg.camera(96, 54)
    rays = np.concatenate([api.camera_rays(cam, 96, 54, sample=s) for s in range(2)])
    ref, _ = oracle.intersect(world.desc, rays, use_bvh=False)
    scene = api.Scene(world)
    try:
        gpu = scene.hit(rays, use_bvh=False)
        _compare(gpu, ref, f"{name} list")
    finally:
        scene.close()


@pytest.mark.parametrize("name", ["Basic", "Checker", "Perlin", "Earth", "SimpleLight", "CornellBox"])
def test_reference_scenes_bvh(name, worlds, oracle):
    """The reference's own small scenes, including the one-shape world whose root is a leaf."""
    world = worlds(name)
    look = {"CornellBox": ((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 40.0)}.get(name, ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0))
    cam = api.Camera.new(look[0], look[1], (0, 1, 0), look[2], 16 / 9, 0.1, 10.0, 0.0, 1.0)
    pixels = np.arange(0, 128 * 72, 5, dtype=np.uint32)
    rays, ref = oracle.trace_rays(world.desc, cam, api.SKY, 128, 72, pixels, (0, 1))
    scene = api.Scene(world)
    try:
        for trav in (abi.TRAVERSE_REFERENCE, abi.TRAVERSE_FAST):
            _compare(scene.hit(rays, traversal=trav), ref, f"{name} traversal {trav}")
    finally:
        scene.close()


def test_camera_rays_match_oracle(oracle):
    import ctypes as C

    cfg = configs.C1
    cam = cfg.camera(64, 36)
    gpu = api.camera_rays(cam, 64, 36, sample=3, seed=99)
    ref = np.zeros(64 * 36, dtype=api.RAY_DTYPE)
    r = abi.Ray()
    for idx in range(64 * 36):
        oracle.lib().oracle_camera_ray(C.byref(cam.c), 64, 36, idx, 3, 99, C.byref(r))
        ref[idx] = (tuple(r.origin), tuple(r.direction), r.time)
    for f in ("origin", "direction", "time"):
        assert np.array_equal(gpu[f], ref[f]), f


from golden_cases import _mixed_scene, _random_rays  # noqa: E402


@pytest.mark.parametrize("boxes", ["as built", "shrunk", "scrambled"])
def test_reference_boxes_decide_which_shapes_are_reached(boxes, oracle):
    """The reference reaches a shape only through the boxes above it (bvh.rs:61-66), whatever they
    are.  'shrunk' scales every box toward a point (boxes still nest but cut into their shapes),
    'scrambled' moves every bound at random (boxes no longer nest): in both, many rays hit a shape
    geometrically that the reference never tests, and the answer must still be the reference's."""
    built = _mixed_scene()
    rng = np.random.default_rng(11)
    for i in range(built.desc.n_nodes):
        m = built.nodes[i]
        if m.is_leaf:
            continue
        for a in range(3):
            if boxes == "shrunk":
                m.bmin[a] = 1.0 + (m.bmin[a] - 1.0) * 0.8
                m.bmax[a] = 1.0 + (m.bmax[a] - 1.0) * 0.8
            elif boxes == "scrambled":
                m.bmin[a] += rng.uniform(-1.0, 2.0)
                m.bmax[a] += rng.uniform(-2.0, 1.0)
    rays = _random_rays(200_000, 5)
    ref, _ = oracle.intersect(built.desc, rays)
    everything, _ = oracle.intersect(built.desc, rays, use_bvh=False)
    if boxes != "as built":
        # the altered boxes do change answers: the test would be vacuous otherwise
        assert np.count_nonzero(ref["prim"] != everything["prim"]) > 1000
    scene = api.Scene(built.desc)
    try:
        for trav, label in ((abi.TRAVERSE_REFERENCE, "reference-order"), (abi.TRAVERSE_FAST, "fast")):
            _compare(scene.hit(rays, traversal=trav), ref, f"{boxes} boxes, {label}")
    finally:
        scene.close()


def test_ray_times_outside_the_shutter_and_finite_range(oracle):
    """Moving spheres are boxed over their own time span; a ray at any other time, and a query
    with a finite (t_min, t_max), still get the reference's answer."""
    built = _mixed_scene(seed=4)
    rays = _random_rays(100_000, 6, times=(-3.0, 4.0))
    scene = api.Scene(built.desc)
    try:
        for t_min, t_max in ((0.001, float("inf")), (0.25, 1.0), (-5.0, 0.5)):
            ref, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            _compare(scene.hit(rays, t_min=t_min, t_max=t_max), ref, f"range ({t_min}, {t_max})")
    finally:
        scene.close()


def _scaled_scene(scale, seed=12):
    """The mixed scene with every length multiplied by `scale` (times untouched)."""
    import scene_util as su

    rng = np.random.default_rng(seed)
    shapes = []
    for k in range(64):
        c = rng.uniform(-20.0, 20.0, 3) * scale
        kind = k % 5
        if kind == 0:
            shapes.append(su.moving_sphere(c, c + rng.uniform(-1.0, 1.0, 3) * scale, rng.uniform(0.3, 1.5) * scale, 0.0, 1.0))
        elif kind == 1:
            a = rng.uniform(0.5, 3.0, 2) * scale
            shapes.append(su.shape(abi.SHAPE_XY_RECT + int(rng.integers(0, 3)), 0, c[0], c[0] + a[0], c[1], c[1] + a[1], c[2]))
        elif kind == 2:
            shapes.append(su.shape(abi.SHAPE_CUBE, 0, *c, *(c + rng.uniform(0.5, 3.0, 3) * scale)))
        else:
            shapes.append(su.sphere(*c, rng.uniform(0.3, 1.5) * scale))
    return su.Built(shapes, seed=seed)


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1.0, 1e9, 1e25, 1e60, 1e200])
def test_scenes_of_any_size(scale, oracle):
    """The culling boxes are f32: a scene far outside that range (or far inside its denormals) must
    lose speed at most, never an answer.  Rays aimed into the scene from its own scale, and the
    same rays from very far away (|origin| up to 2^40 scene sizes)."""
    built = _scaled_scene(scale)
    rays = _random_rays(30_000, 8)
    rays["origin"] *= scale
    rays["direction"] *= scale
    far = rays.copy()
    far["origin"][::2] -= far["direction"][::2] * 1.0e6
    far["origin"][1::4] -= far["direction"][1::4] * 1.0e11
    scene = api.Scene(built.desc)
    try:
        for label, rs in (("near", rays), ("far", far)):
            ref, _ = oracle.intersect(built.desc, rs, t_min=0.001, t_max=float("inf"))
            _compare(scene.hit(rs), ref, f"scale {scale} {label}")
        assert np.count_nonzero(oracle.intersect(built.desc, rays)[0]["prim"] >= 0) > 1000
    finally:
        scene.close()


def test_degenerate_rays(oracle):
    """Zero, infinite and NaN components in a ray: whatever the reference's arithmetic makes of them."""
    built = _mixed_scene(seed=5)
    rays = _random_rays(4096, 9)
    special = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-320, 1e308, -1e308]
    rng = np.random.default_rng(10)
    for i in range(rays.shape[0]):
        field = "direction" if i % 2 else "origin"
        rays[field][i, rng.integers(0, 3)] = special[i % len(special)]
        if i % 7 == 0:
            rays["direction"][i, rng.integers(0, 3)] = special[(i // 7) % len(special)]
    rays["time"][::5] = np.nan
    scene = api.Scene(built.desc)
    try:
        ref, _ = oracle.intersect(built.desc, rays)
        gpu = scene.hit(rays)
        gpu_ref = scene.hit(rays, traversal=abi.TRAVERSE_REFERENCE)
        _compare(gpu, ref, "degenerate rays, fast")
        _compare(gpu_ref, ref, "degenerate rays, reference-order")
    finally:
        scene.close()


def test_ties_go_to_the_shape_the_reference_picks(oracle):
    """Every shape twice, coplanar overlapping rects, cubes sharing a face: hits at exactly equal t.
    Under the BVH the reference keeps the right subtree's hit (bvh.rs:70-74), i.e. the larger index
    of the sorted shapes; in a cube the later side (hittable.rs:432-445)."""
    import scene_util as su

    rng = np.random.default_rng(2)
    shapes = []
    for k in range(24):
        c = np.round(rng.uniform(-10.0, 10.0, 3))
        if k % 3 == 0:
            s = su.sphere(*c, float(rng.integers(1, 4)))
        elif k % 3 == 1:
            s = su.shape(abi.SHAPE_XY_RECT + k % 2 * 2, 0, c[0], c[0] + 4.0, c[1], c[1] + 3.0, c[2])
        else:
            s = su.shape(abi.SHAPE_CUBE, 0, *c, *(c + 2.0))
        shapes += [s, s]
        if k % 3 == 1:  # a coplanar rect overlapping half of it
            shapes.append(su.shape(abi.SHAPE_XY_RECT + k % 2 * 2, 0, c[0] + 2.0, c[0] + 6.0, c[1], c[1] + 3.0, c[2]))
        if k % 3 == 2:  # a cube sharing the face x = c.x + 2
            shapes.append(su.shape(abi.SHAPE_CUBE, 0, c[0] + 2.0, c[1], c[2], c[0] + 4.0, c[1] + 2.0, c[2] + 2.0))
    built = su.Built(shapes, seed=4)
    n = 60_000
    rays = np.zeros(n, dtype=api.RAY_DTYPE)
    rays["origin"] = np.round(rng.uniform(-15.0, 15.0, (n, 3)) * 4.0) / 4.0
    rays["direction"] = np.round(rng.uniform(-10.0, 10.0, (n, 3)) * 2.0) / 2.0 - rays["origin"]
    rays["direction"][: n // 3] = np.sign(rays["direction"][: n // 3]) * np.array([[0.0, 0.0, 1.0]])  # along z, onto edges and corners
    rays["direction"][n // 3: n // 2, 0] = 0.0
    ref, _ = oracle.intersect(built.desc, rays)
    first, _ = oracle.intersect(built.desc, rays, use_bvh=False)
    assert np.count_nonzero(ref["prim"] != first["prim"]) > 1000  # ties do occur, and the list resolves them differently
    scene = api.Scene(built.desc)
    try:
        _compare(scene.hit(rays), ref, "ties, fast")
        _compare(scene.hit(rays, traversal=abi.TRAVERSE_REFERENCE), ref, "ties, reference-order")
        _compare(scene.hit(rays, use_bvh=False), first, "ties, list")
    finally:
        scene.close()


def test_cubes_of_every_description(oracle):
    """Cubes with unordered corners (some faces can then never be hit, hittable.rs:189-197), flat cubes,
    cubes touching along faces and edges, huge and tiny ones side by side; rays from outside, from inside,
    from points on the faces, along edges and through corners.  The fast traversal names the faces
    that can matter from f32 bounds before it tests them exactly: no answer may change."""
    import scene_util as su

    rng = np.random.default_rng(21)
    shapes = []
    for k in range(90):
        c = np.round(rng.uniform(-12.0, 12.0, 3) * 2.0) / 2.0
        size = np.round(rng.uniform(0.5, 4.0, 3) * 2.0) / 2.0
        p0, p1 = c.copy(), c + size
        if k % 6 == 1:   # unordered on one or two axes
            a = rng.integers(0, 3)
            p0[a], p1[a] = p1[a], p0[a]
            if k % 12 == 1:
                b = (a + 1) % 3
                p0[b], p1[b] = p1[b], p0[b]
        elif k % 6 == 2:  # flat
            p1[rng.integers(0, 3)] = p0[rng.integers(0, 3)]
        elif k % 6 == 3:  # tiny next to everything else
            p1 = p0 + 1.0e-6
        elif k % 6 == 4:  # large
            p1 = p0 + 40.0
        shapes.append(su.shape(abi.SHAPE_CUBE, 0, *p0, *p1))
        if k % 5 == 0:    # a neighbour sharing the face x = p1.x
            shapes.append(su.shape(abi.SHAPE_CUBE, 0, p1[0], p0[1], p0[2], p1[0] + 1.5, p1[1], p1[2]))
    built = su.Built(shapes, seed=9)
    n = 400_000
    rays = np.zeros(n, dtype=api.RAY_DTYPE)
    rays["origin"] = rng.uniform(-20.0, 20.0, (n, 3))
    target = rng.uniform(-14.0, 18.0, (n, 3))
    rays["direction"] = (target - rays["origin"]) * rng.uniform(0.1, 4.0, (n, 1))
    q = n // 8
    # onto the half-unit lattice the corners live on: edges, corners, coplanar faces
    rays["origin"][:q] = np.round(rays["origin"][:q] * 2.0) / 2.0
    rays["direction"][:q] = np.round(target[:q] * 2.0) / 2.0 - rays["origin"][:q]
    # diagonal directions through lattice points
    rays["direction"][q:2 * q] = rng.choice([-1.0, 1.0, 2.0, -0.5], (q, 3))
    rays["origin"][q:2 * q] = np.round(rays["origin"][q:2 * q] * 2.0) / 2.0
    # starting on a face of some cube (what a scattered ray does)
    for i in range(2 * q, 3 * q):
        s = built.shapes[i % built.desc.n_shapes]
        lo, hi = np.array(s.f[0:3]), np.array(s.f[3:6])
        o = lo + (hi - lo) * rng.uniform(0.0, 1.0, 3)
        a = rng.integers(0, 3)
        o[a] = (lo, hi)[rng.integers(0, 2)][a]
        rays["origin"][i] = o
    ref, _ = oracle.intersect(built.desc, rays)
    assert np.count_nonzero(ref["prim"] >= 0) > n // 4 and len(np.unique(ref["side"])) == 6
    scene = api.Scene(built.desc)
    try:
        _compare(scene.hit(rays), ref, "cubes, fast")
        _compare(scene.hit(rays, traversal=abi.TRAVERSE_REFERENCE), ref, "cubes, reference-order")
        for t_min, t_max in ((0.5, 2.0), (-3.0, 0.75)):
            ref_r, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            _compare(scene.hit(rays, t_min=t_min, t_max=t_max), ref_r, f"cubes, range ({t_min}, {t_max})")
    finally:
        scene.close()


def _rays_nobody_would_send(n, seed, shapes):
    """Components of any magnitude from 1e-310 to 1e300, zeros of both signs, infinities, NaN, origins on the
    shapes' own parameters, directions along axes."""
    rng = np.random.default_rng(seed)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 1e-300, 1e300, -1e300, 1.0, -1.0, 0.001])
    rays = np.zeros(n, dtype=api.RAY_DTYPE)
    rays["origin"] = rng.uniform(-14.0, 14.0, (n, 3))
    rays["direction"] = rng.uniform(-1.0, 1.0, (n, 3))
    rays["time"] = rng.uniform(-0.5, 1.5, n)
    wild = rng.random(n) < 0.33
    rays["direction"][wild] *= (10.0 ** rng.uniform(-300, 300, (n, 3)))[wild]
    wild = rng.random(n) < 0.15
    rays["origin"][wild] *= (10.0 ** rng.uniform(-30, 30, (n, 3)))[wild]
    pick = rng.integers(0, len(shapes), n)
    for i in np.nonzero(rng.random(n) < 0.2)[0][:2000]:
        f = shapes[pick[i]].f
        rays["origin"][i] = (f[0], f[1], f[2]) if rng.random() < 0.5 else (f[0], f[2], f[4])
    for i in np.nonzero(rng.random(n) < 0.1)[0]:
        d = np.zeros(3)
        d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0, 1e-200, 1e200])
        rays["direction"][i] = d
    for i in np.nonzero(rng.random(n) < 0.25)[0]:
        rays[("origin", "direction")[rng.integers(0, 2)]][i, rng.integers(0, 3)] = special[rng.integers(0, len(special))]
    nasty = rng.random(n) < 0.05
    rays["time"][nasty] = special[rng.integers(0, len(special), np.count_nonzero(nasty))]
    return rays


@pytest.mark.parametrize("seed", range(6))
def test_rays_of_any_magnitude_and_ranges_of_either_sign(seed, oracle):
    """Hittable::hit for rays nobody would send, with the renderer's (0.001, inf), a range that begins below zero and
    a tiny t_min: all three traversals against the oracle.  (A direction component beyond 1e38 has no f32
    reciprocal to cull with, and with a negative t_min the hits at t ~ -1e-43 that such a ray makes count: the
    culling once dropped some of them.  tools/ray_soak.py runs the same over thousands of worlds.)"""
    import test_gpu_fuzz as fz

    kind = seed % 4
    built, _, _, _ = fz._random_world(seed, [(24, 90), (1 + seed % 5, 16), (1 + seed % 4, 11), (300, 900)][kind], planar=kind == 2)
    rays = _rays_nobody_would_send(20000, 77 + seed, built.shapes)
    # the case found by the soak, as it was found (5 shapes: spheres and a ground sphere)
    rays["origin"][0] = (0.7430995077367477, -11.200546938540313, 12.48976460943209)
    rays["direction"][0] = (1.460231731918684e-72, -8.41632469090724e+43, 7.040029465134746e+43)
    rays["time"][0] = 1e300
    scene = api.Scene(built.desc)
    try:
        for t_min, t_max in ((0.001, float("inf")), (-5.0, 40.0), (1e-9, 1e30)):
            ref, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            ref_list, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max, use_bvh=False)
            _compare(scene.hit(rays, t_min, t_max), ref, f"wild rays, fast, ({t_min}, {t_max})")
            _compare(scene.hit(rays, t_min, t_max, traversal=abi.TRAVERSE_REFERENCE), ref, f"wild rays, reference-order, ({t_min}, {t_max})")
            _compare(scene.hit(rays, t_min, t_max, use_bvh=False), ref_list, f"wild rays, list, ({t_min}, {t_max})")
    finally:
        scene.close()
