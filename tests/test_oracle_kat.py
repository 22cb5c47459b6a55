"""Known answers for the CPU oracle (oracle/oracle.cpp), derived by hand from the
reference's formulas.  The reference ships no tests or golden vectors; the one
known-answer table it holds is the comment of sphere_uv (src/hittable.rs:738-740),
checked first.  Everything the GPU parity tests compare against goes through the
functions pinned here."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib
import scene_util as su
from ray_tracer_amd import _abi as abi

INF = float("inf")


@pytest.fixture(scope="module")
def L():
    return oracle_lib.lib()


def d3(*v):
    return (C.c_double * 3)(*[float(x) for x in v])


# ------------------------------------------------------------- reference's own table
@pytest.mark.parametrize("p,uv", [
    ((1, 0, 0), (0.50, 0.50)), ((-1, 0, 0), (0.00, 0.50)),
    ((0, 1, 0), (0.50, 1.00)), ((0, -1, 0), (0.50, 0.00)),
    ((0, 0, 1), (0.25, 0.50)), ((0, 0, -1), (0.75, 0.50)),
])
def test_sphere_uv_table_of_the_reference(L, p, uv):
    u, v = C.c_double(), C.c_double()
    L.oracle_sphere_uv(d3(*p), C.byref(u), C.byref(v))
    assert u.value == pytest.approx(uv[0], abs=1e-15)
    assert v.value == pytest.approx(uv[1], abs=1e-15)


# --------------------------------------------------------------------------- shapes
def test_sphere_from_outside_and_from_inside():
    b = su.Built([su.sphere(0, 0, -1, 0.5)], bvh=False)
    h = su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1)), use_bvh=False)
    assert h["t"] == 0.5 and h["normal"] == [0.0, 0.0, 1.0] and h["front"]
    assert h["p"] == [0.0, 0.0, -0.5]
    # from the centre: first root -0.5 is below t_min, second root 0.5; outward normal (0,0,-1)
    # points along the ray, so the stored normal is flipped and the hit is a back face
    h = su.hit_detail(b, su.ray((0, 0, -1), (0, 0, -1)), use_bvh=False)
    assert h["t"] == 0.5 and h["normal"] == [0.0, 0.0, 1.0] and not h["front"]
    # strict bounds (hittable.rs:118,124): a root equal to t_max is no hit
    assert su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1)), t_max=0.5, use_bvh=False) is None
    assert su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1)), t_min=0.5, t_max=1.5, use_bvh=False) is None
    assert su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1)), t_min=0.5, t_max=1.5000001, use_bvh=False)["t"] == 1.5


def test_negative_radius_flips_the_normal():
    # the hollow glass sphere of the Basic scene (world.rs:39): (p - c) / r with r < 0
    b = su.Built([su.sphere(0, 0, -1, -0.5)], bvh=False)
    h = su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1)), use_bvh=False)
    assert h["t"] == 0.5 and not h["front"] and h["normal"] == [0.0, 0.0, 1.0]


def test_xy_rect_of_the_cornell_box():
    r = su.shape(abi.SHAPE_XY_RECT, 0, 0, 555, 0, 555, 555)
    b = su.Built([r], bvh=False)
    h = su.hit_detail(b, su.ray((278, 278, -800), (0, 0, 1)), use_bvh=False)
    assert h["t"] == 1355.0
    assert h["u"] == 278 / 555 and h["v"] == 278 / 555
    assert h["normal"] == [0.0, 0.0, -1.0] and not h["front"]  # outward normal (0,0,1) points along the ray
    assert su.hit_detail(b, su.ray((600, 278, -800), (0, 0, 1)), use_bvh=False) is None
    # parallel ray: t = (k - oz) / 0 = inf, rejected by the range test
    assert su.hit_detail(b, su.ray((278, 278, -800), (1, 0, 0)), use_bvh=False) is None


def test_xz_and_yz_rects():
    xz = su.Built([su.shape(abi.SHAPE_XZ_RECT, 0, -1, 1, -2, 2, 3.0)], bvh=False)
    h = su.hit_detail(xz, su.ray((0.5, 0, 1), (0, 1, 0)), use_bvh=False)
    assert h["t"] == 3.0 and h["u"] == 0.75 and h["v"] == 0.75 and h["normal"] == [0.0, -1.0, 0.0]
    yz = su.Built([su.shape(abi.SHAPE_YZ_RECT, 0, -1, 1, -2, 2, -4.0)], bvh=False)
    h = su.hit_detail(yz, su.ray((0, 0, 0), (-2, 0, 0)), use_bvh=False)
    assert h["t"] == 2.0 and h["u"] == 0.5 and h["v"] == 0.5 and h["normal"] == [1.0, 0.0, 0.0] and h["front"]


def test_cube_sides_and_edge_tie():
    cube = su.Built([su.shape(abi.SHAPE_CUBE, 0, 0, 0, 0, 1, 1, 1)], bvh=False)
    # straight at the z = 0 face: side 0 of Cube::new (hittable.rs:405-412)
    h = su.hit_detail(cube, su.ray((0.5, 0.5, -1), (0, 0, 1)), use_bvh=False)
    assert h["t"] == 1.0 and h["side"] == 0
    # from above at the y = 1 face: side 2
    h = su.hit_detail(cube, su.ray((0.5, 3, 0.5), (0, -1, 0)), use_bvh=False)
    assert h["t"] == 2.0 and h["side"] == 2
    # through the edge x = 0, z = 0 along (1,0,1): sides 0 (z = 0) and 5 (x = 0) both give t = 1;
    # the loop passes the running minimum as t_max and a rect accepts t == t_max, so the later side wins
    h = su.hit_detail(cube, su.ray((-1, 0.5, -1), (1, 0, 1)), use_bvh=False)
    assert h["t"] == 1.0 and h["side"] == 5


def test_moving_sphere_is_at_the_midpoint_at_half_time():
    b = su.Built([su.moving_sphere((0, 0, -2), (0, 2, -2), 0.5, 0.0, 1.0)], bvh=False)
    assert su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1), time=0.0), use_bvh=False)["t"] == 1.5
    assert su.hit_detail(b, su.ray((0, 0, 0), (0, 0, -1), time=0.5), use_bvh=False) is None  # centre (0,1,-2)
    h = su.hit_detail(b, su.ray((0, 1, 0), (0, 0, -1), time=0.5), use_bvh=False)
    assert h["t"] == 1.5 and h["normal"] == [0.0, 0.0, 1.0]


def test_bounding_boxes(L):
    lo, hi = d3(0, 0, 0), d3(0, 0, 0)
    L.oracle_shape_bbox(C.byref(su.sphere(1, 2, 3, 0.5)), 0.0, 1.0, lo, hi)
    assert list(lo) == [0.5, 1.5, 2.5] and list(hi) == [1.5, 2.5, 3.5]
    # negative radius: the box comes out inverted and is carried as it is (hittable.rs:139-144)
    L.oracle_shape_bbox(C.byref(su.sphere(1, 2, 3, -0.5)), 0.0, 1.0, lo, hi)
    assert list(lo) == [1.5, 2.5, 3.5] and list(hi) == [0.5, 1.5, 2.5]
    # rects are padded by 0.0001 on their flat axis (hittable.rs:205-210)
    L.oracle_shape_bbox(C.byref(su.shape(abi.SHAPE_XY_RECT, 0, 0, 555, 0, 555, 555)), 0.0, 1.0, lo, hi)
    assert list(lo) == [0.0, 0.0, 555 - 0.0001] and list(hi) == [555.0, 555.0, 555 + 0.0001]
    L.oracle_shape_bbox(C.byref(su.moving_sphere((0, 0, 0), (2, 0, 0), 1.0, 0.0, 1.0)), 0.0, 1.0, lo, hi)
    assert list(lo) == [-1.0, -1.0, -1.0] and list(hi) == [3.0, 1.0, 1.0]


# ----------------------------------------------------------------------------- AABB
def test_aabb_slab_test_including_the_infinity_and_nan_paths(L):
    lo, hi = d3(-1, -1, -1), d3(1, 1, 1)

    def hit(o, d, t_min=0.001, t_max=INF):
        return L.oracle_aabb_hit(lo, hi, C.byref(su.ray(o, d)), t_min, t_max)

    assert hit((0, 0, -5), (0, 0, 1)) == 1
    assert hit((0, 0, -5), (0, 0, -1)) == 0
    assert hit((0, 0, -5), (0, 0, 1), t_max=3.9) == 0    # box starts at t = 4
    assert hit((0, 0, -5), (0, 0, 1), t_min=6.0) == 0    # box ends at t = 6; max_t <= min_t
    # d.x = 0: 1/0 = +inf, t0 = -inf and t1 = +inf inside the slab
    assert hit((0.5, 0, -5), (0, 0, 1)) == 1
    # outside the slab both are +inf (or both -inf): empty interval
    assert hit((2.0, 0, -5), (0, 0, 1)) == 0
    assert hit((-2.0, 0, -5), (0, 0, 1)) == 0
    # on the face, 0 * inf = NaN: f64::max / f64::min drop the NaN and the slab does not constrain
    assert hit((1.0, 0, -5), (0, 0, 1)) == 1
    # negative direction swaps t0 and t1 (aabb.rs:38-40)
    assert hit((5, 5, 5), (-1, -1, -1)) == 1


# ------------------------------------------------------------------------ materials
def test_schlick_reflect_refract(L):
    assert L.oracle_schlick(1.0, 1.5) == pytest.approx(0.04, abs=1e-16)
    assert L.oracle_schlick(0.0, 1.5) == 1.0
    out = d3(0, 0, 0)
    L.oracle_reflect(d3(1, -1, 0), d3(0, 1, 0), out)
    assert list(out) == [1.0, 1.0, 0.0]
    # normal incidence: the direction is unchanged whatever the ratio
    L.oracle_refract(d3(0, -1, 0), d3(0, 1, 0), 1.0 / 1.5, out)
    assert list(out) == [0.0, -1.0, 0.0]
    # 45 degrees into glass: sin(theta') = sin(45 deg) / 1.5
    s = math.sqrt(0.5)
    L.oracle_refract(d3(s, -s, 0), d3(0, 1, 0), 1.0 / 1.5, out)
    assert out[0] == pytest.approx(s / 1.5, rel=1e-15)
    assert out[0] ** 2 + out[1] ** 2 == pytest.approx(1.0, rel=1e-15) and out[1] < 0


def _scatter(L, built, r, stream=12345, stage=1):
    out, emitted = (C.c_double * 10)(), d3(0, 0, 0)
    rc = L.oracle_scatter(C.byref(built.desc), C.byref(r), 0, stream, stage, out, emitted)
    return rc, list(out), list(emitted)


def test_dielectric_total_internal_reflection_and_metal_mirror(L):
    glass = su.Built([su.shape(abi.SHAPE_XZ_RECT, 0, -10, 10, -10, 10, 0.0)],
                     materials=[su.material(abi.MAT_DIELECTRIC, param=1.5)], bvh=False)
    # from below (inside the medium: the outward normal (0,1,0) points along the ray) at 60 degrees:
    # ratio * sin = 1.5 * 0.866 > 1, so the ray must reflect whatever the random draw is
    s, c = math.sin(math.radians(60)), math.cos(math.radians(60))
    for stream in (1, 2, 3, 4, 5):
        rc, out, _ = _scatter(L, glass, su.ray((0, -1, 0), (s, c, 0)), stream=stream)
        assert rc == 1
        assert out[3] == pytest.approx(s, rel=1e-15) and out[4] == pytest.approx(-c, rel=1e-15)
        assert out[7:10] == [1.0, 1.0, 1.0]
    mirror = su.Built([su.shape(abi.SHAPE_XZ_RECT, 0, -10, 10, -10, 10, 0.0)],
                      materials=[su.material(abi.MAT_METAL, albedo=(0.8, 0.6, 0.2), param=0.0)], bvh=False)
    rc, out, _ = _scatter(L, mirror, su.ray((0, 1, 0), (1, -1, 0)))
    assert rc == 1
    assert out[0:3] == [1.0, 0.0, 0.0]
    assert out[3:6] == pytest.approx([s2 := math.sqrt(0.5), s2, 0.0], rel=1e-15)
    assert out[7:10] == [0.8, 0.6, 0.2]


def test_diffuse_light_emits_and_does_not_scatter(L):
    light = su.Built([su.shape(abi.SHAPE_XZ_RECT, 0, -1, 1, -1, 1, 5.0)],
                     materials=[su.material(abi.MAT_DIFFUSE_LIGHT, texture=0)], textures=[su.constant(15, 15, 15)],
                     bvh=False)
    rc, _, emitted = _scatter(L, light, su.ray((0, 0, 0), (0, 1, 0)))
    assert rc == 0 and emitted == [15.0, 15.0, 15.0]


def test_lambertian_scatters_into_the_unit_sphere_around_p_plus_n(L):
    ground = su.Built([su.shape(abi.SHAPE_XZ_RECT, 0, -10, 10, -10, 10, 0.0)], textures=[su.constant(0.1, 0.2, 0.3)],
                      bvh=False)
    for stream in range(20):
        rc, out, emitted = _scatter(L, ground, su.ray((0, 1, 0), (0.3, -1, 0.1)), stream=stream)
        assert rc == 1 and emitted == [0.0, 0.0, 0.0]
        d = np.array(out[3:6]) - np.array([0.0, 1.0, 0.0])  # direction - normal = point in the unit sphere
        assert d @ d < 1.0
        assert out[7:10] == [0.1, 0.2, 0.3]
        assert out[0:3] == pytest.approx([0.3, 0.0, 0.1], abs=1e-15)


# ------------------------------------------------------------------------- textures
def test_checker_sign_and_image_texel(L):
    tex = [su.checker(1, 2), su.constant(1, 1, 1), su.constant(0, 0, 0)]
    b = su.Built([su.sphere(0, 0, 0, 1)], textures=tex, bvh=False)
    out = d3(0, 0, 0)
    q = math.pi / 20  # sin(10 q) = 1 on every axis: product +1 -> even
    L.oracle_texture_value(C.byref(b.desc), 0, 0.0, 0.0, d3(q, q, q), out)
    assert list(out) == [1.0, 1.0, 1.0]
    L.oracle_texture_value(C.byref(b.desc), 0, 0.0, 0.0, d3(-q, q, q), out)  # product -1 -> odd
    assert list(out) == [0.0, 0.0, 0.0]
    # image: u = v = 0.5 on a 4x2 image -> i = 2, j = (1 - 0.5) * 2 = 1 (texture.rs:106-122)
    img = np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3) * 10
    b = su.Built([su.sphere(0, 0, 0, 1)], textures=[su.image_texture(0)], images=[img], bvh=False)
    L.oracle_texture_value(C.byref(b.desc), 0, 0.5, 0.5, d3(0, 0, 0), out)
    assert list(out) == [float(x) * (1.0 / 255.0) for x in img[1, 2]]
    # u = 1, v = 0 clamp to the last column, last row
    L.oracle_texture_value(C.byref(b.desc), 0, 1.0, 0.0, d3(0, 0, 0), out)
    assert list(out) == [float(x) * (1.0 / 255.0) for x in img[1, 3]]
    L.oracle_texture_value(C.byref(b.desc), 0, 0.0, 1.0, d3(0, 0, 0), out)
    assert list(out) == [float(x) * (1.0 / 255.0) for x in img[0, 0]]


def test_perlin_noise_is_zero_on_the_lattice_and_bounded(L):
    pl = abi.Perlin()
    L.oracle_perlin_new(99, C.byref(pl))
    # every gradient is a unit vector (perlin.rs:88-98) and the permutations are permutations
    v = np.ctypeslib.as_array(pl.vecs).reshape(256, 3)
    assert np.allclose((v * v).sum(axis=1), 1.0, rtol=1e-14)
    for perm in (pl.perm_x, pl.perm_y, pl.perm_z):
        assert sorted(perm) == list(range(256))
    # at lattice points all eight weights (u - i, v - j, w - k) vanish for the corner with weight 1
    for p in ((0, 0, 0), (3, -2, 7), (255, 256, -257)):
        assert L.oracle_perlin_noise(C.byref(pl), d3(*p)) == 0.0
    rng = np.random.default_rng(1)
    vals = [L.oracle_perlin_noise(C.byref(pl), d3(*rng.uniform(-50, 50, 3))) for _ in range(2000)]
    assert max(abs(x) for x in vals) <= math.sqrt(3) / 2 + 1e-12 and np.std(vals) > 0.1
    # turbulence = |sum of octaves| (perlin.rs:38-50)
    p = d3(0.3, 1.7, -2.2)
    acc, w, q = 0.0, 1.0, np.array([0.3, 1.7, -2.2])
    for _ in range(7):
        acc += w * L.oracle_perlin_noise(C.byref(pl), d3(*q))
        w *= 0.5
        q = q * 2.0
    assert L.oracle_perlin_turbulence(C.byref(pl), p, 7) == abs(acc)


# ------------------------------------------------------------------------------ BVH
def test_bvh_build_shape_of_the_tree():
    shapes = [su.sphere(float(i), 0.0, 0.0, 0.25) for i in (3, 1, 4, 0, 5, 9, 2, 6)]
    b = su.Built(shapes, seed=3)
    n = len(shapes)
    nodes = list(b.nodes)
    assert len(nodes) == 2 * n - 1
    leaves = [m for m in nodes if m.is_leaf]
    assert sorted(m.shape for m in leaves) == list(range(n))
    root = nodes[-1]  # root last (bvh.rs:33-35)
    assert not root.is_leaf
    assert list(root.bmin) == [-0.25, -0.25, -0.25] and list(root.bmax) == [9.25, 0.25, 0.25]
    # children come before their parent (post-order)
    for i, m in enumerate(nodes):
        if not m.is_leaf:
            assert m.left < i and m.right < i


def test_coincident_spheres_tie_goes_to_the_larger_index_under_bvh_and_the_first_in_a_list():
    # two identical spheres: under the BVH left wins only if left.t < right.t (bvh.rs:70-74), so the
    # right (larger index) one is returned; the linear list keeps the first (strict t < closest)
    two = [su.sphere(0, 0, -1, 0.5, material=0), su.sphere(0, 0, -1, 0.5, material=0)]
    b = su.Built(two)
    r = su.ray((0, 0, 0), (0, 0, -1))
    assert su.hit_detail(b, r, use_bvh=True)["prim"] == 1
    assert su.hit_detail(b, r, use_bvh=False)["prim"] == 0


def test_bvh_and_linear_list_agree_on_random_rays():
    rng = np.random.default_rng(5)
    shapes = [su.sphere(*rng.uniform(-5, 5, 3), rng.uniform(0.2, 1.0)) for _ in range(40)]
    shapes += [su.shape(abi.SHAPE_CUBE, 0, -1, -1, -1, 0.5, 0.7, 0.9),
               su.shape(abi.SHAPE_XY_RECT, 0, -3, 3, -3, 3, 2.0),
               su.moving_sphere((2, 2, 2), (2, 3, 2), 0.6, 0.0, 1.0)]
    b = su.Built(shapes)
    rays = np.zeros(4000, dtype=oracle_lib.RAY_DTYPE)
    rays["origin"] = rng.uniform(-8, 8, (4000, 3))
    rays["direction"] = rng.normal(size=(4000, 3))
    rays["time"] = rng.uniform(0, 1, 4000)
    with_bvh, _ = oracle_lib.intersect(b.desc, rays, use_bvh=True)
    linear, _ = oracle_lib.intersect(b.desc, rays, use_bvh=False)
    assert np.count_nonzero(with_bvh["prim"] >= 0) > 500
    assert np.array_equal(with_bvh["prim"], linear["prim"])
    assert np.array_equal(with_bvh["t"], linear["t"])
    assert np.array_equal(with_bvh["side"], linear["side"])


# -------------------------------------------------------------------------- resolve
def test_resolve_is_mean_sqrt_scale_and_saturating_cast():
    # renderer.rs:85-89 and the `as u8` of renderer.rs:42
    sums = np.array([[0.0, 1.0, 4.0], [0.25, 100.0, -1.0], [np.nan, np.inf, 0.9999]], dtype=np.float64)
    out = oracle_lib.resolve(sums, 1)
    assert out.tolist() == [[0, 255, 255], [127, 255, 0], [0, 255, 255]]
    out = oracle_lib.resolve(sums * 4, 4)
    assert out.tolist() == [[0, 255, 255], [127, 255, 0], [0, 255, 255]]


def test_parallel_trace_and_intersect_equal_the_serial_ones(oracle):
    """oracle_trace_rays on many pixels and oracle_intersect on many rays use all cores; what they
    return is the serial loops' output, in the same order."""
    from ray_tracer_amd import configs

    cfg = configs.C1
    world = cfg.world()
    w, h = 96, 54
    cam = cfg.camera(w, h)
    pixels = np.arange(w * h, dtype=np.uint32)
    rays, hits = oracle.trace_rays(world.desc, cam, cfg.background, w, h, pixels, (0, 2))
    parts = [oracle.trace_rays(world.desc, cam, cfg.background, w, h, pixels[k:k + 500], (0, 2)) for k in range(0, w * h, 500)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]).view(np.uint8), rays.view(np.uint8))
    assert np.array_equal(np.concatenate([p[1] for p in parts]).view(np.uint8), hits.view(np.uint8))
    again, _ = oracle.intersect(world.desc, rays)  # >= 4096 rays: parallel
    small = np.concatenate([oracle.intersect(world.desc, rays[k:k + 1000])[0] for k in range(0, rays.shape[0], 1000)])
    assert np.array_equal(again.view(np.uint8), hits.view(np.uint8)) and np.array_equal(small.view(np.uint8), hits.view(np.uint8))
