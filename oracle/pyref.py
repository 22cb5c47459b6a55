"""A SECOND restatement of the reference's render hot path, in plain Python, written from the Rust sources alone
(not from oracle.cpp) and as close to their text as Python allows.  Test infrastructure only: tests/test_oracle_pyref.py
runs it beside the C++ oracle on small cases -- closest hits, rendered sums, BVH builds -- so that a slip of the pen
in one restatement shows as a difference between the two.  It is slow (one Python call per arithmetic operation)
and nothing else may import it.

Python floats are IEEE doubles and `a * b + c` is two roundings, as in Rust; math.sqrt is correctly rounded; math.sin,
acos, atan2, tan are the host's libm, which the C++ oracle calls too.

What is NOT the reference's: the random numbers.  The reference draws from rand 0.7.3's SmallRng seeded from the
OS (renderer.rs:30,47), which no test pins; both restatements and the device draw from include/mirt_rng.h instead,
at the same call sites in the same order (PathRng below states the convention).  Hits carry the index of the shape
and the side of a cube in addition to the reference's fields, so that answers can be compared.

Scenes arrive as the flat arrays of include/mirt.h (mirt_scene_desc), shapes in the order the reference's BVH build
left them in.
"""
import math

MASK = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
INF = float("inf")

# ---------------------------------------------------------------- include/mirt_rng.h, restated


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def rng_stream(seed, pixel, sample):
    a = mix64((seed + GOLDEN * (pixel + 1)) & MASK)
    return mix64((a + 0xD1B54A32D192ED03 * (sample + 1)) & MASK)


def bits_to_1_2(bits):
    import struct

    return struct.unpack("<d", struct.pack("<Q", (bits & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000))[0]


class PathRng:
    """Stands in for `rng: &mut T` on the render path.  Draw k of stage s of one camera path; stage 0 is the
    camera (draws 0, 1: the jitter of renderer.rs:79-80; draw 2: the shutter time of camera.rs:73; draws 3, 4, ...:
    the pairs of camera.rs:78-88), stage b + 1 the material met at bounce b."""

    def __init__(self, stream):
        self.stream, self.stage, self.k = stream, 0, 0

    def set_stage(self, stage):
        self.stage, self.k = stage, 0

    def draw_at(self, k):
        ctr = (self.stage << 32) | k
        return bits_to_1_2(mix64((self.stream + GOLDEN * (ctr + 1)) & MASK)) - 1.0

    def random_double(self):  # util.rs:8-10: gen_range(0.0, 1.0)
        x = self.draw_at(self.k)
        self.k += 1
        return x


class SeqRng:
    """mirt_seq: the sequential generator of host-side set-up work (stands in for `rng: &mut T` in bvh.rs)."""

    def __init__(self, seed):
        self.state = mix64(seed ^ 0x6A09E667F3BCC909)

    def u64(self):
        self.state = (self.state + GOLDEN) & MASK
        return mix64(self.state)

    def gen_range_int(self, n):  # gen_range(0, n)
        return ((self.u64() >> 32) * n) >> 32


# ---------------------------------------------------------------- vec3.rs (f64 triples; every operator by its impl)


def v_add(a, b):  # vec3.rs:146-156
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def v_sub(a, b):  # vec3.rs:326-336
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def v_mul(a, b):  # vec3.rs:190-200
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def v_scale(a, k):  # vec3.rs:226-236 (&Vec3 * f64); f64 * Vec3 and Vec3 * f64 both end here: component * k
    return (a[0] * k, a[1] * k, a[2] * k)


def v_div(a, k):  # vec3.rs:290-300
    return (a[0] / k, a[1] / k, a[2] / k)


def v_neg(a):  # vec3.rs:370-379
    return (-a[0], -a[1], -a[2])


def dot(a, b):  # vec3.rs:88-90
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def square_length(a):  # vec3.rs:71-73
    return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]


def unit(a):  # vec3.rs:75-78
    k = 1.0 / math.sqrt(square_length(a))
    return (a[0] * k, a[1] * k, a[2] * k)


def f64_max(a, b):  # f64::max: the other operand if one is NaN
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def f64_min(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def div(a, b):
    """a / b with IEEE results where Python raises."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -INF if neg else INF


def sqrt(x):
    if x != x or x < 0.0:
        return float("nan")
    return math.sqrt(x) if x != INF else INF


def sin(x):
    if x != x or abs(x) == INF:
        return float("nan")
    return math.sin(x)


# ---------------------------------------------------------------- ray.rs


class Ray:
    def __init__(self, origin, direction, time):
        self.origin, self.direction, self.time = origin, direction, time

    def point_at(self, t):  # ray.rs:31-33: &origin + t * &direction
        return v_add(self.origin, v_scale(self.direction, t))


# ---------------------------------------------------------------- aabb.rs


class AABB:
    def __init__(self, mn, mx):
        self.min, self.max = mn, mx

    def hit(self, ray, t_min, t_max):  # aabb.rs:24-51
        min_t, max_t = t_min, t_max
        for c in range(3):
            direction_component = div(1.0, ray.direction[c])
            t0 = (self.min[c] - ray.origin[c]) * direction_component
            t1 = (self.max[c] - ray.origin[c]) * direction_component
            if direction_component < 0.0:
                t0, t1 = t1, t0
            min_t = f64_max(t0, min_t)
            max_t = f64_min(t1, max_t)
            if max_t <= min_t:
                return False
        return True

    def surrounding_box(self, other):  # aabb.rs:53-71
        small = tuple(f64_min(self.min[c], other.min[c]) for c in range(3))
        big = tuple(f64_max(self.max[c], other.max[c]) for c in range(3))
        return AABB(small, big)


# ---------------------------------------------------------------- hittable.rs

SPHERE, XY_RECT, XZ_RECT, YZ_RECT, CUBE, MOVING_SPHERE = range(6)  # MIRT_SHAPE_* of include/mirt.h


class Hit:
    def __init__(self, t, u, v, p, ray, outward_normal, material, prim, side=0):  # hittable.rs:27-52
        self.front_facing = dot(ray.direction, outward_normal) < 0.0
        self.normal = outward_normal if self.front_facing else v_neg(outward_normal)
        self.t, self.u, self.v, self.p, self.material = t, u, v, p, material
        self.prim, self.side = prim, side


def sphere_uv(p):  # hittable.rs:732-749
    theta = math.acos(-p[1]) if -1.0 <= -p[1] <= 1.0 else float("nan")
    phi = math.atan2(-p[2], p[0]) + math.pi
    return phi / (2.0 * math.pi), theta / math.pi


def sphere_hit_at(center, radius, material, ray, t_min, t_max, prim):  # hittable.rs:105-137 and 488-521
    oc = v_sub(ray.origin, center)
    a = dot(ray.direction, ray.direction)
    b = dot(oc, ray.direction)
    c = dot(oc, oc) - radius * radius
    discriminant = b * b - a * c
    if discriminant < 0.0:
        return None
    discriminant = sqrt(discriminant)
    f = div(-b - discriminant, a)
    g = div(-b + discriminant, a)
    if f < t_max and f > t_min:
        t = f
    elif g < t_max and g > t_min:
        t = g
    else:
        return None
    p = ray.point_at(t)
    outward_normal = tuple(div(x, radius) for x in v_sub(p, center))
    u, v = sphere_uv(outward_normal) if all(x == x for x in outward_normal) else (float("nan"), float("nan"))
    return Hit(t, u, v, p, ray, outward_normal, material, prim)


def rect_hit(kind, a0, a1, b0, b1, k, material, ray, t_min, t_max, prim, side=0):
    """XyRect / XzRect / YzRect::hit, hittable.rs:185-207, 249-271, 313-335: K the axis of the plane, A and B the
    other two in the rect's own order."""
    K, A, B = {XY_RECT: (2, 0, 1), XZ_RECT: (1, 0, 2), YZ_RECT: (0, 1, 2)}[kind]
    t = div(k - ray.origin[K], ray.direction[K])
    if t < t_min or t > t_max:
        return None
    a = ray.origin[A] + t * ray.direction[A]
    b = ray.origin[B] + t * ray.direction[B]
    if a < a0 or a > a1 or b < b0 or b > b1:
        return None
    u = div(a - a0, a1 - a0)
    v = div(b - b0, b1 - b0)
    p = ray.point_at(t)
    outward_normal = {XY_RECT: (0.0, 0.0, 1.0), XZ_RECT: (0.0, 1.0, 0.0), YZ_RECT: (1.0, 0.0, 0.0)}[kind]
    return Hit(t, u, v, p, ray, outward_normal, material, prim, side)


def rect_box(kind, a0, a1, b0, b1, k):  # hittable.rs:209-214, 273-278, 337-342
    if kind == XY_RECT:
        return AABB((a0, b0, k - 0.0001), (a1, b1, k + 0.0001))
    if kind == XZ_RECT:
        return AABB((a0, k - 0.0001, b0), (a1, k + 0.0001, b1))
    return AABB((k - 0.0001, a0, b0), (k + 0.0001, a1, b1))


def cube_sides(f):  # Cube::new, hittable.rs:404-412
    p0, p1 = (f[0], f[1], f[2]), (f[3], f[4], f[5])
    return [
        (XY_RECT, p0[0], p1[0], p0[1], p1[1], p0[2]),
        (XY_RECT, p0[0], p1[0], p0[1], p1[1], p1[2]),
        (XZ_RECT, p0[0], p1[0], p0[2], p1[2], p1[1]),
        (XZ_RECT, p0[0], p1[0], p0[2], p1[2], p0[1]),
        (YZ_RECT, p0[1], p1[1], p0[2], p1[2], p1[0]),
        (YZ_RECT, p0[1], p1[1], p0[2], p1[2], p0[0]),
    ]


def moving_center(f, time):  # Moving::center, hittable.rs:550-561; f = c0(3), radius, c1(3), time_start, time_end
    center_initial = (f[0], f[1], f[2])
    elapsed_time = time - f[7]
    movement_time = f[8] - f[7]
    distance = v_sub((f[4], f[5], f[6]), center_initial)
    return v_add(center_initial, v_scale(distance, div(elapsed_time, movement_time)))


def shape_hit(shape, index, ray, t_min, t_max):  # Shape::hit, hittable.rs:625-635
    kind, f, material = shape
    if kind == SPHERE:
        return sphere_hit_at((f[0], f[1], f[2]), f[3], material, ray, t_min, t_max, index)
    if kind in (XY_RECT, XZ_RECT, YZ_RECT):
        return rect_hit(kind, f[0], f[1], f[2], f[3], f[4], material, ray, t_min, t_max, index)
    if kind == CUBE:  # hittable.rs:432-445
        min_distance, nearest_hit = t_max, None
        for side, (rk, a0, a1, b0, b1, k) in enumerate(cube_sides(f)):
            hit = rect_hit(rk, a0, a1, b0, b1, k, material, ray, t_min, min_distance, index, side)
            if hit is not None:
                min_distance, nearest_hit = hit.t, hit
        return nearest_hit
    # Moving<Sphere>: the centre is evaluated twice with the same arguments (hittable.rs:490, 514)
    return sphere_hit_at(moving_center(f, ray.time), f[3], material, ray, t_min, t_max, index)


def shape_box(shape, time_start, time_end):  # Shape::bounding_box, hittable.rs:637-664
    kind, f, _ = shape
    if kind == SPHERE:  # hittable.rs:139-144
        c, r = (f[0], f[1], f[2]), f[3]
        return AABB(v_sub(c, (r, r, r)), v_add(c, (r, r, r)))
    if kind in (XY_RECT, XZ_RECT, YZ_RECT):
        return rect_box(kind, f[0], f[1], f[2], f[3], f[4])
    if kind == CUBE:  # hittable.rs:447-465
        sides = cube_sides(f)
        aabb = rect_box(*sides[0])
        for s in sides[1:]:
            aabb = aabb.surrounding_box(rect_box(*s))
        return aabb
    # Moving<Sphere>, hittable.rs:523-547
    r = f[3]
    center_initial, center_final = moving_center(f, time_start), moving_center(f, time_end)
    box_of = lambda c: AABB(v_sub(c, (r, r, r)), v_add(c, (r, r, r)))  # noqa: E731
    if center_initial == center_final:
        return box_of(center_initial)
    return box_of(center_initial).surrounding_box(box_of(center_final))


def list_hit(shapes, ray, t_min, t_max):  # hittable.rs:698-710
    min_distance, nearest_hit = t_max, None
    for index, shape in enumerate(shapes):
        hit = shape_hit(shape, index, ray, t_min, min_distance)
        if hit is not None:
            min_distance, nearest_hit = hit.t, hit
    return nearest_hit


def list_box(shapes, time_start, time_end):  # hittable.rs:712-730
    aabb = shape_box(shapes[0], time_start, time_end)
    for shape in shapes[1:]:
        aabb = aabb.surrounding_box(shape_box(shape, time_start, time_end))
    return aabb


# ---------------------------------------------------------------- bvh.rs


def indexed_hit(world, member, ray, t_min, t_max):  # bvh.rs:37-80
    if member["leaf"]:
        return shape_hit(world.shapes[member["shape"]], member["shape"], ray, t_min, t_max)
    if not member["bounds"].hit(ray, t_min, t_max):
        return None
    left_hit = indexed_hit(world, world.nodes[member["left"]], ray, t_min, t_max)
    right_hit = indexed_hit(world, world.nodes[member["right"]], ray, t_min, t_max)
    if left_hit is not None and right_hit is not None:
        return left_hit if left_hit.t < right_hit.t else right_hit
    return left_hit if left_hit is not None else right_hit


def build_tree(rng, shapes, time_initial, time_final):  # bvh.rs:112-135; sorts `shapes` in place
    nodes = []
    root = build_tree_internal(rng, shapes, 0, len(shapes), nodes, 0, len(shapes), time_initial, time_final)
    nodes.append(root)
    return nodes


def _sort_slice(shapes, lo, hi, axis, time_initial, time_final):
    """bvh.rs:147-166: slice::sort_by is a stable merge sort that only asks whether the comparison says Less, and the
    closure says Less exactly when a.min[axis] - b.min[axis] < 0.0: an insertion that moves an element left past
    every element it is less than, and no further, is that order (for keys that are not NaN)."""
    part = shapes[lo:hi]
    keys = [shape_box(s, time_initial, time_final).min[axis] for s in part]
    order = []
    for i in range(len(part)):
        pos = len(order)
        while pos > 0 and keys[i] - keys[order[pos - 1]] < 0.0:
            pos -= 1
        order.insert(pos, i)
    shapes[lo:hi] = [part[i] for i in order]


def build_tree_internal(rng, shapes, lo, hi, nodes, offset, n, time_initial, time_final):  # bvh.rs:137-237
    axis = rng.gen_range_int(3)
    _sort_slice(shapes, lo, hi, axis, time_initial, time_final)
    if n == 1:
        return {"leaf": True, "shape": offset}
    if n == 2:
        bounds = shape_box(shapes[lo], time_initial, time_final).surrounding_box(shape_box(shapes[lo + 1], time_initial, time_final))
        left_index = len(nodes)
        nodes.append({"leaf": True, "shape": offset})
        right_index = len(nodes)
        nodes.append({"leaf": True, "shape": offset + 1})
        return {"leaf": False, "left": left_index, "right": right_index, "bounds": bounds}
    halving_point = n // 2
    left = build_tree_internal(rng, shapes, lo, lo + halving_point, nodes, offset, halving_point, time_initial, time_final)
    left_index = len(nodes)
    nodes.append(left)
    right = build_tree_internal(rng, shapes, lo + halving_point, hi, nodes, offset + halving_point, n - halving_point, time_initial, time_final)
    right_index = len(nodes)
    nodes.append(right)
    bounds = list_box(shapes[lo:hi], time_initial, time_final)
    return {"leaf": False, "left": left_index, "right": right_index, "bounds": bounds}


# ---------------------------------------------------------------- perlin.rs, texture.rs

TEX_CONSTANT, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = range(4)
MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_METAL, MAT_DIFFUSE_LIGHT = range(4)


def as_i32(x):  # `as i32`: truncating, saturating, NaN -> 0
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def as_u32(x):
    if x != x or x <= 0.0:
        return 0
    if x >= 4294967295.0:
        return 4294967295
    return int(x)


def wrap_i32(x):  # i32 addition as release Rust does it
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def perlin_noise(pl, p):  # perlin.rs:52-85
    fl = [math.floor(x) if abs(x) != INF and x == x else x for x in p]
    u, v, w = p[0] - fl[0], p[1] - fl[1], p[2] - fl[2]
    i, j, k = as_i32(fl[0]), as_i32(fl[1]), as_i32(fl[2])
    sample = [[[None, None], [None, None]], [[None, None], [None, None]]]
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                ii = wrap_i32(i + di) & 255
                jj = wrap_i32(j + dj) & 255
                kk = wrap_i32(k + dk) & 255
                index = pl["perm_x"][ii] ^ pl["perm_y"][jj] ^ pl["perm_z"][kk]
                sample[di][dj][dk] = pl["vecs"][index]
    return trilinear_interp(sample, u, v, w)


def trilinear_interp(sample, u, v, w):  # perlin.rs:100-122
    uu = u * u * (3.0 - 2.0 * u)
    vv = v * v * (3.0 - 2.0 * v)
    ww = w * w * (3.0 - 2.0 * w)
    accumulator = 0.0
    for i in range(2):
        for j in range(2):
            for k in range(2):
                weight = (u - float(i), v - float(j), w - float(k))
                color = dot(weight, sample[i][j][k])
                fi = float(i) * uu + (1.0 - float(i)) * (1.0 - uu)
                fj = float(j) * vv + (1.0 - float(j)) * (1.0 - vv)
                fk = float(k) * ww + (1.0 - float(k)) * (1.0 - ww)
                accumulator += color * fi * fj * fk
    return accumulator


def turbulence(pl, p, depth):  # perlin.rs:38-50
    current_p, weight, accumulator = p, 1.0, 0.0
    for _ in range(depth):
        accumulator += weight * perlin_noise(pl, current_p)
        weight *= 0.5
        current_p = v_scale(current_p, 2.0)
    return abs(accumulator)


def perlin_new(seed):
    """Perlin::new, perlin.rs:20-36, 88-98, with draws from mirt_seq(seed): 256 x Vec3::random(rng, -1.0, 1.0).unit()
    (vec3.rs:27-33: gen_range per component, x first), then three permutations, each (0..256).rev() shuffled as
    rand 0.7.3's SliceRandom::shuffle does it: for i in (1..len).rev() { swap(i, gen_range(0, i + 1)) }."""
    rng = SeqRng(seed)

    def gen_range(lower, upper):  # mirt_seq_range
        return lower + (upper - lower) * (float(rng.u64() >> 11) * 2.0 ** -53)

    vecs = []
    for _ in range(256):
        x = gen_range(-1.0, 1.0)
        y = gen_range(-1.0, 1.0)
        z = gen_range(-1.0, 1.0)
        vecs.append(unit((x, y, z)))

    def generate_perm():
        vec = list(range(255, -1, -1))
        for i in range(255, 0, -1):
            j = rng.gen_range_int(i + 1)
            vec[i], vec[j] = vec[j], vec[i]
        return vec

    return {"vecs": vecs, "perm_x": generate_perm(), "perm_y": generate_perm(), "perm_z": generate_perm()}


def fclamp(x, lo, hi):  # texture.rs:195-205
    if x < lo:
        x = lo
    if x > hi:
        x = hi
    return x


def texture_value(world, tex, u, v, p):  # texture.rs:184-193
    t = world.textures[tex]
    if t["kind"] == TEX_CONSTANT:
        return t["color"]
    if t["kind"] == TEX_CHECKER:  # texture.rs:45-54
        sines = sin(10.0 * p[0]) * sin(10.0 * p[1]) * sin(10.0 * p[2])
        return texture_value(world, t["odd"] if sines < 0.0 else t["even"], u, v, p)
    if t["kind"] == TEX_NOISE:  # texture.rs:82-87
        turb = t["scale"] * p[2] + 10.0 * turbulence(world.perlins[t["perlin"]], p, 7)
        return v_scale(v_scale((1.0, 1.0, 1.0), 0.5), 1.0 + sin(turb))
    im = world.images[t["image"]]  # texture.rs:106-122
    height, width = im.shape[0], im.shape[1]
    u = fclamp(u, 0.0, 1.0)
    v = 1.0 - fclamp(v, 0.0, 1.0)
    i = min(as_u32(u * float(width)), width - 1)
    j = min(as_u32(v * float(height)), height - 1)
    color_scale = 1.0 / 255.0
    pixel = im[j, i]
    return v_scale((float(pixel[0]), float(pixel[1]), float(pixel[2])), color_scale)


# ---------------------------------------------------------------- material.rs


def random_in_unit_sphere(rng):  # material.rs:211-225
    def gen_p():
        a, b, c = rng.random_double(), rng.random_double(), rng.random_double()
        return v_sub(v_scale((a, b, c), 2.0), (1.0, 1.0, 1.0))

    p = gen_p()
    while square_length(p) >= 1.0:
        p = gen_p()
    return p


def reflect(v, n):  # material.rs:227-229
    return v_sub(v, v_scale(n, 2.0 * dot(v, n)))


def refract(uv, n, etai_over_etat):  # material.rs:231-237
    cos_theta = f64_min(dot(v_neg(uv), n), 1.0)
    perpendicular = v_scale(v_add(uv, v_scale(n, cos_theta)), etai_over_etat)
    parallel = v_scale(n, -sqrt(abs(1.0 - square_length(perpendicular))))
    return v_add(perpendicular, parallel)


def schlick(cosine, ref_idx):  # material.rs:239-243
    r0 = div(1.0 - ref_idx, 1.0 + ref_idx)
    r0 *= r0
    x = 1.0 - cosine
    x2 = x * x  # powi(5): x * (x^2)^2, as compiler-rt's __powidf2 and LLVM's expansion multiply
    return r0 + (1.0 - r0) * (x * (x2 * x2))


def material_emit(world, m, u, v, p):  # material.rs:201-208, 13-15, 149-151
    if m["kind"] == MAT_DIFFUSE_LIGHT:
        return texture_value(world, m["texture"], u, v, p)
    return (0.0, 0.0, 0.0)


def material_scatter(world, m, rng, ray, hit):  # material.rs:191-199 -> (scattered ray, attenuation) or None
    if m["kind"] == MAT_LAMBERTIAN:  # material.rs:51-59
        target = v_add(v_add(hit.p, hit.normal), random_in_unit_sphere(rng))
        scattered = Ray(hit.p, v_sub(target, hit.p), ray.time)
        return scattered, texture_value(world, m["texture"], hit.u, hit.v, hit.p)
    if m["kind"] == MAT_METAL:  # material.rs:74-91 (the stored fuzziness: Metal::new has clamped it already)
        reflected = reflect(unit(ray.direction), hit.normal)
        scattered = Ray(hit.p, v_add(reflected, v_scale(random_in_unit_sphere(rng), m["param"])), ray.time)
        if dot(scattered.direction, hit.normal) > 0.0:
            return scattered, m["albedo"]
        return None
    if m["kind"] == MAT_DIELECTRIC:  # material.rs:104-131
        refraction_ratio = div(1.0, m["param"]) if hit.front_facing else m["param"]
        unit_direction = unit(ray.direction)
        cos_theta = f64_min(dot(v_neg(unit_direction), hit.normal), 1.0)
        sin_theta = sqrt(1.0 - cos_theta * cos_theta)
        cannot_refract = refraction_ratio * sin_theta > 1.0
        if cannot_refract or schlick(cos_theta, refraction_ratio) > rng.random_double():
            direction = reflect(unit_direction, hit.normal)
        else:
            direction = refract(unit_direction, hit.normal, refraction_ratio)
        return Ray(hit.p, direction, ray.time), (1.0, 1.0, 1.0)
    return None  # DiffuseLight, material.rs:144-147


# ---------------------------------------------------------------- camera.rs


def cross(a, b):  # vec3.rs:92-98
    return (a[1] * b[2] - a[2] * b[1], -(a[0] * b[2] - a[2] * b[0]), a[0] * b[1] - a[1] * b[0])


def camera_new(look_from, look_at, v_up, vertical_fov, aspect, aperture, focus_distance, time_start, time_end):  # camera.rs:21-63
    lens_radius = aperture / 2.0
    theta = vertical_fov * math.pi / 180.0
    half_height = math.tan(theta / 2.0)
    half_width = aspect * half_height
    w = unit(v_sub(look_from, look_at))
    u = unit(cross(v_up, w))
    v = cross(w, u)
    origin = look_from
    lower_left_corner = v_sub(v_sub(v_sub(origin, v_scale(u, half_width * focus_distance)), v_scale(v, half_height * focus_distance)), v_scale(w, focus_distance))
    horizontal = v_scale(u, 2.0 * half_width * focus_distance)
    vertical = v_scale(v, 2.0 * half_height * focus_distance)
    return {"origin": origin, "lower_left_corner": lower_left_corner, "horizontal": horizontal, "vertical": vertical, "u": u, "v": v, "w": w,
            "lens_radius": lens_radius, "time_start": time_start, "time_end": time_end}


def random_in_unit_disk(rng):  # camera.rs:78-88
    def gen_p():
        a, b = rng.random_double(), rng.random_double()
        return v_sub(v_scale((a, b, 0.0), 2.0), (1.0, 1.0, 0.0))

    p = gen_p()
    while dot(p, p) >= 1.0:
        p = gen_p()
    return p


def camera_ray(cam, rng, s, t):  # camera.rs:65-76
    # (the sampler gives the shutter time draw 2 of the stage, before the pairs of the lens: PathRng)
    shutter = rng.random_double()
    rd = v_scale(random_in_unit_disk(rng), cam["lens_radius"])
    offset = v_add(v_scale(cam["u"], rd[0]), v_scale(cam["v"], rd[1]))
    direction = v_sub(v_sub(v_add(v_add(cam["lower_left_corner"], v_scale(cam["horizontal"], s)), v_scale(cam["vertical"], t)), cam["origin"]), offset)
    time = cam["time_start"] + shutter * (cam["time_end"] - cam["time_start"])
    return Ray(v_add(cam["origin"], offset), direction, time)


# ---------------------------------------------------------------- renderer.rs


class World:
    """The flat arrays of a mirt_scene_desc (ctypes) as Python values."""

    def __init__(self, desc, images=()):
        d = desc.contents if hasattr(desc, "contents") else desc
        self.shapes = [(int(d.shapes[i].kind), [float(x) for x in d.shapes[i].f], int(d.shapes[i].material)) for i in range(d.n_shapes)]
        self.materials = [{"kind": int(d.materials[i].kind), "texture": int(d.materials[i].texture),
                           "albedo": tuple(float(x) for x in d.materials[i].albedo), "param": float(d.materials[i].param)}
                          for i in range(d.n_materials)]
        self.textures = [{"kind": int(d.textures[i].kind), "even": int(d.textures[i].even), "odd": int(d.textures[i].odd),
                          "perlin": int(d.textures[i].perlin), "image": int(d.textures[i].image),
                          "color": tuple(float(x) for x in d.textures[i].color), "scale": float(d.textures[i].scale)}
                         for i in range(d.n_textures)]
        self.perlins = [{"vecs": [tuple(float(x) for x in d.perlins[i].vecs[k]) for k in range(256)],
                         "perm_x": list(d.perlins[i].perm_x), "perm_y": list(d.perlins[i].perm_y), "perm_z": list(d.perlins[i].perm_z)}
                        for i in range(d.n_perlins)]
        self.images = list(images)  # numpy (H, W, 3) uint8 arrays, in the order of the description's image table
        self.nodes = []
        for i in range(d.n_nodes if d.nodes else 0):
            m = d.nodes[i]
            if m.is_leaf:
                self.nodes.append({"leaf": True, "shape": int(m.shape)})
            else:
                self.nodes.append({"leaf": False, "left": int(m.left), "right": int(m.right),
                                   "bounds": AABB(tuple(float(x) for x in m.bmin), tuple(float(x) for x in m.bmax))})
        self.segments = 0

    def hit(self, ray, t_min, t_max, use_bvh=True):  # renderer.rs:21-25: the BVH, or the plain list of its shapes
        if use_bvh:
            return indexed_hit(self, self.nodes[-1], ray, t_min, t_max)  # bvh.rs:33-35, 84-86: the root is the last member
        return list_hit(self.shapes, ray, t_min, t_max)


def color(world, rng, ray, background, max_depth, bounce=0, use_bvh=True):  # renderer.rs:92-115
    if max_depth == 0:
        return (0.0, 0.0, 0.0)
    world.segments += 1
    hit = world.hit(ray, 0.001, INF, use_bvh)
    if hit is not None:
        m = world.materials[hit.material]
        emitted = material_emit(world, m, hit.u, hit.v, hit.p)
        rng.set_stage(bounce + 1)
        scatter = material_scatter(world, m, rng, ray, hit)
        if scatter is not None:
            scattered, attenuation = scatter
            return v_mul(v_add(emitted, attenuation), color(world, rng, scattered, background, max_depth - 1, bounce + 1, use_bvh))
        return emitted
    return background


def render_sum(world, cam, background, screen_width, screen_height, antialias_iterations, idx, max_depth=50, seed=0x5EED0002, use_bvh=True):
    """renderer.rs:64-84: the sum over the samples of pixel idx (before the division of renderer.rs:85)."""
    i = idx % screen_width
    j = screen_height - 1 - idx // screen_width
    pixel = (0.0, 0.0, 0.0)
    for s in range(antialias_iterations):
        rng = PathRng(rng_stream(seed, idx, s))
        u = (float(i) + rng.random_double()) / float(screen_width)
        v = (float(j) + rng.random_double()) / float(screen_height)
        ray = camera_ray(cam, rng, u, v)
        pixel = v_add(pixel, color(world, rng, ray, background, max_depth, 0, use_bvh))
    return pixel


def as_u8(x):  # `as u8`, renderer.rs:42
    if x != x or x <= 0.0:
        return 0
    if x >= 255.0:
        return 255
    return int(x)


def resolve(pixel_sum, antialias_iterations):  # renderer.rs:85-89
    pixel = v_div(pixel_sum, float(antialias_iterations))
    pixel = (sqrt(pixel[0]), sqrt(pixel[1]), sqrt(pixel[2]))
    pixel = v_scale(pixel, 255.99)
    return as_u8(pixel[0]), as_u8(pixel[1]), as_u8(pixel[2])


def camera_of(c):
    """A mirt_camera (ctypes) as a dict of tuples."""
    vec = lambda a: tuple(float(x) for x in a)  # noqa: E731
    return {"origin": vec(c.origin), "lower_left_corner": vec(c.lower_left_corner), "horizontal": vec(c.horizontal),
            "vertical": vec(c.vertical), "u": vec(c.u), "v": vec(c.v), "w": vec(c.w), "lens_radius": float(c.lens_radius),
            "time_start": float(c.time_start), "time_end": float(c.time_end)}
