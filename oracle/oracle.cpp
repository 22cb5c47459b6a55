// oracle.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A CPU restatement, in f64 and in the reference's own operation order, of the
// render hot path of mbStavola/ray_tracer.  It exists so that tests, the smoke
// check and the CPU-baseline leg of the benchmark have something to compare the
// HIP path against; nothing in the product library links, loads or calls it.
//
// PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden
// vectors or fixtures for this path (only the six sphere_uv pairs in the
// comment at src/hittable.rs:738-740, which tests/ checks against this file),
// it is written in Rust and no Rust toolchain exists in this environment, so
// this restatement could not be validated against a run of the reference.  It
// is pinned by hand-derived known-answer tests instead (tests/test_oracle_kat.py).
//
// Built with -ffp-contract=off: Rust never fuses a*b+c, so neither may we.
// Every function cites the reference lines it follows.  Random numbers come
// from the counter-based generator of include/mirt_rng.h (the reference's
// entropy-seeded SmallRng cannot be reproduced); the ORDER and NUMBER of draws
// at each call site follow the reference.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/mirt.h"
#include "../include/mirt_rng.h"

namespace {

// ---------------------------------------------------------------- vec3.rs
struct Vec3 {
    double e[3];
    Vec3() : e{0.0, 0.0, 0.0} {}
    Vec3(double a, double b, double c) : e{a, b, c} {}
    explicit Vec3(const double* p) : e{p[0], p[1], p[2]} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double operator[](int i) const { return e[i]; }
    // vec3.rs:71-73
    double square_length() const { return x() * x() + y() * y() + z() * z(); }
    // vec3.rs:66-68
    double length() const { return std::sqrt(square_length()); }
    // vec3.rs:75-78: multiply by the reciprocal, not divide
    Vec3 unit() const {
        double k = 1.0 / length();
        return Vec3(x() * k, y() * k, z() * k);
    }
    // vec3.rs:88-90
    double dot(const Vec3& o) const { return x() * o.x() + y() * o.y() + z() * o.z(); }
    // vec3.rs:92-98
    Vec3 cross(const Vec3& o) const {
        double cx = y() * o.z() - z() * o.y();
        double cy = -(x() * o.z() - z() * o.x());
        double cz = x() * o.y() - y() * o.x();
        return Vec3(cx, cy, cz);
    }
    bool operator==(const Vec3& o) const { return e[0] == o.e[0] && e[1] == o.e[1] && e[2] == o.e[2]; }
};
// vec3.rs:146-156, 190-200, 226-236, 290-300, 326-336, 370-379
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x() + b.x(), a.y() + b.y(), a.z() + b.z()); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x() - b.x(), a.y() - b.y(), a.z() - b.z()); }
inline Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a.x() * b.x(), a.y() * b.y(), a.z() * b.z()); }
inline Vec3 operator*(const Vec3& a, double k) { return Vec3(a.x() * k, a.y() * k, a.z() * k); }
inline Vec3 operator*(double k, const Vec3& a) { return a * k; }  // vec3.rs:246-260: f64 * Vec3 forwards to Vec3 * f64
inline Vec3 operator/(const Vec3& a, double k) { return Vec3(a.x() / k, a.y() / k, a.z() / k); }
inline Vec3 operator-(const Vec3& a) { return Vec3(-a.x(), -a.y(), -a.z()); }

// ----------------------------------------------------------------- ray.rs
struct Ray {
    Vec3 origin, direction;
    double time;
    Ray() : time(0.0) {}
    Ray(const Vec3& o, const Vec3& d, double t) : origin(o), direction(d), time(t) {}
    // ray.rs:31-33
    Vec3 point_at(double t) const { return origin + t * direction; }
};

// Rust f64::max / f64::min: if one operand is NaN the other is returned.
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline double rmin(double a, double b) { return std::fmin(a, b); }

// ---------------------------------------------------------------- aabb.rs
struct AABB {
    Vec3 min, max;
    AABB() {}
    AABB(const Vec3& a, const Vec3& b) : min(a), max(b) {}
    // aabb.rs:24-51
    bool hit(const Ray& ray, double t_min, double t_max) const {
        double min_t = t_min;
        double max_t = t_max;
        for (int c = 0; c < 3; ++c) {
            double min_component = min[c];
            double max_component = max[c];
            double origin_component = ray.origin[c];
            double direction_component = 1.0 / ray.direction[c];
            double t0 = (min_component - origin_component) * direction_component;
            double t1 = (max_component - origin_component) * direction_component;
            if (direction_component < 0.0) std::swap(t0, t1);
            min_t = rmax(t0, min_t);
            max_t = rmin(t1, max_t);
            if (max_t <= min_t) return false;
        }
        return true;
    }
    // aabb.rs:53-71
    AABB surrounding_box(const AABB& o) const {
        Vec3 small(rmin(min.x(), o.min.x()), rmin(min.y(), o.min.y()), rmin(min.z(), o.min.z()));
        Vec3 big(rmax(max.x(), o.max.x()), rmax(max.y(), o.max.y()), rmax(max.z(), o.max.z()));
        return AABB(small, big);
    }
};

// -------------------------------------------------------------- counters
struct Counters {
    uint64_t segments = 0, box_tests = 0, leaf_tests = 0, sphere_tests = 0, rect_tests = 0, shaded = 0, noise_evals = 0;
};
thread_local Counters tl_counters;

// ------------------------------------------------------------- sampler
// Stands in for `rng: &mut T` on the render path: draw k of stage `stage` of
// one camera path (include/mirt_rng.h).
struct PathRng {
    uint64_t stream;
    uint32_t stage;
    uint32_t k;
    void set_stage(uint32_t s) {
        stage = s;
        k = 0;
    }
    // util.rs:8-10
    double random_double() { return mirt_rng_double(stream, stage, k++); }
};

// ------------------------------------------------------------ hittable.rs
struct Hit {
    double t, u, v;
    Vec3 p;
    bool front_facing;
    Vec3 normal;
    uint32_t material;
    int32_t prim;  // filled by the caller that knows the index
    int32_t side;
};

// hittable.rs:27-52
inline Hit make_hit(double t, double u, double v, const Vec3& p, const Ray& ray, const Vec3& outward_normal,
                    uint32_t material) {
    Hit h;
    h.front_facing = ray.direction.dot(outward_normal) < 0.0;
    h.normal = h.front_facing ? outward_normal : -outward_normal;
    h.t = t;
    h.u = u;
    h.v = v;
    h.p = p;
    h.material = material;
    h.prim = -1;
    h.side = 0;
    return h;
}

// hittable.rs:732-749
inline void sphere_uv(const Vec3& p, double* u, double* v) {
    const double PI = 3.14159265358979323846264338327950288;
    const double TAU = 6.28318530717958647692528676655900577;
    double theta = std::acos(-p.y());
    double phi = std::atan2(-p.z(), p.x()) + PI;
    *u = phi / TAU;
    *v = theta / PI;
}

// hittable.rs:105-137 (center fixed) and 487-521 (center(time) evaluated twice)
inline std::optional<Hit> sphere_hit_at(const Vec3& center_for_oc, const Vec3& center_for_normal, double radius,
                                        uint32_t material, const Ray& ray, double t_min, double t_max) {
    tl_counters.sphere_tests++;
    Vec3 oc = ray.origin - center_for_oc;
    double a = ray.direction.dot(ray.direction);
    double b = oc.dot(ray.direction);
    double c = oc.dot(oc) - radius * radius;
    double discriminant = b * b - a * c;
    if (discriminant < 0.0) return std::nullopt;
    discriminant = std::sqrt(discriminant);
    double f = (-b - discriminant) / a;
    double g = (-b + discriminant) / a;
    double t;
    if (f < t_max && f > t_min) {
        t = f;
    } else if (g < t_max && g > t_min) {
        t = g;
    } else {
        return std::nullopt;
    }
    Vec3 p = ray.point_at(t);
    Vec3 outward_normal = (p - center_for_normal) / radius;
    double u, v;
    sphere_uv(outward_normal, &u, &v);
    return make_hit(t, u, v, p, ray, outward_normal, material);
}

// hittable.rs:550-561.  `center_initial` is object.center(0.0) = the sphere's own centre.
inline Vec3 moving_center(const mirt_shape& s, double time) {
    Vec3 center_initial(s.f[0], s.f[1], s.f[2]);
    Vec3 center_final(s.f[4], s.f[5], s.f[6]);
    double elapsed_time = time - s.f[7];
    double movement_time = s.f[8] - s.f[7];
    Vec3 distance = center_final - center_initial;
    return center_initial + ((elapsed_time / movement_time) * distance);
}

// hittable.rs:185-207 (axis 2 = xy), 249-271 (axis 1 = xz), 313-335 (axis 0 = yz).
// `a`,`b` are the two in-plane axes in the order the rect names them.
inline std::optional<Hit> rect_hit(int axis_k, int axis_a, int axis_b, double a0, double a1, double b0, double b1,
                                   double k, uint32_t material, const Ray& ray, double t_min, double t_max) {
    tl_counters.rect_tests++;
    double t = (k - ray.origin[axis_k]) / ray.direction[axis_k];
    if (t < t_min || t > t_max) return std::nullopt;
    double a = ray.origin[axis_a] + t * ray.direction[axis_a];
    double b = ray.origin[axis_b] + t * ray.direction[axis_b];
    if (a < a0 || a > a1 || b < b0 || b > b1) return std::nullopt;
    double u = (a - a0) / (a1 - a0);
    double v = (b - b0) / (b1 - b0);
    Vec3 p = ray.point_at(t);
    Vec3 outward_normal(axis_k == 0 ? 1.0 : 0.0, axis_k == 1 ? 1.0 : 0.0, axis_k == 2 ? 1.0 : 0.0);
    return make_hit(t, u, v, p, ray, outward_normal, material);
}

inline std::optional<Hit> xy_rect_hit(double x0, double x1, double y0, double y1, double k, uint32_t m, const Ray& r,
                                      double t_min, double t_max) {
    return rect_hit(2, 0, 1, x0, x1, y0, y1, k, m, r, t_min, t_max);
}
inline std::optional<Hit> xz_rect_hit(double x0, double x1, double z0, double z1, double k, uint32_t m, const Ray& r,
                                      double t_min, double t_max) {
    return rect_hit(1, 0, 2, x0, x1, z0, z1, k, m, r, t_min, t_max);
}
inline std::optional<Hit> yz_rect_hit(double y0, double y1, double z0, double z1, double k, uint32_t m, const Ray& r,
                                      double t_min, double t_max) {
    return rect_hit(0, 1, 2, y0, y1, z0, z1, k, m, r, t_min, t_max);
}

// One side of a cube, in the order of Cube::new (hittable.rs:405-412).
inline std::optional<Hit> cube_side_hit(const mirt_shape& s, int side, const Ray& ray, double t_min, double t_max) {
    const double p0x = s.f[0], p0y = s.f[1], p0z = s.f[2], p1x = s.f[3], p1y = s.f[4], p1z = s.f[5];
    switch (side) {
        case 0: return xy_rect_hit(p0x, p1x, p0y, p1y, p0z, s.material, ray, t_min, t_max);
        case 1: return xy_rect_hit(p0x, p1x, p0y, p1y, p1z, s.material, ray, t_min, t_max);
        case 2: return xz_rect_hit(p0x, p1x, p0z, p1z, p1y, s.material, ray, t_min, t_max);
        case 3: return xz_rect_hit(p0x, p1x, p0z, p1z, p0y, s.material, ray, t_min, t_max);
        case 4: return yz_rect_hit(p0y, p1y, p0z, p1z, p1x, s.material, ray, t_min, t_max);
        default: return yz_rect_hit(p0y, p1y, p0z, p1z, p0x, s.material, ray, t_min, t_max);
    }
}

// hittable.rs:432-445
inline std::optional<Hit> cube_hit(const mirt_shape& s, const Ray& ray, double t_min, double t_max) {
    double min_distance = t_max;
    std::optional<Hit> nearest;
    for (int side = 0; side < 6; ++side) {
        std::optional<Hit> h = cube_side_hit(s, side, ray, t_min, min_distance);
        if (h) {
            min_distance = h->t;
            h->side = side;
            nearest = h;
        }
    }
    return nearest;
}

// hittable.rs:625-635
inline std::optional<Hit> shape_hit(const mirt_shape& s, const Ray& ray, double t_min, double t_max) {
    tl_counters.leaf_tests++;
    switch (s.kind) {
        case MIRT_SHAPE_SPHERE: {
            Vec3 c(s.f[0], s.f[1], s.f[2]);
            return sphere_hit_at(c, c, s.f[3], s.material, ray, t_min, t_max);
        }
        case MIRT_SHAPE_XY_RECT: return xy_rect_hit(s.f[0], s.f[1], s.f[2], s.f[3], s.f[4], s.material, ray, t_min, t_max);
        case MIRT_SHAPE_XZ_RECT: return xz_rect_hit(s.f[0], s.f[1], s.f[2], s.f[3], s.f[4], s.material, ray, t_min, t_max);
        case MIRT_SHAPE_YZ_RECT: return yz_rect_hit(s.f[0], s.f[1], s.f[2], s.f[3], s.f[4], s.material, ray, t_min, t_max);
        case MIRT_SHAPE_CUBE: return cube_hit(s, ray, t_min, t_max);
        case MIRT_SHAPE_MOVING_SPHERE: {
            Vec3 c1 = moving_center(s, ray.time);  // hittable.rs:490
            Vec3 c2 = moving_center(s, ray.time);  // hittable.rs:514
            return sphere_hit_at(c1, c2, s.f[3], s.material, ray, t_min, t_max);
        }
        default: return std::nullopt;
    }
}

// bounding_box of each shape: hittable.rs:139-144, 209-214, 273-278, 337-342, 447-465, 523-547
inline AABB shape_bbox(const mirt_shape& s, double time_start, double time_end) {
    switch (s.kind) {
        case MIRT_SHAPE_SPHERE: {
            Vec3 c(s.f[0], s.f[1], s.f[2]);
            double r = s.f[3];
            return AABB(c - Vec3(r, r, r), c + Vec3(r, r, r));
        }
        case MIRT_SHAPE_XY_RECT:
            return AABB(Vec3(s.f[0], s.f[2], s.f[4] - 0.0001), Vec3(s.f[1], s.f[3], s.f[4] + 0.0001));
        case MIRT_SHAPE_XZ_RECT:
            return AABB(Vec3(s.f[0], s.f[4] - 0.0001, s.f[2]), Vec3(s.f[1], s.f[4] + 0.0001, s.f[3]));
        case MIRT_SHAPE_YZ_RECT:
            return AABB(Vec3(s.f[4] - 0.0001, s.f[0], s.f[2]), Vec3(s.f[4] + 0.0001, s.f[1], s.f[3]));
        case MIRT_SHAPE_CUBE: {
            const double p0x = s.f[0], p0y = s.f[1], p0z = s.f[2], p1x = s.f[3], p1y = s.f[4], p1z = s.f[5];
            // union of the six side boxes, in side order (hittable.rs:447-465)
            AABB sides[6] = {
                AABB(Vec3(p0x, p0y, p0z - 0.0001), Vec3(p1x, p1y, p0z + 0.0001)),
                AABB(Vec3(p0x, p0y, p1z - 0.0001), Vec3(p1x, p1y, p1z + 0.0001)),
                AABB(Vec3(p0x, p1y - 0.0001, p0z), Vec3(p1x, p1y + 0.0001, p1z)),
                AABB(Vec3(p0x, p0y - 0.0001, p0z), Vec3(p1x, p0y + 0.0001, p1z)),
                AABB(Vec3(p1x - 0.0001, p0y, p0z), Vec3(p1x + 0.0001, p1y, p1z)),
                AABB(Vec3(p0x - 0.0001, p0y, p0z), Vec3(p0x + 0.0001, p1y, p1z)),
            };
            AABB box = sides[0];
            for (int i = 1; i < 6; ++i) box = box.surrounding_box(sides[i]);
            return box;
        }
        case MIRT_SHAPE_MOVING_SPHERE: {
            double r = s.f[3];
            Vec3 ci = moving_center(s, time_start);
            Vec3 cf = moving_center(s, time_end);
            AABB bi(ci - Vec3(r, r, r), ci + Vec3(r, r, r));
            if (ci == cf) return bi;
            AABB bf(cf - Vec3(r, r, r), cf + Vec3(r, r, r));
            return bi.surrounding_box(bf);
        }
        default: return AABB();
    }
}

// ------------------------------------------------------------------ scene
struct Scene {
    const mirt_scene_desc* d;
};

// hittable.rs:698-710 (linear list)
inline std::optional<Hit> list_hit(const Scene& sc, const Ray& ray, double t_min, double t_max) {
    double min_distance = t_max;
    std::optional<Hit> nearest;
    for (uint32_t i = 0; i < sc.d->n_shapes; ++i) {
        std::optional<Hit> h = shape_hit(sc.d->shapes[i], ray, t_min, min_distance);
        if (h) {
            min_distance = h->t;
            h->prim = (int32_t)i;
            nearest = h;
        }
    }
    return nearest;
}

// bvh.rs:37-80
std::optional<Hit> indexed_hit(const Scene& sc, uint32_t member_index, const Ray& ray, double t_min, double t_max) {
    const mirt_bvh_member& m = sc.d->nodes[member_index];
    if (m.is_leaf) {
        std::optional<Hit> h = shape_hit(sc.d->shapes[m.shape], ray, t_min, t_max);
        if (h) h->prim = (int32_t)m.shape;
        return h;
    }
    tl_counters.box_tests++;
    AABB bounds(Vec3(m.bmin), Vec3(m.bmax));
    if (!bounds.hit(ray, t_min, t_max)) return std::nullopt;
    std::optional<Hit> left_hit = indexed_hit(sc, m.left, ray, t_min, t_max);
    std::optional<Hit> right_hit = indexed_hit(sc, m.right, ray, t_min, t_max);
    if (left_hit && right_hit) {
        if (left_hit->t < right_hit->t) return left_hit;
        return right_hit;
    }
    if (left_hit) return left_hit;
    if (right_hit) return right_hit;
    return std::nullopt;
}

// bvh.rs:83-86 / renderer.rs:21-25
inline std::optional<Hit> world_hit(const Scene& sc, bool use_bvh, const Ray& ray, double t_min, double t_max) {
    tl_counters.segments++;
    if (use_bvh) return indexed_hit(sc, sc.d->n_nodes - 1, ray, t_min, t_max);
    return list_hit(sc, ray, t_min, t_max);
}

// --------------------------------------------------------------- perlin.rs
// perlin.rs:100-122
inline double trilinear_interp(const Vec3 sample[2][2][2], double u, double v, double w) {
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accumulator = 0.0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                Vec3 weight(u - (double)i, v - (double)j, w - (double)k);
                double color = weight.dot(sample[i][j][k]);
                double fi = (double)i * uu + (1.0 - (double)i) * (1.0 - uu);
                double fj = (double)j * vv + (1.0 - (double)j) * (1.0 - vv);
                double fk = (double)k * ww + (1.0 - (double)k) * (1.0 - ww);
                accumulator += color * fi * fj * fk;
            }
    return accumulator;
}

// `x as i32` on an f64 in Rust: saturating, NaN -> 0.
inline int32_t f64_as_i32(double x) {
    if (std::isnan(x)) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
// `x as u32` on an f64 in Rust: saturating, NaN -> 0.
inline uint32_t f64_as_u32(double x) {
    if (std::isnan(x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}

// perlin.rs:52-85
inline double perlin_noise(const mirt_perlin& pl, const Vec3& p) {
    double u = p.x() - std::floor(p.x());
    double v = p.y() - std::floor(p.y());
    double w = p.z() - std::floor(p.z());
    int32_t i = f64_as_i32(std::floor(p.x()));
    int32_t j = f64_as_i32(std::floor(p.y()));
    int32_t k = f64_as_i32(std::floor(p.z()));
    Vec3 sample[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                int32_t ii = (int32_t)((uint32_t)i + (uint32_t)di) & 255;
                int32_t jj = (int32_t)((uint32_t)j + (uint32_t)dj) & 255;
                int32_t kk = (int32_t)((uint32_t)k + (uint32_t)dk) & 255;
                int32_t x = pl.perm_x[ii];
                int32_t y = pl.perm_y[jj];
                int32_t z = pl.perm_z[kk];
                size_t index = (size_t)(x ^ y ^ z);
                sample[di][dj][dk] = Vec3(pl.vecs[index]);
            }
    return trilinear_interp(sample, u, v, w);
}

// perlin.rs:38-50
inline double perlin_turbulence(const mirt_perlin& pl, const Vec3& p, int depth) {
    Vec3 current_p = p;
    double weight = 1.0;
    double accumulator = 0.0;
    for (int i = 0; i < depth; ++i) {
        accumulator += weight * perlin_noise(pl, current_p);
        weight *= 0.5;
        current_p = current_p * 2.0;
    }
    return std::fabs(accumulator);
}

// -------------------------------------------------------------- texture.rs
// texture.rs:195-205
inline double fclamp(double x, double lo, double hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// texture.rs:184-193 and the four impls
Vec3 texture_value(const Scene& sc, uint32_t tex_index, double u, double v, const Vec3& p) {
    const mirt_texture& t = sc.d->textures[tex_index];
    switch (t.kind) {
        case MIRT_TEX_CONSTANT:  // texture.rs:24-28
            return Vec3(t.color);
        case MIRT_TEX_CHECKER: {  // texture.rs:45-54
            double sines = std::sin(10.0 * p.x()) * std::sin(10.0 * p.y()) * std::sin(10.0 * p.z());
            if (sines < 0.0) return texture_value(sc, t.odd, u, v, p);
            return texture_value(sc, t.even, u, v, p);
        }
        case MIRT_TEX_NOISE: {  // texture.rs:82-87
            tl_counters.noise_evals++;
            double turbulence = t.scale * p.z() + 10.0 * perlin_turbulence(sc.d->perlins[t.perlin], p, 7);
            return Vec3(1.0, 1.0, 1.0) * 0.5 * (1.0 + std::sin(turbulence));
        }
        case MIRT_TEX_IMAGE: {  // texture.rs:106-122
            const mirt_image& im = sc.d->images[t.image];
            double uc = fclamp(u, 0.0, 1.0);
            double vc = 1.0 - fclamp(v, 0.0, 1.0);
            uint32_t i = f64_as_u32(uc * (double)im.width);
            uint32_t j = f64_as_u32(vc * (double)im.height);
            if (i > im.width - 1) i = im.width - 1;
            if (j > im.height - 1) j = im.height - 1;
            double color_scale = 1.0 / 255.0;
            const uint8_t* px = im.rgb + 3 * ((size_t)j * im.width + i);
            return color_scale * Vec3((double)px[0], (double)px[1], (double)px[2]);
        }
        default: return Vec3();
    }
}

// ------------------------------------------------------------- material.rs
// material.rs:211-225
inline Vec3 random_in_unit_sphere(PathRng& rng) {
    auto gen_p = [&]() {
        double a = rng.random_double();
        double b = rng.random_double();
        double c = rng.random_double();
        return 2.0 * Vec3(a, b, c) - Vec3(1.0, 1.0, 1.0);
    };
    Vec3 p = gen_p();
    while (p.square_length() >= 1.0) p = gen_p();
    return p;
}
// material.rs:227-229
inline Vec3 reflect(const Vec3& v, const Vec3& n) { return v - (2.0 * v.dot(n) * n); }
// material.rs:231-237
inline Vec3 refract(const Vec3& uv, const Vec3& n, double etai_over_etat) {
    double cos_theta = rmin((-uv).dot(n), 1.0);
    Vec3 perpendicular = etai_over_etat * (uv + cos_theta * n);
    Vec3 parallel = -std::sqrt(std::fabs(1.0 - perpendicular.square_length())) * n;
    return perpendicular + parallel;
}
// f64::powi(5): x^5 by squaring = x * (x*x)*(x*x)
inline double powi5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x * x4;
}
// material.rs:239-243
inline double schlick(double cosine, double ref_idx) {
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 *= r0;
    return r0 + (1.0 - r0) * powi5(1.0 - cosine);
}

struct Scatter {
    Ray scattered;
    Vec3 attenuation;
};

// material.rs:191-199 and the four impls
inline std::optional<Scatter> material_scatter(const Scene& sc, const mirt_material& m, PathRng& rng, const Ray& ray,
                                               const Hit& hit) {
    switch (m.kind) {
        case MIRT_MAT_LAMBERTIAN: {  // material.rs:51-59
            Vec3 target = hit.p + hit.normal + random_in_unit_sphere(rng);
            Ray scattered(hit.p, target - hit.p, ray.time);
            Vec3 attenuation = texture_value(sc, m.texture, hit.u, hit.v, hit.p);
            return Scatter{scattered, attenuation};
        }
        case MIRT_MAT_METAL: {  // material.rs:74-91
            Vec3 reflected = reflect(ray.direction.unit(), hit.normal);
            Ray scattered(hit.p, reflected + m.param * random_in_unit_sphere(rng), ray.time);
            if (scattered.direction.dot(hit.normal) > 0.0) return Scatter{scattered, Vec3(m.albedo)};
            return std::nullopt;
        }
        case MIRT_MAT_DIELECTRIC: {  // material.rs:104-131
            double refraction_ratio = hit.front_facing ? 1.0 / m.param : m.param;
            Vec3 unit_direction = ray.direction.unit();
            double cos_theta = rmin((-unit_direction).dot(hit.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            bool cannot_refract = refraction_ratio * sin_theta > 1.0;
            Vec3 direction;
            if (cannot_refract || schlick(cos_theta, refraction_ratio) > rng.random_double()) {
                direction = reflect(unit_direction, hit.normal);
            } else {
                direction = refract(unit_direction, hit.normal, refraction_ratio);
            }
            return Scatter{Ray(hit.p, direction, ray.time), Vec3(1.0, 1.0, 1.0)};
        }
        case MIRT_MAT_DIFFUSE_LIGHT:  // material.rs:144-147
        default: return std::nullopt;
    }
}

// material.rs:201-208, 13-15, 149-151
inline Vec3 material_emit(const Scene& sc, const mirt_material& m, double u, double v, const Vec3& p) {
    if (m.kind == MIRT_MAT_DIFFUSE_LIGHT) return texture_value(sc, m.texture, u, v, p);
    return Vec3();
}

// --------------------------------------------------------------- camera.rs
// camera.rs:78-88
inline Vec3 random_in_unit_disk(PathRng& rng) {
    auto gen_p = [&]() {
        double a = rng.random_double();
        double b = rng.random_double();
        return 2.0 * Vec3(a, b, 0.0) - Vec3(1.0, 1.0, 0.0);
    };
    Vec3 p = gen_p();
    while (p.dot(p) >= 1.0) p = gen_p();
    return p;
}

// camera.rs:65-76
inline Ray camera_ray(const mirt_camera& c, PathRng& rng, double s, double t) {
    Vec3 origin(c.origin), llc(c.lower_left_corner), horizontal(c.horizontal), vertical(c.vertical), cu(c.u), cv(c.v);
    // (the sampler gives the shutter time a draw index of its own, before the lens pairs, so that it does not
    // depend on how many pairs the rejection loop takes: include/mirt_rng.h.  The operations are camera.rs's.)
    const double shutter = rng.random_double();
    Vec3 rd = c.lens_radius * random_in_unit_disk(rng);
    Vec3 offset = cu * rd.x() + cv * rd.y();
    Vec3 direction = llc + s * horizontal + t * vertical - origin - offset;
    double time = c.time_start + shutter * (c.time_end - c.time_start);
    return Ray(origin + offset, direction, time);
}

// ------------------------------------------------------------- renderer.rs
struct TraceSink {
    mirt_ray* rays;
    mirt_hit* hits;
    size_t cap, n;
};

// renderer.rs:92-115.  `bounce` = 50 - max_depth of the reference's call.
Vec3 color(const Scene& sc, bool use_bvh, PathRng& rng, const Ray& ray, const Vec3& background, uint32_t max_depth,
           uint32_t bounce, TraceSink* sink) {
    if (max_depth == 0) return Vec3(0.0, 0.0, 0.0);
    std::optional<Hit> hit = world_hit(sc, use_bvh, ray, 0.001, std::numeric_limits<double>::infinity());
    if (sink && sink->n < sink->cap) {
        mirt_ray& r = sink->rays[sink->n];
        for (int c = 0; c < 3; ++c) {
            r.origin[c] = ray.origin[c];
            r.direction[c] = ray.direction[c];
        }
        r.time = ray.time;
        mirt_hit& h = sink->hits[sink->n];
        h.prim = hit ? hit->prim : -1;
        h.side = hit ? hit->side : 0;
        h.t = hit ? hit->t : 0.0;
        sink->n++;
    }
    if (hit) {
        tl_counters.shaded++;
        const mirt_material& m = sc.d->materials[hit->material];
        Vec3 emitted = material_emit(sc, m, hit->u, hit->v, hit->p);
        rng.set_stage(bounce + 1);
        std::optional<Scatter> scatter = material_scatter(sc, m, rng, ray, *hit);
        if (scatter) {
            return (emitted + scatter->attenuation) *
                   color(sc, use_bvh, rng, scatter->scattered, background, max_depth - 1, bounce + 1, sink);
        }
        return emitted;
    }
    return background;
}

// renderer.rs:64-90 up to (not including) the /spp, sqrt, *255.99 tail: the
// sum over samples.  Returns the f64 sum so that callers can check it too.
Vec3 render_pixel_sum(const Scene& sc, bool use_bvh, const mirt_camera& cam, const Vec3& background, uint32_t width,
                      uint32_t height, uint32_t s_begin, uint32_t s_end, uint32_t max_depth, uint64_t seed,
                      uint32_t idx, TraceSink* sink) {
    uint32_t i = idx % width;
    uint32_t j = height - 1 - idx / width;
    Vec3 pixel;
    for (uint32_t s = s_begin; s < s_end; ++s) {
        PathRng rng{mirt_rng_stream(seed, idx, s), 0, 0};
        double u = ((double)i + rng.random_double()) / (double)width;
        double v = ((double)j + rng.random_double()) / (double)height;
        Ray ray = camera_ray(cam, rng, u, v);
        pixel = pixel + color(sc, use_bvh, rng, ray, background, max_depth, 0, sink);
    }
    return pixel;
}

// renderer.rs:85-89 and the `as u8` of renderer.rs:42 (saturating, truncating, NaN -> 0)
inline uint8_t f64_as_u8(double x) {
    if (std::isnan(x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}
inline void resolve_pixel(const Vec3& sum, uint32_t spp, uint8_t* out) {
    Vec3 pixel = sum / (double)spp;
    pixel = Vec3(std::sqrt(pixel.x()), std::sqrt(pixel.y()), std::sqrt(pixel.z()));
    pixel = pixel * 255.99;
    out[0] = f64_as_u8(pixel.x());
    out[1] = f64_as_u8(pixel.y());
    out[2] = f64_as_u8(pixel.z());
}

// ------------------------------------------------------------------ bvh.rs
// bvh.rs:137-237
mirt_bvh_member build_tree_internal(mirt_seq* rng, mirt_shape* shapes, std::vector<mirt_bvh_member>& nodes,
                                    uint32_t offset, uint32_t n, double time_initial, double time_final) {
    uint32_t axis = mirt_seq_below(rng, 3);  // bvh.rs:146
    // bvh.rs:147-166: a stable sort that only ever asks "is a less than b", with
    // less defined as (a.min[axis] - b.min[axis] < 0.0).
    std::stable_sort(shapes, shapes + n, [&](const mirt_shape& a, const mirt_shape& b) {
        double left_axis = shape_bbox(a, time_initial, time_final).min[(int)axis];
        double right_axis = shape_bbox(b, time_initial, time_final).min[(int)axis];
        return left_axis - right_axis < 0.0;
    });
    mirt_bvh_member m;
    std::memset(&m, 0, sizeof m);
    if (n == 1) {  // bvh.rs:179-180
        m.is_leaf = 1;
        m.shape = offset;
        return m;
    }
    if (n == 2) {  // bvh.rs:181-200
        AABB bounds = shape_bbox(shapes[0], time_initial, time_final)
                          .surrounding_box(shape_bbox(shapes[1], time_initial, time_final));
        mirt_bvh_member l, r;
        std::memset(&l, 0, sizeof l);
        std::memset(&r, 0, sizeof r);
        l.is_leaf = 1;
        l.shape = offset;
        r.is_leaf = 1;
        r.shape = offset + 1;
        m.left = (uint32_t)nodes.size();
        nodes.push_back(l);
        m.right = (uint32_t)nodes.size();
        nodes.push_back(r);
        for (int c = 0; c < 3; ++c) {
            m.bmin[c] = bounds.min[c];
            m.bmax[c] = bounds.max[c];
        }
        return m;
    }
    uint32_t halving_point = n / 2;  // bvh.rs:202
    mirt_bvh_member left = build_tree_internal(rng, shapes, nodes, offset, halving_point, time_initial, time_final);
    m.left = (uint32_t)nodes.size();
    nodes.push_back(left);
    mirt_bvh_member right = build_tree_internal(rng, shapes + halving_point, nodes, offset + halving_point,
                                                n - halving_point, time_initial, time_final);
    m.right = (uint32_t)nodes.size();
    nodes.push_back(right);
    // bvh.rs:228-229 -> hittable.rs:712-730: fold surrounding_box over the whole slice
    AABB bounds = shape_bbox(shapes[0], time_initial, time_final);
    for (uint32_t i = 1; i < n; ++i) bounds = bounds.surrounding_box(shape_bbox(shapes[i], time_initial, time_final));
    for (int c = 0; c < 3; ++c) {
        m.bmin[c] = bounds.min[c];
        m.bmax[c] = bounds.max[c];
    }
    return m;
}

}  // namespace

// ======================================================================
// C entry points (called through ctypes by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg)
// ======================================================================
extern "C" {

typedef struct oracle_counters {
    uint64_t segments, box_tests, leaf_tests, sphere_tests, rect_tests, shaded, noise_evals;
} oracle_counters;

static void counters_out(const Counters& c, oracle_counters* out) {
    if (!out) return;
    out->segments = c.segments;
    out->box_tests = c.box_tests;
    out->leaf_tests = c.leaf_tests;
    out->sphere_tests = c.sphere_tests;
    out->rect_tests = c.rect_tests;
    out->shaded = c.shaded;
    out->noise_evals = c.noise_evals;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Hittable::hit on the BVH (bvh.rs:83-86) or the list (hittable.rs:698-710) for n rays.
int oracle_intersect(const mirt_scene_desc* desc, const mirt_ray* rays, size_t n, double t_min, double t_max,
                     int use_bvh, mirt_hit* hits, oracle_counters* counters) {
    if (!desc || !rays || !hits) return -1;
    if (use_bvh && (!desc->nodes || desc->n_nodes == 0)) return -1;
    Scene sc{desc};
    Counters total;
    // rays are independent: all cores for large batches (the answers do not depend on the split)
#pragma omp parallel if (n >= 4096)
    {
        tl_counters = Counters();
#pragma omp for schedule(dynamic, 1024)
        for (int64_t r = 0; r < (int64_t)n; ++r) {
            Ray ray(Vec3(rays[r].origin), Vec3(rays[r].direction), rays[r].time);
            std::optional<Hit> h = world_hit(sc, use_bvh != 0, ray, t_min, t_max);
            hits[r].prim = h ? h->prim : -1;
            hits[r].side = h ? h->side : 0;
            hits[r].t = h ? h->t : 0.0;
        }
#pragma omp critical
        {
            total.segments += tl_counters.segments;
            total.box_tests += tl_counters.box_tests;
            total.leaf_tests += tl_counters.leaf_tests;
            total.sphere_tests += tl_counters.sphere_tests;
            total.rect_tests += tl_counters.rect_tests;
            total.shaded += tl_counters.shaded;
            total.noise_evals += tl_counters.noise_evals;
        }
    }
    counters_out(total, counters);
    return 0;
}

// Full detail of one Hit (for known-answer tests): out = {t,u,v,px,py,pz,nx,ny,nz,front,material,prim,side}
int oracle_hit_detail(const mirt_scene_desc* desc, const mirt_ray* ray_in, double t_min, double t_max, int use_bvh,
                      double* out13) {
    Scene sc{desc};
    Ray ray(Vec3(ray_in->origin), Vec3(ray_in->direction), ray_in->time);
    std::optional<Hit> h = world_hit(sc, use_bvh != 0, ray, t_min, t_max);
    if (!h) return 0;
    double o[13] = {h->t,       h->u,          h->v,          h->p.x(),
                    h->p.y(),   h->p.z(),      h->normal.x(), h->normal.y(),
                    h->normal.z(), h->front_facing ? 1.0 : 0.0, (double)h->material, (double)h->prim,
                    (double)h->side};
    std::memcpy(out13, o, sizeof o);
    return 1;
}

// AABB::hit (aabb.rs:24-51)
int oracle_aabb_hit(const double bmin[3], const double bmax[3], const mirt_ray* ray_in, double t_min, double t_max) {
    Ray ray(Vec3(ray_in->origin), Vec3(ray_in->direction), ray_in->time);
    return AABB(Vec3(bmin), Vec3(bmax)).hit(ray, t_min, t_max) ? 1 : 0;
}

void oracle_sphere_uv(const double p[3], double* u, double* v) { sphere_uv(Vec3(p), u, v); }

void oracle_shape_bbox(const mirt_shape* s, double time_start, double time_end, double bmin[3], double bmax[3]) {
    AABB b = shape_bbox(*s, time_start, time_end);
    for (int c = 0; c < 3; ++c) {
        bmin[c] = b.min[c];
        bmax[c] = b.max[c];
    }
}

// BoundingVolumeHierarchy::new (bvh.rs:18-27, 112-135): sorts `shapes` in place,
// writes 2n-1 members (root last).  Axis draws come from mirt_seq(seed).
int oracle_bvh_build(mirt_shape* shapes, uint32_t n, double time_initial, double time_final, uint64_t seed,
                     mirt_bvh_member* nodes_out) {
    if (!shapes || n == 0 || !nodes_out) return -1;
    mirt_seq rng = mirt_seq_new(seed);
    std::vector<mirt_bvh_member> nodes;
    nodes.reserve((size_t)n * 2);
    mirt_bvh_member root = build_tree_internal(&rng, shapes, nodes, 0, n, time_initial, time_final);
    nodes.push_back(root);
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(mirt_bvh_member));
    return (int)nodes.size();
}

// Camera::new (camera.rs:21-63)
void oracle_camera_new(const double look_from[3], const double look_at[3], const double v_up[3], double vertical_fov,
                       double aspect, double aperture, double focus_distance, double time_start, double time_end,
                       mirt_camera* out) {
    const double PI = 3.14159265358979323846264338327950288;
    double lens_radius = aperture / 2.0;
    double theta = vertical_fov * PI / 180.0;
    double half_height = std::tan(theta / 2.0);
    double half_width = aspect * half_height;
    Vec3 from(look_from), at(look_at), vup(v_up);
    Vec3 w = (from - at).unit();
    Vec3 u = vup.cross(w).unit();
    Vec3 v = w.cross(u);
    Vec3 origin = from;
    Vec3 llc = origin - (half_width * focus_distance * u) - (half_height * focus_distance * v) - focus_distance * w;
    Vec3 horizontal = 2.0 * half_width * focus_distance * u;
    Vec3 vertical = 2.0 * half_height * focus_distance * v;
    for (int c = 0; c < 3; ++c) {
        out->origin[c] = origin[c];
        out->lower_left_corner[c] = llc[c];
        out->horizontal[c] = horizontal[c];
        out->vertical[c] = vertical[c];
        out->u[c] = u[c];
        out->v[c] = v[c];
        out->w[c] = w[c];
    }
    out->lens_radius = lens_radius;
    out->time_start = time_start;
    out->time_end = time_end;
}

// Camera::ray for (pixel, sample) with the jitter of renderer.rs:79-80
void oracle_camera_ray(const mirt_camera* cam, uint32_t width, uint32_t height, uint32_t idx, uint32_t sample,
                       uint64_t seed, mirt_ray* out) {
    uint32_t i = idx % width;
    uint32_t j = height - 1 - idx / width;
    PathRng rng{mirt_rng_stream(seed, idx, sample), 0, 0};
    double u = ((double)i + rng.random_double()) / (double)width;
    double v = ((double)j + rng.random_double()) / (double)height;
    Ray r = camera_ray(*cam, rng, u, v);
    for (int c = 0; c < 3; ++c) {
        out->origin[c] = r.origin[c];
        out->direction[c] = r.direction[c];
    }
    out->time = r.time;
}

// Perlin::new (perlin.rs:20-36, 88-98) with draws from mirt_seq(seed)
void oracle_perlin_new(uint64_t seed, mirt_perlin* out) {
    mirt_seq rng = mirt_seq_new(seed);
    for (int i = 0; i < 256; ++i) {
        // Vec3::random(rng, -1.0, 1.0).unit()  (vec3.rs:27-33, 75-78)
        double x = mirt_seq_range(&rng, -1.0, 1.0);
        double y = mirt_seq_range(&rng, -1.0, 1.0);
        double z = mirt_seq_range(&rng, -1.0, 1.0);
        Vec3 uvec = Vec3(x, y, z).unit();
        out->vecs[i][0] = uvec.x();
        out->vecs[i][1] = uvec.y();
        out->vecs[i][2] = uvec.z();
    }
    int32_t* perms[3] = {out->perm_x, out->perm_y, out->perm_z};
    for (int t = 0; t < 3; ++t) {
        // (0..256).rev() then SliceRandom::shuffle = Fisher-Yates from the top
        int32_t* p = perms[t];
        for (int i = 0; i < 256; ++i) p[i] = 255 - i;
        for (uint32_t i = 255; i >= 1; --i) {
            uint32_t j = mirt_seq_below(&rng, i + 1);
            std::swap(p[i], p[j]);
        }
    }
}

double oracle_perlin_noise(const mirt_perlin* pl, const double p[3]) { return perlin_noise(*pl, Vec3(p)); }
double oracle_perlin_turbulence(const mirt_perlin* pl, const double p[3], int depth) {
    return perlin_turbulence(*pl, Vec3(p), depth);
}

// Texture::value (texture.rs:184-193)
void oracle_texture_value(const mirt_scene_desc* desc, uint32_t texture, double u, double v, const double p[3],
                          double out[3]) {
    Scene sc{desc};
    Vec3 c = texture_value(sc, texture, u, v, Vec3(p));
    out[0] = c.x();
    out[1] = c.y();
    out[2] = c.z();
}

// Material::scatter on a given ray and its closest hit in the scene; draws from
// stage `stage` of stream(seed, pixel, sample).  Returns 1 if Some.
// out = {ox,oy,oz,dx,dy,dz,time,ar,ag,ab}
int oracle_scatter(const mirt_scene_desc* desc, const mirt_ray* ray_in, int use_bvh, uint64_t stream, uint32_t stage,
                   double out10[10], double emitted[3]) {
    Scene sc{desc};
    Ray ray(Vec3(ray_in->origin), Vec3(ray_in->direction), ray_in->time);
    std::optional<Hit> hit = world_hit(sc, use_bvh != 0, ray, 0.001, std::numeric_limits<double>::infinity());
    if (!hit) return -1;
    const mirt_material& m = desc->materials[hit->material];
    Vec3 e = material_emit(sc, m, hit->u, hit->v, hit->p);
    emitted[0] = e.x();
    emitted[1] = e.y();
    emitted[2] = e.z();
    PathRng rng{stream, stage, 0};
    std::optional<Scatter> s = material_scatter(sc, m, rng, ray, *hit);
    if (!s) return 0;
    for (int c = 0; c < 3; ++c) {
        out10[c] = s->scattered.origin[c];
        out10[3 + c] = s->scattered.direction[c];
        out10[7 + c] = s->attenuation[c];
    }
    out10[6] = s->scattered.time;
    return 1;
}

double oracle_schlick(double cosine, double ref_idx) { return schlick(cosine, ref_idx); }
void oracle_reflect(const double v[3], const double n[3], double out[3]) {
    Vec3 r = reflect(Vec3(v), Vec3(n));
    out[0] = r.x();
    out[1] = r.y();
    out[2] = r.z();
}
void oracle_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
    Vec3 r = refract(Vec3(uv), Vec3(n), ratio);
    out[0] = r.x();
    out[1] = r.y();
    out[2] = r.z();
}

// render_world (renderer.rs:9-62) over pixels [pixel_begin, pixel_end) and
// samples [sample_begin, sample_end).  out_rgb (width*height*3, may be NULL)
// gets the resolved bytes when the whole sample range 0..spp was rendered;
// out_sum (width*height*3 doubles, may be NULL) gets the per-pixel f64 sums.
// nthreads <= 0 means all cores; 1 means the serial branch (renderer.rs:44-61).
int oracle_render(const mirt_scene_desc* desc, const mirt_camera* cam, const double background[3], uint32_t width,
                  uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed, int use_bvh, uint32_t pixel_begin,
                  uint32_t pixel_end, uint32_t sample_begin, uint32_t sample_end, int nthreads, uint8_t* out_rgb,
                  double* out_sum, oracle_counters* counters, double* seconds) {
    if (!desc || !cam || !background) return -1;
    if (use_bvh && (!desc->nodes || desc->n_nodes == 0)) return -1;
    if (pixel_end == 0) pixel_end = width * height;
    if (sample_end == 0) sample_end = spp;
    Scene sc{desc};
    Vec3 bg(background);
    Counters total;
    auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        tl_counters = Counters();
#pragma omp for schedule(dynamic, 16)
        for (int64_t idx = (int64_t)pixel_begin; idx < (int64_t)pixel_end; ++idx) {
            Vec3 sum = render_pixel_sum(sc, use_bvh != 0, *cam, bg, width, height, sample_begin, sample_end, max_depth,
                                        seed, (uint32_t)idx, nullptr);
            if (out_sum) {
                out_sum[3 * idx + 0] = sum.x();
                out_sum[3 * idx + 1] = sum.y();
                out_sum[3 * idx + 2] = sum.z();
            }
            if (out_rgb) resolve_pixel(sum, spp, out_rgb + 3 * idx);
        }
#pragma omp critical
        {
            total.segments += tl_counters.segments;
            total.box_tests += tl_counters.box_tests;
            total.leaf_tests += tl_counters.leaf_tests;
            total.sphere_tests += tl_counters.sphere_tests;
            total.rect_tests += tl_counters.rect_tests;
            total.shaded += tl_counters.shaded;
            total.noise_evals += tl_counters.noise_evals;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    counters_out(total, counters);
    return 0;
}

// The tail of `render` + the u8 cast (renderer.rs:85-89, 42) for n pixels.
void oracle_resolve(const double* sum, size_t n_pixels, uint32_t spp, uint8_t* out_rgb) {
    for (size_t i = 0; i < n_pixels; ++i) resolve_pixel(Vec3(sum + 3 * i), spp, out_rgb + 3 * i);
}

// Every ray handed to world.hit while rendering the given pixels (serial), with
// the hit the oracle found: the primary + secondary ray set for intersection
// parity.  Returns the number of rays recorded (<= cap).
int64_t oracle_trace_rays(const mirt_scene_desc* desc, const mirt_camera* cam, const double background[3],
                          uint32_t width, uint32_t height, uint32_t max_depth, uint64_t seed, int use_bvh,
                          const uint32_t* pixels, size_t n_pixels, uint32_t sample_begin, uint32_t sample_end,
                          mirt_ray* rays, mirt_hit* hits, size_t cap) {
    if (!desc || !cam || !pixels || !rays || !hits) return -1;
    Scene sc{desc};
    Vec3 bg(background);
    if (n_pixels < 512) {
        TraceSink sink{rays, hits, cap, 0};
        for (size_t k = 0; k < n_pixels && sink.n < cap; ++k)
            render_pixel_sum(sc, use_bvh != 0, *cam, bg, width, height, sample_begin, sample_end, max_depth, seed,
                             pixels[k], &sink);
        return (int64_t)sink.n;
    }
    // Many pixels: blocks of pixels on all cores, each into a buffer of its own, put together in
    // pixel order afterwards -- the same sequence of rays the serial loop records.
    const size_t BLOCK_PIXELS = 64;
    const size_t n_blocks = (n_pixels + BLOCK_PIXELS - 1) / BLOCK_PIXELS;
    const size_t worst = BLOCK_PIXELS * (size_t)(sample_end - sample_begin) * max_depth;
    std::vector<std::vector<mirt_ray>> block_rays(n_blocks);
    std::vector<std::vector<mirt_hit>> block_hits(n_blocks);
#pragma omp parallel
    {
        std::vector<mirt_ray> local_rays(worst);
        std::vector<mirt_hit> local_hits(worst);
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)n_blocks; ++b) {
            TraceSink sink{local_rays.data(), local_hits.data(), worst, 0};
            const size_t k_end = std::min(n_pixels, (size_t)(b + 1) * BLOCK_PIXELS);
            for (size_t k = (size_t)b * BLOCK_PIXELS; k < k_end; ++k)
                render_pixel_sum(sc, use_bvh != 0, *cam, bg, width, height, sample_begin, sample_end, max_depth, seed,
                                 pixels[k], &sink);
            block_rays[b].assign(local_rays.begin(), local_rays.begin() + sink.n);
            block_hits[b].assign(local_hits.begin(), local_hits.begin() + sink.n);
        }
    }
    size_t n = 0;
    for (size_t b = 0; b < n_blocks && n < cap; ++b) {
        const size_t take = std::min(block_rays[b].size(), cap - n);
        std::memcpy(rays + n, block_rays[b].data(), take * sizeof(mirt_ray));
        std::memcpy(hits + n, block_hits[b].data(), take * sizeof(mirt_hit));
        n += take;
    }
    return (int64_t)n;
}

}  // extern "C"
