// device_intersect.h -- f64 ray/primitive and ray/box tests and the three ways of
// walking the BVH, as __device__ code for gfx950.
//
// Every expression keeps the operand order of the reference so that, compiled
// with -ffp-contract=off, hit/miss decisions, primitive ids and t carry the same
// bits the reference's f64 arithmetic produces:
//   AABB::hit            src/aabb.rs:24-51
//   Sphere::hit          src/hittable.rs:105-137
//   Xy/Xz/YzRect::hit    src/hittable.rs:185-207, 249-271, 313-335
//   Cube::hit            src/hittable.rs:432-445 (sides in the order of 405-412)
//   Moving<Sphere>::hit  src/hittable.rs:487-521, Moving::center 550-561
//   list hit             src/hittable.rs:698-710
//   BVH indexed_hit      src/bvh.rs:37-80
#pragma once

#include <hip/hip_runtime.h>

#include "device_scene.h"

namespace mirt {

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double k) { return V3{a.x * k, a.y * k, a.z * k}; }
__device__ __forceinline__ V3 operator*(double k, V3 a) { return V3{a.x * k, a.y * k, a.z * k}; }  // vec3.rs:246-260
__device__ __forceinline__ V3 operator/(V3 a, double k) { return V3{a.x / k, a.y / k, a.z / k}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:88-90
__device__ __forceinline__ double square_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// vec3.rs:75-78: multiply by the reciprocal of the length
__device__ __forceinline__ V3 unit(V3 a) {
    double k = 1.0 / sqrt(square_length(a));
    return V3{a.x * k, a.y * k, a.z * k};
}

struct RayD {
    V3 o, d;
    double time;
};

// What a traversal keeps per ray besides the ray itself.
struct RayPre {
    double inv[3];  // 1.0 / d: the reference recomputes it per box (aabb.rs:33); same bits
    double dd;      // d.d: `a` of the sphere quadratic (hittable.rs:108)
    int sgn[3];     // inv < 0.0 (aabb.rs:38)
};

__device__ __forceinline__ RayPre ray_pre(const RayD& r) {
    RayPre p;
    p.inv[0] = 1.0 / r.d.x;
    p.inv[1] = 1.0 / r.d.y;
    p.inv[2] = 1.0 / r.d.z;
    p.sgn[0] = p.inv[0] < 0.0;
    p.sgn[1] = p.inv[1] < 0.0;
    p.sgn[2] = p.inv[2] < 0.0;
    p.dd = dot(r.d, r.d);
    return p;
}

// ---- slab test (aabb.rs:24-51) ----------------------------------------------
// `b` points at [axis][min,max].  Returns pass/fail and the entry distance
// (min_t after the third axis).  The reference exits at the first axis with
// max_t <= min_t; because min_t never decreases and max_t never increases over
// the three axes, that equals testing once after the last axis.  fmax/fmin drop
// a NaN operand like f64::max/min do.
__device__ __forceinline__ bool slab_test(const double* __restrict__ b, const RayD& r, const RayPre& p, double t_min,
                                          double t_max, double* entry) {
    double lo_x = p.sgn[0] ? b[1] : b[0], hi_x = p.sgn[0] ? b[0] : b[1];
    double lo_y = p.sgn[1] ? b[3] : b[2], hi_y = p.sgn[1] ? b[2] : b[3];
    double lo_z = p.sgn[2] ? b[5] : b[4], hi_z = p.sgn[2] ? b[4] : b[5];
    double t0x = (lo_x - r.o.x) * p.inv[0], t1x = (hi_x - r.o.x) * p.inv[0];
    double t0y = (lo_y - r.o.y) * p.inv[1], t1y = (hi_y - r.o.y) * p.inv[1];
    double t0z = (lo_z - r.o.z) * p.inv[2], t1z = (hi_z - r.o.z) * p.inv[2];
    double mn = fmax(t0x, t_min);
    double mx = fmin(t1x, t_max);
    mn = fmax(t0y, mn);
    mx = fmin(t1y, mx);
    mn = fmax(t0z, mn);
    mx = fmin(t1z, mx);
    *entry = mn;
    return !(mx <= mn);
}

// ---- primitives ---------------------------------------------------------------
__device__ __forceinline__ bool sphere_hit(V3 c, double radius, const RayD& r, double dd, double t_min, double t_max,
                                           double* t_out) {
    V3 oc = r.o - c;
    double a = dd;
    double b = dot(oc, r.d);
    double cc = dot(oc, oc) - radius * radius;
    double discriminant = b * b - a * cc;
    if (discriminant < 0.0) return false;
    discriminant = sqrt(discriminant);
    double f = (-b - discriminant) / a;
    if (f < t_max && f > t_min) {
        *t_out = f;
        return true;
    }
    double g = (-b + discriminant) / a;
    if (g < t_max && g > t_min) {
        *t_out = g;
        return true;
    }
    return false;
}

__device__ __forceinline__ V3 moving_center(const double* __restrict__ f, double time) {
    V3 c0 = v3(f[0], f[1], f[2]);
    V3 c1 = v3(f[4], f[5], f[6]);
    double elapsed_time = time - f[7];
    double movement_time = f[8] - f[7];
    V3 distance = c1 - c0;
    return c0 + ((elapsed_time / movement_time) * distance);
}

// K = axis the rect is perpendicular to, A/B = its in-plane axes (in the rect's own order).
template <int K, int A, int B>
__device__ __forceinline__ bool rect_hit(double a0, double a1, double b0, double b1, double k, const RayD& r,
                                         double t_min, double t_max, double* t_out) {
    const double ok = K == 0 ? r.o.x : (K == 1 ? r.o.y : r.o.z);
    const double dk = K == 0 ? r.d.x : (K == 1 ? r.d.y : r.d.z);
    const double oa = A == 0 ? r.o.x : (A == 1 ? r.o.y : r.o.z);
    const double da = A == 0 ? r.d.x : (A == 1 ? r.d.y : r.d.z);
    const double ob = B == 0 ? r.o.x : (B == 1 ? r.o.y : r.o.z);
    const double db = B == 0 ? r.d.x : (B == 1 ? r.d.y : r.d.z);
    double t = (k - ok) / dk;
    if (t < t_min || t > t_max) return false;
    double a = oa + t * da;
    double b = ob + t * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    *t_out = t;
    return true;
}

__device__ __forceinline__ bool cube_hit(const double* __restrict__ f, const RayD& r, double t_min, double t_max,
                                         double* t_out, int* side_out) {
    const double p0x = f[0], p0y = f[1], p0z = f[2], p1x = f[3], p1y = f[4], p1z = f[5];
    double min_distance = t_max;
    bool any = false;
    double t;
    if (rect_hit<2, 0, 1>(p0x, p1x, p0y, p1y, p0z, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 0; any = true; }
    if (rect_hit<2, 0, 1>(p0x, p1x, p0y, p1y, p1z, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 1; any = true; }
    if (rect_hit<1, 0, 2>(p0x, p1x, p0z, p1z, p1y, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 2; any = true; }
    if (rect_hit<1, 0, 2>(p0x, p1x, p0z, p1z, p0y, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 3; any = true; }
    if (rect_hit<0, 1, 2>(p0y, p1y, p0z, p1z, p1x, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 4; any = true; }
    if (rect_hit<0, 1, 2>(p0y, p1y, p0z, p1z, p0x, r, t_min, min_distance, &t)) { min_distance = t; *side_out = 5; any = true; }
    *t_out = min_distance;
    return any;
}

// Shape::hit (hittable.rs:625-635) reduced to what a closest-hit query keeps.
// `f` = the shape's parameters (DShape::f, in global memory or staged in LDS).
__device__ __forceinline__ bool shape_hit_params(const double* __restrict__ f, uint32_t kind, const RayD& r, double dd,
                                                 double t_min, double t_max, double* t_out, int* side_out) {
    *side_out = 0;
    switch (kind) {
        case MIRT_SHAPE_SPHERE: return sphere_hit(v3(f[0], f[1], f[2]), f[3], r, dd, t_min, t_max, t_out);
        case MIRT_SHAPE_XY_RECT: return rect_hit<2, 0, 1>(f[0], f[1], f[2], f[3], f[4], r, t_min, t_max, t_out);
        case MIRT_SHAPE_XZ_RECT: return rect_hit<1, 0, 2>(f[0], f[1], f[2], f[3], f[4], r, t_min, t_max, t_out);
        case MIRT_SHAPE_YZ_RECT: return rect_hit<0, 1, 2>(f[0], f[1], f[2], f[3], f[4], r, t_min, t_max, t_out);
        case MIRT_SHAPE_CUBE: return cube_hit(f, r, t_min, t_max, t_out, side_out);
        case MIRT_SHAPE_MOVING_SPHERE:
            return sphere_hit(moving_center(f, r.time), f[3], r, dd, t_min, t_max, t_out);
        default: return false;
    }
}

// The same for the fast traversal: `p` = the shape's ten traversal parameters and `tk`
// its TraverseKind (device_scene.h).
__device__ __forceinline__ bool tshape_hit(const double* __restrict__ p, uint32_t tk, const RayD& r, double dd,
                                           double t_min, double t_max, double* t_out, int* side_out) {
    *side_out = 0;
    if (tk == TK_SPHERE) {
        // Moving::center (hittable.rs:550-561) with the division by a power of two turned into
        // the multiplication by its exact reciprocal; a static sphere has p[4..6] = 0
        const double frac = (r.time - p[7]) * p[8];
        const V3 c = v3(p[0], p[1], p[2]) + frac * v3(p[4], p[5], p[6]);
        return sphere_hit(c, p[3], r, dd, t_min, t_max, t_out);
    }
    return shape_hit_params(p, tk, r, dd, t_min, t_max, t_out, side_out);
}

__device__ __forceinline__ bool shape_hit(const DShape* __restrict__ shapes, uint32_t index, uint32_t kind,
                                          const RayD& r, double dd, double t_min, double t_max, double* t_out,
                                          int* side_out) {
    return shape_hit_params(shapes[index].f, kind, r, dd, t_min, t_max, t_out, side_out);
}

struct HitRec {
    double t;
    int32_t prim;  // -1 = none
    int32_t side;
};

// ---- linear list (hittable.rs:698-710) ---------------------------------------
__device__ __forceinline__ HitRec list_closest(const DScene& sc, const RayD& r, double t_min, double t_max) {
    HitRec best{0.0, -1, 0};
    double min_distance = t_max;
    const double dd = dot(r.d, r.d);
    for (uint32_t i = 0; i < sc.n_shapes; ++i) {
        double t;
        int side;
        if (shape_hit(sc.shapes, i, sc.shapes[i].kind, r, dd, t_min, min_distance, &t, &side)) {
            min_distance = t;
            best.t = t;
            best.prim = (int32_t)i;
            best.side = side;
        }
    }
    return best;
}

// ---- BVH in the reference's own order (bvh.rs:37-80) ---------------------------
// The recursion of indexed_hit unrolled onto an explicit frame stack in private
// memory: every member whose ancestors' boxes pass is visited, left then right,
// both with the caller's t_max, and the two results are combined exactly as
// bvh.rs:68-79 (left survives only if left.t < right.t).  Not fast; this is the
// form the fast traversal is checked against on the device.
__device__ inline HitRec bvh_closest_reference(const DScene& sc, const RayD& r, double t_min, double t_max) {
    struct Frame {
        uint32_t member;
        uint32_t state;  // 0 = enter, 1 = left done, 2 = right done
        HitRec left;
    };
    Frame stack[MAX_TREE_HEIGHT + 2];
    int sp = 0;
    HitRec result{0.0, -1, 0};
    const double dd = dot(r.d, r.d);
    stack[0].member = sc.n_members - 1;  // root is last, bvh.rs:33-35
    stack[0].state = 0;
    sp = 1;
    while (sp > 0) {
        Frame& fr = stack[sp - 1];
        const mirt_bvh_member& m = sc.members[fr.member];
        if (fr.state == 0) {
            if (m.is_leaf) {
                double t;
                int side;
                const uint32_t kind = sc.shapes[m.shape].kind;
                if (shape_hit(sc.shapes, m.shape, kind, r, dd, t_min, t_max, &t, &side)) {
                    result = HitRec{t, (int32_t)m.shape, side};
                } else {
                    result = HitRec{0.0, -1, 0};
                }
                --sp;
                continue;
            }
            // AABB::hit with a fresh 1.0 / d per axis, aabb.rs:28-48
            double min_t = t_min, max_t = t_max;
            bool pass = true;
            const double o[3] = {r.o.x, r.o.y, r.o.z};
            const double d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double inv = 1.0 / d[c];
                double t0 = (m.bmin[c] - o[c]) * inv;
                double t1 = (m.bmax[c] - o[c]) * inv;
                if (inv < 0.0) {
                    double tmp = t0;
                    t0 = t1;
                    t1 = tmp;
                }
                min_t = fmax(t0, min_t);
                max_t = fmin(t1, max_t);
                if (max_t <= min_t) {
                    pass = false;
                    break;
                }
            }
            if (!pass) {
                result = HitRec{0.0, -1, 0};
                --sp;
                continue;
            }
            fr.state = 1;
            stack[sp].member = m.left;
            stack[sp].state = 0;
            ++sp;
            continue;
        }
        if (fr.state == 1) {
            fr.left = result;
            fr.state = 2;
            stack[sp].member = m.right;
            stack[sp].state = 0;
            ++sp;
            continue;
        }
        // state 2: `result` is the right subtree's answer
        const HitRec right = result;
        const HitRec left = fr.left;
        if (left.prim >= 0 && right.prim >= 0) {
            result = (left.t < right.t) ? left : right;
        } else if (left.prim >= 0) {
            result = left;
        } else {
            result = right;
        }
        --sp;
    }
    return result;
}

// Safety margin of the distance prune of the fast traversal (device_traverse.h): a
// child is skipped only if the slab entry distance of its box exceeds
// best_t * (1 + PRUNE_REL).  Rounding moves either quantity by parts in 1e-16 to,
// for a grazing hit on a very large sphere, parts in 1e-9; 1e-6 dwarfs both and
// costs no measurable pruning.
constexpr double PRUNE_REL = 1.0e-6;

}  // namespace mirt
