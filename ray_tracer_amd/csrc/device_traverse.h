// device_traverse.h -- closest-hit queries against the culling tree and the gates of
// accel_build.h: the walk of one ray cut into steps (RayWalk), and two drivers that run 64
// walks per wave -- traverse_persistent (a queue of rays; the extend kernel and
// mirt_intersect) and the path kernel's own loop in kernels.h.
//
// Why this shape (measured on MI355X, RTIOW scene):
//  * One ray per lane run to completion left 31 % of the lanes active in the average vector
//    instruction: rays of one wave need very different numbers of node visits, and shape tests
//    ran inside the node loop for the few lanes that had reached a leaf.  Hence steps: a lane
//    that meets leaves parks their shapes in registers; the wave runs node visits for the
//    lanes that want one and shape tests for the lanes that have parked some, whichever more
//    lanes wait for, and leaves a loop early when few lanes still want it.
//  * The kernels are bound by vector-instruction issue.  A node visit in the reference's f64
//    arithmetic costs ~40 f64 instructions for two boxes; culling needs no such precision, so
//    the tree's boxes are f32 and two boxes cost 6 packed FMAs and 8 min/max (slab_pair).
//    The reference's f64 boxes are consulted once per ray, for the winning shape (confirm).
//  * When they fit, the tree's nodes (56 B each) and the shapes' parameters (80 B each) are
//    staged in LDS once per workgroup; the near / far bound of each axis is read by address
//    (8-byte read at base + 8 * sign) instead of loading both bounds and selecting.
//  * The per-lane stack lives in LDS as [level][thread] and holds inner-node indices only
//    (16-bit when the tree has < 65536 nodes).
//  * traverse_persistent: waves pull chunks of rays from a global counter and request the
//    next ray of a lane while the current one is traced.
//
// Decisions are the reference's (device_intersect.h, accel_build.h): shapes tested with the
// caller's (t_min, t_max), smallest t wins and equal t goes to the larger shape index, the
// winner must pass the reference's slab test on the boxes above its leaf.
#pragma once

#include <hip/hip_runtime.h>

#include "device_intersect.h"

namespace mirt {

__device__ __forceinline__ uint32_t mbcnt_lt(unsigned long long mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

#ifndef MIRT_NODE_WIDE_LOADS
#define MIRT_NODE_WIDE_LOADS 1  // +1.6 % on the 3,500-shape scene against seven 8-byte loads picked by address
#endif
// A tree too large for LDS is read from global memory as a whole: keeping its top levels in LDS makes every
// node visit of a wave run both ways of reading (lanes above and below the cut), which costs more than the
// LDS reads save (measured on the 3,500-shape scene: -2.6 %; the top of the tree stays in the vector L1 anyway)
#ifndef MIRT_TOP_IN_LDS
#define MIRT_TOP_IN_LDS 0
#endif
constexpr uint32_t CUR_DONE = 0x7FFFFFFFu;
constexpr int NODE_LDS_BYTES = sizeof(CNode);  // 56
constexpr int SHAPE_LDS_BYTES = sizeof(ShapeRec);  // 136
constexpr uint32_t TUNING_DEFAULT = 2u | (64u << 8);  // see traverse_persistent
constexpr uint32_t CHUNK_RAYS = 512;  // most rays a wave takes from the queue per fetch

// Rays per fetch for a queue of n rays shared by `waves` persistent waves: large chunks
// for long queues (few atomics), down to one wave-load for short ones (a late bounce has
// a few thousand live paths: they must spread over many waves, not queue up behind a few).
__host__ __device__ inline uint32_t chunk_rays(uint32_t n, uint32_t waves) {
    uint32_t c = n / (waves * 4u);
    c = c < 64u ? 64u : (c > CHUNK_RAYS ? CHUNK_RAYS : c);
    return c & ~63u;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One 8-byte read at byte address `addr` + OFFSET of LDS, as exactly one ds_read_b64.
// Written as inline assembly because, left to itself, the compiler pairs neighbouring
// 8-byte LDS reads into ds_read2_b64; that instruction banks modulo 32 in four 16-lane
// groups, and with the 64 lanes of a wave in 64 different nodes it measured ~30 LDS cycles
// per instruction on gfx950 against ~2 for ds_read_b64 -- the node loop became LDS-bound.
// The caller must issue lds_wait() before using the value.
template <int OFFSET>
__device__ __forceinline__ f32x2 lds_read_f32x2(uint32_t addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}
template <int OFFSET>
__device__ __forceinline__ double lds_read_f64(uint32_t addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}
template <int OFFSET>
__device__ __forceinline__ uint2 lds_read_u32x2(uint32_t addr) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
template <int OFFSET>
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // nothing that reads the values may be hoisted above the wait
}

enum TraverseStat : int {
    STAT_OUTER_ITERS = 0,  // trips of the outer loop, per wave
    STAT_NODE_ITERS,       // wave-level node visits (trips of the inner loop)
    STAT_NODE_LANES,       // lanes active in them
    STAT_LEAF_ITERS,       // wave-level leaf phases
    STAT_LEAF_LANES,       // lanes with at least one parked leaf in them
    STAT_REFILL_ITERS,     // wave-level refills
    STAT_REFILL_LANES,     // lanes that took a new ray in them
    STAT_SHAPE_TESTS,      // shape tests (lane level)
    STAT_CYC_REFILL,       // shader clocks spent per wave in refill (incl. requesting the next rays)
    STAT_CYC_NODE,         // ... in the inner-node loop
    STAT_CYC_LEAF,         // ... in leaf tests
    STAT_CYC_STORE,        // ... storing results
    STAT_CYC_TOTAL,        // ... in the whole traversal
    STAT_COUNT
};

// Where the tree and the shapes are read from.
struct TraverseLds {
    const char* nodes;   // copies of the first n_top nodes of DScene::nodes (breadth-first: the top of the tree), NODE_LDS_BYTES apart
    const char* shapes;  // SHAPE_LDS_BYTES apart, or nullptr: read DScene::recs
    uint32_t n_top;      // NODES_IN_LDS kernels: all of them
    const char* items = nullptr;  // ScanWalk: copies of DScene::scan_items (`nodes` then holds DScene::scan_nodes)
    uint32_t n_items = 0;
};

// The culling slab test, f32, for the two children of a node at once.
//
// Per ray and axis: inv = 1 / d, and the bound b of a box maps to t = b * inv + c with
//   c_near = -(o * inv) - E,   c_far = -(o * inv) + E,   E = 2^-19 |o * inv|.
// It must never report a box as missed that contains a point o + t d, t_min < t <= best t,
// which the f64 shape tests could accept.  Against the exact (b - o) / d the computed t
// carries relative errors of at most 2^-24 each from rounding d, o, o * inv, c and the FMA to
// f32 and 2^-23 from v_rcp_f32, proportional to |b * inv| or |o * inv|; the first kind is
// covered by the 2^-19 |b| the host moved every bound outward by (accel_build.h: CULL_REL),
// the second by E -- both more than four times what is needed.  Where the arithmetic leaves
// the range of f32 (d = 0, |o * inv| > 3e38) the terms become infinities of the safe sign or
// NaN, and the max/min below drop a NaN operand: the axis then constrains nothing.
struct CullRay {
    // {inv, c_near} per axis and {c_far x, c_far y}, {c_far z, -}: register pairs, so that the packed
    // FMAs of slab_pair read one half of a pair for both their lanes and no copy is ever made
    f32x2 ic[3], far_xy, far_z;
    uint32_t off[3];  // 8 if the ray runs down the axis (near bound = upper) else 0
    // every term finite and far from the ends of f32's range: the same constants then also
    // bound t from the other side (sure_to_cross)
    bool tame;
    __device__ __forceinline__ float inv(int a) const { return ic[a].x; }
    __device__ __forceinline__ float c_near(int a) const { return ic[a].y; }
    __device__ __forceinline__ float c_far(int a) const { return a == 0 ? far_xy.x : a == 1 ? far_xy.y : far_z.x; }
    __device__ __forceinline__ void set(const RayD& r, bool cull) {
        const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
        float size = 0.0f, far[3], inv32[3];
        // A direction component beyond ~1e30 has a reciprocal that f32 holds as a denormal or as zero, and the
        // bounds above speak of normal numbers: such a ray is not culled at all.  (A component of zero, or too
        // small for f32, gives an infinite reciprocal, which only ever drops a constraint; a NaN fails the
        // comparison.)
#pragma unroll
        for (int a = 0; a < 3; ++a) inv32[a] = __builtin_amdgcn_rcpf((float)d[a]);
        cull = cull && fminf(fminf(__builtin_fabsf(inv32[0]), __builtin_fabsf(inv32[1])), __builtin_fabsf(inv32[2])) >= 1.0e-30f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float i32 = inv32[a];
            const float p = (float)o[a] * i32;
            const float e = __builtin_fabsf(p) * 0x1.0p-19f;
            size += __builtin_fabsf(i32) + __builtin_fabsf(p);
            ic[a].x = cull ? i32 : 0.0f;
            ic[a].y = cull ? -p - e : -__builtin_inff();
            far[a] = cull ? -p + e : __builtin_inff();
            off[a] = i32 < 0.0f ? 8u : 0u;
        }
        far_xy = (f32x2){far[0], far[1]};
        far_z = (f32x2){far[2], 0.0f};
        tame = cull && size < 1.0e18f;  // false for a NaN
    }
    // The same constants for a tree whose bounds are halves in grid units (device_scene.h, QNode): the
    // bound g maps to t = g * (step * inv) + c with c = (center - o) * inv -+ E.  Against the exact
    // (center + g * step - o) / d the computed t carries relative errors of at most 6e-7 in all (2^-24 for
    // rounding d to f32 and 2^-23 for v_rcp_f32 on both terms; 2^-24 each for rounding center - o, the product
    // with inv and c -+ E; 2^-24 for the FMA; g itself and g * step are exact), proportional to
    // |g * step * inv| <= half * |step * inv| or to |(center - o) * inv|: E = 2^-18 of their sum, six times
    // what is needed.  A ray whose terms are not all finite, of sane size and in f32's normal range (step * inv
    // must not underflow: the bound rests on its being exact) is not culled at all (`tame` is the culling flag
    // here): with every term finite no NaN arises and nothing rests on one being dropped.
    __device__ __forceinline__ void set_q(const RayD& r, bool cull, const QGrid& grid) {
        const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
        float size = 0.0f, least = __builtin_inff(), inv_q[3], c[3], e[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float i32 = __builtin_amdgcn_rcpf((float)d[a]);
            const float p = (float)(grid.center[a] - o[a]) * i32;
            inv_q[a] = i32 * grid.step[a];
            e[a] = (__builtin_fabsf(p) + __builtin_fabsf(inv_q[a]) * grid.half[a]) * 0x1.0p-18f;
            size += __builtin_fabsf(i32) + __builtin_fabsf(p) + __builtin_fabsf(inv_q[a]) * grid.half[a];
            least = fminf(least, fminf(__builtin_fabsf(inv_q[a]), __builtin_fabsf(i32)));  // (the reciprocal itself must be a normal number too)
            c[a] = p;
            off[a] = i32 < 0.0f ? 16u : 0u;  // the rotation that brings the near bound of a QBox dword into its low half
        }
        tame = cull && size < 1.0e18f && least > 1.0e-30f;  // false for a NaN
        float far[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ic[a].x = tame ? inv_q[a] : 0.0f;
            ic[a].y = tame ? c[a] - e[a] : -__builtin_inff();
            far[a] = tame ? c[a] + e[a] : __builtin_inff();
        }
        far_xy = (f32x2){far[0], far[1]};
        far_z = (f32x2){far[2], 0.0f};
    }
    // Does the ray cross `box` ([axis][lo, hi], f32, already moved inward by the host) somewhere
    // in (t_lo, t_hi), beyond what rounding could undo?  The mirror image of slab_pair: the near
    // bound with the FAR constant bounds the entry from above, the far bound with the NEAR
    // constant the exit from below.  Only for a tame ray (no NaN can arise, none is dropped).
    __device__ __forceinline__ bool sure_to_cross(const float* __restrict__ box, float t_lo, float t_hi) const {
        float in = t_lo, out = t_hi;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = box[2 * a], hi = box[2 * a + 1];
            const float near = off[a] ? hi : lo, far = off[a] ? lo : hi;
            in = fmaxf(in, __builtin_fmaf(near, inv(a), c_far(a)));
            out = fminf(out, __builtin_fmaf(far, inv(a), c_near(a)));
        }
        return tame && out > in;
    }
};

// entry / exit of both children's boxes for the t-range (t_lo, t_hi]; pass = the range is not empty.
// Written out in instructions: the packed FMA takes the ray's constant from one half of a register
// pair for both boxes (op_sel), and the max/min follow without the compiler's re-quieting of values
// it did not compute itself (an FMA's result is never a signalling NaN).
__device__ __forceinline__ void slab_pair(f32x2 nx, f32x2 fx, f32x2 ny, f32x2 fy, f32x2 nz, f32x2 fz, const CullRay& c,
                                          float t_lo, float t_hi, float* e0, float* e1, bool* pass0, bool* pass1) {
    f32x2 tnx, tny, tnz, tfx, tfy, tfz;
    asm("v_pk_fma_f32 %0, %6, %12, %12 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %7, %13, %13 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %8, %14, %14 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %15 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %4, %10, %13, %15 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %5, %11, %14, %16 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(tnx), "=&v"(tny), "=&v"(tnz), "=&v"(tfx), "=&v"(tfy), "=&v"(tfz)
        : "v"(nx), "v"(ny), "v"(nz), "v"(fx), "v"(fy), "v"(fz), "v"(c.ic[0]), "v"(c.ic[1]), "v"(c.ic[2]), "v"(c.far_xy), "v"(c.far_z));
    float in0, in1, out0, out1;
    asm("v_max_f32 %0, %4, %5\n\t"
        "v_max_f32 %1, %7, %8\n\t"
        "v_min_f32 %2, %10, %11\n\t"
        "v_min_f32 %3, %13, %14\n\t"
        "v_max3_f32 %0, %0, %6, %16\n\t"
        "v_max3_f32 %1, %1, %9, %16\n\t"
        "v_min3_f32 %2, %2, %12, %17\n\t"
        "v_min3_f32 %3, %3, %15, %17"
        : "=&v"(in0), "=&v"(in1), "=&v"(out0), "=&v"(out1)
        : "v"(tnx.x), "v"(tny.x), "v"(tnz.x), "v"(tnx.y), "v"(tny.y), "v"(tnz.y), "v"(tfx.x), "v"(tfy.x), "v"(tfz.x), "v"(tfx.y), "v"(tfy.y),
          "v"(tfz.y), "v"(t_lo), "v"(t_hi));
    *e0 = in0;
    *e1 = in1;
    *pass0 = !(out0 < in0);
    *pass1 = !(out1 < in1);
}

// The same for bounds that are halves (QBox): x0 = child 0's x bounds, near in the low and far in the high
// half, and so on; v_fma_mix_f32 converts the half it is pointed at on the way in.
__device__ __forceinline__ void slab_pair_q(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, const CullRay& c,
                                            float t_lo, float t_hi, float* e0, float* e1, bool* pass0, bool* pass1) {
    float tn0x, tn1x, tn0y, tn1y, tn0z, tn1z, tf0x, tf1x, tf0y, tf1y, tf0z, tf1z;
    asm("v_fma_mix_f32 %0, %12, %18, %19 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %13, %18, %19 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %14, %20, %21 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %15, %20, %21 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %16, %22, %23 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %17, %22, %23 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %12, %18, %24 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %7, %13, %18, %24 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %8, %14, %20, %25 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %9, %15, %20, %25 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %16, %22, %26 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %11, %17, %22, %26 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(tn0x), "=&v"(tn1x), "=&v"(tn0y), "=&v"(tn1y), "=&v"(tn0z), "=&v"(tn1z), "=&v"(tf0x), "=&v"(tf1x), "=&v"(tf0y), "=&v"(tf1y),
          "=&v"(tf0z), "=&v"(tf1z)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(z0), "v"(z1), "v"(c.ic[0].x), "v"(c.ic[0].y), "v"(c.ic[1].x), "v"(c.ic[1].y), "v"(c.ic[2].x),
          "v"(c.ic[2].y), "v"(c.far_xy.x), "v"(c.far_xy.y), "v"(c.far_z.x));
    float in0, in1, out0, out1;
    asm("v_max_f32 %0, %4, %5\n\t"
        "v_max_f32 %1, %7, %8\n\t"
        "v_min_f32 %2, %10, %11\n\t"
        "v_min_f32 %3, %13, %14\n\t"
        "v_max3_f32 %0, %0, %6, %16\n\t"
        "v_max3_f32 %1, %1, %9, %16\n\t"
        "v_min3_f32 %2, %2, %12, %17\n\t"
        "v_min3_f32 %3, %3, %15, %17"
        : "=&v"(in0), "=&v"(in1), "=&v"(out0), "=&v"(out1)
        : "v"(tn0x), "v"(tn0y), "v"(tn0z), "v"(tn1x), "v"(tn1y), "v"(tn1z), "v"(tf0x), "v"(tf0y), "v"(tf0z), "v"(tf1x), "v"(tf1y), "v"(tf1z),
          "v"(t_lo), "v"(t_hi));
    *e0 = in0;
    *e1 = in1;
    *pass0 = !(out0 < in0);
    *pass1 = !(out1 < in1);
}

// A value of f32 not below x * (1 + 1e-6) for x >= 0 (not above, for x < 0), and the mirror
// image: the culling range (t_lo, t_hi] must contain every t the f64 tests can accept.
// An infinite x is returned as it is (inf - inf * 2e-6 would be a NaN: a range end that the min / max
// of the culling tests would silently drop).
// The 1e-30 covers what is absolute in the f32 evaluation of a bound's t: products and sums that fall below f32's
// normal range are rounded to multiples of 1.4e-45, whatever their size.
// (added in f32: 1e-30 is a normal number there, and a sum of two f32 values is off by 6e-8 of itself at most)
__device__ __forceinline__ float f32_not_below(double x) { return fabs(x) == __builtin_inf() ? (float)x : (float)(x + fabs(x) * 2.0e-6) + 1.0e-30f; }
__device__ __forceinline__ float f32_not_above(double x) { return fabs(x) == __builtin_inf() ? (float)x : (float)(x - fabs(x) * 2.0e-6) - 1.0e-30f; }

// The reference's slab test (aabb.rs:24-51) on every gate of shape `prim`.  Rarely needed: a
// call, so that its registers are not the kernels'.
// (The ray travels by value: a reference would pin the caller's copy to memory.)
__device__ __noinline__ bool gates_pass(const DGate* __restrict__ gates, const uint32_t* __restrict__ gate_of, uint32_t prim,
                                        double ox, double oy, double oz, double dx, double dy, double dz, double t_min, double t_max) {
    uint32_t g = gate_of[prim];
    if (g == NO_GATE) return true;
    RayD r;
    r.o = v3(ox, oy, oz);
    r.d = v3(dx, dy, dz);
    r.time = 0.0;
    const RayPre p = ray_pre(r);
    bool ok = true;
    do {
        const DGate* __restrict__ gate = gates + g;
        double e;
        ok = slab_test(&gate->box[0][0], r, p, t_min, t_max, &e);
        g = gate->next;
    } while (ok && g != NO_GATE);
    return ok;
}

// The reference's own recursion (device_intersect.h, bvh_closest_reference) for ONE ray, as a call.
// For the rays whose answer is not a matter of geometry: a shape test that returns a NaN t (a rect
// does, for a ray lying in its plane -- wherever in that plane -- or with a NaN component:
// hittable.rs:189-193 compares and every comparison is false) wins or loses against its neighbour
// by position in the tree, not by value (bvh.rs:70-74), and a time that is not finite separates the
// static sphere, which ignores it, from the moving one.  No shortcut reproduces that; nothing else
// needs it.  The hit's side carries the shape's class and kind like a walk's own.
__device__ __noinline__ HitRec reference_answer(const mirt_bvh_member* __restrict__ members, uint32_t n_members,
                                                const DShape* __restrict__ shapes, uint32_t n_shapes, double ox, double oy, double oz,
                                                double dx, double dy, double dz, double time, double t_min, double t_max) {
    DScene sc{};
    sc.members = members;
    sc.n_members = n_members;
    sc.shapes = shapes;
    sc.n_shapes = n_shapes;
    RayD r;
    r.o = v3(ox, oy, oz);
    r.d = v3(dx, dy, dz);
    r.time = time;
    HitRec h = bvh_closest_reference(sc, r, t_min, t_max);
    if (h.prim >= 0) h.side = (int32_t)((uint32_t)h.side | shapes[h.prim].cls << 4 | shapes[h.prim].tk << 7);
    return h;
}

// The walk of ONE ray (one lane) through the tree, cut into steps so that a wave can
// choose, trip by trip, which step to run for which lanes:
//   begin(ray)            per-ray constants of the culling test, root
//   wants_node()          the lane can visit inner node `cur` next
//   wants_leaf()          the lane has parked shapes to test (PARK > 2: both may hold at once)
//   node_step()           both children's boxes of `cur`; parks leaves; moves on or pops
//   leaf_step()           tests one parked shape
//   finished()            neither: `best` is the closest hit (prim < 0: none) -- to be
//                         confirmed by confirm(); if that fails, begin_careful() and walk again
// ODDITIES: rays whose answer depends on the reference's order of evaluation (reference_answer) are
// recognised and handed to it.
// stack: LDS, [level][BLOCK] entries of StackT, this lane's column is `tid`.
// QNODES: the tree is read as QNodes (bounds as halves, 16-bit child references; device_scene.h) from LDS.
template <int BLOCK, class StackT, bool NODES_IN_LDS, bool SHAPES_IN_LDS, bool ODDITIES = true, int PARK = 2, bool QNODES = false>
struct RayWalk {
    static_assert(!QNODES || (NODES_IN_LDS && !SHAPES_IN_LDS && !ODDITIES), "QNodes live in LDS, their shapes' records in global memory");
    // best.side of a hit found by a walk: cube side | ShadeClass << 4 | TraverseKind << 7
    // (bits 23..28 of the leaf reference, moved down as they are)
    static constexpr uint32_t SIDE_MASK = 15u, CLASS_SHIFT = 4, TK_SHIFT = 7;
    RayD r;
    CullRay cull;
    HitRec best;
    float t_hi;   // f32 not below min(t_max, best t): nothing entered beyond it can beat the best hit
    uint32_t cur;
    int sp;
    // shapes parked for the next leaf steps (leaf references), at most PARK, in registers; q[0] is the
    // next out.  A node step parks up to two (both children), so it needs nq <= PARK - 2: with
    // PARK = 2 shape tests come first, with more a lane may go on visiting nodes before them
    uint32_t q[PARK], nq;
    // PACKED (nodes in LDS, PARK = 4): the parked shapes as four 16-bit references (ref16_of) in one
    // register pair instead, lowest = next out; the LDS copy of the tree holds its leaf references in
    // that form (stage_tree).  Parking is a shift and an or, and there is no count to keep: a reference
    // is never zero, so the low word is zero iff nothing is parked and the high word iff at most two are.
    // QNODES: the references are 16 bits as they come; with PARK = 2 a node visit still waits until nothing is parked.
    static constexpr bool PACKED = NODES_IN_LDS && (PARK == 4 || QNODES);
    static constexpr bool PACKED_EAGER = PACKED && PARK == 4;  // a node visit while at most two shapes are parked
    uint32_t parked_lo, parked_hi;
    bool careful;  // second walk: a hit counts only if its shape's gates pass
    bool weird;    // a NaN t or a time that is not finite was met: see reference_answer

    __device__ __forceinline__ void reset() {
        cur = CUR_DONE;
        sp = 0;
        nq = 0;
        clear_parked();
        best = HitRec{0.0, -1, 0};
        careful = false;
        weird = false;
    }
    __device__ __forceinline__ bool wants_node() const {
        return (PACKED ? (PACKED_EAGER ? parked_hi : parked_lo) == 0u : nq <= (uint32_t)(PARK - 2)) && cur != CUR_DONE;
    }
    __device__ __forceinline__ bool wants_leaf() const { return PACKED ? parked_lo != 0u : nq != 0; }
    // The wave's lanes that want a node visit / a shape test, as masks: one vote per comparison, combined
    // as masks (the vote on a combined flag goes through a vector select and a second comparison).
    __device__ __forceinline__ unsigned long long vote_node() const {
        return __builtin_amdgcn_ballot_w64(PACKED ? (PACKED_EAGER ? parked_hi : parked_lo) == 0u : nq <= (uint32_t)(PARK - 2)) &
               __builtin_amdgcn_ballot_w64(cur != CUR_DONE);
    }
    __device__ __forceinline__ unsigned long long vote_leaf() const { return __builtin_amdgcn_ballot_w64(PACKED ? parked_lo != 0u : nq != 0); }
    __device__ __forceinline__ bool finished() const { return (PACKED ? parked_lo == 0u : nq == 0) && cur == CUR_DONE; }

    __device__ __forceinline__ void clear_parked() {
#pragma unroll
        for (int k = 0; k < PARK; ++k) q[k] = 0;
        parked_lo = parked_hi = 0u;
    }
    // ref: a leaf reference as the node it was read from holds it (PACKED: its low 16 bits are the ref16)
    __device__ __forceinline__ void park(uint32_t ref) {
        if (PACKED) {
            // (hi:lo) << 16 | low half of ref: a funnel shift and a byte permute
            parked_hi = __builtin_amdgcn_alignbit(parked_hi, parked_lo, 16);
            parked_lo = __builtin_amdgcn_perm(parked_lo, ref, 0x05040100u);
            return;
        }
#pragma unroll
        for (int k = PARK - 1; k > 0; --k) q[k] = q[k - 1];
        q[0] = ref;
        ++nq;
    }

    __device__ __forceinline__ void start(const DScene& sc, double t_min, double t_max) {
        t_hi = f32_not_below(t_max);
        cur = CUR_DONE;
        sp = 0;
        nq = 0;
        clear_parked();
        best = HitRec{0.0, -1, 0};
        const uint32_t root = sc.root_ref;
        // a one-shape world: bare leaf (bvh.rs:179-180)
        if (root & REF_LEAF)
            park(QNODES   ? QREF_LEAF | (((root >> REF_KIND_SHIFT) & 7u) << QREF_KIND_SHIFT) | (root & REF_INDEX_MASK)
                 : PACKED ? ref16_of(root)
                          : root);
        else cur = root;
    }

    __device__ __forceinline__ void begin(const DScene& sc, const RayD& ray, double t_min, double t_max) {
        r = ray;
        // a moving sphere's box holds for times inside its own span only; and the boxes say nothing
        // about what the sphere test makes of a ray that starts too far away (accel_build.h, cull_reach)
        const float from = fmaxf(fmaxf(__builtin_fabsf((float)r.o.x), __builtin_fabsf((float)r.o.y)), __builtin_fabsf((float)r.o.z));
        const bool may_cull = r.time >= sc.time_lo && r.time <= sc.time_hi && from <= sc.cull_reach &&
                              r.o.x == r.o.x && r.o.y == r.o.y && r.o.z == r.o.z;
        if (QNODES) cull.set_q(r, may_cull, sc.qgrid);
        else cull.set(r, may_cull);
        careful = false;
        weird = false;
        if (ODDITIES) {
            // a rect "hits" a ray that lies in its plane WHEREVER the ray is ((k - o) / d = 0 / 0: a NaN
            // passes every comparison of hittable.rs:189-193), so no box may keep such a ray away from
            // it: rays with a zero component in their direction, or anything not finite, are not walked
            // at all but answered by reference_answer (confirm)
            const double product = r.d.x * r.d.y * r.d.z;
            const double sum = ((r.o.x + r.o.y) + (r.o.z + r.d.x)) + ((r.d.y + r.d.z) + r.time);
            weird = !(product != 0.0) || !(sum - sum == 0.0);
        }
        if (weird) {
            cur = CUR_DONE;
            sp = 0;
            nq = 0;
            clear_parked();
            best = HitRec{0.0, -1, 0};
            t_hi = 0.0f;
        } else {
            start(sc, t_min, t_max);
        }
    }

    // After finished(): is `best` the reference's answer?  Nearly always the cheap test on the
    // shape's record says so; the reference's own arithmetic decides what is left.
    __device__ __forceinline__ bool confirm(const DScene& sc, const TraverseLds& lds, double t_min, double t_max) {
        if (ODDITIES && weird) {
            best = sc.n_members ? reference_answer(sc.members, sc.n_members, sc.shapes, sc.n_shapes, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y,
                                                   r.d.z, r.time, t_min, t_max)
                                : best;
            weird = false;
            return true;
        }
        if (careful || best.prim < 0) return true;
        const ShapeRec* rec = SHAPES_IN_LDS ? reinterpret_cast<const ShapeRec*>(lds.shapes + (uint32_t)best.prim * SHAPE_LDS_BYTES)
                                            : sc.recs + best.prim;
        if (QNODES) {
            // the record's box is in world coordinates: the ray's constants for those, once per ray
            CullRay plain;
            plain.set(r, cull.tame);
            if (plain.sure_to_cross(&rec->sure[0][0], f32_not_below(t_min), f32_not_above(t_max))) return true;
        } else if (__builtin_expect(cull.sure_to_cross(&rec->sure[0][0], f32_not_below(t_min), f32_not_above(t_max)), 1)) {
            return true;
        }
        return gates_pass(sc.gates, sc.gate_of, (uint32_t)best.prim, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max);
    }
    __device__ __forceinline__ void begin_careful(const DScene& sc, double t_min, double t_max) {
        careful = true;
        start(sc, t_min, t_max);
    }

    // t_min: the query's (the same for every lane: its f32 lower bound lives in a scalar register)
    __device__ __forceinline__ void node_step(const DScene& sc, const TraverseLds& lds, StackT* __restrict__ stack, uint32_t tid, double t_min) {
        const float t_lo = f32_not_above(t_min);
        uint32_t c0, c1;
        float e0, e1;
        bool pass0, pass1;
        if (QNODES) {
            // three 8-byte reads (one per axis: both children's bounds) and the references, issued back to back, one wait
            const uint32_t box = (uint32_t)(size_t)lds.nodes + __umul24(cur, (uint32_t)sizeof(QBox));
            const uint32_t ref = (uint32_t)(size_t)lds.shapes + cur * (uint32_t)sizeof(QRefs);  // (the QRefs follow the QBoxes)
            const uint32_t refs = lds_read_u32<0>(ref);
            const uint2 x = lds_read_u32x2<0>(box), y = lds_read_u32x2<8>(box), z = lds_read_u32x2<16>(box);
            lds_wait();
            c0 = refs & 0xFFFFu;
            c1 = refs >> 16;
            // near bound into the low half, far bound into the high half of each child's dword
            slab_pair_q(__builtin_amdgcn_alignbit(x.x, x.x, cull.off[0]), __builtin_amdgcn_alignbit(x.y, x.y, cull.off[0]),
                        __builtin_amdgcn_alignbit(y.x, y.x, cull.off[1]), __builtin_amdgcn_alignbit(y.y, y.y, cull.off[1]),
                        __builtin_amdgcn_alignbit(z.x, z.x, cull.off[2]), __builtin_amdgcn_alignbit(z.y, z.y, cull.off[2]), cull, t_lo, t_hi, &e0,
                        &e1, &pass0, &pass1);
        } else if (NODES_IN_LDS) {
            // seven 8-byte reads of one node, issued back to back, one wait
            // (LDS byte address = the low half of the generic address of an LDS object)
            const uint32_t node = (uint32_t)(size_t)lds.nodes + __umul24(cur, (uint32_t)NODE_LDS_BYTES);  // (one v_mad_u32_u24)
            const uint32_t ax = node + cull.off[0], bx = node + (cull.off[0] ^ 8u);
            const uint32_t ay = node + cull.off[1], by = node + (cull.off[1] ^ 8u);
            const uint32_t az = node + cull.off[2], bz = node + (cull.off[2] ^ 8u);
            const uint2 refs = lds_read_u32x2<48>(node);
            const f32x2 nx = lds_read_f32x2<0>(ax), fx = lds_read_f32x2<0>(bx);
            const f32x2 ny = lds_read_f32x2<16>(ay), fy = lds_read_f32x2<16>(by);
            const f32x2 nz = lds_read_f32x2<32>(az), fz = lds_read_f32x2<32>(bz);
            lds_wait();
            c0 = refs.x;
            c1 = refs.y;
            slab_pair(nx, fx, ny, fy, nz, fz, cull, t_lo, t_hi, &e0, &e1, &pass0, &pass1);
        } else {
            // a tree too large for LDS: its top lives there, the rest is read from global memory
            f32x2 nx, fx, ny, fy, nz, fz;
            if (MIRT_TOP_IN_LDS && cur < lds.n_top) {
                const uint32_t node = (uint32_t)(size_t)lds.nodes + __umul24(cur, (uint32_t)NODE_LDS_BYTES);  // (one v_mad_u32_u24)
                const uint32_t ax = node + cull.off[0], bx = node + (cull.off[0] ^ 8u);
                const uint32_t ay = node + cull.off[1], by = node + (cull.off[1] ^ 8u);
                const uint32_t az = node + cull.off[2], bz = node + (cull.off[2] ^ 8u);
                const uint2 refs = lds_read_u32x2<48>(node);
                nx = lds_read_f32x2<0>(ax), fx = lds_read_f32x2<0>(bx);
                ny = lds_read_f32x2<16>(ay), fy = lds_read_f32x2<16>(by);
                nz = lds_read_f32x2<32>(az), fz = lds_read_f32x2<32>(bz);
                lds_wait();
                c0 = refs.x;
                c1 = refs.y;
            } else {
                const char* node = reinterpret_cast<const char*>(sc.nodes + cur);
#if MIRT_NODE_WIDE_LOADS
                // the whole node with three 16-byte loads and one of 8, near / far picked in registers: fewer
                // requests to the texture addresser than seven 8-byte ones
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
                const f32x4u bx = *reinterpret_cast<const f32x4u*>(node), by = *reinterpret_cast<const f32x4u*>(node + 16);
                const f32x4u bz = *reinterpret_cast<const f32x4u*>(node + 32);
                const uint2 refs = *reinterpret_cast<const uint2*>(node + 48);
                const bool dx = cull.off[0] != 0u, dy = cull.off[1] != 0u, dz = cull.off[2] != 0u;
                nx = (f32x2){dx ? bx.z : bx.x, dx ? bx.w : bx.y}, fx = (f32x2){dx ? bx.x : bx.z, dx ? bx.y : bx.w};
                ny = (f32x2){dy ? by.z : by.x, dy ? by.w : by.y}, fy = (f32x2){dy ? by.x : by.z, dy ? by.y : by.w};
                nz = (f32x2){dz ? bz.z : bz.x, dz ? bz.w : bz.y}, fz = (f32x2){dz ? bz.x : bz.z, dz ? bz.y : bz.w};
#else
                const uint2 refs = *reinterpret_cast<const uint2*>(node + 48);
                nx = *reinterpret_cast<const f32x2*>(node + cull.off[0]);
                fx = *reinterpret_cast<const f32x2*>(node + (cull.off[0] ^ 8u));
                ny = *reinterpret_cast<const f32x2*>(node + 16 + cull.off[1]);
                fy = *reinterpret_cast<const f32x2*>(node + 16 + (cull.off[1] ^ 8u));
                nz = *reinterpret_cast<const f32x2*>(node + 32 + cull.off[2]);
                fz = *reinterpret_cast<const f32x2*>(node + 32 + (cull.off[2] ^ 8u));
#endif
                c0 = refs.x;
                c1 = refs.y;
            }
            slab_pair(nx, fx, ny, fy, nz, fz, cull, t_lo, t_hi, &e0, &e1, &pass0, &pass1);
        }
        const bool leaf0 = (c0 & (QNODES ? QREF_LEAF : REF_LEAF)) != 0, leaf1 = (c1 & (QNODES ? QREF_LEAF : REF_LEAF)) != 0;
        // (the same decisions written two ways: how the compiler lays out the compares and branches is worth
        // +1.2 % on the RTIOW scene for the first form with the tree in LDS, and -0.9 % on the 3,500-shape scene,
        // whose nodes come from global memory)
        bool go0, go1;
        if (NODES_IN_LDS) {
            go0 = go1 = false;
            if (pass0) {
                if (leaf0) park(c0);
                else go0 = true;
            }
            if (pass1) {
                if (leaf1) park(c1);
                else go1 = true;
            }
        } else {
            if (pass0 && leaf0) park(c0);
            if (pass1 && leaf1) park(c1);
            go0 = pass0 && !leaf0;
            go1 = pass1 && !leaf1;
        }
        if (go0 && go1) {
            const bool first0 = e0 <= e1;
            stack[sp * BLOCK + tid] = (StackT)(first0 ? c1 : c0);
            ++sp;
            cur = first0 ? c0 : c1;
        } else if (go0) {
            cur = c0;
        } else if (go1) {
            cur = c1;
        } else if (sp > 0) {
            --sp;
            cur = (uint32_t)stack[sp * BLOCK + tid];
        } else {
            cur = CUR_DONE;
        }
    }

    __device__ __forceinline__ void leaf_step(const DScene& sc, const TraverseLds& lds, double t_min, double t_max) {
        uint32_t index, tk, class_and_kind;
        if (PACKED) {
            const uint32_t entry = parked_lo & 0xFFFFu;
            parked_lo = __builtin_amdgcn_alignbit(parked_hi, parked_lo, 16);
            parked_hi >>= 16;
            index = entry & (QNODES ? QREF_INDEX_MASK : REF16_INDEX_MASK);
            tk = QNODES ? (entry >> QREF_KIND_SHIFT) & 7u : entry >> REF16_KIND_SHIFT;
            class_and_kind = QNODES ? tk << 3 : entry >> REF16_CLASS_SHIFT;  // QNODES: the class comes from the record, below
        } else {
            const uint32_t entry = q[0];
#pragma unroll
            for (int k = 0; k + 1 < PARK; ++k) q[k] = q[k + 1];
            --nq;
            index = entry & REF_INDEX_MASK;
            tk = (entry >> REF_KIND_SHIFT) & 7u;
            class_and_kind = (entry >> REF_CLASS_SHIFT) & 63u;
        }
        double local[9];
        const double* p = sc.recs[SHAPES_IN_LDS ? 0u : index].f;
        if (QNODES) class_and_kind |= sc.recs[index].material >> REC_CLASS_SHIFT;
        if (SHAPES_IN_LDS) {
            // the nine parameters as nine single reads issued back to back (see lds_read_f32x2)
            const uint32_t rec = (uint32_t)(size_t)lds.shapes + __umul24(index, (uint32_t)SHAPE_LDS_BYTES);
            local[0] = lds_read_f64<0>(rec);
            local[1] = lds_read_f64<8>(rec);
            local[2] = lds_read_f64<16>(rec);
            local[3] = lds_read_f64<24>(rec);
            local[4] = lds_read_f64<32>(rec);
            local[5] = lds_read_f64<40>(rec);
            local[6] = lds_read_f64<48>(rec);
            local[7] = lds_read_f64<56>(rec);
            local[8] = lds_read_f64<64>(rec);
            lds_wait();
            p = local;
        }
        double t;
        int side;
        if (tshape_hit(p, tk, r, dot(r.d, r.d), t_min, t_max, &t, &side)) {
            if (ODDITIES) weird = weird || t != t;
            // smallest t wins; equal t: larger shape index wins (bvh.rs:70-74)
            if (best.prim < 0 || t < best.t || (t == best.t && (int32_t)index > best.prim)) {
                if (__builtin_expect(!careful, 1) || gates_pass(sc.gates, sc.gate_of, index, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max)) {
                    best = HitRec{t, (int32_t)index, (int32_t)((uint32_t)side | class_and_kind << CLASS_SHIFT)};
                    t_hi = fminf(t_hi, f32_not_below(t));
                }
            }
        }
    }
};

#ifdef MIRT_PATHS_STATS
// development build: wave clocks of the two parts of ScanWalk::leaf_step, trips of the second
__device__ unsigned long long g_scan_stats[4];
#endif
// The exact test of one item (a rect, or one face of a cube) of a scene that is rendered without a tree
// (ScanWalk below), each lane its own item.  The three orientations share one piece of code: the axes are data.
// (A free function over the ray's components BY VALUE: written as a member that reads them from the
// walk, the selects between components become loads from a selected address, and the whole walk then
// lives in scratch memory instead of registers.)
// Is a ray sure to start on (within what t_min cuts off of) the rect-kind item at position k of its axis?
// |k - o| < |d| t_min (1 - 1e-9) implies |t| < t_min for the rounded quotient t = (k - o) / d as well.
__device__ __forceinline__ bool scan_starts_on(uint32_t kind, double k, double ox, double oy, double oz, double dx, double dy, double dz,
                                               double t_min) {
    const bool k_is_x = kind == TK_YZ_RECT, k_is_z = kind == TK_XY_RECT;
    const bool rect = k_is_x || k_is_z || kind == TK_XZ_RECT;
    const double ok = k_is_x ? ox : k_is_z ? oz : oy, dk = k_is_x ? dx : k_is_z ? dz : dy;
    return rect && fabs(k - ok) < fabs(dk) * (t_min * (1.0 - 1.0e-9));
}

struct ItemHit {
    bool hit;
    double t;
    uint32_t ref;  // ScanItem::ref
};
__device__ __forceinline__ ItemHit scan_item_test(uint32_t lds_items, uint32_t item, double ox, double oy, double oz, double dx, double dy,
                                                  double dz, double t_min, double t_max) {
    const uint32_t rec = lds_items + item * (uint32_t)sizeof(ScanItem);
    const uint2 ref_word = lds_read_u32x2<40>(rec);
    const double a0 = lds_read_f64<0>(rec), a1 = lds_read_f64<8>(rec), b0 = lds_read_f64<16>(rec);
    const double b1 = lds_read_f64<24>(rec), k = lds_read_f64<32>(rec);
    lds_wait();
    ItemHit out;
    out.ref = ref_word.x;
    // rect_hit (device_intersect.h; hittable.rs:185-207, 249-271, 313-335): K = the axis the rect is
    // perpendicular to, A / B = the other two in their order
    const uint32_t kind = (out.ref >> 8) & 7u;
    const bool k_is_x = kind == TK_YZ_RECT, k_is_z = kind == TK_XY_RECT;
    const double ok = k_is_x ? ox : k_is_z ? oz : oy, dk = k_is_x ? dx : k_is_z ? dz : dy;
    const double oa = k_is_x ? oy : ox, da = k_is_x ? dy : dx;
    const double ob = k_is_z ? oy : oz, db = k_is_z ? dy : dz;
    const double t = (k - ok) / dk;
    out.t = t;
    out.hit = !(t < t_min || t > t_max);
    if (out.hit) {
        const double a = oa + t * da;
        const double b = ob + t * db;
        out.hit = !(a < a0 || a > a1 || b < b0 || b > b1);
    }
    return out;
}

// The same interface for a scene of a handful of rects and cubes (the Cornell box has eight shapes): no
// tree at all.  One "shape test" step does the whole query for all lanes that wait for it:
//   1. the culling box of EVERY item (device_scene.h, ScanItem: a rect, or one face of a cube) against the
//      ray, two boxes at a time with the same f32 slab test as a node visit -- the loop is the same for
//      the whole wave, no lane waits for another's walk -- keeping the two smallest entry parameters;
//   2. the exact test of the item reached first, each lane its own.  A rect's box is the rect, as thick
//      as f32 needs it, so the ray nearly always hits that item, and a hit that lies before the entry
//      of every other item is the closest hit.  Rects of the three orientations share one piece of
//      code (the axes are data), so the trip does not split by orientation;
//   3. for the rays that step 2 did not settle (a small share), the exact test of every item.
// Walking a tree of eight shapes costs more in divergence than it saves in tests, and testing all six
// faces of a cube exactly costs six divisions where two faces can be hit at all.  (Spheres are left to
// the tree: a ray that crosses a sphere's box misses the sphere one time in four, so step 2 settles
// too little; measured slower than the tree from four spheres on.)
// The closest hit among all items is then confirmed against the reference's boxes exactly as
// RayWalk does it (confirm / begin_careful).
template <int BLOCK>
struct ScanWalk {
    static constexpr uint32_t SIDE_MASK = 15u, CLASS_SHIFT = 4, TK_SHIFT = 7;
    static constexpr bool PACKED = false;
    RayD r;
    CullRay cull;
    HitRec best;
    bool pending;  // the scan of this ray is still to be done
    bool careful;  // second scan: a hit counts only if its shape's gates pass
    static constexpr uint32_t NO_ITEM = 255u;
    uint32_t from_item;  // the item the ray starts on (a scattered ray), or NO_ITEM
#ifdef MIRT_PATHS_STATS
    unsigned long long stat[4] = {};
    __device__ __forceinline__ void flush_stats() {
        for (int k = 0; k < 4; ++k)
            if (stat[k]) atomicAdd(&g_scan_stats[k], stat[k]);
    }
#endif

    __device__ __forceinline__ void reset() {
        pending = false;
        careful = false;
        from_item = NO_ITEM;
        best = HitRec{0.0, -1, 0};
    }
    __device__ __forceinline__ bool wants_node() const { return false; }
    __device__ __forceinline__ bool wants_leaf() const { return pending; }
    __device__ __forceinline__ unsigned long long vote_node() const { return 0ull; }
    __device__ __forceinline__ unsigned long long vote_leaf() const { return __builtin_amdgcn_ballot_w64(pending); }
    __device__ __forceinline__ bool finished() const { return !pending; }

    __device__ __forceinline__ void begin(const DScene& sc, const RayD& ray, double t_min, double t_max) {
        r = ray;
        // (when a ray may be culled at all: RayWalk::begin)
        const float from = fmaxf(fmaxf(__builtin_fabsf((float)r.o.x), __builtin_fabsf((float)r.o.y)), __builtin_fabsf((float)r.o.z));
        cull.set(r, r.time >= sc.time_lo && r.time <= sc.time_hi && from <= sc.cull_reach &&
                        r.o.x == r.o.x && r.o.y == r.o.y && r.o.z == r.o.z);
        careful = false;
        pending = true;
        // (a scattered ray: `best` still is the hit it starts from)
        from_item = best.prim >= 0 ? (uint32_t)best.side >> 24 : NO_ITEM;
        best = HitRec{0.0, -1, 0};
    }
    __device__ __forceinline__ bool confirm(const DScene& sc, const TraverseLds& lds, double t_min, double t_max) {
        if (careful || best.prim < 0) return true;
        const ShapeRec* rec = reinterpret_cast<const ShapeRec*>(lds.shapes + (uint32_t)best.prim * SHAPE_LDS_BYTES);
        if (cull.sure_to_cross(&rec->sure[0][0], f32_not_below(t_min), f32_not_above(t_max))) return true;
        return gates_pass(sc.gates, sc.gate_of, (uint32_t)best.prim, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max);
    }
    __device__ __forceinline__ void begin_careful(const DScene& sc, double t_min, double t_max) {
        careful = true;
        pending = true;
        best = HitRec{0.0, -1, 0};
    }
    template <class StackT>
    __device__ __forceinline__ void node_step(const DScene&, const TraverseLds&, StackT*, uint32_t, double) {}

    // The exact test of item `item` (each lane its own), and `best` if it wins.
    __device__ __forceinline__ void exact_test(const DScene& sc, const TraverseLds& lds, uint32_t item, double t_min, double t_max) {
        const ItemHit h = scan_item_test((uint32_t)(size_t)lds.items, item, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max);
        // of equal t the later item wins (bvh.rs:70-74 between shapes, hittable.rs:432-445 between the
        // faces of a cube); best.side carries the item's number above the bits a hit's side uses
        const uint32_t index = h.ref & 255u;
        if (h.hit && (best.prim < 0 || h.t < best.t || (h.t == best.t && item > ((uint32_t)best.side >> 24)))) {
            if (!careful || gates_pass(sc.gates, sc.gate_of, index, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max))
                best = HitRec{h.t, (int32_t)index, (int32_t)((h.ref >> 16) | item << 24)};
        }
    }

    __device__ __forceinline__ void leaf_step(const DScene& sc, const TraverseLds& lds, double t_min, double t_max) {
#ifdef MIRT_PATHS_STATS
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- 0. A ray that starts ON a rect (every scattered ray does) crosses that rect's box, which is as
        // thick as f32 needs it, and the exact test then finds t = (k - o) / d below t_min.  That much is
        // known without the division: |k - o| < |d| t_min (1 - 1e-9) implies |t| < t_min for the rounded
        // quotient too.  Such an item is no candidate, or it would be "reached first" by every ray that
        // leaves it.  (Other rects through the ray's origin -- the neighbouring face at a cube's edge --
        // are rare and take the long way below.)
        uint32_t skip = NO_ITEM;
        if (from_item != NO_ITEM) {
            const uint32_t rec = (uint32_t)(size_t)lds.items + from_item * (uint32_t)sizeof(ScanItem);
            const uint2 ref_word = lds_read_u32x2<40>(rec);
            const double k = lds_read_f64<32>(rec);
            lds_wait();
            skip = scan_starts_on((ref_word.x >> 8) & 7u, k, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min) ? from_item : NO_ITEM;
        }
        // ---- 1. which items can the ray hit at all, and which of them does it reach first?
        // An item's entry parameter (slab_pair) is a lower bound of the t of any hit on it.  Entry and
        // item number travel as one key: the bits of the entry (not negative), its lowest six replaced by
        // the number -- which lowers the bound by less than f32 resolves at 2^-17 and orders as the entry does.
        const float t_lo = f32_not_above(t_min), t_hi = f32_not_below(t_max);
        const uint32_t n_items = lds.n_items, n_pairs = (n_items + 1) >> 1;
        uint32_t first = 0xFFFFFFFFu, second = 0xFFFFFFFFu;  // the two smallest keys
        uint32_t entry_bits = ~63u;
        asm volatile("" : "+v"(entry_bits));  // (in a vector register: and-or in one instruction with the item number as its scalar operand)
        // one address per axis: near bound there, far bound 8 bytes behind it (ScanBoxes)
        uint32_t ax = (uint32_t)(size_t)lds.nodes + cull.off[0];
        uint32_t ay = (uint32_t)(size_t)lds.nodes + cull.off[1] + 24u;
        uint32_t az = (uint32_t)(size_t)lds.nodes + cull.off[2] + 48u;
        for (uint32_t pair = 0; pair < n_pairs; ++pair) {
            const f32x2 nx = lds_read_f32x2<0>(ax), fx = lds_read_f32x2<8>(ax);
            const f32x2 ny = lds_read_f32x2<0>(ay), fy = lds_read_f32x2<8>(ay);
            const f32x2 nz = lds_read_f32x2<0>(az), fz = lds_read_f32x2<8>(az);
            lds_wait();
            ax += (uint32_t)sizeof(ScanBoxes), ay += (uint32_t)sizeof(ScanBoxes), az += (uint32_t)sizeof(ScanBoxes);
            float e0, e1;
            bool pass0, pass1;
            slab_pair(nx, fx, ny, fy, nz, fz, cull, t_lo, t_hi, &e0, &e1, &pass0, &pass1);
            const uint32_t item0 = 2u * pair, item1 = 2u * pair + 1u;
            pass0 = pass0 && skip != item0;
            pass1 = pass1 && skip != item1 && item1 < n_items;  // (an odd count: the last box is there twice)
            // (a NaN entry -- nothing constrains the ray -- becomes 0, "before everything": the maximum of a NaN
            // and a number is the number; slab_pair's results come out of max3 and need no quieting first)
            asm("v_max_f32 %0, 0, %0" : "+v"(e0));
            asm("v_max_f32 %0, 0, %0" : "+v"(e1));
            const uint32_t key0 = pass0 ? (__float_as_uint(e0) & entry_bits) | item0 : 0xFFFFFFFFu;
            const uint32_t key1 = pass1 ? (__float_as_uint(e1) & entry_bits) | item1 : 0xFFFFFFFFu;
            second = max(min(first, second), min(max(first, second), key0));  // the median: second smallest of the three
            first = min(first, key0);
            second = max(min(first, second), min(max(first, second), key1));
            first = min(first, key1);
        }
        const uint32_t first_item = first & 63u;
        const float second_entry = __uint_as_float(second & ~63u);
        // ---- 2. the exact test of the item reached first: a hit there that lies before every other item's
        // entry is the closest hit (and so is none at all when no box is crossed)
#ifdef MIRT_PATHS_STATS
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime();
        unsigned long long trips = 0;
#endif
        bool settled = first == 0xFFFFFFFFu;
        if (!settled) {
#ifdef MIRT_PATHS_STATS
            trips += mbcnt_lt(__builtin_amdgcn_ballot_w64(true)) == 0 ? 1 : 0;
#endif
            exact_test(sc, lds, first_item, t_min, t_max);
            settled = second == 0xFFFFFFFFu || (best.prim >= 0 && best.t < (double)second_entry);
        }
        // ---- 3. otherwise (the box was crossed and the item missed, or two items lie within rounding of each
        // other; a small share of the rays) the long way: the exact test of every item, in order
        if (!settled) {
            best = HitRec{0.0, -1, 0};
            for (uint32_t item = 0; item < n_items; ++item) {
#ifdef MIRT_PATHS_STATS
                trips += mbcnt_lt(__builtin_amdgcn_ballot_w64(true)) == 0 ? 1 : 0;
#endif
                exact_test(sc, lds, item, t_min, t_max);
            }
        }
#ifdef MIRT_PATHS_STATS
        {
            const unsigned long long clk2 = __builtin_amdgcn_s_memtime();
            const bool leader = mbcnt_lt(__builtin_amdgcn_ballot_w64(true)) == 0;
            if (leader) {
                stat[0] += clk1 - clk0;
                stat[1] += clk2 - clk1;
                stat[3] += 1ull;
            }
            stat[2] += trips;
        }
#endif
        pending = false;
    }
};

// Per-wave persistent traversal of a queue of rays.
//   Io::count()                      number of rays in the queue
//   Io::load(i, &ray, &cookie)       ray i of the queue
//   Io::store(i, cookie, t, prim, side, cls)   its closest hit (prim < 0: none) and the shade class of the shape
//   work_counter                     device word, zero at launch: next unclaimed ray
//   stack                            LDS, [level][BLOCK] entries of StackT
//   tuning                           bits 0-7 `stay`: the wave stays in the node loop (or the
//                                    leaf loop) while at least stay/8 of its busy lanes want
//                                    that step (0: until none does); bits 8-15: number of
//                                    lanes without a ray that triggers a refill
//   stats (STATS builds only)        STAT_COUNT device counters
// The wave runs one of three pieces of code at a time -- node visits, shape tests, refill --
// for the lanes that want it.  Measured on RTIOW queues: lanes active per vector instruction
// 59 %; leaving a loop early when few lanes want it and refilling only when the whole wave is
// idle were each worth a few per cent; a scheduler that re-decides after every single step
// lost 20 % to its own overhead.
template <int BLOCK, class StackT, bool NODES_IN_LDS, bool SHAPES_IN_LDS, bool STATS, class Io>
__device__ __forceinline__ void traverse_persistent(const DScene& sc, Io& io, uint32_t* __restrict__ work_counter,
                                                    double t_min, double t_max, TraverseLds lds,
                                                    StackT* __restrict__ stack, uint32_t tuning,
                                                    unsigned long long* stats) {
    unsigned long long st[STAT_COUNT] = {};
    auto now = [&]() -> unsigned long long { return STATS ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_begin = now();
    // wave-level events are counted by the first active lane of the moment and summed over
    // the lanes at the end (a lane's own copy only sees the trips it took part in)
    auto leader = [&]() -> bool { return mbcnt_lt(__ballot(true)) == 0; };
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t n = io.count();
    const uint32_t chunk = chunk_rays(n, gridDim.x * (BLOCK / 64));

    // wave-uniform queue window
    uint32_t next = 0, chunk_end = 0;
    bool exhausted = false;

    // the ray this lane will trace next, requested ahead of time
    bool pf_valid = false;
    uint32_t pf_i = 0, pf_cookie = 0;
    RayD pf_ray{};

    // the ray this lane is tracing
    bool active = false;
    uint32_t i = 0, cookie = 0;
    RayWalk<BLOCK, StackT, NODES_IN_LDS, SHAPES_IN_LDS> walk;
    walk.reset();

    // Requests the next ray for every lane that holds none, while the queue lasts.
    auto prefetch = [&]() {
        unsigned long long need = __ballot(!pf_valid);
        while (need && !exhausted) {
            if (next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n) {
                    exhausted = true;
                    break;
                }
                next = base;
                chunk_end = min(base + chunk, n);
            }
            const uint32_t avail = chunk_end - next;
            const uint32_t rank = mbcnt_lt(need);
            if (!pf_valid && rank < avail) {
                pf_i = next + rank;
                io.load(pf_i, &pf_ray, &pf_cookie);
                pf_valid = true;
            }
            next += min((uint32_t)__popcll(need), avail);
            need = __ballot(!pf_valid);
        }
    };

    const int stay = (int)(tuning & 0xFFu), refill_when_idle = (int)((tuning >> 8) & 0xFFu);
    prefetch();
    for (;;) {
        if (STATS && leader()) st[STAT_OUTER_ITERS] += 1;
        const unsigned long long t0 = now();
        // finished rays leave at once
        if (active && walk.finished()) {
            if (walk.confirm(sc, lds, t_min, t_max)) {
                io.store(i, cookie, walk.best.t, walk.best.prim, walk.best.side & (int32_t)walk.SIDE_MASK,
                         ((uint32_t)walk.best.side >> walk.CLASS_SHIFT) & 7u);
                active = false;
            } else {
                walk.begin_careful(sc, t_min, t_max);  // rare: the winner's gate failed
            }
        }
        const int n_node = __popcll(__ballot(active && walk.wants_node()));
        const int n_leaf = __popcll(__ballot(active && walk.wants_leaf()));
        const int n_ready = __popcll(__ballot(!active && pf_valid));
        const unsigned long long t1 = now();
        if (STATS && leader()) st[STAT_CYC_STORE] += t1 - t0;
        if (n_node + n_leaf == 0 || n_ready >= refill_when_idle) {
            if (n_ready == 0) break;  // nothing in flight, nothing ready: the queue is drained
            if (!active && pf_valid) {
                i = pf_i;
                cookie = pf_cookie;
                pf_valid = false;
                active = true;
                walk.begin(sc, pf_ray, t_min, t_max);
            }
            prefetch();  // request the rays after these; they arrive while the wave traverses
            if (STATS && leader()) {
                st[STAT_REFILL_ITERS] += 1;
                st[STAT_REFILL_LANES] += n_ready;
                st[STAT_CYC_REFILL] += now() - t1;
            }
            continue;
        }
        const int enough = max(1, ((n_node + n_leaf) * stay) >> 3);
        if (n_leaf > n_node) {
            int n_now = n_leaf;
            do {
                if (STATS && leader()) {
                    st[STAT_LEAF_ITERS] += 1;
                    st[STAT_LEAF_LANES] += n_now;
                }
                if (active && walk.wants_leaf()) {
                    if (STATS) st[STAT_SHAPE_TESTS] += 1;
                    walk.leaf_step(sc, lds, t_min, t_max);
                }
                n_now = __popcll(__ballot(active && walk.wants_leaf()));
            } while (n_now >= enough);
            if (STATS && leader()) st[STAT_CYC_LEAF] += now() - t1;
        } else {
            int n_now = n_node;
            do {
                if (STATS && leader()) {
                    st[STAT_NODE_ITERS] += 1;
                    st[STAT_NODE_LANES] += n_now;
                }
                if (active && walk.wants_node()) walk.node_step(sc, lds, stack, tid, t_min);
                n_now = __popcll(__ballot(active && walk.wants_node()));
            } while (n_now >= enough);
            if (STATS && leader()) st[STAT_CYC_NODE] += now() - t1;
        }
    }
    if (STATS && leader()) st[STAT_CYC_TOTAL] = now() - t_begin;
    if (STATS) {
        for (int k = 0; k < STAT_COUNT; ++k) {
            unsigned long long v = st[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) atomicAdd(stats + k, v);
        }
    }
}

}  // namespace mirt
