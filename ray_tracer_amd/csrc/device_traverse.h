// device_traverse.h -- closest-hit queries against the two-boxes-per-node tree of
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
//  * 128-byte node fetches from 64 different addresses per wave instruction kept waves in
//    s_waitcnt most of their cycles.  When they fit, the tree's nodes (104 B each) and the
//    shapes' parameters (80 B each) are staged in LDS once per workgroup.
//  * The slab test reads the near / far bound of each axis by address (8-byte read at
//    base + 8 * sign) instead of loading both bounds and selecting.
//  * The per-lane stack lives in LDS as [level][thread] and holds inner-node indices only
//    (16-bit when the tree has < 65536 nodes).
//  * traverse_persistent: waves pull chunks of rays from a global counter and request the
//    next ray of a lane while the current one is traced.
//
// Decisions are the reference's (device_intersect.h, accel_build.h): exact slab test with
// the caller's (t_min, t_max) at every box that admits something, shapes tested with the
// caller's (t_min, t_max), smallest t wins and equal t goes to the larger shape index, extra
// distance prune with a relative margin.
#pragma once

#include <hip/hip_runtime.h>

#include "device_intersect.h"

namespace mirt {

__device__ __forceinline__ uint32_t mbcnt_lt(unsigned long long mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

constexpr uint32_t CUR_DONE = 0x7FFFFFFFu;
constexpr int NODE_LDS_BYTES = 104;   // box[2][3][2] + child[2] of a DNode
constexpr int SHAPE_LDS_BYTES = 80;   // DShape::f
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

// Slab test (aabb.rs:24-51) on the six bounds of a box, already ordered near/far per axis.
__device__ __forceinline__ bool slab_near_far(double lo_x, double hi_x, double lo_y, double hi_y, double lo_z, double hi_z,
                                              const RayD& r, const double inv[3], double t_min, double t_max,
                                              double* entry) {
    const double t0x = (lo_x - r.o.x) * inv[0], t1x = (hi_x - r.o.x) * inv[0];
    const double t0y = (lo_y - r.o.y) * inv[1], t1y = (hi_y - r.o.y) * inv[1];
    const double t0z = (lo_z - r.o.z) * inv[2], t1z = (hi_z - r.o.z) * inv[2];
    double mn = fmax(t0x, t_min);
    double mx = fmin(t1x, t_max);
    mn = fmax(t0y, mn);
    mx = fmin(t1y, mx);
    mn = fmax(t0z, mn);
    mx = fmin(t1z, mx);
    *entry = mn;
    return !(mx <= mn);
}

// Slab test of child c of a node in GLOBAL memory; near/far bound of each axis picked by
// address: off[a] = 8 if 1/d[a] < 0 else 0.
__device__ __forceinline__ bool slab_by_address(const char* node, int c, const RayD& r, const double inv[3],
                                                const uint32_t off[3], double t_min, double t_max, double* entry) {
    const char* b = node + c * 48;
    const double lo_x = *reinterpret_cast<const double*>(b + off[0]);
    const double hi_x = *reinterpret_cast<const double*>(b + (off[0] ^ 8u));
    const double lo_y = *reinterpret_cast<const double*>(b + 16 + off[1]);
    const double hi_y = *reinterpret_cast<const double*>(b + 16 + (off[1] ^ 8u));
    const double lo_z = *reinterpret_cast<const double*>(b + 32 + off[2]);
    const double hi_z = *reinterpret_cast<const double*>(b + 32 + (off[2] ^ 8u));
    return slab_near_far(lo_x, hi_x, lo_y, hi_y, lo_z, hi_z, r, inv, t_min, t_max, entry);
}

// One 8-byte LDS read at byte address `addr` + OFFSET, as exactly one ds_read_b64.
// Written as inline assembly because, left to itself, the compiler pairs neighbouring
// 8-byte LDS reads into ds_read2_b64; that instruction banks modulo 32 in four 16-lane
// groups, and with the 64 lanes of a wave in 64 different nodes it measured ~30 LDS cycles
// per instruction on gfx950 against ~2 for ds_read_b64 -- the node loop became LDS-bound.
// The caller must issue lds_wait() before using the value.
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
    const char* nodes;   // NODE_LDS_BYTES apart, or nullptr: read DScene::nodes
    const char* shapes;  // SHAPE_LDS_BYTES apart, or nullptr: read DScene::shapes
};

// The walk of ONE ray (one lane) through the tree, cut into steps so that a wave can
// choose, trip by trip, which step to run for which lanes:
//   begin(ray)            1/d, signs, root box
//   wants_node()          the lane's next step is a visit of inner node `cur`
//   wants_leaf()          the lane has parked shapes to test first
//   node_step()           both children's boxes of `cur`; parks leaves; moves on or pops
//   leaf_step()           tests one parked shape
//   finished()            neither: `best` is the closest hit (prim < 0: none)
// stack: LDS, [level][BLOCK] entries of StackT, this lane's column is `tid`.
template <int BLOCK, class StackT, bool NODES_IN_LDS, bool SHAPES_IN_LDS>
struct RayWalk {
    static constexpr uint32_t Q_KIND_SHIFT = 26;
    RayD r;
    double inv[3];
    uint32_t off[3];
    double dd;
    HitRec best;
    // no box entered beyond `limit` can hold a hit that beats the best one: best.t plus a
    // margin, +inf while there is no hit
    double limit;
    uint32_t cur;
    int sp;
    // shapes parked for the next leaf steps: TraverseKind << 26 | index, at most four (two
    // leaf references of up to two shapes each), kept in registers q0 (next out) .. q3
    uint32_t q0, q1, q2, q3, nq;

    __device__ __forceinline__ void reset() {
        cur = CUR_DONE;
        sp = 0;
        nq = 0;
        q0 = q1 = q2 = q3 = 0;
        best = HitRec{0.0, -1, 0};
        limit = __builtin_inf();
    }
    __device__ __forceinline__ bool wants_node() const { return nq == 0 && cur != CUR_DONE; }
    __device__ __forceinline__ bool wants_leaf() const { return nq != 0; }
    __device__ __forceinline__ bool finished() const { return nq == 0 && cur == CUR_DONE; }

    __device__ __forceinline__ void park_shape(uint32_t entry) {
        q3 = q2;
        q2 = q1;
        q1 = q0;
        q0 = entry;
        ++nq;
    }
    __device__ __forceinline__ void park(uint32_t ref) {
        const uint32_t index = ref & REF_INDEX_MASK;
        park_shape((((ref >> REF_KIND_SHIFT) & 7u) << Q_KIND_SHIFT) | index);
        if (ref & REF_PAIR) park_shape((((ref >> REF_KIND2_SHIFT) & 7u) << Q_KIND_SHIFT) | (index + 1));
    }

    __device__ __forceinline__ void begin(const DScene& sc, const RayD& ray, double t_min, double t_max) {
        r = ray;
        inv[0] = 1.0 / r.d.x;  // aabb.rs:33, once per ray instead of once per box
        inv[1] = 1.0 / r.d.y;
        inv[2] = 1.0 / r.d.z;
        off[0] = inv[0] < 0.0 ? 8u : 0u;  // aabb.rs:38
        off[1] = inv[1] < 0.0 ? 8u : 0u;
        off[2] = inv[2] < 0.0 ? 8u : 0u;
        dd = dot(r.d, r.d);
        reset();
        const uint32_t root = sc.root_ref;
        if ((root & REF_LEAF) && !(root & REF_GATED)) {
            // a one-shape world: bare leaf, no box (bvh.rs:179-180)
            park(root);
        } else {
            RayPre p;
            p.inv[0] = inv[0]; p.inv[1] = inv[1]; p.inv[2] = inv[2];
            p.sgn[0] = off[0] != 0; p.sgn[1] = off[1] != 0; p.sgn[2] = off[2] != 0;
            p.dd = dd;
            double e;
            if (slab_test(&sc.root_box[0][0], r, p, t_min, t_max, &e)) {
                if (root & REF_LEAF) park(root);
                else cur = root;
            }
        }
    }

    __device__ __forceinline__ void node_step(const DScene& sc, const TraverseLds& lds, StackT* __restrict__ stack, uint32_t tid,
                                              double t_min, double t_max) {
        uint32_t c0, c1;
        double e0, e1;
        bool pass0, pass1;
        if (NODES_IN_LDS) {
            // thirteen 8-byte reads of one node, issued back to back, one wait
            // (LDS byte address = the low half of the generic address of an LDS object)
            const uint32_t node = (uint32_t)(size_t)lds.nodes + cur * NODE_LDS_BYTES;
            const uint32_t ax = node + off[0], bx = node + (off[0] ^ 8u);
            const uint32_t ay = node + off[1], by = node + (off[1] ^ 8u);
            const uint32_t az = node + off[2], bz = node + (off[2] ^ 8u);
            const uint2 refs = lds_read_u32x2<96>(node);
            const double l0x = lds_read_f64<0>(ax), h0x = lds_read_f64<0>(bx);
            const double l0y = lds_read_f64<16>(ay), h0y = lds_read_f64<16>(by);
            const double l0z = lds_read_f64<32>(az), h0z = lds_read_f64<32>(bz);
            const double l1x = lds_read_f64<48>(ax), h1x = lds_read_f64<48>(bx);
            const double l1y = lds_read_f64<64>(ay), h1y = lds_read_f64<64>(by);
            const double l1z = lds_read_f64<80>(az), h1z = lds_read_f64<80>(bz);
            lds_wait();
            c0 = refs.x;
            c1 = refs.y;
            pass0 = slab_near_far(l0x, h0x, l0y, h0y, l0z, h0z, r, inv, t_min, t_max, &e0);
            pass1 = slab_near_far(l1x, h1x, l1y, h1y, l1z, h1z, r, inv, t_min, t_max, &e1);
        } else {
            const char* node = reinterpret_cast<const char*>(sc.nodes + cur);
            c0 = *reinterpret_cast<const uint32_t*>(node + 96);
            c1 = *reinterpret_cast<const uint32_t*>(node + 100);
            pass0 = slab_by_address(node, 0, r, inv, off, t_min, t_max, &e0);
            pass1 = slab_by_address(node, 1, r, inv, off, t_min, t_max, &e1);
        }
        // nothing entered farther than the best hit (plus margin) can beat it
        const bool near0 = !(e0 > limit), near1 = !(e1 > limit);
        const bool leaf0 = (c0 & REF_LEAF) != 0, leaf1 = (c1 & REF_LEAF) != 0;
        // a gated leaf is tested iff its box passes; an ungated one whatever its box says
        if (leaf0 && near0 && (pass0 || !(c0 & REF_GATED))) park(c0);
        if (leaf1 && near1 && (pass1 || !(c1 & REF_GATED))) park(c1);
        const bool go0 = pass0 && near0 && !leaf0;
        const bool go1 = pass1 && near1 && !leaf1;
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
        const uint32_t entry = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        --nq;
        const uint32_t index = entry & REF_INDEX_MASK;
        const uint32_t tk = entry >> Q_KIND_SHIFT;
        const double* p = SHAPES_IN_LDS ? reinterpret_cast<const double*>(lds.shapes + index * SHAPE_LDS_BYTES)
                                        : sc.tparams + (size_t)index * TPARAMS_PER_SHAPE;
        double t;
        int side;
        if (tshape_hit(p, tk, r, dd, t_min, t_max, &t, &side)) {
            // smallest t wins; equal t: larger shape index wins (bvh.rs:70-74)
            if (best.prim < 0 || t < best.t || (t == best.t && (int32_t)index > best.prim)) {
                best = HitRec{t, (int32_t)index, side};
                limit = t + fabs(t) * PRUNE_REL;
            }
        }
    }
};

// Per-wave persistent traversal of a queue of rays.
//   Io::count()                      number of rays in the queue
//   Io::load(i, &ray, &cookie)       ray i of the queue
//   Io::store(i, cookie, hit)        its closest hit
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
            io.store(i, cookie, walk.best);
            active = false;
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
                if (active && walk.wants_node()) walk.node_step(sc, lds, stack, tid, t_min, t_max);
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

// The same walk for ONE ray per lane, all lanes of the wave at once, no queue: the wave
// alternates node visits and shape tests, whichever more lanes are waiting for, until
// every lane with `have` set has its closest hit in walk.best.
template <int BLOCK, class StackT, bool NODES_IN_LDS, bool SHAPES_IN_LDS>
__device__ __forceinline__ void traverse_wave(const DScene& sc, bool have, const RayD& ray, double t_min, double t_max,
                                              TraverseLds lds, StackT* __restrict__ stack, uint32_t tid,
                                              RayWalk<BLOCK, StackT, NODES_IN_LDS, SHAPES_IN_LDS>& walk) {
    walk.reset();
    if (have) walk.begin(sc, ray, t_min, t_max);
    for (;;) {
        const bool wn = have && walk.wants_node(), wl = have && walk.wants_leaf();
        const int n_node = __popcll(__ballot(wn)), n_leaf = __popcll(__ballot(wl));
        if (n_node + n_leaf == 0) break;
        if (n_leaf > n_node) {
            if (wl) walk.leaf_step(sc, lds, t_min, t_max);
        } else {
            if (wn) walk.node_step(sc, lds, stack, tid, t_min, t_max);
        }
    }
}

}  // namespace mirt
