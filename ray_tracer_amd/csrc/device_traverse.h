// device_traverse.h -- the closest-hit engine of the extend kernel: persistent
// waves that walk the two-boxes-per-node tree of accel_build.h in "while-while"
// form, 64 rays per wave, refilling finished lanes from the ray queue.
//
// Why this shape (measured on the first, one-ray-per-lane-to-completion kernel):
// only 31 % of the lanes were active in the average vector instruction, because
// rays of one wave need very different numbers of node visits and because the
// sphere/rect tests ran inside the node loop for the few lanes that had reached a
// leaf; and the vector ALU was busy 23 % of the time, the rest spent waiting on
// 128-byte node fetches that hit 64 different L1 lines per wave instruction.
// Hence:
//  * inner loop = node visits only; a lane that meets leaves parks them in two
//    registers and leaves the loop; leaf tests run after it for all parked lanes;
//  * a lane whose ray is finished takes the next ray of the wave's chunk as soon as
//    enough lanes are idle, instead of idling until the slowest ray is done;
//  * when the tree fits, its nodes are staged in LDS (104 B each) and the slab test
//    reads the near/far bound of each axis by address (8-byte ds_read at
//    base + 8 * sign) instead of loading both and selecting;
//  * the per-lane stack lives in LDS as [level][thread] and holds inner-node
//    indices only (16-bit when the tree has < 65536 nodes).
//
// Decisions are those of bvh_closest_fast / the reference (device_intersect.h,
// accel_build.h): exact slab test with the caller's (t_min, t_max) at every box
// that admits something, shapes tested with the caller's (t_min, t_max), smallest
// t wins and equal t goes to the larger shape index, extra distance prune with a
// relative margin.
#pragma once

#include <hip/hip_runtime.h>

#include "device_intersect.h"

namespace mirt {

__device__ __forceinline__ uint32_t mbcnt_lt(unsigned long long mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

constexpr uint32_t CUR_DONE = 0x7FFFFFFFu;
constexpr int NODE_LDS_BYTES = 104;  // box[2][3][2] + child[2] of a DNode
constexpr int REFILL_WHEN_IDLE = 20; // refill as soon as this many lanes have no ray
constexpr uint32_t CHUNK_RAYS = 512; // most rays a wave takes from the queue per fetch

// Rays per fetch for a queue of n rays shared by `waves` persistent waves: large chunks
// for long queues (few atomics), down to one wave-load for short ones (a late bounce has
// a few thousand live paths: they must spread over many waves, not queue up behind a few).
__host__ __device__ inline uint32_t chunk_rays(uint32_t n, uint32_t waves) {
    uint32_t c = n / (waves * 4u);
    c = c < 64u ? 64u : (c > CHUNK_RAYS ? CHUNK_RAYS : c);
    return c & ~63u;
}

// Slab test of child c of a node whose bytes start at `node` (LDS or global).
// near/far bound of each axis picked by address: off[a] = 8 if 1/d[a] < 0 else 0.
template <class NodePtr>
__device__ __forceinline__ bool slab_by_address(NodePtr node, int c, const RayD& r, const double inv[3],
                                                const uint32_t off[3], double t_min, double t_max, double* entry) {
    const char* b = reinterpret_cast<const char*>(node) + c * 48;
    const double lo_x = *reinterpret_cast<const double*>(b + off[0]);
    const double hi_x = *reinterpret_cast<const double*>(b + (off[0] ^ 8u));
    const double lo_y = *reinterpret_cast<const double*>(b + 16 + off[1]);
    const double hi_y = *reinterpret_cast<const double*>(b + 16 + (off[1] ^ 8u));
    const double lo_z = *reinterpret_cast<const double*>(b + 32 + off[2]);
    const double hi_z = *reinterpret_cast<const double*>(b + 32 + (off[2] ^ 8u));
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

// Per-wave persistent traversal.
//   Io::count()                      number of rays in the queue
//   Io::load(i, &ray, &cookie)       ray i of the queue
//   Io::store(i, cookie, hit)        its closest hit
//   work_counter                     device word, zero at launch: next unclaimed ray
//   lds_nodes                        nodes staged in LDS (NODE_LDS_BYTES apart) or nullptr
//   stack                            LDS, [level][BLOCK] entries of StackT
template <int BLOCK, class StackT, bool NODES_IN_LDS, class Io>
__device__ __forceinline__ void traverse_persistent(const DScene& sc, Io& io, uint32_t* __restrict__ work_counter,
                                                    double t_min, double t_max, const char* lds_nodes,
                                                    StackT* __restrict__ stack) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t n = io.count();
    const uint32_t chunk = chunk_rays(n, gridDim.x * (BLOCK / 64));

    // wave-uniform queue window
    uint32_t next = 0, chunk_end = 0;
    bool exhausted = false;

    // per-lane ray state
    bool active = false;
    uint32_t i = 0, cookie = 0;
    RayD r{};
    double inv[3] = {0, 0, 0};
    uint32_t off[3] = {0, 0, 0};
    double dd = 0.0;
    HitRec best{0.0, -1, 0};
    uint32_t cur = CUR_DONE, pend0 = 0, pend1 = 0;
    int sp = 0;

    auto consider_one = [&](uint32_t index, uint32_t kind) {
        double t;
        int side;
        if (shape_hit(sc.shapes, index, kind, r, dd, t_min, t_max, &t, &side)) {
            // smallest t wins; equal t: larger shape index wins (bvh.rs:70-74)
            if (best.prim < 0 || t < best.t || (t == best.t && (int32_t)index > best.prim)) best = HitRec{t, (int32_t)index, side};
        }
    };
    auto consider = [&](uint32_t ref) {
        const uint32_t index = ref & REF_INDEX_MASK;
        consider_one(index, (ref >> REF_KIND_SHIFT) & 7u);
        if (ref & REF_PAIR) consider_one(index + 1, (ref >> REF_KIND2_SHIFT) & 7u);
    };

    for (;;) {
        // ------------------------------------------------------------ refill
        unsigned long long act = __ballot(active);
        const int n_idle = 64 - __popcll(act);
        if (!exhausted && n_idle >= (act ? REFILL_WHEN_IDLE : 1)) {
            if (next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n) {
                    exhausted = true;
                } else {
                    next = base;
                    chunk_end = min(base + chunk, n);
                }
            }
            if (!exhausted) {
                const uint32_t avail = chunk_end - next;
                const uint32_t rank = mbcnt_lt(~act);
                if (!active && rank < avail) {
                    i = next + rank;
                    io.load(i, &r, &cookie);
                    inv[0] = 1.0 / r.d.x;  // aabb.rs:33, once per ray instead of once per box
                    inv[1] = 1.0 / r.d.y;
                    inv[2] = 1.0 / r.d.z;
                    off[0] = inv[0] < 0.0 ? 8u : 0u;  // aabb.rs:38
                    off[1] = inv[1] < 0.0 ? 8u : 0u;
                    off[2] = inv[2] < 0.0 ? 8u : 0u;
                    dd = dot(r.d, r.d);
                    best = HitRec{0.0, -1, 0};
                    sp = 0;
                    pend0 = pend1 = 0;
                    active = true;
                    const uint32_t root = sc.root_ref;
                    if ((root & REF_LEAF) && !(root & REF_GATED)) {
                        // a one-shape world: bare leaf, no box (bvh.rs:179-180)
                        pend0 = root;
                        cur = CUR_DONE;
                    } else {
                        RayPre p;
                        p.inv[0] = inv[0]; p.inv[1] = inv[1]; p.inv[2] = inv[2];
                        p.sgn[0] = off[0] != 0; p.sgn[1] = off[1] != 0; p.sgn[2] = off[2] != 0;
                        p.dd = dd;
                        double e;
                        if (!slab_test(&sc.root_box[0][0], r, p, t_min, t_max, &e)) {
                            cur = CUR_DONE;
                        } else if (root & REF_LEAF) {
                            pend0 = root;
                            cur = CUR_DONE;
                        } else {
                            cur = root;
                        }
                    }
                }
                next += min((uint32_t)n_idle, avail);
                act = __ballot(active);
            }
        }
        if (act == 0) {
            if (exhausted) break;
            continue;
        }

        // ------------------------------------------------- inner-node visits
        while (active && cur != CUR_DONE && pend0 == 0) {
            uint32_t c0, c1;
            double e0, e1;
            bool pass0, pass1;
            if (NODES_IN_LDS) {
                const char* node = lds_nodes + cur * NODE_LDS_BYTES;
                c0 = *reinterpret_cast<const uint32_t*>(node + 96);
                c1 = *reinterpret_cast<const uint32_t*>(node + 100);
                pass0 = slab_by_address(node, 0, r, inv, off, t_min, t_max, &e0);
                pass1 = slab_by_address(node, 1, r, inv, off, t_min, t_max, &e1);
            } else {
                const char* node = reinterpret_cast<const char*>(sc.nodes + cur);
                c0 = *reinterpret_cast<const uint32_t*>(node + 96);
                c1 = *reinterpret_cast<const uint32_t*>(node + 100);
                pass0 = slab_by_address(node, 0, r, inv, off, t_min, t_max, &e0);
                pass1 = slab_by_address(node, 1, r, inv, off, t_min, t_max, &e1);
            }
            if (best.prim >= 0) {
                // nothing entered farther than the best hit (plus margin) can beat it
                const double limit = best.t + fabs(best.t) * PRUNE_REL;
                const bool far0 = e0 > limit, far1 = e1 > limit;
                // an ungated leaf must still be tested unless distance rules it out as well
                if (far0) pass0 = false;
                if (far1) pass1 = false;
                if (far0 && (c0 & REF_LEAF)) c0 = REF_LEAF | REF_GATED;  // gated + failed = dropped
                if (far1 && (c1 & REF_LEAF)) c1 = REF_LEAF | REF_GATED;
            }
            if (c0 & REF_LEAF) {
                if (pass0 || !(c0 & REF_GATED)) pend0 = c0;
                pass0 = false;
            }
            if (c1 & REF_LEAF) {
                if (pass1 || !(c1 & REF_GATED)) {
                    if (pend0 == 0) pend0 = c1;
                    else pend1 = c1;
                }
                pass1 = false;
            }
            if (pass0 && pass1) {
                const bool near0 = e0 <= e1;
                stack[sp * BLOCK + tid] = (StackT)(near0 ? c1 : c0);
                ++sp;
                cur = near0 ? c0 : c1;
            } else if (pass0) {
                cur = c0;
            } else if (pass1) {
                cur = c1;
            } else if (sp > 0) {
                --sp;
                cur = (uint32_t)stack[sp * BLOCK + tid];
            } else {
                cur = CUR_DONE;
            }
        }

        // ------------------------------------------------------- leaf tests
        if (active && pend0 != 0) {
            consider(pend0);
            if (pend1 != 0) consider(pend1);
            pend0 = pend1 = 0;
        }

        // ------------------------------------------------------------ done
        if (active && cur == CUR_DONE) {
            io.store(i, cookie, best);
            active = false;
        }
    }
}

}  // namespace mirt
