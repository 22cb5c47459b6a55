// Whole paths, regrouped across the waves of a workgroup by what they wait for.
//
// k_paths gives a path one lane from its camera ray to its end; a wave runs one piece of code at a
// time, so lanes whose paths wait for another piece sit idle (about 40 % of them).  Here the waves of
// the workgroup (one per CU) take fixed roles, and paths move between them through two rings in LDS:
//   walking waves   hold one walk per lane (RayWalk); a finished walk with a hit goes to the HIT ring,
//                   one without ends its path there and then (a store) and raises the count of paths
//                   to be made; empty lanes take the next ray from the RAY ring
//   shading waves   take 64 hits at a time, shade them with all 64 lanes (shade_path), put the rays of
//                   the surviving paths into the RAY ring and count the ended ones; and make camera
//                   rays, 64 at a time, for the count of paths to be made
// A ring: entries numbered n = 0, 1, ... live in slot n % CAP; `tail` counts entries reserved by
// writers, `head` entries reserved by readers, and seq[slot] says whose turn the slot is: 2r = free for
// the entry of round r = n / CAP, 2r + 1 = that entry is written.  A writer reserves only while
// tail - head < CAP, so whoever still reads its slot has at least reserved it and will free it shortly.
// No wave ever waits with a full ring behind it on both sides: walking waves never block on a ring
// (they try, and go on walking), and the paths in flight are fewer than the places outside the rings
// (n_total <= lanes of the walking waves + one batch), so a shading wave that waits for room in the
// RAY ring is waited for by walking lanes that are empty and will take from it.
// Results are those of k_paths: a path's random numbers depend on (pixel, sample, bounce) only.
// State of this kernel: correct (same images, same numbers of queries), not yet faster.  Measured on
// RTIOW: vector-instruction issue busy 85 % (k_paths: 95 %), 39 of 64 lanes active per instruction
// (38), 101 vector instructions per path and wave (95).  The waves of a SIMD share its issue slots
// equally whatever their role, so the roles must be split in the ratio of their work (about 6.6
// shading waves of 16 on RTIOW), and the walking lanes only stay full with several batches more in
// flight than they hold, for which these rings (LDS is nearly full) are too small.
#pragma once

namespace mirt {

constexpr uint32_t RING_CAP = 128;            // entries per ring (power of two)
constexpr uint32_t RAY_FIELDS = 12, HIT_FIELDS = 14;  // 8-byte fields per entry, field-major: [field][slot]
enum { CTL_RAY_TAIL, CTL_RAY_HEAD, CTL_HIT_TAIL, CTL_HIT_HEAD, CTL_TO_MAKE, CTL_IN_FLIGHT, CTL_NEXT, CTL_N };

__host__ __device__ constexpr size_t regroup_ring_bytes() { return (size_t)RING_CAP * 8u * (RAY_FIELDS + HIT_FIELDS); }

// (volatile: other waves write these words)
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

// Reserves `count` entries for writing if the ring has room; the first entry's number, or ~0u.  Wave-uniform.
__device__ __forceinline__ uint32_t ring_reserve_write(uint32_t* tail, const uint32_t* head, uint32_t count, uint32_t lane) {
    uint32_t first = ~0u;
    if (lane == 0) {
        for (;;) {
            const uint32_t t = lds_peek(tail), h = lds_peek(head);
            if (t + count - h > RING_CAP) break;
            if (atomicCAS(tail, t, t + count) == t) {
                first = t;
                break;
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(first);
}
// Reserves up to `want` written-or-reserved entries for reading, at least `at_least`; *first = the first
// entry's number; returns how many (0: not enough there).  Wave-uniform.
__device__ __forceinline__ uint32_t ring_reserve_read(const uint32_t* tail, uint32_t* head, uint32_t want, uint32_t at_least,
                                                      uint32_t lane, uint32_t* first) {
    uint32_t got = 0, from = 0;
    if (lane == 0) {
        for (;;) {
            const uint32_t h = lds_peek(head), t = lds_peek(tail);
            const uint32_t there = t - h;
            const uint32_t take = there < want ? there : want;
            if (take == 0 || take < at_least) break;
            if (atomicCAS(head, h, h + take) == h) {
                got = take;
                from = h;
                break;
            }
        }
    }
    *first = __builtin_amdgcn_readfirstlane(from);
    return __builtin_amdgcn_readfirstlane(got);
}
__device__ __forceinline__ void ring_wait(const uint32_t* seq, uint32_t slot, uint32_t value) {
    while (lds_peek(seq + slot) != value) __builtin_amdgcn_s_sleep(1);
}

template <int TEXTURES>
__global__ __launch_bounds__(paths_block(TEXTURES)) __attribute__((amdgpu_waves_per_eu(paths_waves(TEXTURES)))) void k_paths_regroup(
    DScene sc, Paths P, RenderParams rp, uint32_t stack_levels, uint32_t n_walking, uint32_t n_total, uint32_t tuning) {
    constexpr int PATHS_BLOCK = paths_block(TEXTURES);
    constexpr bool HAS_NOISE = TEXTURES >= TEXTURES_ALL;
    typedef uint16_t StackT;
    extern __shared__ __align__(16) char smem[];
    __shared__ mirt_perlin s_perlin[HAS_NOISE ? PERLIN_STAGED : 1];
    __shared__ mirt_texture s_textures[TEXTURES > TEXTURES_CONSTANT ? TEXTURES_STAGED : 1];
    __shared__ unsigned long long s_scratch[PATHS_BLOCK / 64][IN_SPHERE_SCRATCH];
    __shared__ uint32_t s_ctl[CTL_N], s_ray_seq[RING_CAP], s_hit_seq[RING_CAP];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_staged = HAS_NOISE && sc.perlins ? (uint32_t)PERLIN_STAGED : 0u;
    if (n_staged) stage_perlin<PATHS_BLOCK>(sc, s_perlin);
    const mirt_perlin* lds_tables = HAS_NOISE ? s_perlin : nullptr;
    const mirt_texture* textures = sc.shade_textures;
    if (TEXTURES > TEXTURES_CONSTANT && sc.n_shade_textures <= (uint32_t)TEXTURES_STAGED) {
        const uint32_t words = sc.n_shade_textures * (uint32_t)(sizeof(mirt_texture) / 4);
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.shade_textures);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_textures);
        for (uint32_t w = tid; w < words; w += PATHS_BLOCK) dst[w] = src[w];
        textures = s_textures;
    }
    // this workgroup's share of the pass: paths [range_begin, range_end) (index i = sample i / n_pixels of pixel
    // i % n_pixels: every share holds whole frames' worth of pixels, so the shares cost the same)
    const uint32_t per_group = (rp.n_paths + gridDim.x - 1) / gridDim.x;
    const uint32_t range_begin = min(blockIdx.x * per_group, rp.n_paths), range_end = min(range_begin + per_group, rp.n_paths);
    if (tid < CTL_N) s_ctl[tid] = tid == CTL_TO_MAKE ? n_total : tid == CTL_NEXT ? range_begin : 0u;
    if (tid < RING_CAP) s_ray_seq[tid] = s_hit_seq[tid] = 0u;
    StackT* stack = reinterpret_cast<StackT*>(smem);
    char* lds_nodes = smem + trav_stack_bytes(stack_levels, sizeof(StackT), PATHS_BLOCK);
    char* lds_shapes = lds_nodes + trav_nodes_bytes(sc.n_nodes);
    unsigned long long* ray_ring = reinterpret_cast<unsigned long long*>(lds_shapes + (((size_t)sc.n_shapes * SHAPE_LDS_BYTES + 15) & ~(size_t)15));
    unsigned long long* hit_ring = ray_ring + RAY_FIELDS * RING_CAP;
    typedef RayWalk<PATHS_BLOCK, StackT, true, true, false, 4> Walk;
    stage_tree<PATHS_BLOCK, true, true, Walk::PACKED>(sc, lds_nodes, lds_shapes, sc.n_nodes);  // (ends with a barrier)
    const TraverseLds lds{lds_nodes, lds_shapes, sc.n_nodes};
    const V3 background = v3(rp.background[0], rp.background[1], rp.background[2]);
    uint32_t segments = 0;  // closest-hit queries answered by this wave (the same in every lane)
    auto as_bits = [](double x) { return (unsigned long long)__double_as_longlong(x); };
    auto as_double = [](unsigned long long u) { return __longlong_as_double((long long)u); };
    // (a maker counts its paths as in flight BEFORE it takes their indices, and these two are read in this order)
    auto exhausted = [&]() { return __atomic_load_n(s_ctl + CTL_NEXT, __ATOMIC_ACQUIRE) >= range_end; };
    auto done = [&]() { return exhausted() && __atomic_load_n(s_ctl + CTL_IN_FLIGHT, __ATOMIC_ACQUIRE) == 0u; };

    if (wave < n_walking) {
        // ------------------------------------------------------------------ a walking wave
        const int stay = (int)(tuning & 0xFFu), push_at = (int)((tuning >> 8) & 0xFFu), pop_at = (int)((tuning >> 16) & 0xFFu);
        bool have = false;
        uint32_t slot = 0, bounce = 0;
        V3 beta = v3(0.0, 0.0, 0.0);
        uint64_t stream = 0;
        Walk walk;
        walk.reset();
        for (;;) {
            unsigned long long m_have = vote(have), m_node = walk.vote_node(), m_leaf = walk.vote_leaf();
            const unsigned long long m_finished = m_have & ~(m_node | m_leaf);
            const int busy = (int)__popcll(m_node | m_leaf);
            bool changed = false;
            if (m_finished && ((int)__popcll(m_finished) >= push_at || busy == 0)) {
                // finished walks leave: a confirmed hit into the HIT ring, a miss ends its path here
                const bool finished = have && walk.finished();
                if (finished && !walk.confirm(sc, lds, rp.t_min, rp.t_max)) walk.begin_careful(sc, rp.t_min, rp.t_max);
                const bool leaving = have && walk.finished();
                const bool missed = leaving && walk.best.prim < 0, hit = leaving && walk.best.prim >= 0;
                const bool missed_later = missed && bounce != 0u, hit_later = hit && bounce != 0u;  // not a path's first query
                const unsigned long long m_missed = vote(missed), m_hit = vote(hit);
                const uint32_t n_hit = (uint32_t)__popcll(m_hit), n_missed = (uint32_t)__popcll(m_missed);
                if (n_missed) {
                    if (missed) {
                        store_radiance(P, slot, beta * background);
                        have = false;
                        walk.reset();
                    }
                    if (lane == 0) {
                        atomicAdd(s_ctl + CTL_TO_MAKE, n_missed);
                        atomicSub(s_ctl + CTL_IN_FLIGHT, n_missed);
                    }
                    segments += (uint32_t)__popcll(vote(missed_later));
                    changed = true;
                }
                if (n_hit) {
                    const uint32_t first = ring_reserve_write(s_ctl + CTL_HIT_TAIL, s_ctl + CTL_HIT_HEAD, n_hit, lane);
                    if (first != ~0u) {
                        if (hit) {
                            const uint32_t n = first + mbcnt_lt(m_hit), s = n & (RING_CAP - 1u), round = n / RING_CAP;
                            ring_wait(s_hit_seq, s, 2u * round);
                            unsigned long long* e = hit_ring + s;
                            e[0 * RING_CAP] = as_bits(walk.r.o.x); e[1 * RING_CAP] = as_bits(walk.r.o.y); e[2 * RING_CAP] = as_bits(walk.r.o.z);
                            e[3 * RING_CAP] = as_bits(walk.r.d.x); e[4 * RING_CAP] = as_bits(walk.r.d.y); e[5 * RING_CAP] = as_bits(walk.r.d.z);
                            e[6 * RING_CAP] = as_bits(walk.r.time);
                            e[7 * RING_CAP] = as_bits(beta.x); e[8 * RING_CAP] = as_bits(beta.y); e[9 * RING_CAP] = as_bits(beta.z);
                            e[10 * RING_CAP] = stream;
                            e[11 * RING_CAP] = (unsigned long long)slot | ((unsigned long long)bounce << 32);
                            e[12 * RING_CAP] = as_bits(walk.best.t);
                            e[13 * RING_CAP] = (unsigned long long)(uint32_t)walk.best.prim | ((unsigned long long)(uint32_t)walk.best.side << 32);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __atomic_store_n(s_hit_seq + s, 2u * round + 1u, __ATOMIC_RELAXED);
                            have = false;
                            walk.reset();
                        }
                        segments += (uint32_t)__popcll(vote(hit_later));
                        changed = true;
                    }
                }
            }
            if (changed) m_have = vote(have);
            const unsigned long long need = ~m_have;
            if ((int)__popcll(need) >= pop_at || need == ~0ull) {
                uint32_t first;
                const uint32_t got = ring_reserve_read(s_ctl + CTL_RAY_TAIL, s_ctl + CTL_RAY_HEAD, (uint32_t)__popcll(need), 1u, lane, &first);
                const uint32_t rank = mbcnt_lt(need);
                if (!have && rank < got) {
                    const uint32_t n = first + rank, s = n & (RING_CAP - 1u), round = n / RING_CAP;
                    ring_wait(s_ray_seq, s, 2u * round + 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const unsigned long long* e = ray_ring + s;
                    RayD r;
                    r.o = v3(as_double(e[0 * RING_CAP]), as_double(e[1 * RING_CAP]), as_double(e[2 * RING_CAP]));
                    r.d = v3(as_double(e[3 * RING_CAP]), as_double(e[4 * RING_CAP]), as_double(e[5 * RING_CAP]));
                    r.time = as_double(e[6 * RING_CAP]);
                    beta = v3(as_double(e[7 * RING_CAP]), as_double(e[8 * RING_CAP]), as_double(e[9 * RING_CAP]));
                    stream = e[10 * RING_CAP];
                    const unsigned long long sb = e[11 * RING_CAP];
                    slot = (uint32_t)sb;
                    bounce = (uint32_t)(sb >> 32);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __atomic_store_n(s_ray_seq + s, 2u * round + 2u, __ATOMIC_RELAXED);
                    have = true;
                    walk.begin(sc, r, rp.t_min, rp.t_max);
                }
                if (got == 0 && need == ~0ull) {
                    if (done()) break;
                    __builtin_amdgcn_s_sleep(8);
                    continue;
                }
                if (got) continue;  // (the masks are stale)
            }
            if (busy == 0) {
                // nothing to walk, and what is finished found no room: wait for the shading waves
                if (!changed) __builtin_amdgcn_s_sleep(4);
                continue;
            }
            const int n_node = (int)__popcll(m_node), n_leaf = (int)__popcll(m_leaf);
            const int busy_share = (busy * stay) >> 3;
            const int enough = busy_share > 1 ? busy_share : 1;
            if (n_leaf > n_node) {
                bool wants = walk.wants_leaf();
                int n_now;
                do {
                    if (wants) walk.leaf_step(sc, lds, rp.t_min, rp.t_max);
                    wants = walk.wants_leaf();
                    n_now = (int)__popcll(walk.vote_leaf());
                } while (n_now >= enough);
            } else {
                bool wants = walk.wants_node();
                int n_now;
                do {
                    if (wants) walk.node_step(sc, lds, stack, tid, rp.t_min);
                    wants = walk.wants_node();
                    n_now = (int)__popcll(walk.vote_node());
                } while (n_now >= enough);
            }
        }
    } else {
        // ------------------------------------------------------------------ a shading wave
        ShadeClock clock;
        // puts the rays of the lanes with `mine` into the RAY ring (waits for room)
        auto push_rays = [&](bool mine, const RayD& r, const V3& beta, uint64_t stream, uint32_t slot, uint32_t bounce) {
            const unsigned long long m = vote(mine);
            const uint32_t count = (uint32_t)__popcll(m);
            if (count == 0) return;
            uint32_t first;
            while ((first = ring_reserve_write(s_ctl + CTL_RAY_TAIL, s_ctl + CTL_RAY_HEAD, count, lane)) == ~0u) __builtin_amdgcn_s_sleep(4);
            if (mine) {
                const uint32_t k = first + mbcnt_lt(m), s = k & (RING_CAP - 1u), round = k / RING_CAP;
                ring_wait(s_ray_seq, s, 2u * round);
                unsigned long long* e = ray_ring + s;
                e[0 * RING_CAP] = as_bits(r.o.x); e[1 * RING_CAP] = as_bits(r.o.y); e[2 * RING_CAP] = as_bits(r.o.z);
                e[3 * RING_CAP] = as_bits(r.d.x); e[4 * RING_CAP] = as_bits(r.d.y); e[5 * RING_CAP] = as_bits(r.d.z);
                e[6 * RING_CAP] = as_bits(r.time);
                e[7 * RING_CAP] = as_bits(beta.x); e[8 * RING_CAP] = as_bits(beta.y); e[9 * RING_CAP] = as_bits(beta.z);
                e[10 * RING_CAP] = stream;
                e[11 * RING_CAP] = (unsigned long long)slot | ((unsigned long long)bounce << 32);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __atomic_store_n(s_ray_seq + s, 2u * round + 1u, __ATOMIC_RELAXED);
            }
        };
        for (;;) {
            // ---- 64 hits (fewer once no new paths are made and the rings run dry)
            const bool draining = exhausted();
            uint32_t first;
            const uint32_t got = ring_reserve_read(s_ctl + CTL_HIT_TAIL, s_ctl + CTL_HIT_HEAD, 64u, draining ? 1u : 64u, lane, &first);
            if (got) {
                const bool mine = lane < got;
                RayD r{};
                V3 beta = v3(0.0, 0.0, 0.0);
                uint64_t stream = 0;
                uint32_t slot = 0, bounce = 0;
                HitRec h{0.0, -1, 0};
                if (mine) {
                    const uint32_t k = first + lane, s = k & (RING_CAP - 1u), round = k / RING_CAP;
                    ring_wait(s_hit_seq, s, 2u * round + 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const unsigned long long* e = hit_ring + s;
                    r.o = v3(as_double(e[0 * RING_CAP]), as_double(e[1 * RING_CAP]), as_double(e[2 * RING_CAP]));
                    r.d = v3(as_double(e[3 * RING_CAP]), as_double(e[4 * RING_CAP]), as_double(e[5 * RING_CAP]));
                    r.time = as_double(e[6 * RING_CAP]);
                    beta = v3(as_double(e[7 * RING_CAP]), as_double(e[8 * RING_CAP]), as_double(e[9 * RING_CAP]));
                    stream = e[10 * RING_CAP];
                    const unsigned long long sb = e[11 * RING_CAP], ps = e[13 * RING_CAP];
                    slot = (uint32_t)sb;
                    bounce = (uint32_t)(sb >> 32);
                    h.t = as_double(e[12 * RING_CAP]);
                    h.prim = (int32_t)(uint32_t)ps;
                    h.side = (int32_t)(uint32_t)(ps >> 32);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __atomic_store_n(s_hit_seq + s, 2u * round + 2u, __ATOMIC_RELAXED);
                }
                const bool last = bounce + 1 == rp.max_depth;
                const uint32_t cls = ((uint32_t)h.side >> Walk::CLASS_SHIFT) & 7u, tk = ((uint32_t)h.side >> Walk::TK_SHIFT) & 7u;
                const V3 in_sphere = random_in_unit_sphere_wave(mine && wants_in_sphere(cls, last), PathRng{stream, bounce + 1u, 0u}, s_scratch[wave]);
                bool alive = false;
                if (mine) {
                    const uint32_t hit_ref = (uint32_t)h.prim | (((uint32_t)h.side & Walk::SIDE_MASK) << HIT_SIDE_SHIFT);
                    const ShapeRec* rec = reinterpret_cast<const ShapeRec*>(lds.shapes + (uint32_t)h.prim * SHAPE_LDS_BYTES);
                    const ShadeResult s = shade_path<TEXTURES>(sc, cls, r, h.t, hit_ref, tk, rec, textures, beta, in_sphere, stream, bounce, last, background,
                                                               lds_tables, n_staged, clock);
                    alive = s.alive;
                    if (alive) {
                        r.o = s.origin;
                        r.d = s.direction;
                        beta = s.value;
                        ++bounce;
                    } else {
                        store_radiance(P, slot, s.value);
                    }
                }
                const uint32_t n_dead = got - (uint32_t)__popcll(vote(alive));
                if (n_dead && lane == 0) {
                    atomicAdd(s_ctl + CTL_TO_MAKE, n_dead);
                    atomicSub(s_ctl + CTL_IN_FLIGHT, n_dead);
                }
                push_rays(alive, r, beta, stream, slot, bounce);
                continue;
            }
            // ---- 64 camera rays for paths that ended (or have not begun)
            const uint32_t to_make = lds_peek(s_ctl + CTL_TO_MAKE);
            if (!draining && (to_make >= 64u || (to_make > 0u && lds_peek(s_ctl + CTL_HIT_TAIL) == lds_peek(s_ctl + CTL_HIT_HEAD)))) {
                uint32_t take = 0, base = 0;
                if (lane == 0) {
                    for (;;) {
                        const uint32_t v = lds_peek(s_ctl + CTL_TO_MAKE);
                        take = v < 64u ? v : 64u;
                        if (take == 0 || atomicCAS(s_ctl + CTL_TO_MAKE, v, v - take) == v) break;
                    }
                    if (take) {
                        atomicAdd(s_ctl + CTL_IN_FLIGHT, take);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        base = atomicAdd(s_ctl + CTL_NEXT, take);
                    }
                }
                take = __builtin_amdgcn_readfirstlane(take);
                base = __builtin_amdgcn_readfirstlane(base);
                if (take == 0) continue;
                const uint32_t made = base < range_end ? min(take, range_end - base) : 0u;
                // (what the share did not hold any more is never made)
                if (lane == 0 && made < take) atomicSub(s_ctl + CTL_IN_FLIGHT, take - made);
                const bool mine = lane < made;
                RayD r{};
                uint64_t stream = 0;
                const uint32_t slot = base + lane;
                if (mine) {
                    const uint32_t idx = rp.px.idx(slot / rp.n_samples);  // (numbered pixel by pixel, as k_paths does)
                    PathRng rng{mirt_rng_stream(rp.seed, idx, rp.sample_base + slot % rp.n_samples), 0u, 0u};
                    r = camera_ray(rp.cam, rp.width, rp.height, idx, rng);
                    stream = rng.stream;
                }
                push_rays(mine, r, v3(1.0, 1.0, 1.0), stream, slot, 0u);
                continue;
            }
            if (done()) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (lane == 0 && segments) {
        // every closest-hit query beyond a path's first (k_accumulate counts one per path)
        atomicAdd(P.segments + 0, (unsigned long long)segments);
        atomicAdd(P.segments + 1, (unsigned long long)segments);
    }
}

}  // namespace mirt
