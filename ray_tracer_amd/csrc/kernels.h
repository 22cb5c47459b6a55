// kernels.h -- the gfx950 kernels of the path tracer.
//
// renderer.rs's per-pixel loop (src/renderer.rs:64-115) is restructured into passes.  One
// pass holds `n_paths` camera paths = B samples of every pixel of the range, path `slot` =
// sample (sample_base + slot / n_pixels) of pixel (slot % n_pixels), and runs either
//
//   paths        every path from its camera ray to its end, one lane per path, all in
//                registers (k_paths)               renderer.rs:74-82, 92-115
//
// or, bounce by bounce over queues of live paths whose state lives in HBM,
//
//   raygen                                          renderer.rs:74-82 + camera.rs:65-76
//   for bounce in 0 .. handover:
//       extend   closest hit of every live path     renderer.rs:103 -> bvh.rs:37-80
//       bin      sort the live queue by shade class (material/texture kind)
//       shade    emit / scatter / terminate, append survivors to the next live
//                queue                              renderer.rs:104-114 -> material.rs
//   paths        from bounce `handover` on (unless handover = max_depth)
//
// then
//
//   accumulate   pixel += sample radiance, in sample order   renderer.rs:83
//
// and after the last pass `resolve` does renderer.rs:85-89 and the u8 cast.
//
// Queues hold slot indices; counts live in device memory (`counters`), so the host enqueues
// a whole pass without reading anything back.  Kernels are launched with a fixed grid and
// stride (or pull) over the work; a kernel whose queue is empty returns at once.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_shade.h"
#include "device_traverse.h"

namespace mirt {

constexpr int BLOCK = 256;               // threads per workgroup (4 waves)
constexpr int WAVES_PER_BLOCK = BLOCK / 64;
constexpr uint32_t HIT_MISS = 0xFFFFFFFFu;
constexpr uint32_t HIT_SIDE_SHIFT = 28;
constexpr uint32_t HIT_PRIM_MASK = 0x0FFFFFFFu;

// Per-pass path state, structure of arrays, `capacity` slots each.
struct Paths {
    double *ox, *oy, *oz, *dx, *dy, *dz, *time;  // current ray of the path
    double *bx, *by, *bz;                        // product of attenuations so far
    double *hit_t;                               // closest hit of the current ray
    uint32_t* hit_ref;                           // shape index | side << 28, or HIT_MISS
    double* radiance;                            // radiance of the finished path: one record of three doubles
                                                 // per path, written once, by one lane
    uint64_t* stream;                            // sampler key of the path
    uint32_t* live[2];                           // live queues (slot indices), ping-pong by bounce parity
    uint8_t* cls_q;                              // shade class of live entry i (queue order)
    uint32_t* cls_queue;                         // SHADE_NUM_CLASSES queues of `capacity` slots
    uint32_t* counters;                          // see Counters
    unsigned long long* segments;                // [0] running total of closest-hit queries, [1] those of k_paths
    uint32_t capacity;
};

// Layout of Paths::counters for a pass of depth D (all zeroed at pass start):
//   [0 .. D]                 live[b]     number of live paths entering bounce b (live[0] unused: = n_paths)
//   [D+1 .. D+1+8D)          cls[b][c]   paths of shade class c at bounce b
struct Counters {
    uint32_t depth;
    __host__ __device__ uint32_t live(uint32_t b) const { return b; }
    __host__ __device__ uint32_t cls(uint32_t b, uint32_t c) const { return depth + 1 + b * SHADE_NUM_CLASSES + c; }
    // [D+1+8D .. D+1+9D)        work[b]     next unclaimed live entry of bounce b (fast extend kernel)
    __host__ __device__ uint32_t work(uint32_t b) const { return depth + 1 + depth * SHADE_NUM_CLASSES + b; }
    __host__ __device__ uint32_t total() const { return depth + 1 + depth * SHADE_NUM_CLASSES + depth; }
};

// The pixels one call renders: those idx in [pixel_begin, pixel_end) whose tile
// ((idx - pixel_begin) / tile_size) is congruent to tile_index modulo tile_count.
// tile_count = 1 selects the whole range.  Local pixel k is the k-th such idx.
struct PixelSet {
    uint32_t pixel_begin;
    uint32_t n_pixels;  // number of selected pixels
    uint32_t tile_size, tile_count, tile_index;
    __host__ __device__ uint32_t idx(uint32_t k) const {
        if (tile_count <= 1) return pixel_begin + k;
        const uint32_t t = k / tile_size;
        return pixel_begin + (t * tile_count + tile_index) * tile_size + k % tile_size;
    }
    // how many pixels of a range of `len` pixels the set selects
    static uint32_t count(uint32_t len, uint32_t tile_size, uint32_t tile_count, uint32_t tile_index) {
        if (tile_count <= 1) return len;
        const uint32_t full = len / tile_size, rem = len % tile_size;
        uint32_t n = full > tile_index ? (full - tile_index + tile_count - 1) / tile_count * tile_size : 0;
        if (rem && full % tile_count == tile_index) n += rem;
        return n;
    }
};

struct RenderParams {
    mirt_camera cam;
    double background[3];
    uint32_t width, height;
    uint32_t max_depth;
    PixelSet px;            // the pixels of this call
    uint32_t n_pixels;      // = px.n_pixels
    uint32_t sample_base;   // first sample index of this pass
    uint32_t n_samples;     // samples per pixel in this pass (B)
    uint32_t n_paths;       // n_pixels * n_samples
    uint64_t seed;
    double t_min, t_max;    // 0.001, +inf (renderer.rs:103)
    int use_bvh;
    int traversal;
    uint32_t handover;      // bounce at which k_paths takes over in this pass (max_depth: it does not)
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }
// The wave's vote as the mask the hardware already holds (HIP's __ballot goes through a vector
// compare of the flag turned into 0 / 1: two vector instructions per vote).
__device__ __forceinline__ unsigned long long vote(bool flag) { return __builtin_amdgcn_ballot_w64(flag); }

// Three doubles into the path's record: finished paths are scattered over the slots, and a record keeps each
// path's result together (three separate arrays measured 13x the algorithmic bytes at the memory side).
constexpr uint32_t RECORD_DOUBLES = 3, RECORD_BYTES = RECORD_DOUBLES * 8;
__device__ __forceinline__ void store_radiance(const Paths& P, uint32_t slot, const V3& c) {
    double* __restrict__ v = P.radiance + (size_t)slot * RECORD_DOUBLES;
    v[0] = c.x;
    v[1] = c.y;
    v[2] = c.z;
}

// ---------------------------------------------------------------- closest hit
// MODE_LIST and MODE_REFERENCE: one ray per lane from start to finish (simple forms,
// kept for the linear-list mode of the reference and for checking the fast one).
// MODE_FAST: the persistent traversal of device_traverse.h.
enum : int { MODE_LIST = 0, MODE_REFERENCE = 1, MODE_FAST = 2 };

template <int MODE>
__device__ __forceinline__ HitRec closest_hit(const DScene& sc, const RayD& r, double t_min, double t_max) {
    if (MODE == MODE_LIST) return list_closest(sc, r, t_min, t_max);
    return bvh_closest_reference(sc, r, t_min, t_max);
}

// Hittable::hit for caller-supplied rays (array of structs, as the C ABI hands them over).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_intersect(DScene sc, const mirt_ray* __restrict__ rays, size_t n,
                                                     double t_min, double t_max, mirt_hit* __restrict__ hits) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        RayD r;
        r.o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        r.time = rays[i].time;
        HitRec h = closest_hit<MODE>(sc, r, t_min, t_max);
        mirt_hit out;
        out.prim = h.prim;
        out.side = h.prim >= 0 ? h.side : 0;
        out.t = h.prim >= 0 ? h.t : 0.0;
        hits[i] = out;
    }
}

// ---- the fast traversal as a kernel ------------------------------------------------
// Workgroup size: 1024 threads (16 waves = 4 per SIMD, one workgroup per CU) when the
// tree is staged in LDS -- one copy serves the most waves -- and 512 otherwise.
constexpr int TRAV_BLOCK_LDS = 1024;
constexpr int TRAV_BLOCK_GLOBAL = 512;

// Dynamic LDS: [stack: levels x TBLOCK entries of StackT][nodes: n_nodes x 56 B][shapes: n_shapes x 136 B].
__host__ __device__ inline size_t trav_stack_bytes(uint32_t levels, size_t entry, int tblock) {
    return (((size_t)levels * tblock * entry) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t trav_nodes_bytes(uint32_t n_nodes) {
    return (((size_t)n_nodes * NODE_LDS_BYTES) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t qnodes_bytes(uint32_t n_nodes) {
    return (((size_t)n_nodes * QNODE_BYTES) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t scan_boxes_bytes(uint32_t n_pairs) {
    return (((size_t)n_pairs * sizeof(ScanBoxes)) + 15) & ~(size_t)15;
}

// Copies the tree's nodes and the shapes' traversal parameters into LDS (whole workgroup).
// REF16: leaf references in the copy become REF_LEAF | ref16_of(reference) (RayWalk::PACKED).
template <int NTHREADS, bool NODES, bool SHAPES, bool REF16 = false>
__device__ __forceinline__ void stage_tree(const DScene& sc, char* lds_nodes, char* lds_shapes, uint32_t n_top) {
    if (NODES) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.nodes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds_nodes);
        const uint32_t words = n_top * (NODE_LDS_BYTES / 4);
        for (uint32_t w = threadIdx.x; w < words; w += NTHREADS) dst[w] = src[w];
        if (REF16) {
            __syncthreads();
            for (uint32_t node = threadIdx.x; node < n_top; node += NTHREADS) {
                uint32_t* child = reinterpret_cast<CNode*>(lds_nodes)[node].child;
                if (child[0] & REF_LEAF) child[0] = REF_LEAF | ref16_of(child[0]);
                if (child[1] & REF_LEAF) child[1] = REF_LEAF | ref16_of(child[1]);
            }
        }
    }
    if (SHAPES) {
        // the shapes' records, as they lie in DScene::recs
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.recs);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds_shapes);
        const uint32_t words = sc.n_shapes * (SHAPE_LDS_BYTES / 4);
        for (uint32_t w = threadIdx.x; w < words; w += NTHREADS) dst[w] = src[w];
    }
    if (NODES || SHAPES) __syncthreads();
}

template <int TBLOCK, class StackT, bool NODES_IN_LDS, bool SHAPES_IN_LDS, bool STATS, class Io>
__global__ __launch_bounds__(TBLOCK) void k_traverse(DScene sc, Io io, uint32_t* __restrict__ work_counter, double t_min,
                                                      double t_max, uint32_t stack_levels, uint32_t tuning,
                                                      unsigned long long* stats) {
    extern __shared__ __align__(16) char smem[];
    const uint32_t n = io.count();
    // waves beyond the number of chunks can never get work: leave before staging anything
    const uint32_t chunk = chunk_rays(n, gridDim.x * (TBLOCK / 64));
    const uint32_t chunks = (n + chunk - 1) / chunk;
    if (blockIdx.x * (TBLOCK / 64) >= chunks) return;
    StackT* stack = reinterpret_cast<StackT*>(smem);
    char* lds_nodes = smem + trav_stack_bytes(stack_levels, sizeof(StackT), TBLOCK);
    char* lds_shapes = lds_nodes + (NODES_IN_LDS ? trav_nodes_bytes(sc.n_nodes) : 0);
    stage_tree<TBLOCK, NODES_IN_LDS, SHAPES_IN_LDS>(sc, lds_nodes, lds_shapes, sc.n_nodes);
    const TraverseLds lds{lds_nodes, lds_shapes, NODES_IN_LDS ? sc.n_nodes : 0u};
    traverse_persistent<TBLOCK, StackT, NODES_IN_LDS, SHAPES_IN_LDS, STATS>(sc, io, work_counter, t_min, t_max, lds, stack, tuning, stats);
}

// Rays and hits as arrays of structs (the parity entry point).
struct IntersectIo {
    const mirt_ray* __restrict__ rays;
    mirt_hit* __restrict__ hits;
    uint32_t n;
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ void load(uint32_t i, RayD* r, uint32_t* cookie) const {
        r->o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r->d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        r->time = rays[i].time;
        *cookie = i;
    }
    __device__ __forceinline__ void store(uint32_t i, uint32_t, double t, int32_t prim, int32_t side, uint32_t) const {
        mirt_hit out;
        out.prim = prim;
        out.side = prim >= 0 ? side : 0;
        out.t = prim >= 0 ? t : 0.0;
        hits[i] = out;
    }
};

// ---------------------------------------------------------------------- raygen
// Slot s holds sample (sample_base + s / n_pixels) of the (s % n_pixels)-th pixel of the set:
// consecutive lanes start neighbouring pixels of one sample index.
__global__ __launch_bounds__(BLOCK) void k_raygen(Paths P, RenderParams rp) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < rp.n_paths; slot += stride) {
        const uint32_t idx = rp.px.idx(slot % rp.n_pixels);
        const uint32_t sample = rp.sample_base + slot / rp.n_pixels;
        PathRng rng{mirt_rng_stream(rp.seed, idx, sample), 0u, 0u};
        RayD r = camera_ray(rp.cam, rp.width, rp.height, idx, rng);
        P.ox[slot] = r.o.x; P.oy[slot] = r.o.y; P.oz[slot] = r.o.z;
        P.dx[slot] = r.d.x; P.dy[slot] = r.d.y; P.dz[slot] = r.d.z;
        P.time[slot] = r.time;
        P.bx[slot] = 1.0; P.by[slot] = 1.0; P.bz[slot] = 1.0;
        P.stream[slot] = rng.stream;
    }
}

// Camera rays only (array of structs), for the intersection parity check.
__global__ __launch_bounds__(BLOCK) void k_camera_rays(mirt_camera cam, uint32_t width, uint32_t height,
                                                       uint32_t pixel_begin, uint32_t n, uint32_t sample,
                                                       uint64_t seed, mirt_ray* __restrict__ rays) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
        const uint32_t idx = pixel_begin + k;
        PathRng rng{mirt_rng_stream(seed, idx, sample), 0u, 0u};
        RayD r = camera_ray(cam, width, height, idx, rng);
        mirt_ray out;
        out.origin[0] = r.o.x; out.origin[1] = r.o.y; out.origin[2] = r.o.z;
        out.direction[0] = r.d.x; out.direction[1] = r.d.y; out.direction[2] = r.d.z;
        out.time = r.time;
        rays[k] = out;
    }
}

// ---------------------------------------------------------------------- extend
// Closest hit of every live path of this bounce.  Entry i of the live queue is
// handled by thread i mod (grid size): no atomics, no barriers.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_extend(DScene sc, Paths P, RenderParams rp, uint32_t bounce) {
    const Counters C{rp.max_depth};
    const uint32_t n = bounce == 0 ? rp.n_paths : P.counters[C.live(bounce)];
    const uint32_t* __restrict__ qin = P.live[bounce & 1];
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = bounce == 0 ? i : qin[i];
        RayD r;
        r.o = v3(P.ox[slot], P.oy[slot], P.oz[slot]);
        r.d = v3(P.dx[slot], P.dy[slot], P.dz[slot]);
        r.time = P.time[slot];
        HitRec h = closest_hit<MODE>(sc, r, rp.t_min, rp.t_max);
        uint32_t cls = SHADE_MISS;
        uint32_t ref = HIT_MISS;
        if (h.prim >= 0) {
            cls = sc.shapes[h.prim].cls;
            ref = (uint32_t)h.prim | ((uint32_t)h.side << HIT_SIDE_SHIFT);
            P.hit_t[slot] = h.t;
        }
        P.hit_ref[slot] = ref;
        P.cls_q[i] = (uint8_t)cls;
    }
}

// The live queue of one bounce as the ray source of the persistent traversal
// (k_traverse<.., ExtendIo>): the extend kernel of the fast mode.
struct ExtendIo {
    Paths P;
    const DShape* __restrict__ shapes;
    const uint32_t* __restrict__ qin;  // live queue of this bounce (unused at bounce 0: entry i = slot i)
    const uint32_t* __restrict__ n_live;  // its length in device memory (null at bounce 0)
    uint32_t n_paths;
    uint32_t first_bounce;
    __device__ __forceinline__ uint32_t count() const { return first_bounce ? n_paths : *n_live; }
    __device__ __forceinline__ void load(uint32_t i, RayD* r, uint32_t* cookie) const {
        const uint32_t slot = first_bounce ? i : qin[i];
        r->o = v3(P.ox[slot], P.oy[slot], P.oz[slot]);
        r->d = v3(P.dx[slot], P.dy[slot], P.dz[slot]);
        r->time = P.time[slot];
        *cookie = slot;
    }
    __device__ __forceinline__ void store(uint32_t i, uint32_t slot, double t, int32_t prim, int32_t side, uint32_t shape_cls) const {
        uint32_t cls = SHADE_MISS;
        uint32_t ref = HIT_MISS;
        if (prim >= 0) {
            cls = shape_cls;
            ref = (uint32_t)prim | ((uint32_t)side << HIT_SIDE_SHIFT);
            P.hit_t[slot] = t;
        }
        P.hit_ref[slot] = ref;
        P.cls_q[i] = (uint8_t)cls;
    }
};

// ------------------------------------------------------------------------- bin
// Sorts the live queue of this bounce by shade class: a block takes a tile of
// BIN_TILE consecutive entries, counts its classes with wave ballots, reserves
// one contiguous range per class with one atomic per class per block, and
// writes the slot indices there in queue order.
constexpr int BIN_ITEMS = 8;
constexpr int BIN_TILE = BLOCK * BIN_ITEMS;

__global__ __launch_bounds__(BLOCK) void k_bin(Paths P, RenderParams rp, uint32_t bounce) {
    __shared__ uint32_t s_wave[WAVES_PER_BLOCK][SHADE_NUM_CLASSES];  // per wave: count, then offset inside the block
    __shared__ uint32_t s_base[SHADE_NUM_CLASSES];                   // block's first index in each class queue
    const Counters C{rp.max_depth};
    const uint32_t n = bounce == 0 ? rp.n_paths : P.counters[C.live(bounce)];
    const uint32_t* __restrict__ qin = P.live[bounce & 1];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_tiles = (n + BIN_TILE - 1) / BIN_TILE;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t base = tile * BIN_TILE + wave * (64 * BIN_ITEMS);
        uint32_t cls[BIN_ITEMS];
        uint32_t wave_count[SHADE_NUM_CLASSES];
#pragma unroll
        for (int c = 0; c < SHADE_NUM_CLASSES; ++c) wave_count[c] = 0;
#pragma unroll
        for (int k = 0; k < BIN_ITEMS; ++k) {
            const uint32_t i = base + k * 64 + lane;
            cls[k] = i < n ? (uint32_t)P.cls_q[i] : 0xFFu;
#pragma unroll
            for (int c = 0; c < SHADE_NUM_CLASSES; ++c) wave_count[c] += __popcll(__ballot(cls[k] == (uint32_t)c));
        }
        if (lane < SHADE_NUM_CLASSES) {
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < SHADE_NUM_CLASSES; ++c) mine = lane == (uint32_t)c ? wave_count[c] : mine;
            s_wave[wave][lane] = mine;
        }
        __syncthreads();
        if (threadIdx.x < SHADE_NUM_CLASSES) {
            const uint32_t c = threadIdx.x;
            uint32_t total = 0;
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
                const uint32_t cnt = s_wave[w][c];
                s_wave[w][c] = total;
                total += cnt;
            }
            s_base[c] = total ? atomicAdd(&P.counters[C.cls(bounce, c)], total) : 0u;
        }
        __syncthreads();
        uint32_t run[SHADE_NUM_CLASSES];
#pragma unroll
        for (int c = 0; c < SHADE_NUM_CLASSES; ++c) run[c] = s_base[c] + s_wave[wave][c];
#pragma unroll
        for (int k = 0; k < BIN_ITEMS; ++k) {
            const uint32_t i = base + k * 64 + lane;
            const uint32_t slot = i < n ? (bounce == 0 ? i : qin[i]) : 0u;
#pragma unroll
            for (int c = 0; c < SHADE_NUM_CLASSES; ++c) {
                const unsigned long long m = __ballot(cls[k] == (uint32_t)c);
                if (cls[k] == (uint32_t)c) P.cls_queue[(size_t)c * P.capacity + run[c] + mbcnt_lt(m)] = slot;
                run[c] += __popcll(m);
            }
        }
        __syncthreads();  // s_wave / s_base are reused by the next tile
    }
}

// ----------------------------------------------------------------------- shade
// Development builds (MIRT_PATHS_STATS): wave clocks of the pieces of one shade trip.
struct ShadeClock {
#ifdef MIRT_PATHS_STATS
    enum { N = 10 };
    unsigned long long clk[N] = {}, last = 0;
    __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        clk[k] += now - last;
        last = now;
    }
#elif defined(MIRT_ISA_MARKS)
    // analysis builds: the same places as comments in the assembly
    __device__ __forceinline__ void start() { asm volatile("; SHADE_MARK start"); }
    __device__ __forceinline__ void mark(int k) { asm volatile("; SHADE_MARK %0" ::"n"(k)); }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
#endif
};

// Will shade_path use a sample of random_in_unit_sphere for a hit of class `cls`?
// (lambertian, material.rs:53; metal, material.rs:79)
__device__ __forceinline__ bool wants_in_sphere(uint32_t cls, bool last) {
    return !last && cls != SHADE_MISS && cls != SHADE_LIGHT && cls != SHADE_DIELECTRIC;
}

// What becomes of a path at a hit (or a miss): renderer.rs:99-116 for one bounce.
struct ShadeResult {
    bool alive;     // the path goes on with the ray (origin, direction)
    V3 origin, direction;
    V3 value;       // alive: the path's new throughput; otherwise: the radiance of the finished path
};

// cls: shade class of the hit shape (SHADE_MISS: none).  `last`: this is the deepest bounce
// allowed; what would be scattered is multiplied by color(.., 0) = 0 (renderer.rs:99-101).
template <int TEXTURES, bool NOISE_INLINE = true>
__device__ __forceinline__ ShadeResult shade_path(const DScene& sc, uint32_t cls, const RayD& r, double hit_t, uint32_t hit_ref,
                                                  uint32_t tk, const ShapeRec* rec, const mirt_texture* textures, const V3& beta,
                                                  const V3& in_sphere, uint64_t stream, uint32_t bounce, bool last, const V3& background, const mirt_perlin* lds_tables, uint32_t n_staged,
                                                  ShadeClock& clock, bool have_turbulence = false, double turbulence = 0.0) {
    ShadeResult out;
    out.alive = false;
    out.value = v3(0.0, 0.0, 0.0);
    if (cls == SHADE_MISS) {
        out.value = beta * background;  // renderer.rs:114
        return out;
    }
    const Surface h = surface_at(r, hit_t, tk, rec->f, (int)(hit_ref >> HIT_SIDE_SHIFT));
    clock.mark(1);
    // the material's constants from the shape's record; a texture that is not a constant is evaluated
    const V3 color = v3(rec->color[0], rec->color[1], rec->color[2]);
    const double param = rec->param;
    const uint32_t texture = rec->texture;
    if (cls == SHADE_LIGHT) {
        // material.rs:144-152: no scatter, emit = texture value
        out.value = beta * (texture == NO_TEXTURE ? color : texture_value<TEXTURES, NOISE_INLINE>(sc, textures, texture, h, lds_tables, n_staged));
        return out;
    }
    if (last) return out;
    // Lanes of one wave may hold hits of different classes (path kernel): what several classes
    // need is computed ONCE for all of them before the class branches -- `in_sphere`, the result
    // of random_in_unit_sphere (first draws of the stage for lambertian and metal alike; the
    // caller had the whole wave work on it, see wants_in_sphere) and the unit direction (metal,
    // dielectric) -- instead of once per class present in the wave.
    PathRng rng{stream, bounce + 1u, 0u};  // dielectric: its one draw is the first of the stage
    const bool is_lambertian = cls != SHADE_METAL && cls != SHADE_DIELECTRIC;
    V3 unit_direction = v3(0.0, 0.0, 0.0);
    clock.mark(2);
    if (!is_lambertian) unit_direction = unit(r.d);
    clock.mark(3);
    V3 dir, attenuation;
    bool scattered = true;
    if (cls == SHADE_METAL) {  // material.rs:74-91
        const V3 reflected = reflect(unit_direction, h.normal);
        dir = reflected + param * in_sphere;
        scattered = dot(dir, h.normal) > 0.0;
        attenuation = color;
        clock.mark(4);
    } else if (cls == SHADE_DIELECTRIC) {  // material.rs:104-131
        const double refraction_ratio = h.front_facing ? 1.0 / param : param;
        const double cos_theta = fmin(dot(-unit_direction, h.normal), 1.0);
        const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        const bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        if (cannot_refract || schlick(cos_theta, refraction_ratio) > rng.random_double()) {
            dir = reflect(unit_direction, h.normal);
        } else {
            dir = refract(unit_direction, h.normal, refraction_ratio);
        }
        attenuation = v3(1.0, 1.0, 1.0);
        clock.mark(5);
    } else {  // the lambertian classes, material.rs:51-59
        const V3 target = h.p + h.normal + in_sphere;
        dir = target - h.p;
        attenuation = cls == SHADE_LAMBERTIAN_CONSTANT ? color
                                                        : texture_value<TEXTURES, NOISE_INLINE>(sc, textures, texture, h, lds_tables, n_staged,
                                                                                  have_turbulence && cls == SHADE_LAMBERTIAN_NOISE, turbulence);
        clock.mark(6);
    }
    if (scattered) {
        out.alive = true;
        out.origin = h.p;
        out.direction = dir;
        out.value = beta * attenuation;  // (emitted + attenuation) with emitted = 0, renderer.rs:107
    }
    return out;
}

// Perlin tables (9 KB each: 256 gradient vectors + 3 permutations) staged in LDS by the
// kernels that shade.  The host pads DScene::perlins to PERLIN_STAGED readable tables.
constexpr int PERLIN_STAGED = 2;
constexpr int TEXTURES_STAGED = 32;  // entries of DScene::shade_textures the path kernel copies to LDS

template <int NTHREADS>
__device__ __forceinline__ void stage_perlin(const DScene& sc, mirt_perlin* dst_tables) {
    const uint32_t words = (uint32_t)(sizeof(mirt_perlin) / 4) * PERLIN_STAGED;
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.perlins);
    uint32_t* dst = reinterpret_cast<uint32_t*>(dst_tables);
    for (uint32_t w = threadIdx.x; w < words; w += NTHREADS) dst[w] = src[w];
    __syncthreads();
}

// One wave = 64 consecutive entries of ONE class queue.  The virtual index space
// is the concatenation of the class queues, each padded to a multiple of 64.
// Survivors are appended to the next live queue with one atomic per block step.
template <bool HAS_NOISE>
__global__ __launch_bounds__(BLOCK) void k_shade(DScene sc, Paths P, RenderParams rp, uint32_t bounce) {
    __shared__ uint32_t s_cnt[2][WAVES_PER_BLOCK];
    __shared__ uint32_t s_base[2];
    __shared__ unsigned long long s_scratch[WAVES_PER_BLOCK][IN_SPHERE_SCRATCH];
    __shared__ mirt_perlin s_perlin[HAS_NOISE ? PERLIN_STAGED : 1];
    const uint32_t n_staged = HAS_NOISE && sc.perlins ? (uint32_t)PERLIN_STAGED : 0u;
    if (n_staged) stage_perlin<BLOCK>(sc, s_perlin);
    const mirt_perlin* lds_tables = HAS_NOISE ? s_perlin : nullptr;

    const Counters C{rp.max_depth};
    uint32_t cnt[SHADE_NUM_CLASSES], pref[SHADE_NUM_CLASSES + 1];
    pref[0] = 0;
#pragma unroll
    for (int c = 0; c < SHADE_NUM_CLASSES; ++c) {
        cnt[c] = P.counters[C.cls(bounce, c)];
        pref[c + 1] = pref[c] + ((cnt[c] + 63u) >> 6);
    }
    const uint32_t total_chunks = pref[SHADE_NUM_CLASSES];
    const bool last = bounce + 1 == rp.max_depth;
    uint32_t* __restrict__ qout = P.live[(bounce + 1) & 1];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const V3 background = v3(rp.background[0], rp.background[1], rp.background[2]);

    uint32_t it = 0;
    for (uint32_t chunk0 = blockIdx.x * WAVES_PER_BLOCK; chunk0 < total_chunks;
         chunk0 += gridDim.x * WAVES_PER_BLOCK, ++it) {
        const uint32_t chunk = chunk0 + wave;
        uint32_t cls = 0, first = 0;
        bool valid = false;
        if (chunk < total_chunks) {
#pragma unroll
            for (int c = 0; c < SHADE_NUM_CLASSES; ++c)
                if (chunk >= pref[c] && chunk < pref[c + 1]) {
                    cls = c;
                    first = (chunk - pref[c]) * 64;
                }
            valid = first + lane < cnt[cls];
        }
        cls = __builtin_amdgcn_readfirstlane(cls);  // wave-uniform by construction
        bool alive = false;
        uint32_t slot = 0;
        if (valid) slot = P.cls_queue[(size_t)cls * P.capacity + first + lane];
        uint64_t stream = 0;
        const bool sampling = valid && wants_in_sphere(cls, last);
        if (valid && cls != SHADE_MISS && cls != SHADE_LIGHT && !last) stream = P.stream[slot];
        const V3 in_sphere = random_in_unit_sphere_wave(sampling, PathRng{stream, bounce + 1u, 0u}, s_scratch[wave]);
        if (valid) {
            const V3 beta = v3(P.bx[slot], P.by[slot], P.bz[slot]);
            RayD r{};
            double hit_t = 0.0;
            uint32_t hit_ref = HIT_MISS, tk = 0;
            const ShapeRec* rec = sc.recs;
            if (cls != SHADE_MISS) {
                r.o = v3(P.ox[slot], P.oy[slot], P.oz[slot]);
                r.d = v3(P.dx[slot], P.dy[slot], P.dz[slot]);
                r.time = P.time[slot];
                hit_t = P.hit_t[slot];
                hit_ref = P.hit_ref[slot];
                tk = sc.shapes[hit_ref & HIT_PRIM_MASK].tk;
                rec = sc.recs + (hit_ref & HIT_PRIM_MASK);
            }
            ShadeClock clock;
            const ShadeResult s = shade_path<HAS_NOISE ? TEXTURES_ALL : TEXTURES_IMAGE>(sc, cls, r, hit_t, hit_ref, tk, rec, sc.shade_textures, beta, in_sphere, stream, bounce, last, background, lds_tables,
                                                                                                 n_staged, clock);
            alive = s.alive;
            if (alive) {
                P.ox[slot] = s.origin.x; P.oy[slot] = s.origin.y; P.oz[slot] = s.origin.z;
                P.dx[slot] = s.direction.x; P.dy[slot] = s.direction.y; P.dz[slot] = s.direction.z;
                P.bx[slot] = s.value.x; P.by[slot] = s.value.y; P.bz[slot] = s.value.z;
            } else {
                store_radiance(P, slot, s.value);
            }
        }
        // ---- append survivors: ballot per wave, one atomic per block step
        const unsigned long long mask = __ballot(alive);
        const uint32_t buf = it & 1;
        if (lane == 0) s_cnt[buf][wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) total += s_cnt[buf][w];
            s_base[buf] = total ? atomicAdd(&P.counters[C.live(bounce + 1)], total) : 0u;
        }
        __syncthreads();
        if (alive) {
            uint32_t off = s_base[buf];
            for (uint32_t w = 0; w < wave; ++w) off += s_cnt[buf][w];
            qout[off + mbcnt_lt(mask)] = slot;
        }
    }
}

// ----------------------------------------------------------------------- paths
// Whole paths, one per lane: camera ray (or the ray a path has when bounce `bounce0` starts),
// closest hit, shade, next bounce ... until the path ends, all in registers; the lane then
// takes the next path.  Nothing but the finished path's radiance goes to memory.
// Measured on RTIOW against a launch per bounce over sorted queues of live paths (extend /
// bin / shade, kept below): 2.3x the samples per second.  Deep bounces hold few paths (1 % of
// them reach bounce 10, some go on to 50), a launch for them costs more in latency than in
// work, and the queues' 200 bytes per path and bounce of HBM traffic buy less coherence than
// they cost.  Results do not depend on the bounce at which this kernel takes over: a path's
// random numbers are a function of (pixel, sample, bounce) alone.
// Workgroup size and waves per SIMD by texture level (registers: 128 / 128 / 160 / 168 VGPRs: four waves per SIMD, or three): one
// workgroup fills the CU and shares one copy of the tree -- all of it with the shapes' records
// (TREE_LDS), or its nodes alone when only they fit (TREE_NODES_LDS: +5..7 % on scenes of 900-1 900 shapes against
// reading them from global memory), or nothing of it (TREE_GLOBAL).
__host__ __device__ constexpr int paths_waves(int textures) { return textures <= TEXTURES_CHECKER ? 4 : 3; }
__host__ __device__ constexpr int paths_block(int textures) { return 256 * paths_waves(textures); }
// Scheduling parameters of k_paths (`tuning`):
//   bits 0-7    as in traverse_persistent
//   bits 8-15   lanes waiting to be shaded that trigger shading
//   bits 16-23  empty lanes that trigger a refill from the queue
//   bits 24-29  node visits are chosen over shape tests unless 8 * (lanes wanting a shape test) >
//               this * (lanes wanting a node visit); 0 = 8: the larger group wins
//   bit 30      walks that ended without a hit count towards the shading trigger
//   bit 31      such walks are finished off by the refill (the lane counts as empty)
// Measured at 64 samples per pixel, relative to 3 | 48 << 8 | 16 << 16 | 1 << 30: bit 31 gives +12 % on
// the textured scene (tree in LDS), -1 % on the Cornell box, -4 % on the 3,500-shape scene; a refill at
// 24 empty lanes instead of 16 another +2.5 % on RTIOW and -2.5 % on the Cornell box (which has its own
// parameters since it is rendered without a tree: the refill is at 26 now, +1.2 % RTIOW against 20, flat from
// 24 to 28, falling off beyond 32).
// With parked shapes most lanes can take either step, so the wave stays with one for as long as an
// eighth of its busy lanes want it (bits 0-7 = 1: +4 % RTIOW, +2 % textured scene against 3).
constexpr uint32_t PATHS_TUNING_DEFAULT = 1u | (48u << 8) | (26u << 16) | (3u << 30);
// a tree too big for LDS: walks are long and differ in length, waiting for 48 finished lanes or 16
// empty ones leaves the others idle for longer (3,500-shape scene: +4 %; scenes whose tree is in LDS
// lose 1-3 % with these)
constexpr uint32_t PATHS_TUNING_BIG_TREE = 3u | (44u << 8) | (8u << 16) | (1u << 30);
// nodes in LDS as QNodes, records in global memory
constexpr uint32_t PATHS_TUNING_QNODES = PATHS_TUNING_BIG_TREE;
// no tree (ScanWalk): after a scan every lane is ready for shading, so shading never waits; the refill
// threshold is what there is to choose
constexpr uint32_t PATHS_TUNING_SCAN = 1u | (1u << 8) | (16u << 16) | (3u << 30);
// Most shapes a scene of rects and cubes may have to be rendered without a tree (and at most SCAN_MAX_ITEMS
// items, a cube counting six).
#ifndef MIRT_SCAN_MAX_SHAPES
#define MIRT_SCAN_MAX_SHAPES 16
#endif
// Shapes a walk may hold parked for later shape tests while it goes on visiting nodes: with 4 (node
// visits allowed while at most 2 are parked) a lane can take part in whichever of the two steps the
// wave runs next, at the price of some visits a found hit would have culled: +4 % RTIOW, +11 % Cornell
// box, +2 % textured scene against 2 (a node visit only with none parked); 6 gains less.  With nodes
// read from global memory the extra visits cost more than the fuller trips gain (-4 %): 2 there.
#ifndef MIRT_PARK
#define MIRT_PARK 4
#endif
#ifndef MIRT_PARK_BIG
#define MIRT_PARK_BIG 2
#endif
// The turbulence of a shading trip's noise-texture hits computed by the whole wave (perlin_turbulence_wave)
// Node visits a wave runs between two votes on who still wants one, for trees that are not in LDS as a whole
#ifndef MIRT_NODE_VISITS_PER_VOTE
#define MIRT_NODE_VISITS_PER_VOTE 2
#endif
#ifndef MIRT_TURBULENCE_WAVE
#define MIRT_TURBULENCE_WAVE 1
#endif
// ... and with the nodes in LDS as QNodes and the records in global memory
#ifndef MIRT_PARK_QNODES
#define MIRT_PARK_QNODES 4
#endif
// Wave priorities (s_setprio, 0-3: a SIMD issues from the ready wave of the highest priority) by what a wave of
// k_paths runs.  The longer a piece's uninterrupted f64 arithmetic, the lower its priority: a wave in node visits
// (some forty instructions between two waits for LDS) then never queues behind a neighbour's refill or shading
// trip (thousands of instructions that nothing waits for), which fills the gaps instead.  Measured against all
// waves equal: +4 % RTIOW, +3.3 % on the 3 507-shape scene, +2 % Cornell box, +1.2 % textured scene; the reverse
// (shading first) loses 3-7 %; any order with refill and shading below the walk is within 0.5 % of this one.
#ifndef MIRT_PRIO_WALK
#define MIRT_PRIO_WALK 3    // the votes between trips
#endif
#ifndef MIRT_PRIO_NODE
#define MIRT_PRIO_NODE 3    // node visits
#endif
#ifndef MIRT_PRIO_LEAF
#define MIRT_PRIO_LEAF 2    // shape tests (f64 divisions and square roots)
#endif
#ifndef MIRT_PRIO_SHADE
#define MIRT_PRIO_SHADE 1
#endif
#ifndef MIRT_PRIO_REFILL
#define MIRT_PRIO_REFILL 0  // camera rays: sampler arithmetic, nothing anybody waits for
#endif

// A kernel argument read from the argument block (constant address space: scalar loads) at byte `offset`.
typedef const __attribute__((address_space(4))) uint32_t* KernargWords;
template <class T>
__device__ __forceinline__ T load_kernarg(KernargWords args, size_t offset) {
    static_assert(sizeof(T) % 8 == 0 && alignof(T) <= 8, "whole 8-byte words");
    T value;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&value);
    const __attribute__((address_space(4))) unsigned long long* src =
        (const __attribute__((address_space(4))) unsigned long long*)(args + offset / 4);
#pragma unroll
    for (size_t w = 0; w < sizeof(T) / 8; ++w) dst[w] = src[w];
    return value;
}

static_assert(sizeof(DScene) % 8 == 0 && sizeof(Paths) % 8 == 0 && alignof(DScene) == 8 && alignof(Paths) == 8 && alignof(RenderParams) == 8,
              "k_paths finds its second and third argument at sizeof(DScene) and sizeof(DScene) + sizeof(Paths) of the argument block");

// TREE: where the culling tree and the shapes' records are read from.
enum : int {
    TREE_GLOBAL = 0,     // both from global memory
    TREE_LDS = 1,        // both staged in LDS
    TREE_NODES_LDS = 2,  // the nodes in LDS, the records (2.4 times their size) from global memory
    TREE_SCAN = 3,       // a handful of rects and cubes: no tree, every ray against the box of every rect and face (ScanWalk)
    TREE_QNODES_LDS = 4  // the nodes in LDS as QNodes (28 bytes, bounds as halves), the records from global memory
};
template <class StackT, int TEXTURES, int TREE>
__global__ __launch_bounds__(paths_block(TEXTURES)) __attribute__((amdgpu_waves_per_eu(paths_waves(TEXTURES)))) void k_paths(
    DScene sc, Paths P, RenderParams rp, uint32_t bounce0, uint32_t stack_levels, uint32_t tuning, uint32_t n_top) {
    constexpr int PATHS_BLOCK = paths_block(TEXTURES);
    constexpr bool HAS_NOISE = TEXTURES >= TEXTURES_ALL;
    constexpr bool SCAN = TREE == TREE_SCAN;
    constexpr bool QNODES = TREE == TREE_QNODES_LDS;
    constexpr bool TREE_IN_LDS = TREE == TREE_LDS, NODES_IN_LDS = TREE == TREE_LDS || TREE == TREE_NODES_LDS || QNODES;
    constexpr bool RECS_IN_LDS = TREE_IN_LDS || SCAN;  // the shapes' records, as shading reads them
    // dynamic LDS: [level][PATHS_BLOCK] stack entries, the first n_top nodes of the tree (TREE_IN_LDS: all of
    // them), and (TREE_IN_LDS) the shapes' records
    extern __shared__ __align__(16) char smem[];
    __shared__ mirt_perlin s_perlin[HAS_NOISE ? PERLIN_STAGED : 1];
    const uint32_t n_staged = HAS_NOISE && sc.perlins ? (uint32_t)PERLIN_STAGED : 0u;
    if (n_staged) stage_perlin<PATHS_BLOCK>(sc, s_perlin);
    const mirt_perlin* lds_tables = HAS_NOISE ? s_perlin : nullptr;
    // the textures shading can meet (one checker and its two colours, say), in LDS when they are few
    __shared__ mirt_texture s_textures[TEXTURES > TEXTURES_CONSTANT ? TEXTURES_STAGED : 1];
    __shared__ unsigned long long s_scratch[PATHS_BLOCK / 64][IN_SPHERE_SCRATCH];
    __shared__ double s_turbulence[HAS_NOISE ? PATHS_BLOCK / 64 : 1][HAS_NOISE ? 64 : 1];  // perlin_turbulence_wave's terms
    static_assert(IN_SPHERE_SCRATCH >= 4 * TURBULENCE_POINTS, "the in-sphere scratch doubles as perlin_turbulence_wave's points");
    const mirt_texture* textures = sc.shade_textures;
    if (TEXTURES > TEXTURES_CONSTANT && sc.n_shade_textures <= (uint32_t)TEXTURES_STAGED) {
        const uint32_t words = sc.n_shade_textures * (uint32_t)(sizeof(mirt_texture) / 4);
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.shade_textures);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_textures);
        for (uint32_t w = threadIdx.x; w < words; w += PATHS_BLOCK) dst[w] = src[w];
        textures = s_textures;
        __syncthreads();
    }
    StackT* stack = reinterpret_cast<StackT*>(smem);
    char* lds_nodes = smem + trav_stack_bytes(stack_levels, sizeof(StackT), PATHS_BLOCK);
    char* lds_shapes = lds_nodes + (TREE == TREE_SCAN ? scan_boxes_bytes(n_top) : QNODES ? qnodes_bytes(n_top) : trav_nodes_bytes(n_top));
    typedef RayWalk<PATHS_BLOCK, StackT, NODES_IN_LDS, TREE_IN_LDS, false, TREE_IN_LDS ? MIRT_PARK : QNODES ? MIRT_PARK_QNODES : MIRT_PARK_BIG, QNODES> TreeWalk;
    typedef typename std::conditional<SCAN, ScanWalk<PATHS_BLOCK>, TreeWalk>::type Walk;
    static_assert(!Walk::PACKED || TREE_IN_LDS || QNODES, "16-bit parked references: a tree that is in LDS as a whole (at most 1 024 shapes), or QNodes (at most 4 096)");
    // SCAN: n_top = the pairs of item boxes (DScene::scan_boxes), the items follow the shapes' records
    char* lds_items = lds_shapes + (((size_t)sc.n_shapes * SHAPE_LDS_BYTES + 15) & ~(size_t)15);
    if (SCAN) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.scan_boxes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds_nodes);
        for (uint32_t w = threadIdx.x; w < n_top * (uint32_t)(sizeof(ScanBoxes) / 4); w += PATHS_BLOCK) dst[w] = src[w];
        src = reinterpret_cast<const uint32_t*>(sc.scan_items);
        dst = reinterpret_cast<uint32_t*>(lds_items);
        for (uint32_t w = threadIdx.x; w < sc.n_scan_items * (uint32_t)(sizeof(ScanItem) / 4); w += PATHS_BLOCK) dst[w] = src[w];
    }
    if (QNODES) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.qnodes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds_nodes);
        for (uint32_t w = threadIdx.x; w < n_top * (QNODE_BYTES / 4); w += PATHS_BLOCK) dst[w] = src[w];
        __syncthreads();
    } else {
        stage_tree<PATHS_BLOCK, !SCAN, RECS_IN_LDS, Walk::PACKED>(sc, lds_nodes, lds_shapes, n_top);
    }
    // (QNODES: no records in LDS; `shapes` is where the QRefs begin)
    const TraverseLds lds{lds_nodes, QNODES ? lds_nodes + (size_t)n_top * sizeof(QBox) : lds_shapes, n_top, SCAN ? lds_items : nullptr,
                          SCAN ? sc.n_scan_items : 0u};

    const Counters C{rp.max_depth};
    const uint32_t n = bounce0 == 0 ? rp.n_paths : P.counters[C.live(bounce0)];
    uint32_t* __restrict__ work = P.counters + C.work(bounce0);
    const V3 background = v3(rp.background[0], rp.background[1], rp.background[2]);
    const uint32_t tid = threadIdx.x, lane = lane_id();

    // A lane is empty (!have), traversing (have, walk not finished) or waiting to be shaded
    // (have, walk finished).  Per trip the wave refills empty lanes, then runs the piece of
    // code -- node visits, shape tests, shading -- that enough lanes are waiting for.
    bool have = false, exhausted = false;
    uint32_t next = 0, chunk_end = 0;  // the wave's chunk of the queue (wave-uniform)
    // (a long queue is taken in longer chunks: consecutive paths are neighbouring pixels, and a wave that stays in one
    // corner of the image keeps its lanes' rays together -- +0.75 % on the 500-spp frame; on a short queue the waves
    // would finish further apart than that is worth: -2 % at 64 spp)
    const uint32_t n_waves = gridDim.x * (PATHS_BLOCK / 64);
    const uint32_t chunk = n / n_waves >= 65536u ? 2u * CHUNK_RAYS : chunk_rays(n, n_waves);
    uint32_t slot = 0, bounce = 0;
    uint32_t extra_segments = 0;  // closest-hit queries of this wave beyond the first of each path (the same in every lane)
    V3 beta = v3(0.0, 0.0, 0.0);
    uint64_t stream = 0;
    // (ODDITIES off: a path's rays come from the camera and from scattering, with the camera's finite time
    // and never inside the plane of the rect they leave, so no shape test returns a NaN t; the hook for
    // those cases costs 4 % here and is kept for caller-supplied rays, mirt_intersect)
    Walk walk;
    walk.reset();
    const int stay = (int)(tuning & 0xFFu), shade_at = (int)((tuning >> 8) & 0xFFu);
    const int refill_at = (int)((tuning >> 16) & 0xFFu);
    const int leaf_bias = ((tuning >> 24) & 63u) ? (int)((tuning >> 24) & 63u) : 8;
    const bool count_missed = (tuning >> 30) & 1u, refill_missed = (tuning >> 31) & 1u;
#ifdef MIRT_PATHS_STATS
    // development build: wave clocks and lane counts per piece of code, summed into P.segments[2..]
    enum { PS_REFILL, PS_SHADE, PS_NODE, PS_LEAF, PS_N };
    unsigned long long ps_clk[PS_N] = {}, ps_trips[PS_N] = {}, ps_lanes[PS_N] = {};
    ShadeClock shade_clock;
    unsigned long long ps_outer = 0;
    unsigned long long ps_idle[3] = {};  // in node trips: lanes holding a hit that waits for shading, empty lanes, lanes that can only test shapes
    const unsigned long long ps_begin = __builtin_amdgcn_s_memtime();
    // (the 100 MHz counter that all waves share: when the launch began, when the queue ran dry, when each wave ended)
    const unsigned long long ps_real_begin = __builtin_amdgcn_s_memrealtime();
    bool ps_saw_dry = false;
#define PS_START() const unsigned long long ps_t0 = __builtin_amdgcn_s_memtime()
#define PS_STOP(k, lanes)                                                   \
    do {                                                                    \
        ps_clk[k] += __builtin_amdgcn_s_memtime() - ps_t0;                  \
        ps_trips[k] += 1;                                                   \
        ps_lanes[k] += (lanes);                                             \
    } while (0)
#else
    ShadeClock shade_clock;
#define PS_START() do {} while (0)
#define PS_STOP(k, lanes) do {} while (0)
#endif
    __builtin_amdgcn_s_setprio(MIRT_PRIO_WALK);
    for (;;) {
#ifdef MIRT_PATHS_STATS
        ++ps_outer;
#endif
        // The wave's lanes by what they wait for (an empty lane's walk is reset: it wants nothing and holds no hit).
        unsigned long long m_have = vote(have), m_node = walk.vote_node(), m_leaf = walk.vote_leaf();
        unsigned long long m_hit = vote(walk.best.prim >= 0);
        unsigned long long m_finished = m_have & ~(m_node | m_leaf);
        // A walk that ended without a hit ends its path with the background (renderer.rs:114): a store,
        // done by the refill that gives the lane its next path -- such lanes count as empty.  Shading
        // trips are then made of hits, whose code the missed lanes would sit through idle.
        const unsigned long long need = ~m_have | (refill_missed ? m_finished & ~m_hit : 0ull);
        // one refill serves many lanes: wait until enough are empty.  Path indices come from the
        // wave's own chunk of the queue; one global atomic fetches the next chunk.
        if (!exhausted && (__popcll(need) >= refill_at || need == ~0ull)) {
            const bool missed = refill_missed && have && walk.finished() && walk.best.prim < 0;
            PS_START();
            __builtin_amdgcn_s_setprio(MIRT_PRIO_REFILL);
            extra_segments += (uint32_t)__popcll(vote(missed && bounce != bounce0));
            if (missed) {
                store_radiance(P, slot, beta * background);
                have = false;
                walk.reset();
            }
            if (__builtin_expect(next >= chunk_end, 0)) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                next = base;
                chunk_end = min(base + chunk, n);
                exhausted = base >= n;
#ifdef MIRT_PATHS_STATS
                if (exhausted && !ps_saw_dry) {
                    ps_saw_dry = true;
                    if (lane == 0) atomicMax(P.segments + 2 + 16 + 8 + 6 + 0, (1ull << 62) - (unsigned long long)__builtin_amdgcn_s_memrealtime());
                }
#endif
            }
            const uint32_t avail = exhausted ? 0u : chunk_end - next;
            const uint32_t rank = mbcnt_lt(need);
            const uint32_t i = next + rank;
            if (!have && rank < avail) {
                // What only a refill reads -- the camera, the pixel set, the queue pipeline's arrays: some
                // hundred scalar registers' worth -- is read from the kernel's argument block HERE, through
                // a pointer the compiler cannot see through, instead of being held (and spilled) around
                // the whole loop.
                KernargWords args = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(args));
                RayD r;
                if (bounce0 == 0) {
                    const RenderParams q = load_kernarg<RenderParams>(args, sizeof(DScene) + sizeof(Paths));
                    // Path i of the pass = sample (i % n_samples) of the (i / n_samples)-th pixel: the 64 paths a wave
                    // takes together are samples of one pixel (or of a few neighbours), so its lanes walk the tree in
                    // step down to the first hit and mostly meet the same material there: -1 to -6 % of the kernel's
                    // time against taking the pass sample by sample, as k_raygen numbers it.  The radiance record
                    // is the path's index either way; k_accumulate_by_pixel reads this numbering.
                    slot = i;
                    const uint32_t k = slot / q.n_samples;
                    const uint32_t idx = q.px.idx(k);
                    PathRng rng{mirt_rng_stream(q.seed, idx, q.sample_base + (slot - k * q.n_samples)), 0u, 0u};
                    r = camera_ray(q.cam, q.width, q.height, idx, rng);
                    beta = v3(1.0, 1.0, 1.0);
                    stream = rng.stream;
                } else {
                    const Paths Q = load_kernarg<Paths>(args, sizeof(DScene));
                    slot = ((bounce0 & 1) ? Q.live[1] : Q.live[0])[i];
                    r.o = v3(Q.ox[slot], Q.oy[slot], Q.oz[slot]);
                    r.d = v3(Q.dx[slot], Q.dy[slot], Q.dz[slot]);
                    r.time = Q.time[slot];
                    beta = v3(Q.bx[slot], Q.by[slot], Q.bz[slot]);
                    stream = Q.stream[slot];
                }
                bounce = bounce0;
                have = true;
                walk.begin(sc, r, rp.t_min, rp.t_max);
            }
            next += min((uint32_t)__popcll(need), avail);
            m_have = vote(have), m_node = walk.vote_node(), m_leaf = walk.vote_leaf();
            m_hit = vote(walk.best.prim >= 0);
            m_finished = m_have & ~(m_node | m_leaf);
            __builtin_amdgcn_s_setprio(MIRT_PRIO_WALK);
            PS_STOP(PS_REFILL, __popcll(need));
        }
        const int n_node = __popcll(m_node), n_leaf = __popcll(m_leaf);
        const int n_finished = __popcll(m_finished), n_shade = __popcll(m_finished & m_hit);
        if (n_node + n_leaf + n_finished == 0) {
            if (exhausted) break;  // no path in flight and the queue is drained
            continue;              // every lane is empty: the next trip refills
        }
        if ((count_missed ? n_finished : n_shade) >= shade_at || n_node + n_leaf == 0) {
            PS_START();
            shade_clock.start();
            __builtin_amdgcn_s_setprio(MIRT_PRIO_SHADE);
            if (have && walk.finished() && __builtin_expect(!walk.confirm(sc, lds, rp.t_min, rp.t_max), 0)) walk.begin_careful(sc, rp.t_min, rp.t_max);
            shade_clock.mark(0);
            // the samples of random_in_unit_sphere the hits of this trip will use: the whole wave works on them
            const bool shading = have && walk.finished();
            const bool last = bounce + 1 == rp.max_depth;
            const uint32_t cls_now = walk.best.prim >= 0 ? ((uint32_t)walk.best.side >> walk.CLASS_SHIFT) & 7u : (uint32_t)SHADE_MISS;
            // the turbulence of the hits on noise textures (a lambertian whose texture IS a noise texture with
            // its table staged: class SHADE_LAMBERTIAN_NOISE): the whole wave works on those, too
            bool have_turbulence = false;
            double turbulence = 0.0;
            if (HAS_NOISE && MIRT_TURBULENCE_WAVE) {
                uint32_t table = 0;
                if (shading && !last && cls_now == SHADE_LAMBERTIAN_NOISE) {
                    const ShapeRec* rec = RECS_IN_LDS ? reinterpret_cast<const ShapeRec*>(lds.shapes + (uint32_t)walk.best.prim * SHAPE_LDS_BYTES)
                                                      : sc.recs + walk.best.prim;
                    table = load_texture(textures, rec->texture).perlin;
                    have_turbulence = table < n_staged;
                }
                if (vote(have_turbulence) != 0ull)
                    turbulence = perlin_turbulence_wave(have_turbulence, walk.r.o + walk.best.t * walk.r.d, table, lds_tables, s_scratch[tid >> 6],
                                                        s_turbulence[tid >> 6]);
            }
            const V3 in_sphere = random_in_unit_sphere_wave(shading && wants_in_sphere(cls_now, last), PathRng{stream, bounce + 1u, 0u},
                                                            s_scratch[tid >> 6]);
            extra_segments += (uint32_t)__popcll(vote(shading && bounce != bounce0));
            if (shading) {
                const HitRec h = walk.best;
                uint32_t cls = SHADE_MISS, hit_ref = HIT_MISS, tk = 0;
                // (a record in LDS is read with LDS instructions only if every value this pointer can take lies in LDS:
                // mixed with a global address it becomes a generic pointer, whose loads travel the long way round
                // and wait for every memory operation in flight)
                const ShapeRec* rec = RECS_IN_LDS ? reinterpret_cast<const ShapeRec*>(lds.shapes) : sc.recs;
                if (h.prim >= 0) {
                    cls = ((uint32_t)h.side >> walk.CLASS_SHIFT) & 7u;
                    tk = ((uint32_t)h.side >> walk.TK_SHIFT) & 7u;
                    hit_ref = (uint32_t)h.prim | (((uint32_t)h.side & walk.SIDE_MASK) << HIT_SIDE_SHIFT);
                    rec = RECS_IN_LDS ? reinterpret_cast<const ShapeRec*>(lds.shapes + (uint32_t)h.prim * SHAPE_LDS_BYTES) : sc.recs + h.prim;
                }
                const ShadeResult s = shade_path<TEXTURES, !MIRT_TURBULENCE_WAVE>(sc, cls, walk.r, h.t, hit_ref, tk, rec, textures, beta, in_sphere, stream, bounce, last, background,
                                                                                  lds_tables, n_staged, shade_clock, have_turbulence, turbulence);
                shade_clock.mark(7);
                // (the store comes last: whatever the next instructions wait for in memory, they would
                // wait for it too)
                if (s.alive) {
                    RayD r;
                    r.o = s.origin;
                    r.d = s.direction;
                    r.time = walk.r.time;
                    beta = s.value;
                    ++bounce;
                    walk.begin(sc, r, rp.t_min, rp.t_max);
                }
                shade_clock.mark(8);
                if (!s.alive) {
                    store_radiance(P, slot, s.value);
                    have = false;
                    walk.reset();
                }
            }
            shade_clock.mark(9);
            __builtin_amdgcn_s_setprio(MIRT_PRIO_WALK);
            PS_STOP(PS_SHADE, n_finished);
            continue;
        }
        const int busy_share = ((int)__popcll(m_node | m_leaf) * stay) >> 3;
        const int enough = busy_share > 1 ? busy_share : 1;
        if (n_leaf * 8 > n_node * leaf_bias) {
            // (the lane's own wish travels round the loop as a flag: evaluated once per trip, for the vote
            // and for the next trip's branch alike)
            int n_now = n_leaf;
            bool wants = walk.wants_leaf();
            __builtin_amdgcn_s_setprio(MIRT_PRIO_LEAF);
            do {
                PS_START();
                if (wants) walk.leaf_step(sc, lds, rp.t_min, rp.t_max);
                PS_STOP(PS_LEAF, n_now);
                wants = walk.wants_leaf();
                n_now = __popcll(walk.vote_leaf());
            } while (n_now >= enough);
            __builtin_amdgcn_s_setprio(MIRT_PRIO_WALK);
        } else {
            int n_now = n_node;
            bool wants = walk.wants_node();
            __builtin_amdgcn_s_setprio(MIRT_PRIO_NODE);
            do {
                PS_START();
#ifdef MIRT_PATHS_STATS
                ps_idle[0] += __popcll(vote(have && walk.finished() && walk.best.prim >= 0));
                ps_idle[1] += __popcll(vote(!have || (walk.finished() && walk.best.prim < 0)));
                ps_idle[2] += __popcll(vote(have && !walk.finished() && !walk.wants_node()));
#endif
                if (wants) walk.node_step(sc, lds, stack, tid, rp.t_min);
                // (a second visit for the lanes that still want one before the wave is asked again: +1.3 % on the
                // 3 507-shape scene, whose walks are long; nothing on trees that are in LDS as a whole)
                if (MIRT_NODE_VISITS_PER_VOTE > 1 && !TREE_IN_LDS && walk.wants_node()) walk.node_step(sc, lds, stack, tid, rp.t_min);
                PS_STOP(PS_NODE, n_now);
                wants = walk.wants_node();
                n_now = __popcll(walk.vote_node());
            } while (n_now >= enough);
            __builtin_amdgcn_s_setprio(MIRT_PRIO_WALK);
        }
    }
    if (lane == 0 && extra_segments) {
        // (k_accumulate counts the segments of this kernel's first bounce with those of the queues)
        atomicAdd(P.segments + 0, (unsigned long long)extra_segments);
        atomicAdd(P.segments + 1, (unsigned long long)extra_segments);
    }
#ifdef MIRT_PATHS_STATS
    if constexpr (SCAN) walk.flush_stats();
    if (lane == 0) {
        for (int k = 0; k < PS_N; ++k) {
            atomicAdd(P.segments + 2 + 3 * k, ps_clk[k]);
            atomicAdd(P.segments + 3 + 3 * k, ps_trips[k]);
            atomicAdd(P.segments + 4 + 3 * k, ps_lanes[k]);
        }
        atomicAdd(P.segments + 2 + 3 * PS_N, (unsigned long long)(__builtin_amdgcn_s_memtime() - ps_begin));
        for (int k = 0; k < ShadeClock::N; ++k) atomicAdd(P.segments + 3 + 3 * PS_N + k, shade_clock.clk[k]);
        atomicAdd(P.segments + 3 + 3 * PS_N + ShadeClock::N, ps_outer);
        for (int k = 0; k < 3; ++k) atomicAdd(P.segments + 4 + 3 * PS_N + ShadeClock::N + k, ps_idle[k]);
        const unsigned long long ps_real_end = __builtin_amdgcn_s_memrealtime();
        atomicMax(P.segments + 2 + 16 + 8 + 6 + 1, (1ull << 62) - ps_real_begin);  // the earliest begin
        atomicMax(P.segments + 2 + 16 + 8 + 6 + 2, ps_real_end);                    // the latest end
        atomicAdd(P.segments + 2 + 16 + 8 + 6 + 3, ps_real_end - ps_real_begin);    // sum of the waves' lifetimes
        atomicAdd(P.segments + 2 + 16 + 8 + 6 + 4, 1ull);                           // waves
    }
#endif
#undef PS_START
#undef PS_STOP
}

}  // namespace mirt
#include "paths_regroup.h"
namespace mirt {

// ------------------------------------------------------------------ accumulate
// pixel += color(sample) for the samples of this pass, in sample order
// (renderer.rs:78-84), into the caller's f64 accumulation image.
__global__ __launch_bounds__(BLOCK) void k_accumulate(Paths P, RenderParams rp, double* __restrict__ accum) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < rp.n_pixels; k += stride) {
        const size_t idx = rp.px.idx(k);
        double r = accum[3 * idx + 0], g = accum[3 * idx + 1], b = accum[3 * idx + 2];
        for (uint32_t s = 0; s < rp.n_samples; ++s) {
            const size_t slot = (size_t)s * rp.n_pixels + k;
            const double* __restrict__ v = P.radiance + slot * RECORD_DOUBLES;
            r += v[0];
            g += v[1];
            b += v[2];
        }
        accum[3 * idx + 0] = r;
        accum[3 * idx + 1] = g;
        accum[3 * idx + 2] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const Counters C{rp.max_depth};
        unsigned long long seg = rp.n_paths;
        // live(b) stays zero for the bounces k_paths ran beyond its first; it adds those itself
        for (uint32_t b = 1; b < rp.max_depth; ++b) seg += P.counters[C.live(b)];
        atomicAdd(P.segments + 0, seg);
        if (rp.handover < rp.max_depth)
            atomicAdd(P.segments + 1, (unsigned long long)(rp.handover == 0 ? rp.n_paths : P.counters[C.live(rp.handover)]));
    }
}

// The same for a pass whose records are numbered pixel by pixel (k_paths from the camera ray: sample s of the k-th
// pixel in record k * n_samples + s).  A wave takes 64 pixels; their records for eight samples at a time -- 192
// contiguous bytes per pixel -- are read with every lane on its own 8 bytes of consecutive runs (whole cache lines but
// for the ends) and handed through LDS to the lane that owns the pixel, which adds them in sample order.
constexpr uint32_t ACC_TILE = 8;                                        // samples per tile
constexpr uint32_t ACC_RUN_WORDS = ACC_TILE * RECORD_DOUBLES;           // doubles of one pixel's run in a tile
constexpr uint32_t ACC_PIXEL_BYTES = ACC_RUN_WORDS * 8 + 8;             // LDS bytes per pixel of a tile (rows 50 banks apart)
constexpr uint32_t ACC_WAVES = 4;
__global__ __launch_bounds__(ACC_WAVES * 64) void k_accumulate_by_pixel(Paths P, RenderParams rp, double* __restrict__ accum) {
    __shared__ __align__(16) char s_tile[ACC_WAVES][64 * ACC_PIXEL_BYTES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* tile = s_tile[wave];
    const uint32_t n_groups = (rp.n_pixels + 63u) >> 6, S = rp.n_samples;
    const double* __restrict__ records = P.radiance;
    for (uint32_t group = blockIdx.x * ACC_WAVES + wave; group < n_groups; group += gridDim.x * ACC_WAVES) {
        const uint32_t k0 = group << 6, k = k0 + lane;
        const uint32_t pixels = min(64u, rp.n_pixels - k0);
        const bool mine = k < rp.n_pixels;
        const size_t idx = mine ? rp.px.idx(k) : 0;
        double r = 0.0, g = 0.0, b = 0.0;
        if (mine) r = accum[3 * idx + 0], g = accum[3 * idx + 1], b = accum[3 * idx + 2];
        for (uint32_t s0 = 0; s0 < S; s0 += ACC_TILE) {
            const uint32_t T = min(ACC_TILE, S - s0);
            if (pixels == 64u && T == ACC_TILE) {
                // twenty-four loads in flight per lane, then as many LDS writes
                double v[ACC_RUN_WORDS];
#pragma unroll
                for (uint32_t j = 0; j < ACC_RUN_WORDS; ++j) {
                    const uint32_t word = j * 64u + lane, pl = word / ACC_RUN_WORDS, w = word - pl * ACC_RUN_WORDS;
                    v[j] = records[((size_t)(k0 + pl) * S + s0) * RECORD_DOUBLES + w];
                }
#pragma unroll
                for (uint32_t j = 0; j < ACC_RUN_WORDS; ++j) {
                    const uint32_t word = j * 64u + lane, pl = word / ACC_RUN_WORDS, w = word - pl * ACC_RUN_WORDS;
                    *reinterpret_cast<double*>(tile + pl * ACC_PIXEL_BYTES + w * 8u) = v[j];
                }
            } else {
                const uint32_t per_pixel = RECORD_DOUBLES * T;  // doubles of one pixel's run
                for (uint32_t word = lane; word < pixels * per_pixel; word += 64u) {
                    const uint32_t pl = word / per_pixel, w = word - pl * per_pixel;
                    *reinterpret_cast<double*>(tile + pl * ACC_PIXEL_BYTES + w * 8u) = records[((size_t)(k0 + pl) * S + s0) * RECORD_DOUBLES + w];
                }
            }
            // (one wave writes and reads its own tile: its LDS operations complete in program order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (mine) {
                const double* __restrict__ mine_rows = reinterpret_cast<const double*>(tile + lane * ACC_PIXEL_BYTES);
                for (uint32_t t = 0; t < T; ++t) {
                    r += mine_rows[RECORD_DOUBLES * t + 0];
                    g += mine_rows[RECORD_DOUBLES * t + 1];
                    b += mine_rows[RECORD_DOUBLES * t + 2];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (mine) accum[3 * idx + 0] = r, accum[3 * idx + 1] = g, accum[3 * idx + 2] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (this numbering belongs to passes that k_paths runs from the camera ray: see k_accumulate)
        atomicAdd(P.segments + 0, (unsigned long long)rp.n_paths);
        atomicAdd(P.segments + 1, (unsigned long long)rp.n_paths);
    }
}

// --------------------------------------------------------------------- resolve
// renderer.rs:85-89 and the `as u8` of renderer.rs:42 (saturating, truncating, NaN -> 0).
__device__ __forceinline__ uint8_t f64_as_u8(double x) {
    if (x != x) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)(int)x;
}

// out[k] = bytes of the k-th pixel of the set.
__global__ __launch_bounds__(BLOCK) void k_resolve(const double* __restrict__ accum, PixelSet px, uint32_t spp,
                                                   uint8_t* __restrict__ out) {
    const uint32_t stride = gridDim.x * BLOCK;
    const double n = (double)spp;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < px.n_pixels; k += stride) {
        const size_t idx = px.idx(k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = accum[3 * idx + c] / n;
            v = sqrt(v);
            v = v * 255.99;
            out[3 * (size_t)k + c] = f64_as_u8(v);
        }
    }
}

}  // namespace mirt
