// multi_device.h -- one frame over several devices from ONE process: mirt_render_multi.
// Included at the end of mirt.hip (same translation unit: it calls render_accumulate_impl).
//
// The reference spreads pixels over the threads of one process (rayon, src/renderer.rs:27-43);
// pixels and samples are independent, so a frame spreads over devices the same way.  One host
// thread drives each rank (= one entry of the caller's scene list, on that scene's device):
//
//   tile split    rank r renders the tiles t with t % n == r of the pixel order (a tile = one
//                 image row by default) and copies them to their places in the caller's buffer.
//                 No exchange between devices.  Bytes equal the one-device render's.
//   sample split  rank r renders every pixel for its contiguous share of the samples into an f64
//                 image on its device; then every rank OWNS a contiguous slice of the pixels, sums
//                 that slice of all ranks' images and resolves it to bytes (a reduce-scatter, then a
//                 gather into the caller's buffer through the host copies the bytes need anyway).
//                 The sum is either the library's own -- peer copies of the slices (peer access is asked
//                 for and switched on per pair of devices; a pair without it goes through host memory),
//                 added in rank order, so the bytes do not depend on timing -- or RCCL's
//                 ncclReduceScatter (ring order; needs distinct devices).  The ranks' buffers are kept
//                 between calls (mirt_multi_trim frees them).
#pragma once

#include <dlfcn.h>

#include <map>
#include <thread>

namespace mirt {

constexpr uint32_t MULTI_MAX_RANKS = 16;

struct SlicePtrs {
    const double* p[MULTI_MAX_RANKS];
};

// out[3k + c] = bytes of pixel k of the slice, from the sum over `n` images (in index order) of
// value 3k + c of each image's slice.  renderer.rs:85-89 and the `as u8` of renderer.rs:42.
__global__ __launch_bounds__(BLOCK) void k_sum_resolve(SlicePtrs src, uint32_t n, uint32_t n_values, uint32_t spp,
                                                       uint8_t* __restrict__ out) {
    const uint32_t stride = gridDim.x * BLOCK;
    const double count = (double)spp;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_values; i += stride) {
        double v = src.p[0][i];
        for (uint32_t q = 1; q < n; ++q) v += src.p[q][i];
        v = v / count;
        v = sqrt(v);
        v = v * 255.99;
        out[i] = f64_as_u8(v);
    }
}

}  // namespace mirt

namespace {

// ---- RCCL, bound at run time: only callers that ask for MIRT_REDUCE_RCCL load it
struct Rccl {
    typedef int (*CommInitAll)(void** comms, int ndev, const int* devlist);
    typedef int (*CommDestroy)(void* comm);
    typedef int (*ReduceScatter)(const void* send, void* recv, size_t recvcount, int datatype, int op, void* comm, hipStream_t stream);
    typedef const char* (*GetErrorString)(int);
    void* handle = nullptr;
    CommInitAll comm_init_all = nullptr;
    CommDestroy comm_destroy = nullptr;
    ReduceScatter reduce_scatter = nullptr;
    GetErrorString error_string = nullptr;
    static constexpr int DOUBLE = 8, SUM = 0;  // ncclDouble, ncclSum (rccl.h)
};

std::mutex g_rccl_mutex;
Rccl g_rccl;
std::map<std::vector<int>, std::vector<void*>> g_rccl_comms;  // device list -> communicators, kept to process end

int rccl_load() {
    if (g_rccl.handle) return MIRT_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* name : names)
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(MIRT_ERR_UNSUPPORTED, "MIRT_REDUCE_RCCL: librccl.so could not be loaded: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.comm_init_all = (Rccl::CommInitAll)dlsym(h, "ncclCommInitAll");
    r.comm_destroy = (Rccl::CommDestroy)dlsym(h, "ncclCommDestroy");
    r.reduce_scatter = (Rccl::ReduceScatter)dlsym(h, "ncclReduceScatter");
    r.error_string = (Rccl::GetErrorString)dlsym(h, "ncclGetErrorString");
    if (!r.comm_init_all || !r.comm_destroy || !r.reduce_scatter || !r.error_string) {
        dlclose(h);
        return fail(MIRT_ERR_UNSUPPORTED, "MIRT_REDUCE_RCCL: librccl.so lacks ncclCommInitAll / ncclReduceScatter");
    }
    g_rccl = r;
    return MIRT_OK;
}

// One communicator per rank over `devices` (distinct ordinals), created on first use.
int rccl_comms_for(const std::vector<int>& devices, std::vector<void*>* out) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = rccl_load();
    if (rc != MIRT_OK) return rc;
    auto it = g_rccl_comms.find(devices);
    if (it == g_rccl_comms.end()) {
        std::vector<void*> comms(devices.size(), nullptr);
        const int e = g_rccl.comm_init_all(comms.data(), (int)devices.size(), devices.data());
        if (e != 0) return fail(MIRT_ERR_HIP, "ncclCommInitAll over %zu devices failed: %s", devices.size(), g_rccl.error_string(e));
        it = g_rccl_comms.emplace(devices, comms).first;
    }
    *out = it->second;
    return MIRT_OK;
}

// ---- what the sample split keeps between calls, per rank slot: the rank's f64 image (padded to n equal
// slices), its staging area and the bytes of its slice.  Guarded by g_multi_mutex for the whole of a call's
// sample split; mirt_multi_trim gives the memory back.
struct MultiSlot {
    int device = -1;
    double* accum = nullptr;
    size_t accum_values = 0;
    double* staging = nullptr;
    size_t staging_values = 0;
    uint8_t* bytes = nullptr;
    size_t bytes_size = 0;
    // Call with `dev` current.
    void drop() {
        (void)hipFree(accum);
        (void)hipFree(staging);
        (void)hipFree(bytes);
        accum = staging = nullptr;
        bytes = nullptr;
        accum_values = staging_values = bytes_size = 0;
    }
    int reserve(int dev, size_t want_accum, size_t want_staging, size_t want_bytes) {
        if (device != dev && device >= 0) {
            DeviceGuard g;
            int e = g.enter(device);
            if (e != MIRT_OK) return e;
            drop();
        }
        device = dev;
        auto grow = [](auto** p, size_t* have, size_t want) -> hipError_t {
            if (*have >= want) return hipSuccess;
            (void)hipFree(*p);
            *p = nullptr;
            *have = 0;
            hipError_t e = hipMalloc((void**)p, want * sizeof(**p));
            if (e == hipSuccess) *have = want;
            return e;
        };
        HIP_TRY(grow(&accum, &accum_values, want_accum));
        HIP_TRY(grow(&staging, &staging_values, want_staging));
        HIP_TRY(grow(&bytes, &bytes_size, want_bytes));
        return MIRT_OK;
    }
};
std::mutex g_multi_mutex;
MultiSlot g_multi_slots[mirt::MULTI_MAX_RANKS];

// Can device `to` read device `from`'s memory directly?  Asked of the runtime and, if so, switched on (once
// per pair and process; "already on" is as good).
int peer_access(int to, int from, uint8_t* direct) {
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, to, from));
    *direct = 0;
    if (!can) return MIRT_OK;
    DeviceGuard g;
    int e = g.enter(to);
    if (e != MIRT_OK) return e;
    const hipError_t h = hipDeviceEnablePeerAccess(from, 0);
    if (h != hipSuccess && h != hipErrorPeerAccessAlreadyEnabled)
        return fail(MIRT_ERR_HIP, "hipDeviceEnablePeerAccess(%d) on device %d failed: %s", from, to, hipGetErrorString(h));
    (void)hipGetLastError();  // (the "already enabled" status is sticky)
    *direct = 1;
    return MIRT_OK;
}

struct RankResult {
    int rc = MIRT_OK;
    std::string error;
};

// Runs fn(rank) on one thread per rank and returns the first failure (with its thread's message).
template <class F>
int run_ranks(uint32_t n, F fn) {
    std::vector<RankResult> results(n);
    std::vector<std::thread> threads;
    threads.reserve(n);
    bool started_all = true;
    for (uint32_t r = 0; r < n && started_all; ++r) {
        try {
            threads.emplace_back([&, r] {
                results[r].rc = no_throw([&] { return fn(r); });
                if (results[r].rc != MIRT_OK) results[r].error = tl_error;
            });
        } catch (...) {  // the system would not start another thread
            started_all = false;
        }
    }
    for (auto& t : threads) t.join();
    if (!started_all) return fail(MIRT_ERR_OUT_OF_MEMORY, "could not start a host thread per rank");
    for (uint32_t r = 0; r < n; ++r)
        if (results[r].rc != MIRT_OK) return fail(results[r].rc, "rank %u: %s", r, results[r].error.c_str());
    return MIRT_OK;
}

}  // namespace

static int render_multi_body(const mirt_scene* const* scenes, uint32_t n_scenes, const mirt_camera* camera,
                             const double background[3], uint32_t width, uint32_t height, uint32_t spp,
                             uint32_t max_depth, uint64_t seed, const mirt_render_opts* opts,
                             const mirt_multi_opts* multi, uint8_t* out_rgb, mirt_stats* stats) {
    if (!scenes || n_scenes == 0 || !camera || !background || !out_rgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_scenes > MULTI_MAX_RANKS) return fail(MIRT_ERR_UNSUPPORTED, "more than %u scenes", MULTI_MAX_RANKS);
    for (uint32_t r = 0; r < n_scenes; ++r)
        if (!scenes[r]) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene %u is null", r);
    if (width == 0 || height == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    if (spp == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "spp must be > 0");
    mirt_multi_opts m;
    std::memset(&m, 0, sizeof m);
    if (multi) {
        if (multi->struct_size != sizeof(mirt_multi_opts))
            return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_multi_opts.struct_size is %u, expected %zu", multi->struct_size, sizeof(mirt_multi_opts));
        m = *multi;
    }
    if (m.split != MIRT_SPLIT_TILE && m.split != MIRT_SPLIT_SAMPLE) return fail(MIRT_ERR_INVALID_ARGUMENT, "unknown split %d", m.split);
    if (m.reduce != MIRT_REDUCE_PEER && m.reduce != MIRT_REDUCE_RCCL) return fail(MIRT_ERR_INVALID_ARGUMENT, "unknown reduce %d", m.reduce);
    mirt_render_opts base;
    std::memset(&base, 0, sizeof base);
    base.struct_size = sizeof base;
    base.use_bvh = 1;
    base.traversal = MIRT_TRAVERSE_FAST;
    if (opts) {
        if (opts->struct_size != sizeof(mirt_render_opts))
            return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_render_opts.struct_size is %u, expected %zu", opts->struct_size, sizeof(mirt_render_opts));
        base = *opts;
        if (base.pixel_begin || base.pixel_end || base.sample_begin || base.sample_end || base.tile_count > 1)
            return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_render_multi splits the frame itself: leave the pixel, sample and tile fields of opts zero");
    }
    base.device = -1;
    const uint32_t n = n_scenes, n_pixels = width * height;
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<mirt_stats> rank_stats(n);
    for (auto& s : rank_stats) std::memset(&s, 0, sizeof s);
    int rc = MIRT_OK;

    if (m.split == MIRT_SPLIT_TILE) {
        rc = run_ranks(n, [&](uint32_t r) -> int {
            mirt_render_opts o = base;
            o.tile_size = m.tile_size ? m.tile_size : width;
            o.tile_count = n;
            o.tile_index = r;
            return mirt_render(scenes[r], camera, background, width, height, spp, max_depth, seed, &o, out_rgb, &rank_stats[r]);
        });
    } else {
        // ---- sample split.  Rank r: samples [r * spp / n, (r + 1) * spp / n), pixel slice r of n for the sum.
        const uint32_t slice_pixels = (n_pixels + n - 1) / n;
        const size_t slice_values = (size_t)slice_pixels * 3, padded_values = slice_values * n;
        std::vector<int> devices(n);
        for (uint32_t r = 0; r < n; ++r) devices[r] = scenes[r]->device;
        std::vector<void*> comms;
        if (m.reduce == MIRT_REDUCE_RCCL) {
            std::vector<int> sorted = devices;
            std::sort(sorted.begin(), sorted.end());
            if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
                return fail(MIRT_ERR_UNSUPPORTED, "MIRT_REDUCE_RCCL needs every scene on a device of its own; use MIRT_REDUCE_PEER");
            if ((rc = rccl_comms_for(devices, &comms)) != MIRT_OK) return rc;
        }
        // The ranks' images, staging areas and byte buffers are kept between calls (g_multi_slots: 2x-2.9x the
        // f64 image per rank at 3840x2160 would otherwise be allocated and freed inside what a caller times);
        // calls that split by samples take turns.
        std::lock_guard<std::mutex> turn(g_multi_mutex);
        // Every device must be usable before any rank starts: a rank that cannot enter its device would
        // leave the others waiting in the collective.
        for (uint32_t r = 0; r < n; ++r) {
            DeviceGuard g;
            if ((rc = g.enter(devices[r])) != MIRT_OK) return rc;
        }
        // Which slices can travel device to device: peer access asked for and switched on per pair, once.
        std::vector<uint8_t> direct((size_t)n * n, 1);
        if (m.reduce == MIRT_REDUCE_PEER)
            for (uint32_t r = 0; r < n; ++r)
                for (uint32_t q = 0; q < n; ++q) {
                    if (devices[q] == devices[r]) continue;
                    if ((rc = peer_access(devices[r], devices[q], &direct[(size_t)r * n + q])) != MIRT_OK) return rc;
                }
        const size_t staging_values = m.reduce == MIRT_REDUCE_RCCL ? slice_values : slice_values * (n - 1);
        rc = run_ranks(n, [&](uint32_t r) -> int {
            DeviceGuard g;
            int e = g.enter(devices[r]);
            if (e != MIRT_OK) return e;
            MultiSlot& mem = g_multi_slots[r];
            if ((e = mem.reserve(devices[r], padded_values, std::max<size_t>(staging_values, 1), slice_values)) != MIRT_OK) return e;
            HIP_TRY(hipMemsetAsync(mem.accum, 0, padded_values * sizeof(double), 0));
            const uint32_t lo = (uint32_t)((uint64_t)r * spp / n), hi = (uint32_t)((uint64_t)(r + 1) * spp / n);
            if (lo == hi) {  // fewer samples than ranks: this rank adds nothing but takes part in the sum
                HIP_TRY(hipStreamSynchronize(0));
                return MIRT_OK;
            }
            mirt_render_opts o = base;
            o.sample_begin = lo;
            o.sample_end = hi;
            return render_accumulate_impl(scenes[r], camera, background, width, height, spp, max_depth, seed, &o, mem.accum, &rank_stats[r]);
        });
        if (rc == MIRT_OK)
            rc = run_ranks(n, [&](uint32_t r) -> int {
                DeviceGuard g;
                int e = g.enter(devices[r]);
                if (e != MIRT_OK) return e;
                MultiSlot& mem = g_multi_slots[r];
                const uint32_t first = std::min(r * slice_pixels, n_pixels), count = std::min(slice_pixels, n_pixels - first);
                const size_t off = (size_t)first * 3, values = (size_t)count * 3;
                SlicePtrs src;
                std::memset(&src, 0, sizeof src);
                uint32_t n_src = n;
                if (m.reduce == MIRT_REDUCE_RCCL) {
                    // (every rank calls the collective, also one whose slice of a tiny frame is empty: its peers wait in it)
                    const int ne = g_rccl.reduce_scatter(mem.accum, mem.staging, slice_values, Rccl::DOUBLE, Rccl::SUM, comms[r], 0);
                    if (ne != 0) return fail(MIRT_ERR_HIP, "ncclReduceScatter failed: %s", g_rccl.error_string(ne));
                    src.p[0] = mem.staging;
                    n_src = 1;
                    if (count == 0) {
                        HIP_TRY(hipStreamSynchronize(0));
                        return MIRT_OK;
                    }
                } else {
                    if (count == 0) return MIRT_OK;
                    double* stage = mem.staging;
                    std::vector<double> bounce;  // only for a pair of devices without peer access
                    for (uint32_t q = 0; q < n; ++q) {
                        const double* theirs = g_multi_slots[q].accum + off;
                        if (devices[q] == devices[r]) {  // same device: read in place
                            src.p[q] = theirs;
                            continue;
                        }
                        if (direct[(size_t)r * n + q]) {
                            HIP_TRY(hipMemcpyPeerAsync(stage, devices[r], theirs, devices[q], values * sizeof(double), 0));
                        } else {
                            // no peer access between the two: through host memory, said out loud
                            bounce.resize(values);
                            HIP_TRY(hipMemcpy(bounce.data(), theirs, values * sizeof(double), hipMemcpyDeviceToHost));
                            HIP_TRY(hipMemcpy(stage, bounce.data(), values * sizeof(double), hipMemcpyHostToDevice));
                        }
                        src.p[q] = stage;
                        stage += slice_values;
                    }
                }
                const int grid = (int)std::min<size_t>((values + BLOCK - 1) / BLOCK, 4096);
                hipLaunchKernelGGL(k_sum_resolve, dim3(grid), dim3(BLOCK), 0, 0, src, n_src, (uint32_t)values, spp, mem.bytes);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpy(out_rgb + off, mem.bytes, values, hipMemcpyDeviceToHost));
                return MIRT_OK;
            });
        if (rc != MIRT_OK) {
            // whatever a failed rank left in flight must not outlive the call in buffers the next call reuses
            for (uint32_t r = 0; r < n; ++r) {
                DeviceGuard g;
                if (g.enter(devices[r]) == MIRT_OK) (void)hipDeviceSynchronize();
            }
        }
    }
    if (rc != MIRT_OK) return rc;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    if (stats)
        for (uint32_t r = 0; r < n; ++r) {
            stats[r] = rank_stats[r];
            stats[r].ms_total = ms;
        }
    return MIRT_OK;
}

extern "C" int mirt_render_multi(const mirt_scene* const* scenes, uint32_t n_scenes, const mirt_camera* camera,
                                 const double background[3], uint32_t width, uint32_t height, uint32_t spp,
                                 uint32_t max_depth, uint64_t seed, const mirt_render_opts* opts,
                                 const mirt_multi_opts* multi, uint8_t* out_rgb, mirt_stats* stats) {
    return no_throw([&] { return render_multi_body(scenes, n_scenes, camera, background, width, height, spp, max_depth, seed, opts, multi, out_rgb, stats); });
}

extern "C" int mirt_multi_trim(void) {
    return no_throw([&]() -> int {
        std::lock_guard<std::mutex> turn(g_multi_mutex);
        for (MultiSlot& slot : g_multi_slots) {
            if (slot.device < 0) continue;
            DeviceGuard g;
            int e = g.enter(slot.device);
            if (e != MIRT_OK) return e;
            slot.drop();
            slot.device = -1;
        }
        return MIRT_OK;
    });
}
