// mirt.hip -- the C ABI of include/mirt.h over the gfx950 kernels of kernels.h:
// scene validation + upload, the closest-hit entry points, and the pass loop
// that replaces render_world (reference src/renderer.rs:9-62).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "accel_build.h"
#include "kernels.h"

using namespace mirt;

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string tl_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tl_error = buf;
    return code;
}

// Nothing may propagate across the C boundary: an entry point whose body allocates host memory (vectors,
// threads) runs it through this.
template <class F>
int no_throw(F body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(MIRT_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(MIRT_ERR_HIP, "unexpected failure: %s", e.what());
    } catch (...) {
        return fail(MIRT_ERR_HIP, "unexpected failure");
    }
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_HIP,       \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Frees a set of device allocations when it goes out of scope.
struct DeviceArena {
    std::vector<void*> ptrs;
    ~DeviceArena() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
    void release() { ptrs.clear(); }
};

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    int enter(int device) {
        if (device < 0) return MIRT_OK;
        if (hipGetDevice(&prev) != hipSuccess) return fail(MIRT_ERR_NO_DEVICE, "hipGetDevice failed: no HIP device");
        if (prev != device) {
            if (hipSetDevice(device) != hipSuccess) return fail(MIRT_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
            active = true;
        }
        return MIRT_OK;
    }
    ~DeviceGuard() {
        if (active) (void)hipSetDevice(prev);
    }
};
}  // namespace

struct mirt_scene {
    int device = 0;
    DScene d{};
    std::vector<void*> allocations;
    uint32_t height = 0;        // levels of inner nodes on the longest root-to-leaf chain
    bool has_bvh = false;
    bool has_noise = false;
    int texture_level = 0;      // TextureLevel: the highest texture kind the scene holds
    int cu_count = 0;
    // One block of device memory a render call left behind for the next one (the per-pass
    // buffers: 2 GB at the default pass size); a call takes it if it is large enough and
    // gives back the larger of what it holds and what is cached.  Guarded by ws_mutex:
    // calls on one scene may run concurrently, each then works in a block of its own.
    mutable std::mutex ws_mutex;
    mutable void* ws_ptr = nullptr;
    mutable size_t ws_bytes = 0;
    // ... and a second one, kept the same way: the f64 accumulation image and the resolved bytes of mirt_render
    // (23 + 3 MB for a 1200x800 frame: allocating and freeing them costs a call some 0.15 ms)
    mutable void* frame_ptr = nullptr;
    mutable size_t frame_bytes = 0;
};

namespace {
struct Workspace {
    const mirt_scene* scene = nullptr;
    void* ptr = nullptr;
    size_t bytes = 0;
    bool frame = false;  // the scene's second block (mirt_scene::frame_ptr) instead of its first
    hipError_t take(const mirt_scene* sc, size_t want, bool frame_block = false) {
        scene = sc;
        frame = frame_block;
        {
            std::lock_guard<std::mutex> lock(sc->ws_mutex);
            void*& kept = frame ? sc->frame_ptr : sc->ws_ptr;
            size_t& kept_bytes = frame ? sc->frame_bytes : sc->ws_bytes;
            if (kept && kept_bytes >= want) {
                ptr = kept;
                bytes = kept_bytes;
                kept = nullptr;
                kept_bytes = 0;
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess) bytes = want;
        else ptr = nullptr;
        return e;
    }
    ~Workspace() {
        if (!ptr) return;
        void* drop = ptr;
        {
            std::lock_guard<std::mutex> lock(scene->ws_mutex);
            void*& kept = frame ? scene->frame_ptr : scene->ws_ptr;
            size_t& kept_bytes = frame ? scene->frame_bytes : scene->ws_bytes;
            if (bytes > kept_bytes) {
                drop = kept;
                kept = ptr;
                kept_bytes = bytes;
            }
        }
        if (drop) (void)hipFree(drop);
    }
};
}  // namespace

// ---- the array behind mirt_render_accumulate_static: one pass's radiance records, counters and
// totals (the block the other entry points allocate).  Every device the library runs on holds its own
// copy at its own address until the process ends.
constexpr uint32_t STATIC_PASS_PATHS = MIRT_STATIC_PASS_PATHS;
constexpr size_t STATIC_TAIL_BYTES = 16 * 1024;  // counters (at most 2551 words at depth 255) + totals
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)STATIC_PASS_PATHS * 32 + STATIC_TAIL_BYTES;  // (room for 32-byte records; they are 24)
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {
// In host code `scratch` is only a handle for hipGetSymbolAddress; kernels, memsets and copies
// get the address it returns for the device that is current, kept here per device.  The address is
// asked for by the first mirt_render_accumulate_static call on a device (nothing else is bound by
// MAX_DEVICES); the array itself belongs to the code object and occupies every device the library's
// kernels are loaded on, used or not.
constexpr int MAX_DEVICES = 64;
struct ScratchOfDevice {
    std::mutex setup;        // guards ptr
    void* ptr = nullptr;
    std::mutex in_use;       // one call at a time per device: the array is shared
};
ScratchOfDevice g_scratch[MAX_DEVICES];

// Call with `device` current.  Asks the runtime once per device.
int scratch_address(int device, void** out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(MIRT_ERR_UNSUPPORTED, "device ordinal %d is beyond %d", device, MAX_DEVICES - 1);
    ScratchOfDevice& s = g_scratch[device];
    std::lock_guard<std::mutex> lock(s.setup);
    if (!s.ptr) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(MIRT_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
        s.ptr = p;
    }
    *out = s.ptr;
    return MIRT_OK;
}
}  // namespace

extern "C" {

uint32_t mirt_abi_version(void) { return 1; }

#ifndef MIRT_BUILD_ID
#define MIRT_BUILD_ID "unknown"
#endif
const char* mirt_build_id(void) { return MIRT_BUILD_ID; }

const char* mirt_last_error(void) { return tl_error.c_str(); }

int mirt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------ scene create
static uint32_t shade_class_of(const mirt_scene_desc* desc, const mirt_material& m) {
    switch (m.kind) {
        case MIRT_MAT_DIELECTRIC: return SHADE_DIELECTRIC;
        case MIRT_MAT_METAL: return SHADE_METAL;
        case MIRT_MAT_DIFFUSE_LIGHT: return SHADE_LIGHT;
        default: break;
    }
    switch (desc->textures[m.texture].kind) {
        case MIRT_TEX_CHECKER: return SHADE_LAMBERTIAN_CHECKER;
        case MIRT_TEX_NOISE: return SHADE_LAMBERTIAN_NOISE;
        case MIRT_TEX_IMAGE: return SHADE_LAMBERTIAN_IMAGE;
        default: return SHADE_LAMBERTIAN_CONSTANT;
    }
}

static int validate_desc(const mirt_scene_desc* d) {
    if (!d) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene description is null");
    if (!d->shapes || d->n_shapes == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene has no shapes");
    if (d->n_shapes > MAX_SHAPES) return fail(MIRT_ERR_UNSUPPORTED, "more than %u shapes", MAX_SHAPES);
    if (!d->materials || d->n_materials == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene has no materials");
    if (d->n_materials > REC_MATERIAL_MASK) return fail(MIRT_ERR_UNSUPPORTED, "more than %u materials", REC_MATERIAL_MASK);
    if (d->n_textures && !d->textures) return fail(MIRT_ERR_INVALID_ARGUMENT, "textures is null");
    if (d->n_perlins && !d->perlins) return fail(MIRT_ERR_INVALID_ARGUMENT, "perlins is null");
    if (d->n_images && !d->images) return fail(MIRT_ERR_INVALID_ARGUMENT, "images is null");
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mirt_texture& t = d->textures[i];
        switch (t.kind) {
            case MIRT_TEX_CONSTANT: break;
            case MIRT_TEX_CHECKER:
                // children must precede the parent: keeps the texture graph acyclic
                if (t.even >= i || t.odd >= i)
                    return fail(MIRT_ERR_INVALID_ARGUMENT, "texture %u: checker children must have smaller indices", i);
                break;
            case MIRT_TEX_NOISE:
                if (t.perlin >= d->n_perlins) return fail(MIRT_ERR_INVALID_ARGUMENT, "texture %u: bad perlin index", i);
                break;
            case MIRT_TEX_IMAGE:
                if (t.image >= d->n_images) return fail(MIRT_ERR_INVALID_ARGUMENT, "texture %u: bad image index", i);
                break;
            default: return fail(MIRT_ERR_INVALID_ARGUMENT, "texture %u: unknown kind %u", i, t.kind);
        }
    }
    for (uint32_t i = 0; i < d->n_perlins; ++i)
        for (int k = 0; k < 256; ++k) {
            const mirt_perlin& p = d->perlins[i];
            if ((uint32_t)p.perm_x[k] > 255u || (uint32_t)p.perm_y[k] > 255u || (uint32_t)p.perm_z[k] > 255u)
                return fail(MIRT_ERR_INVALID_ARGUMENT, "perlin %u: permutation entry out of 0..255", i);
        }
    for (uint32_t i = 0; i < d->n_images; ++i)
        if (!d->images[i].rgb || d->images[i].width == 0 || d->images[i].height == 0)
            return fail(MIRT_ERR_INVALID_ARGUMENT, "image %u: empty", i);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mirt_material& m = d->materials[i];
        if (m.kind > MIRT_MAT_DIFFUSE_LIGHT) return fail(MIRT_ERR_INVALID_ARGUMENT, "material %u: unknown kind %u", i, m.kind);
        if ((m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_DIFFUSE_LIGHT) && m.texture >= d->n_textures)
            return fail(MIRT_ERR_INVALID_ARGUMENT, "material %u: bad texture index", i);
    }
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const mirt_shape& s = d->shapes[i];
        if (s.kind > MIRT_SHAPE_MOVING_SPHERE) return fail(MIRT_ERR_INVALID_ARGUMENT, "shape %u: unknown kind %u", i, s.kind);
        if (s.material >= d->n_materials) return fail(MIRT_ERR_INVALID_ARGUMENT, "shape %u: bad material index", i);
    }
    if (d->nodes) {
        if (d->n_nodes != 2 * d->n_shapes - 1)
            return fail(MIRT_ERR_INVALID_ARGUMENT, "BVH has %u members, expected 2n-1 = %u", d->n_nodes, 2 * d->n_shapes - 1);
        std::vector<uint8_t> shape_seen(d->n_shapes, 0), member_used(d->n_nodes, 0);
        for (uint32_t i = 0; i < d->n_nodes; ++i) {
            const mirt_bvh_member& m = d->nodes[i];
            if (m.is_leaf) {
                if (m.shape >= d->n_shapes || shape_seen[m.shape])
                    return fail(MIRT_ERR_INVALID_ARGUMENT, "BVH member %u: bad or repeated shape index", i);
                shape_seen[m.shape] = 1;
            } else {
                // post-order: children precede their parent (bvh.rs:204-236)
                if (m.left >= i || m.right >= i || m.left == m.right || member_used[m.left] || member_used[m.right])
                    return fail(MIRT_ERR_INVALID_ARGUMENT, "BVH member %u: children must be distinct, unshared earlier members", i);
                member_used[m.left] = member_used[m.right] = 1;
            }
        }
        for (uint32_t i = 0; i + 1 < d->n_nodes; ++i)
            if (!member_used[i]) return fail(MIRT_ERR_INVALID_ARGUMENT, "BVH member %u is not reachable from the root", i);
        // Leaves from left to right must be shapes 0, 1, 2, ...: the reference's build leaves them so
        // (bvh.rs:24-26,179-236), and "equal t: the right subtree wins" (bvh.rs:70-74) is "the larger shape
        // index wins" only then -- the fast traversal breaks ties by index.
        std::vector<uint32_t> todo;
        todo.push_back(d->n_nodes - 1);
        uint32_t next_shape = 0;
        while (!todo.empty()) {
            const mirt_bvh_member& m = d->nodes[todo.back()];
            todo.pop_back();
            if (m.is_leaf) {
                if (m.shape != next_shape)
                    return fail(MIRT_ERR_INVALID_ARGUMENT, "BVH leaves must hold shapes 0, 1, 2, ... from left to right: found shape %u where %u belongs", m.shape, next_shape);
                ++next_shape;
            } else {
                todo.push_back(m.right);
                todo.push_back(m.left);
            }
        }
    }
    return MIRT_OK;
}

static int scene_create_body(const mirt_scene_desc* desc, int device, mirt_scene** out) {
    if (!out) return fail(MIRT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc != MIRT_OK) return rc;
    if (mirt_device_count() == 0) return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible");
    DeviceGuard guard;
    if ((rc = guard.enter(device)) != MIRT_OK) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));

    // ---- host-side flattening
    std::vector<DShape> shapes(desc->n_shapes);
    bool has_noise = false;
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        std::memset(&shapes[i], 0, sizeof(DShape));
        std::memcpy(shapes[i].f, s.f, sizeof s.f);
        shapes[i].kind = s.kind;
        shapes[i].material = s.material;
        shapes[i].cls = shade_class_of(desc, desc->materials[s.material]);
    }
    int texture_level = TEXTURES_CONSTANT;
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        has_noise |= desc->textures[i].kind == MIRT_TEX_NOISE;
        switch (desc->textures[i].kind) {
            case MIRT_TEX_CHECKER: texture_level = std::max<int>(texture_level, TEXTURES_CHECKER); break;
            case MIRT_TEX_IMAGE: texture_level = std::max<int>(texture_level, TEXTURES_IMAGE); break;
            case MIRT_TEX_NOISE: texture_level = std::max<int>(texture_level, TEXTURES_ALL); break;
            default: break;
        }
    }

    std::vector<uint32_t> tkinds;
    std::vector<ShapeRec> recs;
    std::vector<mirt_texture> shade_textures;
    traversal_shapes(desc, &tkinds, &recs);
    shading_records(desc, &recs, &shade_textures);
    std::vector<uint32_t> ref_bits(desc->n_shapes);  // what a leaf reference says about its shape
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        shapes[i].tk = tkinds[i];
        ref_bits[i] = (tkinds[i] << REF_KIND_SHIFT) | (shapes[i].cls << REF_CLASS_SHIFT);
        recs[i].material |= shapes[i].cls << REC_CLASS_SHIFT;
    }
    BuiltTree tree;
    if (desc->nodes) {
        tree = build_fast_tree(desc, ref_bits.data(), MAX_TREE_HEIGHT);
        if (tree.height > (uint32_t)MAX_TREE_HEIGHT)
            return fail(MIRT_ERR_UNSUPPORTED, "BVH is %u levels deep; at most %d are supported", tree.height, MAX_TREE_HEIGHT);
        // the reference-order kernel walks the caller's own members: bound their depth too
        std::vector<uint32_t> level(desc->n_nodes, 1);
        uint32_t deepest = 1;
        for (uint32_t i = desc->n_nodes; i-- > 0;) {
            const mirt_bvh_member& m = desc->nodes[i];
            deepest = std::max(deepest, level[i]);
            if (!m.is_leaf) level[m.left] = level[m.right] = level[i] + 1;
        }
        if (deepest > (uint32_t)MAX_TREE_HEIGHT)
            return fail(MIRT_ERR_UNSUPPORTED, "BVH is %u levels deep; at most %d are supported", deepest, MAX_TREE_HEIGHT);
    }
    if (desc->nodes) sure_boxes(tree, &recs);
    QTree qtree;
    if (desc->nodes) qtree = build_qtree(tree, desc->n_shapes);
    std::vector<ScanBoxes> scan_boxes;
    std::vector<ScanItem> scan_items;
    if (desc->nodes && desc->n_shapes <= 4 * MIRT_SCAN_MAX_SHAPES) build_scan_items(desc, recs, ref_bits.data(), &scan_boxes, &scan_items);
    const std::vector<CNode>& nodes = tree.nodes;
    const uint32_t root_ref = tree.root_ref, height = tree.height;

    // ---- upload
    std::unique_ptr<mirt_scene> owner(new (std::nothrow) mirt_scene());
    mirt_scene* sc = owner.get();
    if (!sc) return fail(MIRT_ERR_OUT_OF_MEMORY, "out of host memory");
    sc->device = dev;
    sc->has_bvh = desc->nodes != nullptr;
    sc->has_noise = has_noise;
    sc->texture_level = texture_level;
    sc->height = height;
    DeviceArena arena;
    auto upload = [&](auto** dptr, const auto* src, size_t count, size_t min_count = 1) -> hipError_t {
        using T = std::remove_const_t<std::remove_pointer_t<decltype(src)>>;
        T* p = nullptr;
        hipError_t e = arena.alloc(&p, std::max(count, min_count));
        if (e != hipSuccess) return e;
        if (std::max(count, min_count) > count) {
            e = hipMemset(p, 0, std::max(count, min_count) * sizeof(T));
            if (e != hipSuccess) return e;
        }
        if (count) e = hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
        *dptr = p;
        return e;
    };
#define SCENE_TRY(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                                    \
        }                                                                                                 \
    } while (0)
    DShape* d_shapes = nullptr;
    ShapeRec* d_recs = nullptr;
    mirt_texture* d_shade_textures = nullptr;
    CNode* d_nodes = nullptr;
    DGate* d_gates = nullptr;
    uint32_t* d_gate_of = nullptr;
    mirt_bvh_member* d_members = nullptr;
    mirt_material* d_materials = nullptr;
    mirt_texture* d_textures = nullptr;
    mirt_perlin* d_perlins = nullptr;
    mirt_image* d_images = nullptr;
    SCENE_TRY(upload(&d_shapes, shapes.data(), shapes.size()));
    SCENE_TRY(upload(&d_recs, recs.data(), recs.size()));
    SCENE_TRY(upload(&d_shade_textures, shade_textures.data(), shade_textures.size()));
    SCENE_TRY(upload(&d_nodes, nodes.data(), nodes.size()));
    unsigned char* d_qnodes = nullptr;
    const std::vector<unsigned char> qbytes = qtree.bytes();
    if (!qbytes.empty()) SCENE_TRY(upload(&d_qnodes, qbytes.data(), qbytes.size()));
    ScanBoxes* d_scan_boxes = nullptr;
    ScanItem* d_scan_items = nullptr;
    SCENE_TRY(upload(&d_scan_boxes, scan_boxes.data(), scan_boxes.size()));
    SCENE_TRY(upload(&d_scan_items, scan_items.data(), scan_items.size()));
    SCENE_TRY(upload(&d_gates, tree.gates.data(), tree.gates.size()));
    SCENE_TRY(upload(&d_gate_of, tree.gate_of.data(), tree.gate_of.size()));
    SCENE_TRY(upload(&d_members, desc->nodes, desc->nodes ? desc->n_nodes : 0));
    SCENE_TRY(upload(&d_materials, desc->materials, desc->n_materials));
    SCENE_TRY(upload(&d_textures, desc->textures, desc->n_textures));
    // the shade kernel stages the first two Perlin tables into LDS: keep two readable
    SCENE_TRY(upload(&d_perlins, desc->perlins, desc->n_perlins, 2));
    std::vector<mirt_image> images(desc->n_images);
    for (uint32_t i = 0; i < desc->n_images; ++i) {
        images[i] = desc->images[i];
        uint8_t* d_rgb = nullptr;
        SCENE_TRY(upload(&d_rgb, desc->images[i].rgb, (size_t)images[i].width * images[i].height * 3));
        images[i].rgb = d_rgb;
    }
    SCENE_TRY(upload(&d_images, images.data(), images.size()));
#undef SCENE_TRY
    sc->d.shapes = d_shapes;
    sc->d.recs = d_recs;
    sc->d.shade_textures = d_shade_textures;
    sc->d.n_shade_textures = (uint32_t)shade_textures.size();
    sc->d.nodes = d_nodes;
    sc->d.scan_boxes = d_scan_boxes;
    sc->d.scan_items = d_scan_items;
    sc->d.n_scan_items = (uint32_t)scan_items.size();
    sc->d.qnodes = reinterpret_cast<const QBox*>(d_qnodes);
    sc->d.qgrid = qtree.grid;
    sc->d.gates = d_gates;
    sc->d.gate_of = d_gate_of;
    sc->d.time_lo = tree.time_lo;
    sc->d.time_hi = tree.time_hi;
    sc->d.cull_reach = tree.cull_reach;
    sc->d.members = d_members;
    sc->d.materials = d_materials;
    sc->d.textures = d_textures;
    sc->d.perlins = d_perlins;
    sc->d.images = d_images;
    sc->d.n_shapes = desc->n_shapes;
    sc->d.n_members = desc->nodes ? desc->n_nodes : 0;
    sc->d.n_nodes = (uint32_t)nodes.size();
    sc->d.root_ref = root_ref;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) sc->cu_count = prop.multiProcessorCount;
    if (sc->cu_count <= 0) sc->cu_count = 256;
    sc->allocations = arena.ptrs;
    arena.release();
    *out = owner.release();
    return MIRT_OK;
}

int mirt_scene_create(const mirt_scene_desc* desc, int device, mirt_scene** out) {
    return no_throw([&] { return scene_create_body(desc, device, out); });
}

void mirt_scene_destroy(mirt_scene* scene) {
    if (!scene) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != scene->device) (void)hipSetDevice(scene->device);
    for (void* p : scene->allocations) (void)hipFree(p);
    if (scene->ws_ptr) (void)hipFree(scene->ws_ptr);
    if (scene->frame_ptr) (void)hipFree(scene->frame_ptr);
    if (prev >= 0 && prev != scene->device) (void)hipSetDevice(prev);
    delete scene;
}

int mirt_scene_trim(mirt_scene* scene) {
    if (!scene) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene is null");
    void *drop = nullptr, *drop_frame = nullptr;
    {
        std::lock_guard<std::mutex> lock(scene->ws_mutex);
        drop = scene->ws_ptr;
        scene->ws_ptr = nullptr;
        scene->ws_bytes = 0;
        drop_frame = scene->frame_ptr;
        scene->frame_ptr = nullptr;
        scene->frame_bytes = 0;
    }
    if (!drop && !drop_frame) return MIRT_OK;
    DeviceGuard guard;
    int rc = guard.enter(scene->device);
    if (rc != MIRT_OK) return rc;
    if (drop) HIP_TRY(hipFree(drop));
    if (drop_frame) HIP_TRY(hipFree(drop_frame));
    return MIRT_OK;
}

// -------------------------------------------------------------- intersect
static int pick_mode(const mirt_scene* scene, int use_bvh, int traversal, int* mode) {
    if (!use_bvh) {
        *mode = MODE_LIST;
        return MIRT_OK;
    }
    if (!scene->has_bvh) return fail(MIRT_ERR_INVALID_ARGUMENT, "use_bvh = 1 but the scene was created without BVH members");
    if (traversal == MIRT_TRAVERSE_REFERENCE) *mode = MODE_REFERENCE;
    else if (traversal == MIRT_TRAVERSE_FAST) *mode = MODE_FAST;
    else return fail(MIRT_ERR_INVALID_ARGUMENT, "unknown traversal mode %d", traversal);
    return MIRT_OK;
}

static int grid_for(const mirt_scene* scene, size_t work_items, int blocks_per_cu) {
    const size_t want = (work_items + BLOCK - 1) / BLOCK;
    const size_t cap = (size_t)scene->cu_count * (size_t)blocks_per_cu;
    return (int)std::max<size_t>(1, std::min(want, cap));
}

}  // extern "C"

namespace {

// How the fast traversal kernel is launched for one scene: stack entry width, what is
// staged in LDS, workgroup size, dynamic LDS bytes.
struct TraversePlan {
    bool stack16 = true;
    bool nodes_in_lds = false;
    bool shapes_in_lds = false;
    int block = TRAV_BLOCK_GLOBAL;
    size_t lds_bytes = 0;
    uint32_t levels = 1;
};

// One 1024-thread workgroup per CU may use this much of the CU's 160 KB of LDS.
constexpr size_t TRAV_LDS_BUDGET = 150 * 1024;

TraversePlan plan_traverse(const mirt_scene* scene) {
    TraversePlan p;
    p.levels = std::max<uint32_t>(scene->height, 1);
    p.stack16 = scene->d.n_nodes < 65536u;
    const size_t entry = p.stack16 ? 2 : 4;
    const size_t nodes = trav_nodes_bytes(scene->d.n_nodes);
    const size_t shapes = (size_t)scene->d.n_shapes * SHAPE_LDS_BYTES;
    const size_t stack_lds = trav_stack_bytes(p.levels, entry, TRAV_BLOCK_LDS);
    if (scene->d.n_nodes > 0 && stack_lds + nodes <= TRAV_LDS_BUDGET) {
        p.nodes_in_lds = true;
        p.block = TRAV_BLOCK_LDS;
        p.lds_bytes = stack_lds + nodes;
        if (p.lds_bytes + shapes <= TRAV_LDS_BUDGET) {
            p.shapes_in_lds = true;
            p.lds_bytes += shapes;
        }
    } else {
        p.block = TRAV_BLOCK_GLOBAL;
        p.lds_bytes = trav_stack_bytes(p.levels, entry, TRAV_BLOCK_GLOBAL);
    }
    return p;
}

template <int TBLOCK, class StackT, bool NLDS, bool SLDS, bool STATS, class Io>
hipError_t launch_traverse_as(const mirt_scene* scene, const TraversePlan& plan, const Io& io, uint32_t* work_counter,
                              double t_min, double t_max, hipStream_t stream, unsigned long long* stats = nullptr) {
    auto kernel = k_traverse<TBLOCK, StackT, NLDS, SLDS, STATS, Io>;
    // grid = every workgroup the device can hold at once (persistent waves pull work themselves)
    static thread_local size_t sized_bytes = ~(size_t)0;
    static thread_local int per_cu = 1;
    if (sized_bytes != plan.lds_bytes) {
        if (plan.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
            if (e != hipSuccess) return e;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, TBLOCK, plan.lds_bytes) != hipSuccess || n <= 0) n = 1;
        per_cu = std::min(n, 4);
        sized_bytes = plan.lds_bytes;
    }
    const int grid = scene->cu_count * per_cu;
    // development aid: MIRT_TRAV_TUNING overrides the scheduling parameters of the kernel
    static const uint32_t tuning = [] {
        const char* s = std::getenv("MIRT_TRAV_TUNING");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : TUNING_DEFAULT;
    }();
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(TBLOCK), plan.lds_bytes, stream, scene->d, io, work_counter, t_min, t_max, plan.levels, tuning, stats);
    return hipGetLastError();
}

template <class StackT, bool STATS, class Io>
hipError_t launch_traverse_stack(const mirt_scene* scene, const TraversePlan& plan, const Io& io, uint32_t* work_counter,
                                 double t_min, double t_max, hipStream_t stream, unsigned long long* stats) {
    if (plan.nodes_in_lds && plan.shapes_in_lds)
        return launch_traverse_as<TRAV_BLOCK_LDS, StackT, true, true, STATS>(scene, plan, io, work_counter, t_min, t_max, stream, stats);
    if (plan.nodes_in_lds)
        return launch_traverse_as<TRAV_BLOCK_LDS, StackT, true, false, STATS>(scene, plan, io, work_counter, t_min, t_max, stream, stats);
    return launch_traverse_as<TRAV_BLOCK_GLOBAL, StackT, false, false, STATS>(scene, plan, io, work_counter, t_min, t_max, stream, stats);
}

template <class Io>
hipError_t launch_traverse(const mirt_scene* scene, const Io& io, uint32_t* work_counter, double t_min, double t_max,
                           hipStream_t stream) {
    const TraversePlan plan = plan_traverse(scene);
    if (plan.stack16) return launch_traverse_stack<uint16_t, false>(scene, plan, io, work_counter, t_min, t_max, stream, nullptr);
    return launch_traverse_stack<uint32_t, false>(scene, plan, io, work_counter, t_min, t_max, stream, nullptr);
}

// Development aid (MIRT_TRAV_STATS=1 in the environment): the same traversal with lane
// utilisation counters, printed to stderr.  16-bit stacks only.
hipError_t launch_traverse_with_stats(const mirt_scene* scene, const IntersectIo& io, uint32_t* work_counter, double t_min,
                                      double t_max) {
    const TraversePlan plan = plan_traverse(scene);
    if (!plan.stack16) return hipErrorNotSupported;
    unsigned long long* d_stats = nullptr;
    hipError_t e = hipMalloc((void**)&d_stats, STAT_COUNT * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    (void)hipMemset(d_stats, 0, STAT_COUNT * sizeof(unsigned long long));
    e = launch_traverse_stack<uint16_t, true>(scene, plan, io, work_counter, t_min, t_max, 0, d_stats);
    unsigned long long h[STAT_COUNT] = {0};
    if (e == hipSuccess) e = hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d_stats);
    if (e == hipSuccess)
        std::fprintf(stderr,
                     "[mirt traverse stats] rays=%u nodes=%u lds_nodes=%d lds_shapes=%d block=%d levels=%u lds_bytes=%zu | outer=%llu | node iters=%llu lanes/iter=%.1f | "
                     "leaf iters=%llu lanes/iter=%.1f | refill iters=%llu lanes/iter=%.1f | shape tests/ray=%.2f node visits/ray=%.2f\n",
                     io.n, scene->d.n_nodes, (int)plan.nodes_in_lds, (int)plan.shapes_in_lds, plan.block, plan.levels, plan.lds_bytes, h[STAT_OUTER_ITERS], h[STAT_NODE_ITERS],
                     h[STAT_NODE_ITERS] ? (double)h[STAT_NODE_LANES] / h[STAT_NODE_ITERS] : 0.0, h[STAT_LEAF_ITERS],
                     h[STAT_LEAF_ITERS] ? (double)h[STAT_LEAF_LANES] / h[STAT_LEAF_ITERS] : 0.0, h[STAT_REFILL_ITERS],
                     h[STAT_REFILL_ITERS] ? (double)h[STAT_REFILL_LANES] / h[STAT_REFILL_ITERS] : 0.0,
                     (double)h[STAT_SHAPE_TESTS] / io.n, (double)h[STAT_NODE_LANES] / io.n);
    if (e == hipSuccess && h[STAT_CYC_TOTAL])
        std::fprintf(stderr,
                     "[mirt traverse stats] wave clocks: refill %.1f%% (%.0f per refill) | node loop %.1f%% (%.0f per visit) | leaf %.1f%% "
                     "(%.0f per phase) | store %.1f%% | other %.1f%%\n",
                     100.0 * h[STAT_CYC_REFILL] / h[STAT_CYC_TOTAL], (double)h[STAT_CYC_REFILL] / std::max(1ull, h[STAT_REFILL_ITERS]),
                     100.0 * h[STAT_CYC_NODE] / h[STAT_CYC_TOTAL], (double)h[STAT_CYC_NODE] / std::max(1ull, h[STAT_NODE_ITERS]),
                     100.0 * h[STAT_CYC_LEAF] / h[STAT_CYC_TOTAL], (double)h[STAT_CYC_LEAF] / std::max(1ull, h[STAT_LEAF_ITERS]),
                     100.0 * h[STAT_CYC_STORE] / h[STAT_CYC_TOTAL],
                     100.0 * (1.0 - (double)(h[STAT_CYC_REFILL] + h[STAT_CYC_NODE] + h[STAT_CYC_LEAF] + h[STAT_CYC_STORE]) / h[STAT_CYC_TOTAL]));
    return e;
}

}  // namespace

extern "C" {

static int mirt_intersect_device_body(const mirt_scene* scene, const mirt_ray* d_rays, size_t n, double t_min, double t_max,
                          int use_bvh, int traversal, mirt_hit* d_hits) {
    if (!scene) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene is null");
    if (n == 0) return MIRT_OK;
    if (!d_rays || !d_hits) return fail(MIRT_ERR_INVALID_ARGUMENT, "rays or hits is null");
    int mode = 0, rc = pick_mode(scene, use_bvh, traversal, &mode);
    if (rc != MIRT_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(scene->device)) != MIRT_OK) return rc;
    const int grid = grid_for(scene, n, 8);
    DeviceArena arena;
    switch (mode) {
        case MODE_LIST: hipLaunchKernelGGL(k_intersect<MODE_LIST>, dim3(grid), dim3(BLOCK), 0, 0, scene->d, d_rays, n, t_min, t_max, d_hits); break;
        case MODE_REFERENCE: hipLaunchKernelGGL(k_intersect<MODE_REFERENCE>, dim3(grid), dim3(BLOCK), 0, 0, scene->d, d_rays, n, t_min, t_max, d_hits); break;
        default: {
            if (n > 0xFFFFFFFFull - CHUNK_RAYS * 4096ull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^32 rays in one call");
            uint32_t* d_work = nullptr;
            HIP_TRY(arena.alloc(&d_work, (size_t)1));
            HIP_TRY(hipMemsetAsync(d_work, 0, sizeof(uint32_t), 0));
            IntersectIo io{d_rays, d_hits, (uint32_t)n};
            const char* want_stats = std::getenv("MIRT_TRAV_STATS");
            if (want_stats && want_stats[0] == '1') HIP_TRY(launch_traverse_with_stats(scene, io, d_work, t_min, t_max));
            else HIP_TRY(launch_traverse(scene, io, d_work, t_min, t_max, 0));
            break;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return MIRT_OK;
}

static int mirt_intersect_body(const mirt_scene* scene, const mirt_ray* rays, size_t n, double t_min, double t_max, int use_bvh,
                   int traversal, mirt_hit* hits) {
    if (!scene) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene is null");
    if (n == 0) return MIRT_OK;
    if (!rays || !hits) return fail(MIRT_ERR_INVALID_ARGUMENT, "rays or hits is null");
    int mode = 0, rc = pick_mode(scene, use_bvh, traversal, &mode);
    if (rc != MIRT_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(scene->device)) != MIRT_OK) return rc;
    DeviceArena arena;
    mirt_ray* d_rays = nullptr;
    mirt_hit* d_hits = nullptr;
    HIP_TRY(arena.alloc(&d_rays, n));
    HIP_TRY(arena.alloc(&d_hits, n));
    HIP_TRY(hipMemcpy(d_rays, rays, n * sizeof(mirt_ray), hipMemcpyHostToDevice));
    rc = mirt_intersect_device_body(scene, d_rays, n, t_min, t_max, use_bvh, traversal, d_hits);
    if (rc != MIRT_OK) return rc;
    HIP_TRY(hipMemcpy(hits, d_hits, n * sizeof(mirt_hit), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

static int mirt_camera_rays_body(const mirt_camera* camera, uint32_t width, uint32_t height, uint32_t pixel_begin,
                     uint32_t pixel_end, uint32_t sample, uint64_t seed, int device, mirt_ray* rays) {
    if (!camera || !rays) return fail(MIRT_ERR_INVALID_ARGUMENT, "camera or rays is null");
    if (width == 0 || height == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    if (pixel_end == 0) pixel_end = width * height;
    if (pixel_begin >= pixel_end || pixel_end > width * height) return fail(MIRT_ERR_INVALID_ARGUMENT, "bad pixel range");
    if (mirt_device_count() == 0) return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != MIRT_OK) return rc;
    const uint32_t n = pixel_end - pixel_begin;
    DeviceArena arena;
    mirt_ray* d_rays = nullptr;
    HIP_TRY(arena.alloc(&d_rays, (size_t)n));
    const int grid = (int)std::min<size_t>(((size_t)n + BLOCK - 1) / BLOCK, 4096);
    hipLaunchKernelGGL(k_camera_rays, dim3(grid), dim3(BLOCK), 0, 0, *camera, width, height, pixel_begin, n, sample, seed, d_rays);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rays, d_rays, (size_t)n * sizeof(mirt_ray), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ render
#ifndef MIRT_NODES_IN_LDS
#define MIRT_NODES_IN_LDS 1
#endif


namespace {

struct EventTimer {
    // Pairs of HIP events around kernel launches of one pass, summed per class after the pass.
    enum Class { RAYGEN = 0, EXTEND, BIN, SHADE, ACCUMULATE, PATHS, N_CLASSES };
    bool enabled = false;
    std::vector<hipEvent_t> pool;
    std::vector<int> cls_of_pair;
    size_t used = 0;
    double ms[N_CLASSES] = {0, 0, 0, 0, 0, 0};

    hipError_t begin(hipStream_t s, int cls) {
        if (!enabled) return hipSuccess;
        while (pool.size() < used + 2) {
            hipEvent_t e;
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) return err;
            pool.push_back(e);
        }
        cls_of_pair.push_back(cls);
        return hipEventRecord(pool[used], s);
    }
    hipError_t end(hipStream_t s) {
        if (!enabled) return hipSuccess;
        hipError_t err = hipEventRecord(pool[used + 1], s);
        used += 2;
        return err;
    }
    hipError_t collect() {  // call after the stream has been synchronised
        if (!enabled) return hipSuccess;
        for (size_t p = 0; p < used / 2; ++p) {
            float t = 0.f;
            hipError_t err = hipEventElapsedTime(&t, pool[2 * p], pool[2 * p + 1]);
            if (err != hipSuccess) return err;
            ms[cls_of_pair[p]] += t;
        }
        used = 0;
        cls_of_pair.clear();
        return hipSuccess;
    }
    ~EventTimer() {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    }
};

template <class K>
int blocks_per_cu(K kernel, int fallback) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, BLOCK, 0) != hipSuccess || n <= 0) return fallback;
    return std::min(n, 8);
}

// The one workgroup per CU of the path kernel may use this much of the CU's 160 KB of LDS, static arrays included.
constexpr size_t PATHS_LDS_BUDGET = 158 * 1024;

// The path kernel's launch shape for one scene.
struct PathsPlan {
    bool stack16 = true;
    bool tree_in_lds = false;
    bool nodes_in_lds = false;  // the nodes alone (records from global memory)
    bool qnodes_in_lds = false; // ... as QNodes
    bool scan = false;          // no tree: every shape tested for every ray
    int block = 256;
    size_t lds_bytes = 0;
    uint32_t levels = 1;
    uint32_t n_top = 0;  // nodes from the top of the tree kept in LDS
    int per_cu = 1;
    const void* kernel = nullptr;
};

template <class StackT, int TREE>
const void* paths_kernel_for(int textures) {
    switch (textures) {
        case TEXTURES_CONSTANT: return (const void*)k_paths<StackT, TEXTURES_CONSTANT, TREE>;
        case TEXTURES_CHECKER: return (const void*)k_paths<StackT, TEXTURES_CHECKER, TREE>;
        case TEXTURES_IMAGE: return (const void*)k_paths<StackT, TEXTURES_IMAGE, TREE>;
        default: return (const void*)k_paths<StackT, TEXTURES_ALL, TREE>;
    }
}

hipError_t plan_paths(const mirt_scene* scene, PathsPlan* out) {
    PathsPlan p;
    const int textures = scene->texture_level;
    p.levels = std::max<uint32_t>(scene->height, 1);
    p.stack16 = scene->d.n_nodes < 65536u;
    p.block = paths_block(textures);
    const size_t nodes = trav_nodes_bytes(scene->d.n_nodes), recs = (size_t)scene->d.n_shapes * SHAPE_LDS_BYTES;
    const size_t fixed = (textures >= TEXTURES_ALL ? sizeof(mirt_perlin) * PERLIN_STAGED : 0) +
                         (textures > TEXTURES_CONSTANT ? sizeof(mirt_texture) * TEXTURES_STAGED : 0) +
                         sizeof(unsigned long long) * IN_SPHERE_SCRATCH * (p.block / 64) +
                         (textures >= TEXTURES_ALL ? sizeof(double) * 64 * (p.block / 64) : 0);
    const size_t stack = trav_stack_bytes(p.levels, p.stack16 ? 2 : 4, p.block);
    // ONE workgroup per CU, as many waves as the registers allow, sharing one copy of the tree: all
    // of it and the shapes' records if they fit, else as many nodes from the top as do
    // (REF16_MAX_SHAPES: the kernel with the tree in LDS parks shapes as 16-bit references)
    p.tree_in_lds = scene->d.n_nodes > 0 && p.stack16 && scene->d.n_shapes <= REF16_MAX_SHAPES && stack + nodes + recs + fixed <= PATHS_LDS_BUDGET;
    // development aid: MIRT_SCAN_MAX overrides the largest scene rendered without a tree
    static const uint32_t scan_max = [] {
        const char* s = std::getenv("MIRT_SCAN_MAX");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : (uint32_t)MIRT_SCAN_MAX_SHAPES;
    }();
    if (scene->d.n_scan_items && scene->d.n_shapes <= scan_max) {
        p.scan = true;
        p.tree_in_lds = false;
        p.levels = 0;
        p.n_top = (scene->d.n_scan_items + 1) / 2;
        p.lds_bytes = scan_boxes_bytes(p.n_top) + ((recs + 15) & ~(size_t)15) + (size_t)scene->d.n_scan_items * sizeof(ScanItem);
        p.kernel = paths_kernel_for<uint16_t, TREE_SCAN>(textures);
    } else if (p.tree_in_lds) {
        p.n_top = scene->d.n_nodes;
        p.lds_bytes = stack + nodes + recs;
        p.kernel = paths_kernel_for<uint16_t, TREE_LDS>(textures);
    } else {
        const size_t room = PATHS_LDS_BUDGET > stack + fixed ? PATHS_LDS_BUDGET - stack - fixed : 0;
        // development aid: MIRT_QNODES=0 keeps QNodes out, =2 prefers them to 56-byte nodes that would fit too
        static const int qnodes_mode = [] {
            const char* s = std::getenv("MIRT_QNODES");
            return s ? std::atoi(s) : 1;
        }();
        const bool qnodes_fit = qnodes_mode && p.stack16 && scene->d.qnodes && qnodes_bytes(scene->d.n_nodes) <= room;
        if (qnodes_fit && (qnodes_mode == 2 || nodes > room)) {
            p.qnodes_in_lds = true;
            p.n_top = scene->d.n_nodes;
            p.lds_bytes = stack + qnodes_bytes(scene->d.n_nodes);
            p.kernel = paths_kernel_for<uint16_t, TREE_QNODES_LDS>(textures);
        } else if (MIRT_NODES_IN_LDS && p.stack16 && scene->d.n_nodes > 0 && nodes <= room) {
            // the nodes fit on their own: every node visit from LDS, the records from global memory
            p.nodes_in_lds = true;
            p.n_top = scene->d.n_nodes;
            p.lds_bytes = stack + nodes;
            p.kernel = paths_kernel_for<uint16_t, TREE_NODES_LDS>(textures);
        } else {
            p.n_top = MIRT_TOP_IN_LDS ? (uint32_t)std::min<size_t>(scene->d.n_nodes, room / NODE_LDS_BYTES) : 0u;
            p.lds_bytes = stack + trav_nodes_bytes(p.n_top);
            p.kernel = p.stack16 ? paths_kernel_for<uint16_t, TREE_GLOBAL>(textures) : paths_kernel_for<uint32_t, TREE_GLOBAL>(textures);
        }
    }
    if (p.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(p.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, p.kernel, p.block, p.lds_bytes) != hipSuccess || n <= 0) n = 1;
    p.per_cu = std::min(n, 8);
    // development aid: MIRT_PLAN_VERBOSE=1 says on stderr how the path kernel is launched
    static const bool verbose = [] {
        const char* s = std::getenv("MIRT_PLAN_VERBOSE");
        return s && s[0] == '1';
    }();
    if (verbose)
        std::fprintf(stderr, "[mirt plan] shapes=%u nodes=%u levels=%u block=%d lds=%zu tree=%s per_cu=%d\n", scene->d.n_shapes, scene->d.n_nodes, p.levels,
                     p.block, p.lds_bytes, p.scan ? "scan" : p.tree_in_lds ? "lds" : p.qnodes_in_lds ? "qnodes" : p.nodes_in_lds ? "nodes-lds" : "global", p.per_cu);
    *out = p;
    return hipSuccess;
}

// expected: paths that will probably enter (the kernel copes with any number).
hipError_t launch_paths(const mirt_scene* scene, const PathsPlan& plan, const Paths& P, const RenderParams& rp, uint32_t bounce,
                        uint32_t expected, hipStream_t stream) {
    // few paths: spread them over as many waves as there are, a path's bounces are sequential
    const size_t want = ((size_t)expected + plan.block - 1) / plan.block;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(want, (size_t)scene->cu_count * plan.per_cu));
    // development aid: MIRT_PATHS_TUNING overrides the scheduling parameters of the kernel
    static const uint32_t tuning_override = [] {
        const char* s = std::getenv("MIRT_PATHS_TUNING");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : 0u;
    }();
    const uint32_t tuning_default = tuning_override ? tuning_override
                                    : plan.scan     ? PATHS_TUNING_SCAN
                                    : plan.tree_in_lds ? PATHS_TUNING_DEFAULT
                                    : plan.qnodes_in_lds ? PATHS_TUNING_QNODES
                                                       : PATHS_TUNING_BIG_TREE;
    DScene sc = scene->d;
    Paths paths = P;
    RenderParams params = rp;
    // development switch: MIRT_PATHS_REGROUP = number of walking waves selects the kernel that regroups
    // paths across the waves of a workgroup (paths_regroup.h), for scenes whose tree is in LDS.  Same
    // results; measured slower than k_paths so far (RTIOW: 3000 against 5000 Msamples/s with 10 walking
    // waves; 3750 with 384 more paths in flight than walking lanes, MIRT_REGROUP_EXTRA, which is beyond
    // what the kernel's argument against deadlock covers: more than 64 is at the developer's risk).
    static const uint32_t regroup_walking = [] {
        const char* s = std::getenv("MIRT_PATHS_REGROUP");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : 0u;
    }();
    static const uint32_t regroup_tuning = [] {
        const char* s = std::getenv("MIRT_REGROUP_TUNING");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : (1u | (16u << 8) | (16u << 16));
    }();
    static const uint32_t regroup_extra = [] {
        const char* s = std::getenv("MIRT_REGROUP_EXTRA");
        return s ? (uint32_t)std::strtoul(s, nullptr, 0) : 64u;
    }();
    if (regroup_walking && plan.tree_in_lds && bounce == 0) {
        const int textures = scene->texture_level;
        const void* kernel = textures == TEXTURES_CONSTANT ? (const void*)k_paths_regroup<TEXTURES_CONSTANT>
                             : textures == TEXTURES_CHECKER ? (const void*)k_paths_regroup<TEXTURES_CHECKER>
                             : textures == TEXTURES_IMAGE   ? (const void*)k_paths_regroup<TEXTURES_IMAGE>
                                                            : (const void*)k_paths_regroup<TEXTURES_ALL>;
        const size_t lds_bytes = ((plan.lds_bytes + 15) & ~(size_t)15) + regroup_ring_bytes();
        const uint32_t waves = (uint32_t)plan.block / 64u;
        uint32_t walking = std::min(regroup_walking, waves - 1u), n_total = walking * 64u + regroup_extra, levels = plan.levels, tuning = regroup_tuning;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        void* args[] = {&sc, &paths, &params, &levels, &walking, &n_total, &tuning};
        return hipLaunchKernel(kernel, dim3(grid), dim3(plan.block), args, lds_bytes, stream);
    }
    uint32_t levels = plan.levels, tuning = tuning_default, n_top = plan.n_top;
    void* args[] = {&sc, &paths, &params, &bounce, &levels, &tuning, &n_top};
    return hipLaunchKernel(plan.kernel, dim3(grid), dim3(plan.block), args, plan.lds_bytes, stream);
}

int render_accumulate_impl(const mirt_scene* scene, const mirt_camera* camera, const double background[3],
                           uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed,
                           const mirt_render_opts* opts, double* d_accum, mirt_stats* stats, bool static_scratch = false) {
    if (!scene || !camera || !background || !d_accum) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    if (width == 0 || height == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    if (spp == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "spp must be > 0");
    if (max_depth == 0 || max_depth > 255) return fail(MIRT_ERR_INVALID_ARGUMENT, "max_depth must be in 1..255 (the reference's is a u8)");
    mirt_render_opts o;
    std::memset(&o, 0, sizeof o);
    o.device = -1;
    o.use_bvh = 1;
    o.traversal = MIRT_TRAVERSE_FAST;
    if (opts) {
        if (opts->struct_size != sizeof(mirt_render_opts))
            return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_render_opts.struct_size is %u, expected %zu", opts->struct_size, sizeof(mirt_render_opts));
        o = *opts;
    }
    const uint32_t n_pix_total = width * height;
    if (o.pixel_end == 0) o.pixel_end = n_pix_total;
    if (o.sample_end == 0) o.sample_end = spp;
    if (o.pixel_begin >= o.pixel_end || o.pixel_end > n_pix_total) return fail(MIRT_ERR_INVALID_ARGUMENT, "bad pixel range");
    if (o.sample_begin >= o.sample_end || o.sample_end > spp) return fail(MIRT_ERR_INVALID_ARGUMENT, "bad sample range");
    if (o.device >= 0 && o.device != scene->device)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "opts.device = %d but the scene lives on device %d", o.device, scene->device);
    int mode = 0, rc = pick_mode(scene, o.use_bvh, o.traversal, &mode);
    if (rc != MIRT_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(scene->device)) != MIRT_OK) return rc;

    const auto t_start = std::chrono::steady_clock::now();
    if (o.tile_count > 1 && (o.tile_size == 0 || o.tile_index >= o.tile_count))
        return fail(MIRT_ERR_INVALID_ARGUMENT, "bad tile split: size %u, count %u, index %u", o.tile_size, o.tile_count, o.tile_index);
    PixelSet px;
    px.pixel_begin = o.pixel_begin;
    px.tile_size = o.tile_size;
    px.tile_count = o.tile_count;
    px.tile_index = o.tile_index;
    px.n_pixels = PixelSet::count(o.pixel_end - o.pixel_begin, o.tile_size, o.tile_count, o.tile_index);
    const uint32_t n_pixels = px.n_pixels;
    if (n_pixels == 0) {  // this tile index owns nothing of the range
        if (stats) std::memset(stats, 0, sizeof *stats);
        return MIRT_OK;
    }
    const uint32_t n_samples_total = o.sample_end - o.sample_begin;

    // ---- where the launches per bounce hand over to the path kernel (mirt_render_opts.handover_bounce)
    uint32_t handover = max_depth;  // max_depth = never
    bool probing = false;           // MIRT_HANDOVER_PROBE: read the live counts back during the first pass
    if (mode == MODE_FAST) {
        if (o.handover_bounce >= 0) handover = std::min<uint32_t>((uint32_t)o.handover_bounce, max_depth);
        else if (o.handover_bounce == MIRT_HANDOVER_PROBE) probing = true;
        else if (o.handover_bounce != MIRT_HANDOVER_NEVER)
            return fail(MIRT_ERR_INVALID_ARGUMENT, "bad handover_bounce %d", o.handover_bounce);
    }
    const bool from_start = handover == 0;  // no queues of live paths, no per-path state but the radiance
    if (static_scratch && (!from_start || o.profile))
        return fail(MIRT_ERR_UNSUPPORTED, "mirt_render_accumulate_static serves use_bvh = 1, MIRT_TRAVERSE_FAST, handover_bounce = 0, profile = 0 only");
    if (static_scratch && n_pixels > STATIC_PASS_PATHS)
        return fail(MIRT_ERR_UNSUPPORTED, "%u pixels selected, at most %u fit a pass of mirt_render_accumulate_static: split the frame into pixel ranges",
                    n_pixels, STATIC_PASS_PATHS);

    // Default pass: 2^29 paths when the path kernel runs from the camera ray (13 GB of radiance
    // records: each launch ends with a tail of a few deep paths, ~0.8 ms on RTIOW, so few large
    // launches beat many small ones: +7.5 % over 2^26); 2^26 for the per-bounce pipeline (173 B of path state
    // per path + the record: against 2^23 an eighth of the launches, most of which find almost empty queues:
    // 1 369 instead of 787 Msamples/s on RTIOW); either way at most a quarter of the device memory that is free.
    uint64_t target = o.paths_per_pass ? o.paths_per_pass : (from_start ? (1ull << 29) : (1ull << 26));
    const size_t bytes_per_path = from_start ? RECORD_BYTES : 176 + RECORD_BYTES;
    if (static_scratch) target = std::min<uint64_t>(target, STATIC_PASS_PATHS);
    if (!o.paths_per_pass && !static_scratch) {
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
            {
                std::lock_guard<std::mutex> lock(scene->ws_mutex);
                free_bytes += scene->ws_bytes;  // what the scene already holds for this purpose
            }
            while (target > (1ull << 22) && target * bytes_per_path > free_bytes / 4) target >>= 1;
        }
    }
    uint32_t per_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_samples_total, target / n_pixels));
    const uint64_t capacity64 = (uint64_t)n_pixels * per_pass;
    if (capacity64 > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "pass of %llu paths is too large", (unsigned long long)capacity64);
    const uint32_t capacity = (uint32_t)capacity64;

    // ---- per-pass state: radiance records, counters and totals in one block (kept by the
    // scene between calls), the queue pipeline's path state in arrays of this call
    const Counters C{max_depth};
    constexpr size_t N_SEGMENT_WORDS = 2 + 16 + 8 + 6 + 6;  // [2..] development statistics of k_paths (MIRT_PATHS_STATS builds)
    auto round256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    static_assert((256 + 255 * (SHADE_NUM_CLASSES + 1)) * sizeof(uint32_t) + 256 + N_SEGMENT_WORDS * sizeof(unsigned long long) + 256 <= STATIC_TAIL_BYTES,
                  "counters at the largest depth and the totals fit behind the static array's records");
    const size_t radiance_bytes = round256((size_t)capacity * RECORD_BYTES);
    const size_t counters_bytes = round256((size_t)C.total() * sizeof(uint32_t));
    const size_t segments_bytes = round256(N_SEGMENT_WORDS * sizeof(unsigned long long));
    Workspace ws;  // declared before the stream work below ends: released after the final synchronise
    std::unique_lock<std::mutex> scratch_lock;  // held to the end of the call when the shared array is used
    void* block = nullptr;
    if (static_scratch) {
        if (radiance_bytes + counters_bytes + segments_bytes > STATIC_SCRATCH_BYTES)
            return fail(MIRT_ERR_UNSUPPORTED, "a pass of %u paths does not fit the static array", capacity);
        if ((rc = scratch_address(scene->device, &block)) != MIRT_OK) return rc;
        scratch_lock = std::unique_lock<std::mutex>(g_scratch[scene->device].in_use);
    } else {
        HIP_TRY(ws.take(scene, radiance_bytes + counters_bytes + segments_bytes));
        block = ws.ptr;
    }
    DeviceArena arena;
    Paths P;
    std::memset(&P, 0, sizeof P);
    P.capacity = capacity;
    P.radiance = reinterpret_cast<double*>(block);
    P.counters = reinterpret_cast<uint32_t*>(static_cast<char*>(block) + radiance_bytes);
    P.segments = reinterpret_cast<unsigned long long*>(static_cast<char*>(block) + radiance_bytes + counters_bytes);
    if (!from_start) {
        double** f64_arrays[] = {&P.ox, &P.oy, &P.oz, &P.dx, &P.dy, &P.dz, &P.time, &P.bx, &P.by, &P.bz, &P.hit_t};
        for (double** a : f64_arrays) HIP_TRY(arena.alloc(a, (size_t)capacity));
        HIP_TRY(arena.alloc(&P.hit_ref, (size_t)capacity));
        HIP_TRY(arena.alloc(&P.stream, (size_t)capacity));
        HIP_TRY(arena.alloc(&P.live[0], (size_t)capacity));
        HIP_TRY(arena.alloc(&P.live[1], (size_t)capacity));
        HIP_TRY(arena.alloc(&P.cls_q, (size_t)capacity));
        HIP_TRY(arena.alloc(&P.cls_queue, (size_t)capacity * SHADE_NUM_CLASSES));
    }
    hipStream_t stream = 0;
    HIP_TRY(hipMemsetAsync(P.segments, 0, N_SEGMENT_WORDS * sizeof(unsigned long long), stream));

    RenderParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.cam = *camera;
    for (int c = 0; c < 3; ++c) rp.background[c] = background[c];
    rp.width = width;
    rp.height = height;
    rp.max_depth = max_depth;
    rp.px = px;
    rp.n_pixels = n_pixels;
    rp.seed = seed;
    rp.t_min = 0.001;
    rp.t_max = std::numeric_limits<double>::infinity();
    rp.use_bvh = o.use_bvh;
    rp.traversal = o.traversal;

    int occ_extend = 4;
    switch (mode) {
        case MODE_LIST: occ_extend = blocks_per_cu(k_extend<MODE_LIST>, 4); break;
        case MODE_REFERENCE: occ_extend = blocks_per_cu(k_extend<MODE_REFERENCE>, 2); break;
        default: break;  // the fast kernel sizes its own persistent grid (launch_traverse)
    }
    const int occ_shade = scene->has_noise ? blocks_per_cu(k_shade<true>, 2) : blocks_per_cu(k_shade<false>, 4);

    PathsPlan paths_plan;
    if (mode == MODE_FAST) HIP_TRY(plan_paths(scene, &paths_plan));
    uint32_t handover_paths = o.handover_paths ? o.handover_paths : std::max<uint32_t>(capacity / 2, 1);
    uint32_t paths_estimate = capacity;  // paths expected to enter the path kernel (sizes its grid)

    EventTimer timer;
    timer.enabled = o.profile != 0;
    uint64_t passes = 0, launches_extend = 0, launches_shade = 0, launches_paths = 0;

    for (uint32_t s0 = o.sample_begin; s0 < o.sample_end; s0 += per_pass) {
        rp.sample_base = s0;
        rp.n_samples = std::min(per_pass, o.sample_end - s0);
        rp.n_paths = n_pixels * rp.n_samples;
        HIP_TRY(hipMemsetAsync(P.counters, 0, (size_t)C.total() * sizeof(uint32_t), stream));
        rp.handover = max_depth;

        if (!from_start) {
            HIP_TRY(timer.begin(stream, EventTimer::RAYGEN));
            hipLaunchKernelGGL(k_raygen, dim3(grid_for(scene, rp.n_paths, 8)), dim3(BLOCK), 0, stream, P, rp);
            HIP_TRY(timer.end(stream));
        }

        const int grid_extend = grid_for(scene, rp.n_paths, occ_extend);
        const int grid_shade = grid_for(scene, rp.n_paths, occ_shade);
        const int grid_bin = (int)std::max<size_t>(1, std::min<size_t>(((size_t)rp.n_paths + BIN_TILE - 1) / BIN_TILE, (size_t)scene->cu_count * 8));
        for (uint32_t bounce = 0; bounce < max_depth; ++bounce) {
            if (probing && bounce > 0) {
                // one small read-back per bounce, first pass only: how many paths are left?
                uint32_t live = 0;
                HIP_TRY(hipMemcpyAsync(&live, P.counters + C.live(bounce), sizeof live, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (live <= handover_paths) {
                    handover = bounce;
                    paths_estimate = std::max<uint32_t>(live, 1);
                    probing = false;
                }
            }
            if (bounce >= handover) {
                rp.handover = bounce;
                HIP_TRY(timer.begin(stream, EventTimer::PATHS));
                HIP_TRY(launch_paths(scene, paths_plan, P, rp, bounce, std::min(paths_estimate, rp.n_paths), stream));
                HIP_TRY(timer.end(stream));
                ++launches_paths;
                break;
            }
            HIP_TRY(timer.begin(stream, EventTimer::EXTEND));
            switch (mode) {
                case MODE_LIST: hipLaunchKernelGGL(k_extend<MODE_LIST>, dim3(grid_extend), dim3(BLOCK), 0, stream, scene->d, P, rp, bounce); break;
                case MODE_REFERENCE: hipLaunchKernelGGL(k_extend<MODE_REFERENCE>, dim3(grid_extend), dim3(BLOCK), 0, stream, scene->d, P, rp, bounce); break;
                default: {
                    ExtendIo io{P, scene->d.shapes, P.live[bounce & 1], P.counters + C.live(bounce), rp.n_paths, bounce == 0 ? 1u : 0u};
                    HIP_TRY(launch_traverse(scene, io, P.counters + C.work(bounce), rp.t_min, rp.t_max, stream));
                    break;
                }
            }
            HIP_TRY(timer.end(stream));
            HIP_TRY(timer.begin(stream, EventTimer::BIN));
            hipLaunchKernelGGL(k_bin, dim3(grid_bin), dim3(BLOCK), 0, stream, P, rp, bounce);
            HIP_TRY(timer.end(stream));
            HIP_TRY(timer.begin(stream, EventTimer::SHADE));
            if (scene->has_noise) hipLaunchKernelGGL(k_shade<true>, dim3(grid_shade), dim3(BLOCK), 0, stream, scene->d, P, rp, bounce);
            else hipLaunchKernelGGL(k_shade<false>, dim3(grid_shade), dim3(BLOCK), 0, stream, scene->d, P, rp, bounce);
            HIP_TRY(timer.end(stream));
            ++launches_extend;
            ++launches_shade;
        }
        HIP_TRY(timer.begin(stream, EventTimer::ACCUMULATE));
        if (rp.handover == 0)  // k_paths from the camera ray numbers its records pixel by pixel
            hipLaunchKernelGGL(k_accumulate_by_pixel, dim3(std::min<uint32_t>((n_pixels + 64 * ACC_WAVES - 1) / (64 * ACC_WAVES), (uint32_t)scene->cu_count * 2u)),
                               dim3(ACC_WAVES * 64), 0, stream, P, rp, d_accum);
        else
            hipLaunchKernelGGL(k_accumulate, dim3(grid_for(scene, n_pixels, 8)), dim3(BLOCK), 0, stream, P, rp, d_accum);
        HIP_TRY(timer.end(stream));
        HIP_TRY(hipGetLastError());
        if (timer.enabled) {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(timer.collect());
        }
        ++passes;
        probing = false;  // paths that never thin out: stay with a launch per bounce
    }
    unsigned long long segments[N_SEGMENT_WORDS] = {0};
    HIP_TRY(hipMemcpyAsync(segments, P.segments, sizeof segments, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
#ifdef MIRT_PATHS_STATS
    {
        const char* names[4] = {"refill", "shade", "node", "leaf"};
        const double total = (double)segments[2 + 12];
        double sum = 0;
        for (int k = 0; k < 4; ++k) {
            const double clk = (double)segments[2 + 3 * k], trips = (double)segments[3 + 3 * k], lanes = (double)segments[4 + 3 * k];
            sum += clk;
            std::fprintf(stderr, "[k_paths] %-6s %5.1f %% of wave clocks, %10.0f trips, %7.0f clocks per trip, %4.1f lanes per trip\n",
                         names[k], 100.0 * clk / total, trips, trips ? clk / trips : 0.0, trips ? lanes / trips : 0.0);
        }
        std::fprintf(stderr, "[k_paths] other  %5.1f %%, %.0f trips of the outer loop\n", 100.0 * (total - sum) / total, (double)segments[2 + 13 + 10]);
        {
            const double node_trips = (double)segments[3 + 3 * 2];
            if (node_trips > 0)
                std::fprintf(stderr, "[k_paths]   in a node trip: %.1f lanes hold a hit that waits for shading, %.1f are empty, %.1f can only test shapes\n",
                             (double)segments[2 + 13 + 11] / node_trips, (double)segments[2 + 13 + 12] / node_trips, (double)segments[2 + 13 + 13] / node_trips);
        }
        {
            const unsigned long long* w = segments + 2 + 16 + 8 + 6;
            if (w[4]) {
                const double begin = (double)((1ull << 62) - w[1]), dry = (double)((1ull << 62) - w[0]), end = (double)w[2];
                std::fprintf(stderr, "[k_paths]   launch: %.3f ms from the first wave's start to the last wave's end; the queue ran dry %.3f ms before the end; "
                                     "a wave lived %.3f ms on average (100 MHz clock, last launch of the call)\n",
                             (end - begin) * 1e-5, (end - dry) * 1e-5, (double)w[3] / (double)w[4] * 1e-5);
            }
        }
        const char* parts[10] = {"confirm", "surface", "in_sphere", "unit dir", "metal", "dielectric", "lambertian+texture", "result", "next ray", "store"};
        for (int k = 0; k < 10; ++k)
            std::fprintf(stderr, "[k_paths]   shade: %-20s %5.1f %% of wave clocks\n", parts[k], 100.0 * (double)segments[2 + 13 + k] / total);
        unsigned long long scan[4] = {};
        if (hipMemcpyFromSymbol(scan, HIP_SYMBOL(g_scan_stats), sizeof scan) == hipSuccess && scan[3])
            std::fprintf(stderr, "[k_paths]   scan: %.0f clocks for the boxes, %.0f for the exact tests, %.2f trips of exact tests per scan\n",
                         (double)scan[0] / scan[3], (double)scan[1] / scan[3], (double)scan[2] / scan[3]);
    }
#endif

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->samples = (uint64_t)n_pixels * n_samples_total;
        stats->segments = segments[0];
        stats->segments_paths = segments[1];
        stats->passes = passes;
        stats->launches_extend = launches_extend;
        stats->launches_shade = launches_shade;
        stats->ms_raygen = timer.ms[EventTimer::RAYGEN];
        stats->ms_extend = timer.ms[EventTimer::EXTEND];
        stats->ms_bin = timer.ms[EventTimer::BIN];
        stats->ms_shade = timer.ms[EventTimer::SHADE];
        stats->ms_accumulate = timer.ms[EventTimer::ACCUMULATE];
        stats->ms_paths = timer.ms[EventTimer::PATHS];
        stats->launches_paths = launches_paths;
        stats->handover_bounce = handover;
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    return MIRT_OK;
}

}  // namespace

extern "C" {

int mirt_render_accumulate(const mirt_scene* scene, const mirt_camera* camera, const double background[3],
                           uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed,
                           const mirt_render_opts* opts, double* d_accum, mirt_stats* stats) {
    return no_throw([&] { return render_accumulate_impl(scene, camera, background, width, height, spp, max_depth, seed, opts, d_accum, stats); });
}

static int accum_values(uint32_t width, uint32_t height, size_t* n) {
    if (width == 0 || height == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    *n = (size_t)width * height * 3;
    return MIRT_OK;
}

int mirt_accum_alloc(uint32_t width, uint32_t height, int device, double** d_accum) {
    if (!d_accum) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    *d_accum = nullptr;
    size_t n = 0;
    int rc = accum_values(width, height, &n);
    if (rc != MIRT_OK) return rc;
    if (mirt_device_count() == 0) return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible");
    DeviceGuard guard;
    if ((rc = guard.enter(device)) != MIRT_OK) return rc;
    double* p = nullptr;
    HIP_TRY(hipMalloc((void**)&p, n * sizeof(double)));
    hipError_t e = hipMemset(p, 0, n * sizeof(double));
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(MIRT_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    *d_accum = p;
    return MIRT_OK;
}

void mirt_accum_free(double* d_accum, int device) {
    if (!d_accum) return;
    DeviceGuard guard;
    if (guard.enter(device) != MIRT_OK) return;
    (void)hipFree(d_accum);
}

int mirt_accum_read(const double* d_accum, uint32_t width, uint32_t height, int device, double* host) {
    if (!d_accum || !host) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    size_t n = 0;
    int rc = accum_values(width, height, &n);
    if (rc != MIRT_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(device)) != MIRT_OK) return rc;
    HIP_TRY(hipMemcpy(host, d_accum, n * sizeof(double), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

int mirt_accum_write(double* d_accum, uint32_t width, uint32_t height, int device, const double* host) {
    if (!d_accum || !host) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    size_t n = 0;
    int rc = accum_values(width, height, &n);
    if (rc != MIRT_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(device)) != MIRT_OK) return rc;
    HIP_TRY(hipMemcpy(d_accum, host, n * sizeof(double), hipMemcpyHostToDevice));
    return MIRT_OK;
}

int mirt_render_accumulate_static(const mirt_scene* scene, const mirt_camera* camera, const double background[3],
                                  uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed,
                                  const mirt_render_opts* opts, double* d_accum, mirt_stats* stats) {
    return no_throw([&] { return render_accumulate_impl(scene, camera, background, width, height, spp, max_depth, seed, opts, d_accum, stats, true); });
}

// d_bytes: device memory for the bytes of the selected pixels (3 per pixel of the range), or null: allocated here
static int resolve_tiles_body(const double* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint32_t pixel_begin,
                              uint32_t pixel_end, uint32_t tile_size, uint32_t tile_count, uint32_t tile_index, int device,
                              uint8_t* out_rgb, uint8_t* d_bytes = nullptr) {
    if (!d_accum || !out_rgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    if (width == 0 || height == 0 || spp == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image or spp = 0");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(MIRT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    if (pixel_end == 0) pixel_end = width * height;
    if (pixel_begin >= pixel_end || pixel_end > width * height) return fail(MIRT_ERR_INVALID_ARGUMENT, "bad pixel range");
    if (tile_count > 1 && (tile_size == 0 || tile_index >= tile_count)) return fail(MIRT_ERR_INVALID_ARGUMENT, "bad tile split");
    if (mirt_device_count() == 0) return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != MIRT_OK) return rc;
    PixelSet px;
    px.pixel_begin = pixel_begin;
    px.tile_size = tile_size;
    px.tile_count = tile_count;
    px.tile_index = tile_index;
    px.n_pixels = PixelSet::count(pixel_end - pixel_begin, tile_size, tile_count, tile_index);
    const uint32_t n = px.n_pixels;
    if (n == 0) return MIRT_OK;
    DeviceArena arena;
    uint8_t* d_out = d_bytes;
    if (!d_out) HIP_TRY(arena.alloc(&d_out, (size_t)n * 3));
    const int grid = (int)std::min<size_t>(((size_t)n + BLOCK - 1) / BLOCK, 4096);
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(BLOCK), 0, 0, d_accum, px, spp, d_out);
    HIP_TRY(hipGetLastError());
    if (tile_count <= 1) {
        HIP_TRY(hipMemcpy(out_rgb + (size_t)pixel_begin * 3, d_out, (size_t)n * 3, hipMemcpyDeviceToHost));
        return MIRT_OK;
    }
    std::vector<uint8_t> compact((size_t)n * 3);
    HIP_TRY(hipMemcpy(compact.data(), d_out, compact.size(), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < n; k += tile_size) {  // one run of consecutive pixels per tile
        const uint32_t run = std::min(tile_size, n - k);
        std::memcpy(out_rgb + (size_t)px.idx(k) * 3, compact.data() + (size_t)k * 3, (size_t)run * 3);
    }
    return MIRT_OK;
}

int mirt_resolve_tiles(const double* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint32_t pixel_begin,
                       uint32_t pixel_end, uint32_t tile_size, uint32_t tile_count, uint32_t tile_index, int device,
                       uint8_t* out_rgb) {
    return no_throw([&] { return resolve_tiles_body(d_accum, width, height, spp, pixel_begin, pixel_end, tile_size, tile_count, tile_index, device, out_rgb); });
}

int mirt_resolve(const double* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint32_t pixel_begin,
                 uint32_t pixel_end, int device, uint8_t* out_rgb) {
    return mirt_resolve_tiles(d_accum, width, height, spp, pixel_begin, pixel_end, 0, 1, 0, device, out_rgb);
}

static int mirt_render_body(const mirt_scene* scene, const mirt_camera* camera, const double background[3], uint32_t width,
                uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed, const mirt_render_opts* opts,
                uint8_t* out_rgb, mirt_stats* stats) {
    if (!scene || !out_rgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "null argument");
    if (opts && opts->struct_size == sizeof(mirt_render_opts) && (opts->sample_begin != 0 || (opts->sample_end != 0 && opts->sample_end != spp)))
        return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_render needs the whole sample range; use mirt_render_accumulate + mirt_resolve to split by samples");
    if (width == 0 || height == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "empty image");
    const auto t_start = std::chrono::steady_clock::now();
    DeviceGuard guard;
    int rc = guard.enter(scene->device);
    if (rc != MIRT_OK) return rc;
    // the f64 accumulation image and, behind it, the resolved bytes: one block that stays with the scene
    const size_t n_values = (size_t)width * height * 3;
    const size_t accum_bytes = (n_values * sizeof(double) + 255) & ~(size_t)255;
    Workspace frame;
    HIP_TRY(frame.take(scene, accum_bytes + n_values, true));
    double* d_accum = static_cast<double*>(frame.ptr);
    uint8_t* d_bytes = static_cast<uint8_t*>(frame.ptr) + accum_bytes;
    HIP_TRY(hipMemsetAsync(d_accum, 0, n_values * sizeof(double), 0));
    rc = render_accumulate_impl(scene, camera, background, width, height, spp, max_depth, seed, opts, d_accum, stats);
    if (rc != MIRT_OK) return rc;
    const uint32_t pb = opts ? opts->pixel_begin : 0, pe = opts ? opts->pixel_end : 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = opts && opts->profile && stats;
    if (prof) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, 0));
    }
    rc = resolve_tiles_body(d_accum, width, height, spp, pb, pe, opts ? opts->tile_size : 0, opts ? opts->tile_count : 1,
                            opts ? opts->tile_index : 0, scene->device, out_rgb, d_bytes);
    if (prof) {
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_resolve = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    if (rc != MIRT_OK) return rc;
    if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    return MIRT_OK;
}

}  // extern "C"

extern "C" {

int mirt_intersect_device(const mirt_scene* scene, const mirt_ray* d_rays, size_t n, double t_min, double t_max,
                          int use_bvh, int traversal, mirt_hit* d_hits) {
    return no_throw([&] { return mirt_intersect_device_body(scene, d_rays, n, t_min, t_max, use_bvh, traversal, d_hits); });
}

int mirt_intersect(const mirt_scene* scene, const mirt_ray* rays, size_t n, double t_min, double t_max, int use_bvh,
                   int traversal, mirt_hit* hits) {
    return no_throw([&] { return mirt_intersect_body(scene, rays, n, t_min, t_max, use_bvh, traversal, hits); });
}

int mirt_camera_rays(const mirt_camera* camera, uint32_t width, uint32_t height, uint32_t pixel_begin,
                     uint32_t pixel_end, uint32_t sample, uint64_t seed, int device, mirt_ray* rays) {
    return no_throw([&] { return mirt_camera_rays_body(camera, width, height, pixel_begin, pixel_end, sample, seed, device, rays); });
}

int mirt_render(const mirt_scene* scene, const mirt_camera* camera, const double background[3], uint32_t width,
                uint32_t height, uint32_t spp, uint32_t max_depth, uint64_t seed, const mirt_render_opts* opts,
                uint8_t* out_rgb, mirt_stats* stats) {
    return no_throw([&] { return mirt_render_body(scene, camera, background, width, height, spp, max_depth, seed, opts, out_rgb, stats); });
}

}  // extern "C"

#include "multi_device.h"
