// render.cpp -- render_world (reference src/renderer.rs:9-62) for the C++ host:
// flattens the world, hands it to libmirt and returns the RGB bytes.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "scene.hpp"

namespace rt {

MultiDevice& multi_device() {
    static MultiDevice settings;
    return settings;
}

namespace {
// One scene per device of a list, destroyed together.
struct SceneList {
    std::vector<mirt_scene*> scenes;
    ~SceneList() {
        for (mirt_scene* s : scenes) mirt_scene_destroy(s);
    }
};
}  // namespace

std::vector<uint8_t> render_world(const Vec3& background, const BoundingVolumeHierarchy& world, const Camera& camera,
                                  size_t screen_width, size_t screen_height, size_t antialias_iterations,
                                  bool render_parallel, bool use_bounding_volume, uint64_t seed, mirt_stats* stats) {
    (void)render_parallel;  // rayon vs serial iteration (renderer.rs:27-61) has no device meaning
    std::unique_ptr<FlatScene> flat = flatten(world);
    // The reference spreads the pixels over the threads of the process (renderer.rs:27-43); here the frame is
    // spread over the devices that are visible (multi_device(): how many, and how the frame is split).
    const MultiDevice& md = multi_device();
    const int visible = mirt_device_count();
    const int n_devices = std::max(1, md.devices > 0 ? std::min(md.devices, visible) : visible);
    SceneList list;
    for (int d = 0; d < n_devices; ++d) {
        mirt_scene* scene = nullptr;
        if (mirt_scene_create(&flat->desc, n_devices > 1 ? d : -1, &scene) != MIRT_OK)
            throw std::runtime_error(std::string("mirt_scene_create: ") + mirt_last_error());
        list.scenes.push_back(scene);
    }
    std::vector<uint8_t> out(screen_width * screen_height * 3);
    const double bg[3] = {background.x, background.y, background.z};
    mirt_render_opts opts{};
    opts.struct_size = sizeof opts;
    opts.device = -1;
    opts.use_bvh = use_bounding_volume ? 1 : 0;
    opts.traversal = MIRT_TRAVERSE_FAST;
    // max depth 50 is hard-coded in the reference (renderer.rs:83)
    if (n_devices == 1) {
        if (mirt_render(list.scenes[0], &camera.c, bg, (uint32_t)screen_width, (uint32_t)screen_height, (uint32_t)antialias_iterations, 50,
                        seed, &opts, out.data(), stats) != MIRT_OK)
            throw std::runtime_error(std::string("mirt_render: ") + mirt_last_error());
        return out;
    }
    mirt_multi_opts multi{};
    multi.struct_size = sizeof multi;
    multi.split = md.sample_split ? MIRT_SPLIT_SAMPLE : MIRT_SPLIT_TILE;
    multi.reduce = md.rccl ? MIRT_REDUCE_RCCL : MIRT_REDUCE_PEER;
    std::vector<mirt_stats> rank_stats(n_devices);
    if (mirt_render_multi(list.scenes.data(), (uint32_t)n_devices, &camera.c, bg, (uint32_t)screen_width, (uint32_t)screen_height,
                          (uint32_t)antialias_iterations, 50, seed, &opts, &multi, out.data(), rank_stats.data()) != MIRT_OK)
        throw std::runtime_error(std::string("mirt_render_multi: ") + mirt_last_error());
    (void)mirt_multi_trim();
    if (stats) {
        *stats = rank_stats[0];
        for (int d = 1; d < n_devices; ++d) {
            stats->samples += rank_stats[d].samples;
            stats->segments += rank_stats[d].segments;
            stats->segments_paths += rank_stats[d].segments_paths;
            stats->passes += rank_stats[d].passes;
        }
    }
    return out;
}

}  // namespace rt

// ---- render_world in instalments (the sample loop of renderer.rs:77-84 cut at sample boundaries)

namespace rt {
namespace {

// Checkpoint file: this header, then width * height * 3 doubles (pixel-major RGB sums), all little-endian
// as every host this library runs on stores them.  The struct has no padding (checked below).
constexpr char CHECKPOINT_MAGIC[9] = "MIRTACC2";
struct CheckpointHeader {
    char magic[8];
    uint32_t width, height, spp, samples_done, max_depth, use_bvh;
    uint64_t seed;
    uint64_t samples, segments;  // totals of the instalments so far
    uint64_t scene_hash;         // render_identity(): the flattened scene, the camera and the background
};
static_assert(sizeof(CheckpointHeader) == 8 + 6 * 4 + 4 * 8, "CheckpointHeader is written as it lies in memory: no padding");

// FNV-1a over the bytes that decide what a render computes apart from its size, sample count and seed:
// every array of the flattened scene (textures field by field: they carry a padding word; images by
// their size and pixels), the camera's derived fields and the background.
struct Fnv {
    uint64_t h = 0xCBF29CE484222325ull;
    void bytes(const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) {
            h ^= b[i];
            h *= 0x100000001B3ull;
        }
    }
    template <class T>
    void value(const T& v) {
        bytes(&v, sizeof v);
    }
};
uint64_t render_identity(const mirt_scene_desc& d, const mirt_camera& camera, const double background[3]) {
    Fnv f;
    f.value(d.n_shapes);
    f.bytes(d.shapes, sizeof(mirt_shape) * d.n_shapes);
    f.value(d.n_materials);
    f.bytes(d.materials, sizeof(mirt_material) * d.n_materials);
    f.value(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const mirt_texture& t = d.textures[i];
        f.value(t.kind), f.value(t.even), f.value(t.odd), f.value(t.perlin), f.value(t.image);
        f.bytes(t.color, sizeof t.color);
        f.value(t.scale);
    }
    f.value(d.n_perlins);
    f.bytes(d.perlins, sizeof(mirt_perlin) * d.n_perlins);
    f.value(d.n_images);
    for (uint32_t i = 0; i < d.n_images; ++i) {
        f.value(d.images[i].width), f.value(d.images[i].height);
        f.bytes(d.images[i].rgb, (size_t)d.images[i].width * d.images[i].height * 3);
    }
    f.value(d.n_nodes);
    if (d.nodes) f.bytes(d.nodes, sizeof(mirt_bvh_member) * d.n_nodes);
    f.value(camera);
    f.bytes(background, 3 * sizeof(double));
    return f.h;
}

// False: no such file.  The header is checked against `want` (frame, samples per pixel, seed, mode, scene)
// BEFORE anything is sized from it.
bool read_checkpoint(const std::string& path, const CheckpointHeader& want, CheckpointHeader* h, std::vector<double>* sums) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool is_one = std::fread(h, sizeof *h, 1, f) == 1 && std::memcmp(h->magic, CHECKPOINT_MAGIC, 8) == 0;
    std::string why;
    if (!is_one) {
        why = "is not an accumulation image of this version";
    } else if (h->width != want.width || h->height != want.height || h->spp != want.spp || h->seed != want.seed ||
               h->max_depth != want.max_depth || h->use_bvh != want.use_bvh || h->samples_done > want.spp) {
        why = "belongs to another render (frame, samples per pixel, seed or mode differ)";
    } else if (h->scene_hash != want.scene_hash) {
        why = "belongs to another render (the scene, the camera or the background differ)";
    } else {
        sums->resize((size_t)want.width * want.height * 3);
        if (std::fread(sums->data(), sizeof(double), sums->size(), f) != sums->size()) why = "is not a complete accumulation image";
    }
    std::fclose(f);
    if (!why.empty()) throw std::runtime_error("checkpoint " + path + " " + why);
    return true;
}

void write_checkpoint(const std::string& path, const CheckpointHeader& h, const std::vector<double>& sums) {
    const std::string tmp = path + ".part";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write checkpoint " + tmp);
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(sums.data(), sizeof(double), sums.size(), f) == sums.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot write checkpoint " + path);
}

struct SceneHandle {
    mirt_scene* scene = nullptr;
    double* accum = nullptr;
    ~SceneHandle() {
        mirt_accum_free(accum, -1);
        mirt_scene_destroy(scene);
    }
};

}  // namespace

bool checkpoint_seed(const std::string& path, uint64_t* render_seed) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    CheckpointHeader h{};
    const bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, CHECKPOINT_MAGIC, 8) == 0;
    std::fclose(f);
    if (ok) *render_seed = h.seed;
    return ok;
}

std::vector<uint8_t> render_world_resumable(const Vec3& background, const BoundingVolumeHierarchy& world, const Camera& camera,
                                            size_t screen_width, size_t screen_height, size_t antialias_iterations,
                                            bool use_bounding_volume, uint64_t seed, const Checkpointing& ckpt,
                                            mirt_stats* stats, size_t* samples_done, bool* finished) {
    const uint32_t w = (uint32_t)screen_width, h = (uint32_t)screen_height, spp = (uint32_t)antialias_iterations;
    std::unique_ptr<FlatScene> flat = flatten(world);
    SceneHandle dev;
    if (mirt_scene_create(&flat->desc, -1, &dev.scene) != MIRT_OK)
        throw std::runtime_error(std::string("mirt_scene_create: ") + mirt_last_error());
    if (mirt_accum_alloc(w, h, -1, &dev.accum) != MIRT_OK) throw std::runtime_error(std::string("mirt_accum_alloc: ") + mirt_last_error());

    const double bg[3] = {background.x, background.y, background.z};
    CheckpointHeader head{};
    std::memcpy(head.magic, CHECKPOINT_MAGIC, 8);
    head.width = w;
    head.height = h;
    head.spp = spp;
    head.max_depth = 50;  // renderer.rs:83
    head.use_bvh = use_bounding_volume ? 1 : 0;
    head.seed = seed;
    head.scene_hash = render_identity(flat->desc, camera.c, bg);
    std::vector<double> sums;
    CheckpointHeader found{};
    if (!ckpt.path.empty() && read_checkpoint(ckpt.path, head, &found, &sums)) {
        head = found;
        if (mirt_accum_write(dev.accum, w, h, -1, sums.data()) != MIRT_OK)
            throw std::runtime_error(std::string("mirt_accum_write: ") + mirt_last_error());
    }
    sums.resize((size_t)w * h * 3);

    mirt_stats total{};
    uint32_t done = head.samples_done;
    const uint32_t goal = ckpt.stop_after ? std::min<uint32_t>(spp, (uint32_t)ckpt.stop_after) : spp;
    while (done < goal) {
        const uint32_t step = ckpt.every ? (uint32_t)std::min<size_t>(ckpt.every, goal - done) : goal - done;
        mirt_render_opts opts{};
        opts.struct_size = sizeof opts;
        opts.device = -1;
        opts.use_bvh = head.use_bvh;
        opts.traversal = MIRT_TRAVERSE_FAST;
        opts.sample_begin = done;
        opts.sample_end = done + step;
        mirt_stats st{};
        if (mirt_render_accumulate(dev.scene, &camera.c, bg, w, h, spp, head.max_depth, seed, &opts, dev.accum, &st) != MIRT_OK)
            throw std::runtime_error(std::string("mirt_render_accumulate: ") + mirt_last_error());
        done += step;
        total.samples += st.samples;
        total.segments += st.segments;
        total.passes += st.passes;
        total.ms_total += st.ms_total;
        if (!ckpt.path.empty()) {
            if (mirt_accum_read(dev.accum, w, h, -1, sums.data()) != MIRT_OK)
                throw std::runtime_error(std::string("mirt_accum_read: ") + mirt_last_error());
            head.samples_done = done;
            head.samples += st.samples;
            head.segments += st.segments;
            write_checkpoint(ckpt.path, head, sums);
        }
    }
    std::vector<uint8_t> out((size_t)w * h * 3);
    // an unfinished frame is shown as the mean of the samples it has (pixel / samples so far)
    if (mirt_resolve(dev.accum, w, h, done ? done : 1, 0, 0, -1, out.data()) != MIRT_OK)
        throw std::runtime_error(std::string("mirt_resolve: ") + mirt_last_error());
    if (stats) *stats = total;
    if (samples_done) *samples_done = done;
    if (finished) *finished = done == spp;
    return out;
}

}  // namespace rt
