// render.cpp -- render_world (reference src/renderer.rs:9-62) for the C++ host:
// flattens the world, hands it to libmirt and returns the RGB bytes.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "scene.hpp"

namespace rt {

std::vector<uint8_t> render_world(const Vec3& background, const BoundingVolumeHierarchy& world, const Camera& camera,
                                  size_t screen_width, size_t screen_height, size_t antialias_iterations,
                                  bool render_parallel, bool use_bounding_volume, uint64_t seed, mirt_stats* stats) {
    (void)render_parallel;  // rayon vs serial iteration (renderer.rs:27-61) has no device meaning
    std::unique_ptr<FlatScene> flat = flatten(world);
    mirt_scene* scene = nullptr;
    if (mirt_scene_create(&flat->desc, -1, &scene) != MIRT_OK)
        throw std::runtime_error(std::string("mirt_scene_create: ") + mirt_last_error());
    std::vector<uint8_t> out(screen_width * screen_height * 3);
    const double bg[3] = {background.x, background.y, background.z};
    mirt_render_opts opts{};
    opts.struct_size = sizeof opts;
    opts.device = -1;
    opts.use_bvh = use_bounding_volume ? 1 : 0;
    opts.traversal = MIRT_TRAVERSE_FAST;
    // max depth 50 is hard-coded in the reference (renderer.rs:83)
    const int rc = mirt_render(scene, &camera.c, bg, (uint32_t)screen_width, (uint32_t)screen_height,
                               (uint32_t)antialias_iterations, 50, seed, &opts, out.data(), stats);
    mirt_scene_destroy(scene);
    if (rc != MIRT_OK) throw std::runtime_error(std::string("mirt_render: ") + mirt_last_error());
    return out;
}

}  // namespace rt

// ---- render_world in instalments (the sample loop of renderer.rs:77-84 cut at sample boundaries)

namespace rt {
namespace {

// Checkpoint file: this header, then width * height * 3 doubles (pixel-major RGB sums).
struct CheckpointHeader {
    char magic[8];  // "MIRTACC1"
    uint32_t width, height, spp, samples_done, max_depth, use_bvh;
    uint64_t seed;
    uint64_t samples, segments;  // totals of the instalments so far
};

bool read_checkpoint(const std::string& path, CheckpointHeader* h, std::vector<double>* sums) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = std::fread(h, sizeof *h, 1, f) == 1 && std::memcmp(h->magic, "MIRTACC1", 8) == 0;
    if (ok) {
        sums->resize((size_t)h->width * h->height * 3);
        ok = std::fread(sums->data(), sizeof(double), sums->size(), f) == sums->size();
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("checkpoint " + path + " is not a complete accumulation image");
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
    const bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "MIRTACC1", 8) == 0;
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

    CheckpointHeader head{};
    std::memcpy(head.magic, "MIRTACC1", 8);
    head.width = w;
    head.height = h;
    head.spp = spp;
    head.max_depth = 50;  // renderer.rs:83
    head.use_bvh = use_bounding_volume ? 1 : 0;
    head.seed = seed;
    std::vector<double> sums;
    CheckpointHeader found{};
    if (!ckpt.path.empty() && read_checkpoint(ckpt.path, &found, &sums)) {
        if (found.width != w || found.height != h || found.spp != spp || found.seed != seed || found.max_depth != head.max_depth ||
            found.use_bvh != head.use_bvh || found.samples_done > spp)
            throw std::runtime_error("checkpoint " + ckpt.path + " belongs to another render (frame, samples per pixel, seed or mode differ)");
        head = found;
        if (mirt_accum_write(dev.accum, w, h, -1, sums.data()) != MIRT_OK)
            throw std::runtime_error(std::string("mirt_accum_write: ") + mirt_last_error());
    }
    sums.resize((size_t)w * h * 3);

    const double bg[3] = {background.x, background.y, background.z};
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
