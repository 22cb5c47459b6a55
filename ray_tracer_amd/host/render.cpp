// render.cpp -- render_world (reference src/renderer.rs:9-62) for the C++ host:
// flattens the world, hands it to libmirt and returns the RGB bytes.
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
