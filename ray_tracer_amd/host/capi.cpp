// capi.cpp -- plain-C entry points over the host-side scene mirror, so that
// callers without a C++ toolchain (the Python tests and benchmark through
// ctypes) can generate the reference's scenes, cameras and BVHs as the flat
// arrays of include/mirt.h.  Declared in include/mirt_host.h.
#include <cstring>
#include <exception>
#include <string>

#include "../../include/mirt_host.h"
#include "scene.hpp"

namespace {
thread_local std::string tl_host_error;
int host_fail(const std::string& msg) {
    tl_host_error = msg;
    return MIRT_ERR_INVALID_ARGUMENT;
}
}  // namespace

struct mirt_host_world {
    std::unique_ptr<rt::FlatScene> flat;
};

extern "C" {

const char* mirt_host_last_error(void) { return tl_host_error.c_str(); }

int mirt_host_world_create(const char* scene_name, uint64_t max_objects, uint64_t seed, double time_initial,
                           double time_final, const mirt_image* earth_image, const char* earth_path,
                           mirt_host_world** out) {
    if (!scene_name || !out) return host_fail("scene_name or out is null");
    *out = nullptr;
    try {
        rt::WorldConfig cfg;
        if (!rt::world_kind_from_name(scene_name, &cfg.scene)) return host_fail(std::string("unknown scene ") + scene_name);
        cfg.max_objects = (size_t)max_objects;
        if (earth_path) cfg.earth_path = earth_path;
        if (earth_image && earth_image->rgb) {
            auto img = std::make_shared<rt::Image>();
            img->width = earth_image->width;
            img->height = earth_image->height;
            img->rgb.assign(earth_image->rgb, earth_image->rgb + (size_t)img->width * img->height * 3);
            cfg.earth_image = img;
        }
        rt::Rng rng(seed);  // one generator for scene and BVH, like main.rs:63-70
        rt::BoundingVolumeHierarchy world = rt::gen_world(rng, cfg, time_initial, time_final);
        auto* w = new mirt_host_world();
        w->flat = rt::flatten(world);
        *out = w;
        return MIRT_OK;
    } catch (const std::exception& e) {
        return host_fail(e.what());
    }
}

const mirt_scene_desc* mirt_host_world_desc(const mirt_host_world* world) { return world ? &world->flat->desc : nullptr; }

void mirt_host_world_destroy(mirt_host_world* world) { delete world; }

int mirt_host_camera_new(const double look_from[3], const double look_at[3], const double v_up[3],
                         double vertical_fov, double aspect, double aperture, double focus_distance,
                         double time_start, double time_end, mirt_camera* out) {
    if (!look_from || !look_at || !v_up || !out) return host_fail("null argument");
    rt::Camera c = rt::Camera::create(rt::Vec3(look_from[0], look_from[1], look_from[2]),
                                      rt::Vec3(look_at[0], look_at[1], look_at[2]), rt::Vec3(v_up[0], v_up[1], v_up[2]),
                                      vertical_fov, aspect, aperture, focus_distance, time_start, time_end);
    *out = c.c;
    return MIRT_OK;
}

int mirt_host_bvh_build(mirt_shape* shapes, uint32_t n, double time_initial, double time_final, uint64_t seed,
                        mirt_bvh_member* nodes_out) {
    if (!shapes || !nodes_out || n == 0) return host_fail("null argument or empty shape list");
    try {
        rt::Rng rng(seed);
        std::vector<mirt_bvh_member> nodes;
        rt::bvh_build_flat(rng, shapes, n, time_initial, time_final, &nodes);
        std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(mirt_bvh_member));
        return (int)nodes.size();
    } catch (const std::exception& e) {
        return host_fail(e.what());
    }
}

int mirt_host_shape_bbox(const mirt_shape* shape, double time_start, double time_end, double bmin[3], double bmax[3]) {
    if (!shape || !bmin || !bmax) return host_fail("null argument");
    try {
        rt::flat_shape_bbox(*shape, time_start, time_end, bmin, bmax);
        return MIRT_OK;
    } catch (const std::exception& e) {
        return host_fail(e.what());
    }
}

int mirt_host_perlin_new(uint64_t seed, mirt_perlin* out) {
    if (!out) return host_fail("null argument");
    rt::Rng rng(seed);
    *out = rt::Perlin::create(rng)->tables;
    return MIRT_OK;
}

int mirt_host_ppm_create(const char* output_path, uint32_t image_x, uint32_t image_y, const uint8_t* bytes) {
    if (!output_path || !bytes) return host_fail("null argument");
    std::vector<uint8_t> v(bytes, bytes + (size_t)image_x * image_y * 3);
    return rt::ppm_create(output_path, image_x, image_y, v) ? MIRT_OK : host_fail("cannot write ppm");
}

int mirt_host_ppm_create_binary(const char* output_path, uint32_t image_x, uint32_t image_y, const uint8_t* bytes) {
    if (!output_path || !bytes) return host_fail("null argument");
    std::vector<uint8_t> v(bytes, bytes + (size_t)image_x * image_y * 3);
    return rt::ppm_create_binary(output_path, image_x, image_y, v) ? MIRT_OK : host_fail("cannot write ppm");
}

int mirt_host_image_read(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgb, size_t capacity) {
    if (!path || !width || !height) return host_fail("null argument");
    try {
        rt::Image img;
        std::string why;
        if (!rt::read_image(path, &img, &why)) return host_fail(std::string(path) + ": " + why);
        *width = img.width;
        *height = img.height;
        if (rgb) {
            if (capacity < img.rgb.size()) return host_fail("buffer too small for the image");
            std::memcpy(rgb, img.rgb.data(), img.rgb.size());
        }
        return MIRT_OK;
    } catch (const std::exception& e) {
        return host_fail(e.what());
    }
}

}  // extern "C"
