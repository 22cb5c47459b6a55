/*
 * mirt_host.h -- C entry points of libmirt_host.so: the host-side mirror of the
 * reference's scene set-up (src/world.rs, src/bvh.rs build, src/camera.rs
 * Camera::new, src/perlin.rs Perlin::new, src/ppm.rs), producing the flat arrays
 * that include/mirt.h consumes.  CPU only; usable without a GPU.
 *
 * The reference does this work in Rust before it calls render_world
 * (src/main.rs:37-70); no Rust toolchain exists here, so this library is the
 * caller-side stand-in.  A Rust `ffi` crate would produce the same arrays from
 * the reference's own Shape / Material / Texture / BVHMember values.
 */
#ifndef MIRT_HOST_H
#define MIRT_HOST_H

#include "mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirt_host_world mirt_host_world; /* owns a flattened BoundingVolumeHierarchy */

const char* mirt_host_last_error(void);

/* gen_world (src/world.rs:9-26): builds the named scene and its BVH.
 * scene_name: "Basic" | "Dynamic" | "Checker" | "Perlin" | "Earth" | "SimpleLight" |
 * "CornellBox" (config.rs:182-195), or "Textured" | "NextWeek" (the two scenes the
 * benchmark configurations compose from the same constructors).
 * max_objects: WorldConfig::Dynamic.max_objects.  seed: seeds the stand-in for the
 * reference's entropy-seeded rng (main.rs:64).  earth_image (nullable): a decoded
 * RGB8 image for Texture::image; else earth_path (nullable; baseline JPEG or binary PPM;
 * default ./resources/earthmap.jpg as world.rs:201) is read;
 * on failure the texture falls back to constant (0,1,1) like texture.rs:171-178. */
int mirt_host_world_create(const char* scene_name, uint64_t max_objects, uint64_t seed, double time_initial,
                           double time_final, const mirt_image* earth_image, const char* earth_path,
                           mirt_host_world** out);
const mirt_scene_desc* mirt_host_world_desc(const mirt_host_world* world);
void mirt_host_world_destroy(mirt_host_world* world);

/* Camera::new (src/camera.rs:21-63). */
int mirt_host_camera_new(const double look_from[3], const double look_at[3], const double v_up[3],
                         double vertical_fov, double aspect, double aperture, double focus_distance,
                         double time_start, double time_end, mirt_camera* out);

/* BoundingVolumeHierarchy::new (src/bvh.rs:18-27, 112-237) over flat shapes: sorts
 * `shapes` in place, writes 2n-1 members to nodes_out (root last), returns the
 * member count or a negative status. */
int mirt_host_bvh_build(mirt_shape* shapes, uint32_t n, double time_initial, double time_final, uint64_t seed,
                        mirt_bvh_member* nodes_out);

/* Hittable::bounding_box of one shape (src/hittable.rs:139-144, 209-214, 273-278,
 * 337-342, 447-465, 523-547). */
int mirt_host_shape_bbox(const mirt_shape* shape, double time_start, double time_end, double bmin[3], double bmax[3]);

/* Perlin::new (src/perlin.rs:20-36). */
int mirt_host_perlin_new(uint64_t seed, mirt_perlin* out);

/* ppm::create (src/ppm.rs:6-27): ASCII P3. */
int mirt_host_ppm_create(const char* output_path, uint32_t image_x, uint32_t image_y, const uint8_t* bytes);

/* The file reading behind Texture::image (src/texture.rs:169-181; the reference uses the `image` crate):
 * a baseline JPEG (what resources/earthmap.jpg is) or a binary PPM, to tightly packed RGB8, row 0 = top.
 * With rgb = NULL only width and height are returned; otherwise capacity must be >= width*height*3. */
int mirt_host_image_read(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgb, size_t capacity);

/* The same pixels as binary PPM: "P6\n<w> <h>\n255\n" then 3 bytes per pixel.  Not in the reference
 * (src/ppm.rs writes P3 only, one formatted line per pixel: 8.3 M lines for a 3840x2160 frame). */
int mirt_host_ppm_create_binary(const char* output_path, uint32_t image_x, uint32_t image_y, const uint8_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_HOST_H */
