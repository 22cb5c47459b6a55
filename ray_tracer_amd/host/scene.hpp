// scene.hpp -- host-side mirror of the reference's scene API.
//
// The reference keeps its entry point in Rust (src/main.rs) and builds scenes
// from Shape / Material / Texture constructors (src/hittable.rs:573-623,
// src/material.rs:162-189, src/texture.rs:132-182).  No Rust toolchain exists
// here, so this C++ host offers the same constructors under the same names and
// argument orders, builds the same BoundingVolumeHierarchy, and flattens the
// result into the plain arrays of include/mirt.h -- the arrays a Rust `ffi`
// crate would produce from the reference's own types.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "../../include/mirt_rng.h"

namespace rt {

struct Vec3 {
    double x = 0.0, y = 0.0, z = 0.0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
};

// Stand-in for the `rng: &mut T` the reference threads through scene set-up
// (src/world.rs:9-26, src/bvh.rs:18-27, src/perlin.rs:20-36).
class Rng {
public:
    explicit Rng(uint64_t seed) : s_(mirt_seq_new(seed)) {}
    double random_double() { return mirt_seq_double(&s_); }                       // util.rs:8-10
    double gen_range(double lo, double hi) { return mirt_seq_range(&s_, lo, hi); }  // vec3.rs:28-30
    uint32_t gen_below(uint32_t n) { return mirt_seq_below(&s_, n); }             // bvh.rs:146
    // SliceRandom::shuffle of rand 0.7: Fisher-Yates walking down from the top.
    template <class T>
    void shuffle(std::vector<T>& v) {
        for (size_t i = v.size(); i-- > 1;) {
            size_t j = gen_below((uint32_t)(i + 1));
            std::swap(v[i], v[j]);
        }
    }

private:
    mirt_seq s_;
};

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

// src/perlin.rs:10-17
struct Perlin {
    mirt_perlin tables;
    static std::shared_ptr<Perlin> create(Rng& rng);  // Perlin::new, perlin.rs:20-36
};

// src/texture.rs:124-182
struct Texture {
    uint32_t kind = MIRT_TEX_CONSTANT;
    Vec3 color;
    std::shared_ptr<Texture> even, odd;
    std::shared_ptr<Perlin> perlin;
    double scale = 1.0;
    std::shared_ptr<Image> img;

    static Texture constant(double r, double g, double b);
    static Texture checker_color(double r_even, double g_even, double b_even, double r_odd, double g_odd,
                                 double b_odd);
    static Texture checker(const Texture& even, const Texture& odd);
    static Texture noise(Rng& rng);
    static Texture scaled_noise(Rng& rng, double scale);
    // Texture::image: the reference decodes any format `image` supports; this host
    // reads binary PPM (P6) only, or takes an already decoded image.  A file that
    // cannot be read logs to stderr and yields constant (0,1,1) like texture.rs:171-178.
    static Texture image(const std::string& filepath);
    static Texture image(std::shared_ptr<Image> decoded);
};

// src/material.rs:154-189
struct Material {
    uint32_t kind = MIRT_MAT_LAMBERTIAN;
    Texture texture;
    Vec3 albedo;
    double param = 0.0;

    static Material lambertian(double r, double g, double b);
    static Material textured(const Texture& texture);
    static Material metal(double e0, double e1, double e2, double fuzziness);
    static Material dielectric(double ref_idx);
    static Material diffuse_light(const Texture& texture);
};

// src/hittable.rs:563-623
struct Shape {
    uint32_t kind = MIRT_SHAPE_SPHERE;
    double f[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Material material;

    static Shape sphere(double x, double y, double z, double radius, const Material& m);
    static Shape xy_rect(double x0, double x1, double y0, double y1, double k, const Material& m);
    static Shape xz_rect(double x0, double x1, double z0, double z1, double k, const Material& m);
    static Shape yz_rect(double y0, double y1, double z0, double z1, double k, const Material& m);
    static Shape cube(double x0, double y0, double z0, double x1, double y1, double z1, const Material& m);
    static Shape moving_sphere(double x0, double y0, double z0, double x1, double y1, double z1, double radius,
                               const Material& m, double time_start, double time_end);
    // Hittable::bounding_box for this shape (hittable.rs:139-144, 209-214, 273-278, 337-342, 447-465, 523-547)
    void bounding_box(double time_start, double time_end, double bmin[3], double bmax[3]) const;
};

// src/bvh.rs:11-27, 93-101
struct BoundingVolumeHierarchy {
    std::vector<Shape> shapes;            // sorted in place by the build
    std::vector<mirt_bvh_member> nodes;   // post-order, root last
    static BoundingVolumeHierarchy create(Rng& rng, std::vector<Shape> shapes, double time_initial,
                                          double time_final);  // BoundingVolumeHierarchy::new
};

// The same build over already flattened shapes (sorted in place); nodes get 2n-1 members.
void bvh_build_flat(Rng& rng, mirt_shape* shapes, uint32_t n, double time_initial, double time_final,
                    std::vector<mirt_bvh_member>* nodes);
void flat_shape_bbox(const mirt_shape& s, double time_start, double time_end, double bmin[3], double bmax[3]);

// src/camera.rs:7-63
struct Camera {
    mirt_camera c;
    static Camera create(const Vec3& look_from, const Vec3& look_at, const Vec3& v_up, double vertical_fov,
                         double aspect, double aperture, double focus_distance, double time_start,
                         double time_end);  // Camera::new
};

// src/config.rs:182-195 plus the two scenes the benchmark configurations compose
// from the same constructors.
enum class WorldKind { Basic, Dynamic, Checker, Perlin, Earth, SimpleLight, CornellBox, Textured, NextWeek };

struct WorldConfig {
    WorldKind scene = WorldKind::Basic;
    size_t max_objects = 10;                 // config.rs:199-201
    std::string earth_path = "./resources/earthmap.jpg";  // world.rs:201
    std::shared_ptr<Image> earth_image;      // already decoded image, takes precedence over earth_path
};

bool world_kind_from_name(const std::string& name, WorldKind* out);

// src/world.rs:9-26
std::vector<Shape> gen_shapes(Rng& rng, const WorldConfig& cfg);
BoundingVolumeHierarchy gen_world(Rng& rng, const WorldConfig& cfg, double time_initial, double time_final);

// The flat arrays of include/mirt.h for one world.  Owns all storage that
// `desc` points into.  Materials, textures, Perlin tables and images that the
// reference clones per shape (hittable.rs:406-411, material.rs:40-43,
// texture.rs:56-60) are stored once.
struct FlatScene {
    std::vector<mirt_shape> shapes;
    std::vector<mirt_material> materials;
    std::vector<mirt_texture> textures;
    std::vector<mirt_perlin> perlins;
    std::vector<mirt_image> images;
    std::vector<std::shared_ptr<Image>> image_storage;
    std::vector<mirt_bvh_member> nodes;
    mirt_scene_desc desc;
};
std::unique_ptr<FlatScene> flatten(const BoundingVolumeHierarchy& world);

// How render_world spreads a frame over the devices that are visible (the reference spreads it over the
// threads of the process, renderer.rs:27-43).  Set once by the caller, before rendering.
struct MultiDevice {
    int devices = 0;            // how many devices to use; 0 = all that are visible
    bool sample_split = false;  // false: interleaved image rows per device (no exchange); true: samples per device, summed
    bool rccl = false;          // sample split: sum with ncclReduceScatter instead of peer copies
};
MultiDevice& multi_device();

// src/renderer.rs:9-62: render_world through libmirt (mirt_scene_create + mirt_render, or one scene per
// device + mirt_render_multi when several devices are visible).  render_parallel has no meaning on the
// device and is ignored.  Throws std::runtime_error with mirt_last_error() on failure.
std::vector<uint8_t> render_world(const Vec3& background, const BoundingVolumeHierarchy& world, const Camera& camera,
                                  size_t screen_width, size_t screen_height, size_t antialias_iterations,
                                  bool render_parallel, bool use_bounding_volume, uint64_t seed = 0x5EED0002ull,
                                  mirt_stats* stats = nullptr);

// src/ppm.rs:6-27
bool ppm_create(const std::string& output_path, size_t image_x, size_t image_y, const std::vector<uint8_t>& bytes);
bool ppm_create_binary(const std::string& output_path, size_t image_x, size_t image_y, const std::vector<uint8_t>& bytes);
bool read_ppm_p6(const std::string& path, Image* out);
// Baseline JPEG (what the reference's resources/earthmap.jpg is); *why says what was wrong with the file.
bool read_jpeg_baseline(const std::string& path, Image* out, std::string* why);
// Texture::image's file reading (texture.rs:169-181): a baseline JPEG or a binary PPM, told apart by their first bytes.
bool read_image(const std::string& path, Image* out, std::string* why);

// render_world in instalments with a checkpoint file: after every `every` samples per pixel the f64
// accumulation image is written to `path` (to a temporary name first, then renamed); a run that finds
// `path` continues after the samples it holds.  `stop_after` > 0 ends this run once that many samples
// per pixel are in the checkpoint (the frame is then unfinished: the returned bytes are the mean of
// the samples so far and *finished = false).  The sums, and so the bytes of the finished frame, are
// those of one uninterrupted call bit for bit.  stats adds up over the instalments of THIS run.
struct Checkpointing {
    std::string path;        // empty: no checkpoint file
    size_t every = 0;        // samples per pixel per instalment; 0 = all that are left
    size_t stop_after = 0;   // 0 = run to the end
};
// The sampler seed a checkpoint file was written with (false: no such file, or not a checkpoint).
bool checkpoint_seed(const std::string& path, uint64_t* render_seed);
std::vector<uint8_t> render_world_resumable(const Vec3& background, const BoundingVolumeHierarchy& world, const Camera& camera,
                                            size_t screen_width, size_t screen_height, size_t antialias_iterations,
                                            bool use_bounding_volume, uint64_t seed, const Checkpointing& ckpt,
                                            mirt_stats* stats, size_t* samples_done, bool* finished);

}  // namespace rt
