// world.cpp -- the reference's built-in scenes (src/world.rs:28-246) through the
// mirrored constructors, plus two scenes the benchmark configurations compose
// from the same constructors (the reference API has no rotation, translation
// or participating media, so neither do they).
#include <numeric>
#include <stdexcept>

#include "scene.hpp"

namespace rt {

bool world_kind_from_name(const std::string& name, WorldKind* out) {
    static const struct {
        const char* n;
        WorldKind k;
    } table[] = {{"Basic", WorldKind::Basic},       {"Dynamic", WorldKind::Dynamic},
                 {"Checker", WorldKind::Checker},   {"Perlin", WorldKind::Perlin},
                 {"Earth", WorldKind::Earth},       {"SimpleLight", WorldKind::SimpleLight},
                 {"CornellBox", WorldKind::CornellBox}, {"Textured", WorldKind::Textured},
                 {"NextWeek", WorldKind::NextWeek}};
    for (const auto& e : table)
        if (name == e.n) {
            *out = e.k;
            return true;
        }
    return false;
}

namespace {

Texture earth_texture(const WorldConfig& cfg) {
    if (cfg.earth_image) return Texture::image(cfg.earth_image);
    return Texture::image(cfg.earth_path);
}

// world.rs:28-84
std::vector<Shape> static_world() {
    std::vector<Shape> spheres;
    spheres.push_back(Shape::sphere(0.0, 0.0, -1.0, 0.5, Material::lambertian(0.8, 0.3, 0.3)));
    spheres.push_back(Shape::sphere(0.0, -100.5, -1.0, 100.0, Material::lambertian(0.8, 0.8, 0.0)));
    spheres.push_back(Shape::sphere(1.0, 0.0, -1.0, 0.5, Material::metal(0.8, 0.6, 0.2, 0.0)));
    spheres.push_back(Shape::sphere(-1.0, 0.0, -1.0, 0.5, Material::dielectric(1.5)));
    spheres.push_back(Shape::sphere(-1.0, 0.0, -1.0, -0.45, Material::dielectric(1.5)));
    for (int n = 0; n < 3; ++n) {
        const double i = (double)n;
        const double x = 1.5 + (i * 0.2);
        const double z = -1.5 + (i * 0.2);
        spheres.push_back(Shape::sphere(x, 0.0, z, 0.2, Material::lambertian(i * 0.5, i * 0.5, i * 0.5)));
    }
    for (int n = 0; n < 2; ++n) {
        const double i = (double)n;
        const double x = 1.1 + (i * 0.4) + i;
        const double z = -1.3 - (i * 0.3) + i;
        spheres.push_back(Shape::moving_sphere(x, 0.0, z, x, 0.7, z, 0.2, Material::lambertian(i * 0.5, i * 0.5, i * 0.5),
                                               0.0, 1.0));
    }
    for (int n = 0; n < 3; ++n) {
        const double i = (double)n;
        const double x = 1.2 - (i * 0.8) + i;
        const double z = -1.6 + (i * 0.1) + i;
        spheres.push_back(Shape::sphere(x, 0.0, z, 0.2, Material::metal(0.8, 0.6, 0.2, 0.0)));
    }
    for (int n = 0; n < 2; ++n) {
        const double i = (double)n;
        const double x = 1.4 + (i * 0.2) + i;
        const double z = -1.2 + (i * 0.6) + i;
        spheres.push_back(Shape::sphere(x, 0.0, z, 0.2, Material::dielectric(1.5)));
    }
    return spheres;
}

// world.rs:86-173.  Draw order: shuffle of the outer range; per row a shuffle of
// the inner range; per cell material choice, material parameters, the moving
// coin, then the position.
std::vector<Shape> random_world(Rng& rng, size_t max_objects) {
    max_objects += 4;
    std::vector<Shape> world;
    world.reserve(max_objects);
    world.push_back(Shape::sphere(0.0, -1000.0, 0.0, 1000.0,
                                  Material::textured(Texture::checker_color(0.2, 0.3, 0.1, 0.9, 0.9, 0.9))));
    world.push_back(Shape::sphere(0.0, 1.0, 0.0, 1.0, Material::dielectric(1.5)));
    world.push_back(Shape::sphere(-4.0, 1.0, 0.0, 1.0, Material::lambertian(0.4, 0.2, 0.1)));
    world.push_back(Shape::sphere(4.0, 1.0, 0.0, 1.0, Material::metal(0.7, 0.6, 0.5, 0.0)));

    std::vector<int> outer_range(22);
    std::iota(outer_range.begin(), outer_range.end(), -11);
    rng.shuffle(outer_range);
    for (int a : outer_range) {
        std::vector<int> inner_range(22);
        std::iota(inner_range.begin(), inner_range.end(), -11);
        rng.shuffle(inner_range);
        for (int b : inner_range) {
            if (world.size() == max_objects) break;
            const double material_choice = rng.random_double();
            Material material;
            if (material_choice < 0.8) {
                const double r = rng.random_double() * rng.random_double();
                const double g = rng.random_double() * rng.random_double();
                const double bl = rng.random_double() * rng.random_double();
                material = Material::lambertian(r, g, bl);
            } else if (material_choice < 0.95) {
                const double r = 0.5 * (1.0 + rng.random_double());
                const double g = 0.5 * (1.0 + rng.random_double());
                const double bl = 0.5 * (1.0 + rng.random_double());
                const double fuzz = 0.5 * rng.random_double();
                material = Material::metal(r, g, bl, fuzz);
            } else {
                material = Material::dielectric(1.5);
            }
            const bool is_moving = rng.random_double() < 0.40;
            if (material_choice < 0.8 && is_moving) {
                const double x = (double)a + 0.9 * rng.random_double();
                const double z = (double)b + 0.9 * rng.random_double();
                const double y1 = 0.2 + (0.5 * rng.random_double());
                world.push_back(Shape::moving_sphere(x, 0.2, z, x, y1, z, 0.2, material, 0.0, 1.0));
            } else {
                const double x = (double)a + 0.9 * rng.random_double();
                const double z = (double)b + 0.9 * rng.random_double();
                world.push_back(Shape::sphere(x, 0.2, z, 0.2, material));
            }
        }
    }
    return world;
}

// world.rs:175-182
std::vector<Shape> two_checker_spheres() {
    Texture texture = Texture::checker_color(0.2, 0.3, 0.1, 0.9, 0.9, 0.9);
    return {Shape::sphere(0.0, -10.0, 0.0, 10.0, Material::textured(texture)),
            Shape::sphere(0.0, 10.0, 0.0, 10.0, Material::textured(texture))};
}

// world.rs:184-198
std::vector<Shape> two_perlin_spheres(Rng& rng) {
    Texture texture = Texture::scaled_noise(rng, 4.0);
    return {Shape::sphere(0.0, -1000.0, 0.0, 1000.0, Material::textured(texture)),
            Shape::sphere(0.0, 2.0, 0.0, 2.0, Material::textured(texture))};
}

// world.rs:200-204
std::vector<Shape> earth(const WorldConfig& cfg) {
    return {Shape::sphere(0.0, 0.0, 0.0, 2.0, Material::textured(earth_texture(cfg)))};
}

// world.rs:206-223
std::vector<Shape> simple_light(Rng& rng) {
    Texture noise = Texture::noise(rng);
    Texture color = Texture::constant(4.0, 4.0, 4.0);
    return {Shape::sphere(0.0, -1000.0, 0.0, 1000.0, Material::textured(noise)),
            Shape::sphere(0.0, 2.0, 0.0, 2.0, Material::textured(noise)),
            Shape::xy_rect(3.0, 5.0, 1.0, 3.0, -2.0, Material::diffuse_light(color)),
            Shape::sphere(0.0, 7.0, 0.0, 2.0, Material::diffuse_light(color))};
}

// world.rs:225-246
std::vector<Shape> cornell_box() {
    Material red = Material::lambertian(0.65, 0.05, 0.05);
    Material green = Material::lambertian(0.12, 0.45, 0.15);
    Material white = Material::lambertian(0.73, 0.73, 0.73);
    Material light = Material::diffuse_light(Texture::constant(15.0, 15.0, 15.0));
    return {Shape::yz_rect(0.0, 555.0, 0.0, 555.0, 555.0, green),
            Shape::yz_rect(0.0, 555.0, 0.0, 555.0, 0.0, red),
            Shape::xz_rect(213.0, 343.0, 227.0, 332.0, 554.0, light),
            Shape::xz_rect(0.0, 555.0, 0.0, 555.0, 0.0, white),
            Shape::xz_rect(0.0, 555.0, 0.0, 555.0, 555.0, white),
            Shape::xy_rect(0.0, 555.0, 0.0, 555.0, 555.0, white),
            Shape::cube(130.0, 0.0, 65.0, 295.0, 165.0, 230.0, white),
            Shape::cube(265.0, 0.0, 295.0, 430.0, 330.0, 460.0, white)};
}

// Benchmark configuration 3: checker ground, two Perlin spheres, one earth sphere.
std::vector<Shape> textured_scene(Rng& rng, const WorldConfig& cfg) {
    Texture checker = Texture::checker_color(0.2, 0.3, 0.1, 0.9, 0.9, 0.9);
    Texture marble4 = Texture::scaled_noise(rng, 4.0);
    Texture marble1 = Texture::noise(rng);
    return {Shape::sphere(0.0, -1000.0, 0.0, 1000.0, Material::textured(checker)),
            Shape::sphere(0.0, 2.0, 0.0, 2.0, Material::textured(marble4)),
            Shape::sphere(-5.0, 1.5, 2.0, 1.5, Material::textured(marble1)),
            Shape::sphere(4.5, 2.0, -1.0, 2.0, Material::textured(earth_texture(cfg)))};
}

// Benchmark configuration 5: a "Next Week"-like scene from the reference's own
// constructors: 50 x 50 ground boxes of random height, an area light, a moving
// sphere, glass and metal spheres, an earth sphere, a Perlin sphere and a
// cluster of 1000 small white spheres (3507 shapes).
std::vector<Shape> next_week_scene(Rng& rng, const WorldConfig& cfg) {
    std::vector<Shape> world;
    world.reserve(3507);
    Material ground = Material::lambertian(0.48, 0.83, 0.53);
    const int boxes_per_side = 50;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            const double w = 40.0;
            const double x0 = -1000.0 + (double)i * w;
            const double z0 = -1000.0 + (double)j * w;
            const double y1 = 1.0 + 100.0 * rng.random_double();
            world.push_back(Shape::cube(x0, 0.0, z0, x0 + w, y1, z0 + w, ground));
        }
    world.push_back(Shape::xz_rect(123.0, 423.0, 147.0, 412.0, 554.0,
                                   Material::diffuse_light(Texture::constant(7.0, 7.0, 7.0))));
    world.push_back(Shape::moving_sphere(400.0, 400.0, 200.0, 430.0, 400.0, 200.0, 50.0,
                                         Material::lambertian(0.7, 0.3, 0.1), 0.0, 1.0));
    world.push_back(Shape::sphere(260.0, 150.0, 45.0, 50.0, Material::dielectric(1.5)));
    world.push_back(Shape::sphere(0.0, 150.0, 145.0, 50.0, Material::metal(0.8, 0.8, 0.9, 1.0)));
    world.push_back(Shape::sphere(360.0, 150.0, 145.0, 70.0, Material::dielectric(1.5)));
    world.push_back(Shape::sphere(400.0, 200.0, 400.0, 100.0, Material::textured(earth_texture(cfg))));
    world.push_back(Shape::sphere(220.0, 280.0, 300.0, 80.0, Material::textured(Texture::scaled_noise(rng, 0.1))));
    Material white = Material::lambertian(0.73, 0.73, 0.73);
    for (int k = 0; k < 1000; ++k) {
        const double x = -100.0 + 165.0 * rng.random_double();
        const double y = 270.0 + 165.0 * rng.random_double();
        const double z = 395.0 + 165.0 * rng.random_double();
        world.push_back(Shape::sphere(x, y, z, 10.0, white));
    }
    return world;
}

}  // namespace

std::vector<Shape> gen_shapes(Rng& rng, const WorldConfig& cfg) {
    switch (cfg.scene) {
        case WorldKind::Basic: return static_world();
        case WorldKind::Dynamic: return random_world(rng, cfg.max_objects);
        case WorldKind::Checker: return two_checker_spheres();
        case WorldKind::Perlin: return two_perlin_spheres(rng);
        case WorldKind::Earth: return earth(cfg);
        case WorldKind::SimpleLight: return simple_light(rng);
        case WorldKind::CornellBox: return cornell_box();
        case WorldKind::Textured: return textured_scene(rng, cfg);
        case WorldKind::NextWeek: return next_week_scene(rng, cfg);
    }
    throw std::runtime_error("unknown world kind");
}

BoundingVolumeHierarchy gen_world(Rng& rng, const WorldConfig& cfg, double time_initial, double time_final) {
    return BoundingVolumeHierarchy::create(rng, gen_shapes(rng, cfg), time_initial, time_final);
}

}  // namespace rt
