// main.cpp -- the reference's entry point (src/main.rs:27-102) for the C++ host:
// read the tracer config, build camera and world, render through libmirt, write
// the PPM, printing the same progress lines.
//
// Differences, all forced by the environment or by testability:
//  * the world / sampler seeds come from TRACER_SEED when set (the reference always
//    seeds from entropy, main.rs:64, renderer.rs:30; without the variable so does this);
//  * TRACER_BACKGROUND="r,g,b" overrides the hard-coded black background
//    (main.rs:80-81 keeps the sky colour only as a comment).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "config.hpp"
#include "scene.hpp"

int main() {
    using clock = std::chrono::steady_clock;
    auto elapsed = [](clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); };
    try {
        const char* env_path = std::getenv("TRACER_CONFIG_PATH");
        const std::string config_path = env_path ? env_path : "./resources/tracer.toml";
        const rt::TracerConfig cfg = rt::read_tracer_config(config_path);

        const size_t screen_width = cfg.renderer.screen_width;
        const size_t screen_height = cfg.renderer.screen_height;

        const rt::Vec3 v_up(0.0, 1.0, 0.0);
        const double focus_distance = 10.0;  // main.rs:46
        const rt::Camera camera = rt::Camera::create(
            cfg.camera.look_from, cfg.camera.look_at, v_up, cfg.camera.fov, (double)screen_width / (double)screen_height,
            cfg.camera.aperture, focus_distance, cfg.camera.time_start, cfg.camera.time_end);

        uint64_t seed;
        if (const char* s = std::getenv("TRACER_SEED")) {
            seed = std::strtoull(s, nullptr, 0);
        } else {
            std::random_device rd;
            seed = ((uint64_t)rd() << 32) ^ rd();
        }

        auto world_gen_start = clock::now();
        std::printf("Start world gen\n");
        rt::Rng rng(seed);
        const rt::BoundingVolumeHierarchy world = rt::gen_world(rng, cfg.world, cfg.camera.time_start, cfg.camera.time_end);
        std::printf("End world gen-- took %.2fs\n", elapsed(world_gen_start));

        auto tracing_start = clock::now();
        std::printf("Start tracing\n");
        rt::Vec3 background(0.0, 0.0, 0.0);
        if (const char* bg = std::getenv("TRACER_BACKGROUND")) {
            double r, g, b;
            if (std::sscanf(bg, "%lf,%lf,%lf", &r, &g, &b) == 3) background = rt::Vec3(r, g, b);
        }
        mirt_stats stats{};
        const std::vector<uint8_t> buffer =
            rt::render_world(background, world, camera, screen_width, screen_height, cfg.renderer.antialias_iterations,
                             cfg.renderer.render_parallel, cfg.renderer.use_bounding_volume, seed ^ 0x5EED0002ull, &stats);
        const double tracing = elapsed(tracing_start);
        std::printf("End tracing-- took %.2fs\n", tracing);
        std::printf("  %.3f Msamples/s, %.3f Mrays/s (%llu samples, %llu ray segments)\n", stats.samples / tracing / 1e6,
                    stats.segments / tracing / 1e6, (unsigned long long)stats.samples, (unsigned long long)stats.segments);

        auto ppm_start = clock::now();
        std::printf("Start ppm creation\n");
        if (!rt::ppm_create(cfg.output.output_path, screen_width, screen_height, buffer)) {
            std::fprintf(stderr, "cannot write %s\n", cfg.output.output_path.c_str());
            return 1;
        }
        std::printf("End ppm creation-- took %.2fs\n", elapsed(ppm_start));
        return 0;
    } catch (const std::exception& e) {
        // the reference panics (config.rs:211,216; bvh.rs:153,156,229)
        std::fprintf(stderr, "error: %s\n", e.what());
        return 101;
    }
}
