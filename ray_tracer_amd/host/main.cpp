// main.cpp -- the reference's entry point (src/main.rs:27-102) for the C++ host:
// read the tracer config, build camera and world, render through libmirt, write
// the PPM, printing the same progress lines.
//
// Differences, all forced by the environment or by testability:
//  * the world / sampler seeds come from TRACER_SEED when set (the reference always
//    seeds from entropy, main.rs:64, renderer.rs:30; without the variable so does this);
//  * TRACER_BACKGROUND="r,g,b" overrides the hard-coded black background
//    (main.rs:80-81 keeps the sky colour only as a comment).
//  * long renders can be cut into instalments and survive the end of a process (the reference renders a
//    frame in one call or not at all): --checkpoint FILE (or TRACER_CHECKPOINT) keeps the f64 accumulation
//    image in FILE, written after every --checkpoint-every K samples per pixel (TRACER_CHECKPOINT_EVERY;
//    default: once, at the end of the run), and a run that finds FILE goes on where it stops, with the
//    seed it holds; --stop-after S (TRACER_STOP_AFTER) ends this run once S samples per pixel are done;
//  * several GPUs: a frame rendered in one go is spread over all devices that are visible, by interleaved image
//    rows; --devices N (TRACER_DEVICES) uses the first N, --split sample (TRACER_SPLIT) shares out the samples
//    instead and sums the devices' images, --reduce rccl (TRACER_REDUCE) does that sum with RCCL;
//  * --format p6 (TRACER_OUTPUT_FORMAT, or file_type = "p6" in [output]) writes binary PPM instead of the
//    reference's ASCII P3 (ppm.rs:6-27), which stays the default.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>

#include "config.hpp"
#include "scene.hpp"

namespace {
// --name VALUE on the command line, else the environment variable, else ""
std::string option(int argc, char** argv, const char* flag, const char* env) {
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], flag) == 0) return argv[i + 1];
    const char* e = std::getenv(env);
    return e ? e : "";
}
}  // namespace

int main(int argc, char** argv) {
    using clock = std::chrono::steady_clock;
    auto elapsed = [](clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); };
    try {
        rt::Checkpointing ckpt;
        ckpt.path = option(argc, argv, "--checkpoint", "TRACER_CHECKPOINT");
        ckpt.every = (size_t)std::strtoull(option(argc, argv, "--checkpoint-every", "TRACER_CHECKPOINT_EVERY").c_str(), nullptr, 0);
        ckpt.stop_after = (size_t)std::strtoull(option(argc, argv, "--stop-after", "TRACER_STOP_AFTER").c_str(), nullptr, 0);
        std::string format = option(argc, argv, "--format", "TRACER_OUTPUT_FORMAT");
        const std::string split = option(argc, argv, "--split", "TRACER_SPLIT"), reduce = option(argc, argv, "--reduce", "TRACER_REDUCE");
        if (!split.empty() && split != "tile" && split != "sample") throw std::runtime_error("unknown split " + split + " (tile, sample)");
        if (!reduce.empty() && reduce != "peer" && reduce != "rccl") throw std::runtime_error("unknown reduce " + reduce + " (peer, rccl)");
        rt::multi_device().devices = (int)std::strtol(option(argc, argv, "--devices", "TRACER_DEVICES").c_str(), nullptr, 0);
        rt::multi_device().sample_split = split == "sample";
        rt::multi_device().rccl = reduce == "rccl";

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

        if (format.empty()) format = cfg.output.file_type;
        const bool binary = format == "p6" || format == "P6";
        if (!binary && !format.empty() && format != "p3" && format != "P3" && format != "ppm")
            throw std::runtime_error("unknown output format " + format + " (p3, p6)");

        uint64_t seed, resumed_seed = 0;
        if (const char* s = std::getenv("TRACER_SEED")) {
            seed = std::strtoull(s, nullptr, 0);
        } else if (!ckpt.path.empty() && rt::checkpoint_seed(ckpt.path, &resumed_seed)) {
            seed = resumed_seed ^ 0x5EED0002ull;  // the run that wrote the checkpoint drew it
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
        std::vector<uint8_t> buffer;
        if (ckpt.path.empty() && ckpt.every == 0 && ckpt.stop_after == 0) {
            buffer = rt::render_world(background, world, camera, screen_width, screen_height, cfg.renderer.antialias_iterations,
                                      cfg.renderer.render_parallel, cfg.renderer.use_bounding_volume, seed ^ 0x5EED0002ull, &stats);
        } else {
            size_t done = 0;
            bool finished = false;
            buffer = rt::render_world_resumable(background, world, camera, screen_width, screen_height, cfg.renderer.antialias_iterations,
                                                cfg.renderer.use_bounding_volume, seed ^ 0x5EED0002ull, ckpt, &stats, &done, &finished);
            std::printf("  %zu of %zu samples per pixel done%s\n", done, (size_t)cfg.renderer.antialias_iterations,
                        finished ? "" : " (unfinished: the image is the mean of these; run again to go on)");
        }
        const double tracing = elapsed(tracing_start);
        std::printf("End tracing-- took %.2fs\n", tracing);
        std::printf("  %.3f Msamples/s, %.3f Mrays/s (%llu samples, %llu ray segments)\n", stats.samples / tracing / 1e6,
                    stats.segments / tracing / 1e6, (unsigned long long)stats.samples, (unsigned long long)stats.segments);

        auto ppm_start = clock::now();
        std::printf("Start ppm creation\n");
        if (!(binary ? rt::ppm_create_binary(cfg.output.output_path, screen_width, screen_height, buffer)
                     : rt::ppm_create(cfg.output.output_path, screen_width, screen_height, buffer))) {
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
