// config.hpp -- the subset of TOML the reference's tracer.toml uses, read into
// the same four sections with the same defaults (reference src/config.rs:7-217).
#pragma once

#include <string>

#include "scene.hpp"

namespace rt {

struct RendererConfig {                 // config.rs:105-154
    size_t screen_width = 200;
    size_t screen_height = 100;
    size_t antialias_iterations = 100;
    bool render_parallel = false;
    bool use_bounding_volume = false;
};

struct CameraConfig {                   // config.rs:33-103
    double fov = 15.0;
    double aperture = 0.2;
    Vec3 look_from;
    Vec3 look_at;
    double time_start = 0.0;
    double time_end = 1.0;
};

struct OutputConfig {                   // config.rs:156-180
    std::string output_path = "./resources/output.ppm";
    std::string file_name;
    std::string file_type;
    bool render_window = false;
};

struct TracerConfig {                   // config.rs:7-13
    RendererConfig renderer;
    CameraConfig camera;
    OutputConfig output;
    WorldConfig world;
};

// read_tracer_config (config.rs:210-217).  Throws std::runtime_error where the
// reference panics: unreadable file, malformed input, a missing section
// ([renderer], [camera], [output] and [world] are all required), a missing
// `scene` key, an unknown scene name or an unknown key inside a Vec3 table.
// Unknown keys elsewhere are ignored, like serde does without deny_unknown_fields.
TracerConfig read_tracer_config(const std::string& input_path);
TracerConfig parse_tracer_config(const std::string& text);

}  // namespace rt
