// config.cpp -- reader for the TOML subset that tracer.toml files use: `[section]`
// headers, `key = value` lines with integer / float / boolean / "string" values
// and one-line inline tables `{ x = 1.0, y = 2.0, z = 3.0 }`, `#` comments.
// Mirrors the structure and defaults of the reference's serde structs
// (src/config.rs:7-208) and read_tracer_config (src/config.rs:210-217).
#include "config.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace rt {
namespace {

struct Value {
    enum Kind { Number, Bool, String, Table } kind = Number;
    double number = 0.0;
    bool is_integer = false;
    bool boolean = false;
    std::string string;
    std::map<std::string, Value> table;
};

using Section = std::map<std::string, Value>;

[[noreturn]] void bad(const std::string& what, size_t line) {
    throw std::runtime_error("tracer config: " + what + " (line " + std::to_string(line) + ")");
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// Removes a trailing comment that is not inside a string.
std::string strip_comment(const std::string& s) {
    bool in_string = false;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '"' && (i == 0 || s[i - 1] != '\\')) in_string = !in_string;
        if (s[i] == '#' && !in_string) return s.substr(0, i);
    }
    return s;
}

Value parse_value(const std::string& text, size_t line);

// Splits "a = 1, b = 2" at top-level commas.
std::vector<std::string> split_top_level(const std::string& s, size_t line) {
    std::vector<std::string> parts;
    int depth = 0;
    bool in_string = false;
    std::string cur;
    for (char c : s) {
        if (c == '"') in_string = !in_string;
        if (!in_string && (c == '{' || c == '[')) ++depth;
        if (!in_string && (c == '}' || c == ']')) --depth;
        if (depth < 0) bad("unbalanced brackets", line);
        if (c == ',' && depth == 0 && !in_string) {
            parts.push_back(cur);
            cur.clear();
        } else {
            cur.push_back(c);
        }
    }
    if (!trim(cur).empty()) parts.push_back(cur);
    return parts;
}

void parse_key_value(const std::string& text, size_t line, std::string* key, Value* value) {
    const size_t eq = text.find('=');
    if (eq == std::string::npos) bad("expected key = value", line);
    *key = trim(text.substr(0, eq));
    if (key->empty()) bad("empty key", line);
    if (key->size() >= 2 && key->front() == '"' && key->back() == '"') *key = key->substr(1, key->size() - 2);
    *value = parse_value(trim(text.substr(eq + 1)), line);
}

Value parse_value(const std::string& text, size_t line) {
    Value v;
    if (text.empty()) bad("missing value", line);
    if (text == "true" || text == "false") {
        v.kind = Value::Bool;
        v.boolean = text == "true";
        return v;
    }
    if (text.front() == '"') {
        if (text.size() < 2 || text.back() != '"') bad("unterminated string", line);
        v.kind = Value::String;
        v.string = text.substr(1, text.size() - 2);
        return v;
    }
    if (text.front() == '{') {
        if (text.back() != '}') bad("inline table must end on the same line", line);
        v.kind = Value::Table;
        for (const std::string& part : split_top_level(text.substr(1, text.size() - 2), line)) {
            std::string k;
            Value inner;
            parse_key_value(trim(part), line, &k, &inner);
            v.table[k] = inner;
        }
        return v;
    }
    std::string digits;
    for (char c : text)
        if (c != '_') digits.push_back(c);
    char* end = nullptr;
    v.number = std::strtod(digits.c_str(), &end);
    if (end == digits.c_str() || *end != '\0') bad("cannot parse value '" + text + "'", line);
    v.kind = Value::Number;
    v.is_integer = digits.find_first_of(".eEinfa") == std::string::npos;
    return v;
}

double get_f64(const Section& s, const char* key, double fallback) {
    auto it = s.find(key);
    if (it == s.end()) return fallback;
    // serde's f64 accepts TOML integers as well as floats
    if (it->second.kind != Value::Number) throw std::runtime_error(std::string("tracer config: ") + key + " must be a number");
    return it->second.number;
}

size_t get_usize(const Section& s, const char* key, size_t fallback) {
    auto it = s.find(key);
    if (it == s.end()) return fallback;
    if (it->second.kind != Value::Number || !it->second.is_integer || it->second.number < 0)
        throw std::runtime_error(std::string("tracer config: ") + key + " must be a non-negative integer");
    return (size_t)it->second.number;
}

bool get_bool(const Section& s, const char* key, bool fallback) {
    auto it = s.find(key);
    if (it == s.end()) return fallback;
    if (it->second.kind != Value::Bool) throw std::runtime_error(std::string("tracer config: ") + key + " must be a boolean");
    return it->second.boolean;
}

std::string get_string(const Section& s, const char* key, const std::string& fallback) {
    auto it = s.find(key);
    if (it == s.end()) return fallback;
    if (it->second.kind != Value::String) throw std::runtime_error(std::string("tracer config: ") + key + " must be a string");
    return it->second.string;
}

// Vec3's Deserialize (vec3.rs:101-144): a table with exactly x, y, z; an unknown
// key or a missing one is an error.
Vec3 get_vec3(const Section& s, const char* key, const Vec3& fallback) {
    auto it = s.find(key);
    if (it == s.end()) return fallback;
    if (it->second.kind != Value::Table) throw std::runtime_error(std::string("tracer config: ") + key + " must be a table {x, y, z}");
    Vec3 v;
    bool have[3] = {false, false, false};
    for (const auto& kv : it->second.table) {
        if (kv.second.kind != Value::Number) throw std::runtime_error(std::string("tracer config: ") + key + " components must be numbers");
        if (kv.first == "x") { v.x = kv.second.number; have[0] = true; }
        else if (kv.first == "y") { v.y = kv.second.number; have[1] = true; }
        else if (kv.first == "z") { v.z = kv.second.number; have[2] = true; }
        else throw std::runtime_error(std::string("tracer config: unknown field `") + kv.first + "` in " + key + ", expected x, y, z");
    }
    if (!have[0] || !have[1] || !have[2]) throw std::runtime_error(std::string("tracer config: ") + key + " is missing x, y or z");
    return v;
}

}  // namespace

TracerConfig parse_tracer_config(const std::string& text) {
    std::map<std::string, Section> sections;
    std::string current;
    std::istringstream in(text);
    std::string raw;
    size_t line_no = 0;
    while (std::getline(in, raw)) {
        ++line_no;
        const std::string line = trim(strip_comment(raw));
        if (line.empty()) continue;
        if (line.front() == '[') {
            if (line.back() != ']') bad("malformed section header", line_no);
            current = trim(line.substr(1, line.size() - 2));
            if (current.empty()) bad("empty section name", line_no);
            if (sections.count(current)) bad("section [" + current + "] defined twice", line_no);
            sections[current];
            continue;
        }
        std::string key;
        Value value;
        parse_key_value(line, line_no, &key, &value);
        if (sections[current].count(key)) bad("key " + key + " defined twice", line_no);
        sections[current][key] = value;
    }

    // TracerConfig has no #[serde(default)] on its fields: all four sections are required
    for (const char* required : {"renderer", "camera", "output", "world"})
        if (!sections.count(required)) throw std::runtime_error(std::string("tracer config: missing field `") + required + "`");

    TracerConfig cfg;
    const Section& r = sections["renderer"];
    cfg.renderer.screen_width = get_usize(r, "screen_width", 200);
    cfg.renderer.screen_height = get_usize(r, "screen_height", 100);
    cfg.renderer.antialias_iterations = get_usize(r, "antialias_iterations", 100);
    cfg.renderer.render_parallel = get_bool(r, "render_parallel", false);
    cfg.renderer.use_bounding_volume = get_bool(r, "use_bounding_volume", false);

    const Section& c = sections["camera"];
    cfg.camera.fov = get_f64(c, "fov", 15.0);
    cfg.camera.aperture = get_f64(c, "aperture", 0.2);
    cfg.camera.look_from = get_vec3(c, "look_from", Vec3());
    cfg.camera.look_at = get_vec3(c, "look_at", Vec3());
    cfg.camera.time_start = get_f64(c, "time_start", 0.0);
    cfg.camera.time_end = get_f64(c, "time_end", 1.0);

    const Section& o = sections["output"];
    cfg.output.output_path = get_string(o, "output_path", "./resources/output.ppm");
    cfg.output.file_name = get_string(o, "file_name", "");
    cfg.output.file_type = get_string(o, "file_type", "");
    cfg.output.render_window = get_bool(o, "render_window", false);

    // WorldConfig is internally tagged by `scene` (config.rs:182-195)
    const Section& w = sections["world"];
    auto scene = w.find("scene");
    if (scene == w.end() || scene->second.kind != Value::String) throw std::runtime_error("tracer config: missing field `scene`");
    if (!world_kind_from_name(scene->second.string, &cfg.world.scene))
        throw std::runtime_error("tracer config: unknown variant `" + scene->second.string + "`");
    cfg.world.max_objects = get_usize(w, "max_objects", 10);
    cfg.world.earth_path = get_string(w, "earth_path", cfg.world.earth_path);
    return cfg;
}

TracerConfig read_tracer_config(const std::string& input_path) {
    std::ifstream in(input_path);
    if (!in) throw std::runtime_error("tracer config: cannot open " + input_path);
    std::stringstream ss;
    ss << in.rdbuf();
    return parse_tracer_config(ss.str());
}

}  // namespace rt
