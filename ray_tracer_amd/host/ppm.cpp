// ppm.cpp -- ppm::create (reference src/ppm.rs:6-27): ASCII P3, header
// "P3\n<w> <h>\n255\n", then one "r g b\n" line per pixel in buffer order.
#include <cstdio>
#include <string>

#include "scene.hpp"

namespace rt {

bool ppm_create(const std::string& output_path, size_t image_x, size_t image_y, const std::vector<uint8_t>& bytes) {
    std::string buffer;
    buffer.reserve(bytes.size() * 4 + 32);
    char line[32];
    // ppm.rs:8: chunks(3).chunks(image_x): whole pixels only, rows of image_x pixels
    const size_t n_pixels = bytes.size() / 3;
    for (size_t p = 0; p < n_pixels; ++p) {
        const int len = std::snprintf(line, sizeof line, "%u %u %u\n", (unsigned)bytes[3 * p], (unsigned)bytes[3 * p + 1],
                                      (unsigned)bytes[3 * p + 2]);
        buffer.append(line, (size_t)len);
    }
    std::FILE* f = std::fopen(output_path.c_str(), "wb");
    if (!f) return false;
    const std::string header = "P3\n" + std::to_string(image_x) + " " + std::to_string(image_y) + "\n255\n";
    bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size();
    ok = ok && std::fwrite(buffer.data(), 1, buffer.size(), f) == buffer.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

// Binary PPM (P6) of the same pixels: the header with P6, then the bytes as they are.
bool ppm_create_binary(const std::string& output_path, size_t image_x, size_t image_y, const std::vector<uint8_t>& bytes) {
    std::FILE* f = std::fopen(output_path.c_str(), "wb");
    if (!f) return false;
    const std::string header = "P6\n" + std::to_string(image_x) + " " + std::to_string(image_y) + "\n255\n";
    const size_t n = bytes.size() / 3 * 3;
    bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size();
    ok = ok && std::fwrite(bytes.data(), 1, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

}  // namespace rt
