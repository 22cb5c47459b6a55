// Development tool: mutation fuzzer for the host's readers of files it did not write -- the JPEG / PPM reader
// (host/jpeg.cpp, host/ppm.cpp: world.rs:201 opens resources/earthmap.jpg) and the tracer.toml reader
// (host/config.cpp) -- built with the host sanitizers (no GPU code involved):
//   cd ray_tracer_amd/host && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -I. -I../../include \
//       -o /tmp/host_fuzz ../../tools/host_fuzz.cpp scene.cpp world.cpp ppm.cpp jpeg.cpp config.cpp
//   mkdir -p /tmp/hfuzz && /tmp/host_fuzz img <seed.jpg|seed.ppm> 60000 1 ; /tmp/host_fuzz toml ../../resources/tracer.toml 60000 2
// Each input is the seed file with 1-8 random damages (bytes, bits, truncation, cuts, runs of 00 / FF, copies).
// A finding is a sanitizer report or a decoded image whose size does not match its header; refusals are fine.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "config.hpp"
#include "scene.hpp"
static std::vector<unsigned char> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), {}); }
static void spit(const char* p, const std::vector<unsigned char>& v) { std::ofstream f(p, std::ios::binary); f.write((const char*)v.data(), v.size()); }
int main(int argc, char** argv) {
    const char* kind = argv[1];
    std::vector<unsigned char> seed = slurp(argv[2]);
    long iters = atol(argv[3]);
    std::mt19937_64 rng(atol(argv[4]));
    const std::string tmp = std::string("/tmp/hfuzz/in_") + argv[4] + (strcmp(kind, "toml") ? ".img" : ".toml");
    long ok = 0, bad = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<unsigned char> v = seed;
        int n_mut = 1 + rng() % 8;
        for (int m = 0; m < n_mut && !v.empty(); ++m) {
            switch (rng() % 6) {
                case 0: v[rng() % v.size()] = (unsigned char)rng(); break;
                case 1: v[rng() % v.size()] ^= 1u << (rng() % 8); break;
                case 2: v.resize(rng() % (v.size() + 1)); break;  // truncate
                case 3: { size_t a = rng() % v.size(), len = rng() % 64; v.erase(v.begin() + a, v.begin() + std::min(v.size(), a + len)); break; }
                case 4: { size_t a = rng() % v.size(); unsigned char c = (rng() % 2) ? 0xFF : 0x00; for (size_t k = a; k < v.size() && k < a + rng() % 16; ++k) v[k] = c; break; }
                case 5: { size_t a = rng() % v.size(), b = rng() % v.size(), len = rng() % 128; for (size_t k = 0; k < len && a + k < v.size() && b + k < v.size(); ++k) v[a + k] = v[b + k]; break; }
            }
        }
        spit(tmp.c_str(), v);
        try {
            if (!strcmp(kind, "toml")) {
                rt::TracerConfig c = rt::read_tracer_config(tmp);
                (void)c;
                ++ok;
            } else {
                rt::Image img; std::string why;
                if (rt::read_image(tmp, &img, &why)) { ++ok; if (img.rgb.size() != (size_t)img.width * img.height * 3) { printf("SIZE MISMATCH\n"); return 2; } }
                else ++bad;
            }
        } catch (const std::exception&) { ++bad; }
    }
    printf("%s: %ld inputs, %ld accepted, %ld refused\n", kind, iters, ok, bad);
    return 0;
}
