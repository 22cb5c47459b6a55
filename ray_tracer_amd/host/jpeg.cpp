// jpeg.cpp -- a baseline JPEG reader for the image texture (reference src/texture.rs:169-181:
// Texture::image(path) opens the file with the `image` crate and samples it with get_pixel).
// The reference's earth texture is resources/earthmap.jpg (src/world.rs:201), a baseline JFIF
// file; the host reads it with this decoder so that the scene file the reference ships works
// as it is.
//
// Covers what baseline JFIF files hold: 8-bit samples, Huffman coding, one or three
// components (grey, YCbCr) with sampling factors 1 or 2, restart intervals.  Not covered, and
// refused: progressive and arithmetic-coded files, 12-bit samples, CMYK.  Chroma is
// upsampled by replication; the inverse DCT is computed in double precision and rounded once,
// so a texel may differ by a level from what an integer IDCT (libjpeg, the `image` crate's
// jpeg-decoder) gives: texel values are not pinned by any test of the reference.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "scene.hpp"

namespace rt {
namespace {

struct Huffman {
    // canonical code tables as the standard's decoding procedure wants them (ITU T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    bool present = false;
    void build(const uint8_t counts[16], const uint8_t* vals, int n) {
        std::memcpy(values, vals, (size_t)n);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            k += counts[len - 1];
            code += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    int bit() {
        if (bits == 0) {
            uint8_t b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;  // stuffed zero
                    else {                          // a marker: the entropy-coded segment ends here
                        hit_marker = true;
                        --p;
                        b = 0;
                    }
                }
            }
            acc = b;
            bits = 8;
        }
        --bits;
        return (int)((acc >> bits) & 1u);
    }
    int receive(int n) {
        int v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() {
        acc = 0;
        bits = 0;
        hit_marker = false;
    }
};

int decode_symbol(BitReader& br, const Huffman& h) {
    int code = br.bit();
    for (int len = 1; len <= 16; ++len) {
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
        code = (code << 1) | br.bit();
    }
    return -1;
}

int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

constexpr size_t MAX_PIXELS = (size_t)1 << 28;  // no frame header is believed beyond this
const int ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// 8x8 inverse DCT of dequantised coefficients (row-major), level shift, clamp to 0..255.
void idct_block(const double in[64], uint8_t out[64]) {
    static double basis[8][8];
    static bool ready = false;
    if (!ready) {
        const double PI = 3.14159265358979323846;
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) basis[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * PI / 16.0);
        ready = true;
    }
    double tmp[64];
    for (int v = 0; v < 8; ++v)
        for (int x = 0; x < 8; ++x) {
            double s = 0.0;
            for (int u = 0; u < 8; ++u) s += basis[x][u] * in[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            double s = 0.0;
            for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[v * 8 + x];
            const double level = std::floor(s + 128.0 + 0.5);
            out[y * 8 + x] = (uint8_t)(level < 0.0 ? 0.0 : level > 255.0 ? 255.0 : level);
        }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // in 8x8 blocks, padded to whole MCUs
    std::vector<uint8_t> plane;      // blocks_w * 8 by blocks_h * 8
    int pred = 0;
};

uint8_t clamp_u8(double x) {
    const double r = std::floor(x + 0.5);
    return (uint8_t)(r < 0.0 ? 0.0 : r > 255.0 ? 255.0 : r);
}

}  // namespace

bool read_jpeg_baseline(const std::string& path, Image* out, std::string* why) {
    auto fail = [&](const char* msg) {
        if (why) *why = msg;
        return false;
    };
    std::ifstream in(path, std::ios::binary);
    if (!in) return fail("cannot open the file");
    const std::vector<uint8_t> d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG file");
    double quant[4][64] = {};
    bool have_quant[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, restart_interval = 0;
    size_t i = 2;
    while (i + 4 <= d.size()) {
        if (d[i] != 0xFF) return fail("marker expected");
        const int marker = d[i + 1];
        if (marker == 0xFF) {  // fill byte
            ++i;
            continue;
        }
        if (marker == 0xD9) break;
        const size_t len = ((size_t)d[i + 2] << 8) | d[i + 3];
        if (len < 2 || i + 2 + len > d.size()) return fail("truncated segment");
        const uint8_t* seg = d.data() + i + 4;
        const size_t n = len - 2;
        if (marker == 0xDB) {  // quantisation tables
            size_t k = 0;
            while (k < n) {
                const int precision = seg[k] >> 4, id = seg[k] & 15;
                if (id > 3) return fail("bad quantisation table id");
                ++k;
                if (k + (precision ? 128u : 64u) > n) return fail("truncated quantisation table");
                for (int z = 0; z < 64; ++z) {
                    quant[id][ZIGZAG[z]] = precision ? (double)((seg[k] << 8) | seg[k + 1]) : (double)seg[k];
                    k += precision ? 2 : 1;
                }
                have_quant[id] = true;
            }
        } else if (marker == 0xC4) {  // Huffman tables
            size_t k = 0;
            while (k + 17 <= n) {
                const int cls = seg[k] >> 4, id = seg[k] & 15;
                if (id > 3 || cls > 1) return fail("bad Huffman table id");
                int total = 0;
                for (int l = 0; l < 16; ++l) total += seg[k + 1 + l];
                if (total > 256 || k + 17 + (size_t)total > n) return fail("truncated Huffman table");
                (cls ? ac[id] : dc[id]).build(seg + k + 1, seg + k + 17, total);
                k += 17 + (size_t)total;
            }
        } else if (marker == 0xC0 || marker == 0xC1) {  // baseline / extended sequential, Huffman
            if (n < 6 || seg[0] != 8) return fail("only 8-bit samples are read");
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            const int nc = seg[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc || width <= 0 || height <= 0) return fail("one or three components expected");
            // (a header can claim 65535 x 65535: the planes are sized from it before a single block is decoded)
            if ((size_t)width * (size_t)height > MAX_PIXELS) return fail("more than 2^28 pixels");
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4;
                comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c] & 3;
                if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2) return fail("sampling factors other than 1 and 2");
            }
            // a scan of ONE component is not interleaved: one block per MCU whatever its sampling factors say (T.81 A.2.2)
            if (nc == 1) comps[0].h = comps[0].v = 1;
        } else if (marker == 0xC2 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            return fail("progressive, lossless and arithmetic-coded JPEG files are not read (baseline only)");
        } else if (marker == 0xDD) {
            if (n < 2) return fail("truncated restart interval");
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {  // start of scan: the entropy-coded data follows
            if (comps.empty()) return fail("scan before frame header");
            if (n < 1) return fail("truncated scan header");
            const int ns = seg[0];
            if (ns != (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail("one interleaved scan expected");
            for (int s = 0; s < ns; ++s) {
                Component* comp = nullptr;
                for (auto& c : comps)
                    if (c.id == seg[1 + 2 * s]) comp = &c;
                if (!comp) return fail("scan names an unknown component");
                comp->td = seg[2 + 2 * s] >> 4;
                comp->ta = seg[2 + 2 * s] & 15;
                if (comp->td > 3 || comp->ta > 3 || !dc[comp->td].present || !ac[comp->ta].present || !have_quant[comp->tq])
                    return fail("scan uses a table the file does not define");
            }
            int hmax = 1, vmax = 1;
            for (auto& c : comps) {
                hmax = c.h > hmax ? c.h : hmax;
                vmax = c.v > vmax ? c.v : vmax;
            }
            const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
                c.pred = 0;
            }
            BitReader br{d.data() + i + 2 + len, d.data() + d.size()};
            int until_restart = restart_interval;
            for (int my = 0; my < mcus_y; ++my)
                for (int mx = 0; mx < mcus_x; ++mx) {
                    if (restart_interval && until_restart == 0) {
                        // RSTn: byte-align, skip the marker, reset the predictors
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 >= br.end) return fail("restart marker missing");
                        br.p += 2;
                        for (auto& c : comps) c.pred = 0;
                        until_restart = restart_interval;
                    }
                    for (auto& c : comps)
                        for (int by = 0; by < c.v; ++by)
                            for (int bx = 0; bx < c.h; ++bx) {
                                double coef[64] = {};
                                const int t = decode_symbol(br, dc[c.td]);
                                if (t < 0 || t > 11) return fail("bad DC code");
                                c.pred += extend(br.receive(t), t);
                                coef[0] = c.pred * quant[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(br, ac[c.ta]);
                                    if (rs < 0) return fail("bad AC code");
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) {
                                        if (r != 15) break;  // end of block
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) return fail("AC run past the block");
                                    coef[ZIGZAG[k]] = extend(br.receive(s), s) * quant[c.tq][ZIGZAG[k]];
                                    ++k;
                                }
                                uint8_t px[64];
                                idct_block(coef, px);
                                const int x0 = (mx * c.h + bx) * 8, y0 = (my * c.v + by) * 8, stride = c.blocks_w * 8;
                                for (int y = 0; y < 8; ++y) std::memcpy(&c.plane[(size_t)(y0 + y) * stride + x0], px + 8 * y, 8);
                            }
                    if (restart_interval) --until_restart;
                }
            // ---- planes to RGB (JFIF: Y Cb Cr with full-range BT.601), chroma replicated to full size
            out->width = (uint32_t)width;
            out->height = (uint32_t)height;
            out->rgb.resize((size_t)width * height * 3);
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) {
                    double s[3] = {0.0, 128.0, 128.0};
                    for (size_t c = 0; c < comps.size(); ++c) {
                        const Component& k = comps[c];
                        s[c] = k.plane[(size_t)(y * k.v / vmax) * (k.blocks_w * 8) + (size_t)(x * k.h / hmax)];
                    }
                    uint8_t* p = &out->rgb[((size_t)y * width + x) * 3];
                    if (comps.size() == 1) {
                        p[0] = p[1] = p[2] = (uint8_t)s[0];
                    } else {
                        p[0] = clamp_u8(s[0] + 1.402 * (s[2] - 128.0));
                        p[1] = clamp_u8(s[0] - 0.344136 * (s[1] - 128.0) - 0.714136 * (s[2] - 128.0));
                        p[2] = clamp_u8(s[0] + 1.772 * (s[1] - 128.0));
                    }
                }
            return true;
        }
        i += 2 + len;
    }
    return fail("no image data");
}

}  // namespace rt
