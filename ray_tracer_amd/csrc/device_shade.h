// device_shade.h -- camera rays, hit geometry, textures, Perlin noise and material
// scattering as __device__ f64 code for gfx950, in the reference's operation order:
//   Camera::ray, random_in_unit_disk     src/camera.rs:65-88
//   Hit::new, sphere_uv                  src/hittable.rs:27-52, 732-749
//   Lambertian/Metal/Dielectric scatter  src/material.rs:51-131
//   DiffuseLight                         src/material.rs:144-152
//   random_in_unit_sphere, reflect,
//   refract, schlick                     src/material.rs:211-243
//   Texture::value (4 kinds)             src/texture.rs:24-28, 45-54, 82-87, 106-122
//   Perlin::noise / turbulence           src/perlin.rs:38-86, 100-122
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mirt_rng.h"
#include "checker_sin.h"
#include "device_intersect.h"

namespace mirt {

// Draws of one stage of one camera path (include/mirt_rng.h): draw k of stage s is
// mix64(stream + GOLDEN * ((s << 32 | k) + 1)).  The argument is kept as a running sum, so a
// draw costs the two 64-bit multiplications of the mixer and an addition.
struct PathRng {
    uint64_t stream;
    uint64_t arg;  // stream + GOLDEN * ((stage << 32 | k) + 1) for the next draw
    __device__ __forceinline__ PathRng(uint64_t stream_, uint32_t stage, uint32_t k)
        : stream(stream_), arg(stream_ + MIRT_GOLDEN * ((((uint64_t)stage << 32) | (uint64_t)k) + 1u)) {}
    // a value of [1, 2): random_double() + 1.0, exactly
    __device__ __forceinline__ double random_1_2() {
        const double x = mirt_bits_to_1_2(mirt_mix64(arg));
        arg += MIRT_GOLDEN;
        return x;
    }
    __device__ __forceinline__ double random_double() { return random_1_2() - 1.0; }  // util.rs:8-10
    // random_double() * 2.0 - 1.0 (camera.rs:81, material.rs:214-216 after vec3 arithmetic): with
    // x = random_double() + 1.0 that is 2x - 3, and x - 1, 2(x - 1) and 2(x - 1) - 1 are all
    // exact in f64 (multiples of 2^-51 below 2 in magnitude), so one fused operation gives the
    // same bits as the reference's three roundings
    __device__ __forceinline__ double random_pm1() { return __builtin_fma(random_1_2(), 2.0, -3.0); }
};

// camera.rs:78-88
__device__ __forceinline__ void random_in_unit_disk(PathRng& rng, double* x, double* y) {
    double px, py;
    do {
        px = rng.random_pm1();
        py = rng.random_pm1();
    } while (px * px + py * py + 0.0 * 0.0 >= 1.0);
    *x = px;
    *y = py;
}

// renderer.rs:74-80 + camera.rs:65-76 for pixel idx, with the sampler at stage 0.
__device__ __forceinline__ RayD camera_ray(const mirt_camera& c, uint32_t width, uint32_t height, uint32_t idx,
                                           PathRng& rng) {
    const uint32_t i = idx % width;
    const uint32_t j = height - 1 - idx / width;
    const double s = ((double)i + rng.random_double()) / (double)width;
    const double t = ((double)j + rng.random_double()) / (double)height;
    // (the shutter time is draw 2 of the stage, the lens pairs follow it: include/mirt_rng.h)
    const double shutter = rng.random_double();
    // A pinhole (lens_radius = 0) multiplies whatever the rejection loop of camera.rs:78-88 returns by zero:
    // offset = +-0, origin + offset = origin, direction - offset = direction.  The loop is left out then
    // (three trips of two draws for the slowest lane of a wave: 3-4 % of a frame); its draws are nobody else's.
    double rx = 0.0, ry = 0.0;
    if (c.lens_radius != 0.0) random_in_unit_disk(rng, &rx, &ry);
    // rd = lens_radius * p  (f64 * Vec3 -> component * lens_radius)
    rx = rx * c.lens_radius;
    ry = ry * c.lens_radius;
    const V3 cu = v3(c.u[0], c.u[1], c.u[2]), cv = v3(c.v[0], c.v[1], c.v[2]);
    const V3 origin = v3(c.origin[0], c.origin[1], c.origin[2]);
    const V3 llc = v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    const V3 horizontal = v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    const V3 vertical = v3(c.vertical[0], c.vertical[1], c.vertical[2]);
    const V3 offset = cu * rx + cv * ry;
    const V3 direction = llc + s * horizontal + t * vertical - origin - offset;
    const double time = c.time_start + shutter * (c.time_end - c.time_start);
    return RayD{origin + offset, direction, time};
}

// ------------------------------------------------------------------ hit geometry
struct Surface {
    V3 p;       // ray.point_at(t)
    V3 normal;  // flipped against the ray, hittable.rs:36-41
    bool front_facing;
    // what sphere_uv / the rect u,v need, computed on demand by surface_uv()
    V3 outward_normal;
    uint32_t kind;  // TraverseKind of the surface (a cube's side: that of its rect)
    double ua, ub, a0, a1, b0, b1;  // rects: in-plane hit coordinates and extents
};

// Rebuilds what Hit::new stored for the closest hit {t, side} of ray r on a shape given by
// its TraverseKind and traversal parameters (device_scene.h; global memory or LDS).
__device__ __forceinline__ Surface surface_at(const RayD& r, double t, uint32_t tk, const double* __restrict__ f, int side) {
    Surface h;
    h.p = r.o + t * r.d;  // ray.rs:31-33
    h.ua = h.ub = 0.0;
    uint32_t kind = tk;
    double a0 = f[0], a1 = f[1], b0 = f[2], b1 = f[3];
    if (tk == TK_CUBE) {
        // side -> the rect Cube::new built for it (hittable.rs:405-412)
        const double p0x = f[0], p0y = f[1], p0z = f[2], p1x = f[3], p1y = f[4], p1z = f[5];
        if (side < 2) {
            kind = TK_XY_RECT; a0 = p0x; a1 = p1x; b0 = p0y; b1 = p1y;
        } else if (side < 4) {
            kind = TK_XZ_RECT; a0 = p0x; a1 = p1x; b0 = p0z; b1 = p1z;
        } else {
            kind = TK_YZ_RECT; a0 = p0y; a1 = p1y; b0 = p0z; b1 = p1z;
        }
    }
    switch (kind) {
        case TK_SPHERE: {
            // hittable.rs:130 and 514; the centre as tshape_hit computes it (Moving::center with
            // the division by a power of two as a multiplication; a static sphere moves by zero)
            const double frac = (r.time - f[7]) * f[8];
            const V3 c = v3(f[0], f[1], f[2]) + frac * v3(f[4], f[5], f[6]);
            h.outward_normal = (h.p - c) / f[3];
            break;
        }
        case TK_MOVING_SPHERE:
            h.outward_normal = (h.p - moving_center(f, r.time)) / f[3];  // hittable.rs:514
            break;
        case TK_XY_RECT:
            h.outward_normal = v3(0.0, 0.0, 1.0);
            h.ua = r.o.x + t * r.d.x; h.ub = r.o.y + t * r.d.y;
            break;
        case TK_XZ_RECT:
            h.outward_normal = v3(0.0, 1.0, 0.0);
            h.ua = r.o.x + t * r.d.x; h.ub = r.o.z + t * r.d.z;
            break;
        default:  // TK_YZ_RECT
            h.outward_normal = v3(1.0, 0.0, 0.0);
            h.ua = r.o.y + t * r.d.y; h.ub = r.o.z + t * r.d.z;
            break;
    }
    h.kind = kind;
    h.a0 = a0; h.a1 = a1; h.b0 = b0; h.b1 = b1;
    h.front_facing = dot(r.d, h.outward_normal) < 0.0;
    h.normal = h.front_facing ? h.outward_normal : -h.outward_normal;
    return h;
}

// sphere_uv, hittable.rs:732-749.  A call: acos and atan2 of the library need some eighty vector
// registers, which a kernel that merely CAN meet an image texture should not have to own.
__device__ __noinline__ double2 sphere_uv(double nx, double ny, double nz) {
    const double PI = 3.14159265358979323846264338327950288;
    const double TAU = 6.28318530717958647692528676655900577;
    const double theta = acos(-ny);
    const double phi = atan2(-nz, nx) + PI;
    return make_double2(phi / TAU, theta / PI);
}

// Hit::u(), Hit::v(): only image textures read them, so they are computed only there.
__device__ __forceinline__ void surface_uv(const Surface& h, double* u, double* v) {
    if (h.kind == TK_SPHERE || h.kind == TK_MOVING_SPHERE) {
        const double2 uv = sphere_uv(h.outward_normal.x, h.outward_normal.y, h.outward_normal.z);
        *u = uv.x;
        *v = uv.y;
    } else {
        *u = (h.ua - h.a0) / (h.a1 - h.a0);  // hittable.rs:199-200, 263-264, 327-328
        *v = (h.ub - h.b0) / (h.b1 - h.b0);
    }
}

// ------------------------------------------------------------------------ perlin
// `x as i32` of Rust for an f64 that is already integral: saturating, NaN -> 0.
__device__ __forceinline__ int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    if (x >= 2147483647.0) return 0x7FFFFFFF;
    return (int32_t)x;
}
__device__ __forceinline__ uint32_t f64_as_u32(double x) {
    if (x != x) return 0u;
    if (x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// One 8-byte LDS read as exactly one ds_read_b64 (the compiler would pair neighbours into
// ds_read2_b64, which is many times slower when the 64 lanes read 64 different places).  The
// caller waits (lds_reads_done) before using the values.
template <int OFFSET>
__device__ __forceinline__ double lds_read_single_f64(uint32_t addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}
__device__ __forceinline__ void lds_reads_done() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// perlin.rs:52-85 + 100-122.  Tables may live in LDS (IN_LDS: the gradient vectors of four corners
// are then fetched as twelve single reads, one wait) or in global memory.
// LEAN: the gradient vectors are fetched two corners at a time instead of four (twelve registers fewer,
// twice the waits).
template <bool IN_LDS, bool LEAN = false>
__device__ __forceinline__ double perlin_noise(const double* __restrict__ vecs, const int32_t* __restrict__ perm_x,
                                               const int32_t* __restrict__ perm_y, const int32_t* __restrict__ perm_z,
                                               V3 p) {
    const double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int32_t i = f64_as_i32(fx), j = f64_as_i32(fy), k = f64_as_i32(fz);
    const double uu = u * u * (3.0 - 2.0 * u);
    const double vv = v * v * (3.0 - 2.0 * v);
    const double ww = w * w * (3.0 - 2.0 * w);
    double accumulator = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
        const int32_t px = perm_x[(int32_t)((uint32_t)i + (uint32_t)di) & 255];
        double gv[2][2][3];
        if (IN_LDS && !LEAN) {
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const int32_t py = perm_y[(int32_t)((uint32_t)j + (uint32_t)dj) & 255];
                    const int32_t pz = perm_z[(int32_t)((uint32_t)k + (uint32_t)dk) & 255];
                    const uint32_t at = (uint32_t)(size_t)vecs + 24u * (uint32_t)(px ^ py ^ pz);
                    gv[dj][dk][0] = lds_read_single_f64<0>(at);
                    gv[dj][dk][1] = lds_read_single_f64<8>(at);
                    gv[dj][dk][2] = lds_read_single_f64<16>(at);
                }
            lds_reads_done();
        }
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int32_t py = perm_y[(int32_t)((uint32_t)j + (uint32_t)dj) & 255];
            if (IN_LDS && LEAN) {
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const int32_t pz = perm_z[(int32_t)((uint32_t)k + (uint32_t)dk) & 255];
                    const uint32_t at = (uint32_t)(size_t)vecs + 24u * (uint32_t)(px ^ py ^ pz);
                    gv[dj][dk][0] = lds_read_single_f64<0>(at);
                    gv[dj][dk][1] = lds_read_single_f64<8>(at);
                    gv[dj][dk][2] = lds_read_single_f64<16>(at);
                }
                lds_reads_done();
            }
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const int32_t pz = perm_z[(int32_t)((uint32_t)k + (uint32_t)dk) & 255];
                const int32_t index = px ^ py ^ pz;
                const double* __restrict__ g = IN_LDS ? gv[dj][dk] : vecs + 3 * index;
                // weight.dot(sample): (u-i)*gx + (v-j)*gy + (w-k)*gz
                const double color = (u - (double)di) * g[0] + (v - (double)dj) * g[1] + (w - (double)dk) * g[2];
                // perlin.rs:112-114: i*uu + (1-i)*(1-uu) is exactly uu for i = 1 and exactly
                // 1-uu for i = 0 (uu is finite and >= 0 for any finite p)
                const double wi = di ? uu : (1.0 - uu);
                const double wj = dj ? vv : (1.0 - vv);
                const double wk = dk ? ww : (1.0 - ww);
                accumulator += color * wi * wj * wk;
            }
        }
    }
    return accumulator;
}

// perlin.rs:38-50
template <bool IN_LDS>
__device__ __forceinline__ double perlin_turbulence(const double* __restrict__ vecs, const int32_t* __restrict__ perm_x,
                                                    const int32_t* __restrict__ perm_y,
                                                    const int32_t* __restrict__ perm_z, V3 p, int depth) {
    V3 current_p = p;
    double weight = 1.0;
    double accumulator = 0.0;
#pragma nounroll
    for (int i = 0; i < depth; ++i) {
        accumulator += weight * perlin_noise<IN_LDS>(vecs, perm_x, perm_y, perm_z, current_p);
        weight *= 0.5;
        current_p = current_p * 2.0;
    }
    return fabs(accumulator);
}

// The same through generic pointers (LDS or global memory) and behind a call: for a kernel that meets
// turbulence one lane at a time only in rare cases (perlin_turbulence_wave serves the usual ones) and
// should not carry the inlined code's registers for them.
__device__ __noinline__ double perlin_turbulence_rare(const mirt_perlin* pl, double x, double y, double z) {
    return perlin_turbulence<false>(&pl->vecs[0][0], pl->perm_x, pl->perm_y, pl->perm_z, v3(x, y, z), 7);
}

// perlin_turbulence(p, 7) for the lanes of a wave that `want` it, computed by the WHOLE wave: the seven
// octaves of one point are independent until they are summed, so eight lanes take one point -- lane j of
// the group evaluates octave j (p * 2^j and the weight 2^-j are exact, as the repeated doubling and
// halving of perlin.rs:45-47 are), the eighth idles -- and the point's owner adds the seven terms in
// order, which is the sum the loop of perlin.rs:43-48 forms: same operations, same bits.  Eight points
// per pass; a shading trip that holds a handful of noise hits pays one pass of one octave instead of
// seven octaves run for those few lanes while the others look on.
// Must be called by all 64 lanes of the wave together.
//   want      this lane needs the turbulence at p, from table `table` of `lds_tables` (staged in LDS)
//   points    32 x 8 bytes of LDS owned by this wave (p and the table of the pass's eight points)
//   terms     64 x 8 bytes of LDS owned by this wave
constexpr int TURBULENCE_POINTS = 8;
__device__ __forceinline__ double perlin_turbulence_wave(bool want, const V3& p, uint32_t table, const mirt_perlin* lds_tables,
                                                         unsigned long long* points, double* terms) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t group = lane >> 3, octave = lane & 7u;
    double result = 0.0;
    bool pending = want;
    for (;;) {
        const unsigned long long waiting = __builtin_amdgcn_ballot_w64(pending);
        if (waiting == 0ull) break;
        const uint32_t n = min((uint32_t)__popcll(waiting), (uint32_t)TURBULENCE_POINTS);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(waiting >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)waiting, 0u));
        const bool mine = pending && rank < (uint32_t)TURBULENCE_POINTS;
        if (mine) {
            points[4 * rank + 0] = __builtin_bit_cast(unsigned long long, p.x);
            points[4 * rank + 1] = __builtin_bit_cast(unsigned long long, p.y);
            points[4 * rank + 2] = __builtin_bit_cast(unsigned long long, p.z);
            points[4 * rank + 3] = table;
        }
        if (group < n && octave < 7u) {
            const double up = __hiloint2double((int)((1023u + octave) << 20), 0);    // 2^octave
            const double down = __hiloint2double((int)((1023u - octave) << 20), 0);  // 2^-octave
            const V3 q = v3(__builtin_bit_cast(double, points[4 * group + 0]) * up, __builtin_bit_cast(double, points[4 * group + 1]) * up,
                            __builtin_bit_cast(double, points[4 * group + 2]) * up);
            const mirt_perlin* pl = lds_tables + (uint32_t)points[4 * group + 3];
            terms[lane] = down * perlin_noise<true, true>(&pl->vecs[0][0], pl->perm_x, pl->perm_y, pl->perm_z, q);
        }
        if (mine) {
            double accumulator = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) accumulator += terms[8 * rank + j];
            result = fabs(accumulator);
            pending = false;
        }
    }
    return result;
}

// ---------------------------------------------------------------------- textures
// sin() of the checker: the register-light form for every argument a scene of sane size
// produces, the library's behind a call (so that its code stays out of the kernels' bodies)
// for the rest.
__device__ __noinline__ double sin_rare(double x) { return sin(x); }
__device__ __forceinline__ double checker_sin(double x) { return fabs(x) < SIN_SMALL_LIMIT ? sin_small(x) : sin_rare(x); }
// texture.rs:47-49 evaluated in full: `sines < 0.0`
__device__ __noinline__ bool checker_odd_slow(double x, double y, double z) {
    const double sines = checker_sin(10.0 * x) * checker_sin(10.0 * y) * checker_sin(10.0 * z);
    return sines < 0.0;
}
// The same decision from the signs of the three sines where each is certain (a product of
// three sines of magnitude > 1e-13 neither vanishes nor underflows, and is negative iff an odd
// number of factors is); the full evaluation otherwise.
__device__ __forceinline__ bool checker_odd(const V3& p) {
    const int sx = sin_is_negative_fast(10.0 * p.x);
    const int sy = sin_is_negative_fast(10.0 * p.y);
    const int sz = sin_is_negative_fast(10.0 * p.z);
    if ((sx | sy | sz) < 0) return checker_odd_slow(p.x, p.y, p.z);
    return ((sx ^ sy ^ sz) & 1) != 0;
}

// texture.rs:195-205
__device__ __forceinline__ double fclamp(double x, double lo, double hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// One entry of a texture table in registers.  The table is the scene's (global memory) or a kernel's copy in
// LDS; the pointer says which at run time, and a load through such a pointer travels the long way round
// (and waits for every memory operation in flight).  Here the pointer is asked once, and the entry is read
// with the instructions of the memory it lies in.
struct TextureEntry {
    uint32_t kind, even, odd, perlin, image;
    double color[3], scale;
};
template <int OFFSET>
__device__ __forceinline__ uint2 lds_read_single_u32x2(uint32_t addr) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
static_assert(sizeof(mirt_texture) == 56 && offsetof(mirt_texture, color) == 24 && offsetof(mirt_texture, scale) == 48, "load_texture reads a mirt_texture by offsets");
__device__ __forceinline__ TextureEntry load_texture(const mirt_texture* textures, uint32_t tex) {
    TextureEntry e;
    const mirt_texture* t = textures + tex;
    if (__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)t)) {
        const uint32_t at = (uint32_t)(size_t)t;
        const uint2 a = lds_read_single_u32x2<0>(at), b = lds_read_single_u32x2<8>(at), c = lds_read_single_u32x2<16>(at);
        e.color[0] = lds_read_single_f64<24>(at);
        e.color[1] = lds_read_single_f64<32>(at);
        e.color[2] = lds_read_single_f64<40>(at);
        e.scale = lds_read_single_f64<48>(at);
        lds_reads_done();
        e.kind = a.x, e.even = a.y, e.odd = b.x, e.perlin = b.y, e.image = c.x;
    } else {
        const __attribute__((address_space(1))) mirt_texture* g = (const __attribute__((address_space(1))) mirt_texture*)t;
        e.kind = g->kind, e.even = g->even, e.odd = g->odd, e.perlin = g->perlin, e.image = g->image;
        e.color[0] = g->color[0], e.color[1] = g->color[1], e.color[2] = g->color[2];
        e.scale = g->scale;
    }
    return e;
}

// The texture kinds a kernel is compiled for: every kind up to and including LEVEL in the
// order below.  Each kind costs registers wherever textures are evaluated (measured for the
// path kernel: constant 126 VGPRs, + checker 151, everything 251), so a scene that has no
// noise or image texture runs a kernel that holds more waves per SIMD.
enum TextureLevel : int { TEXTURES_CONSTANT = 0, TEXTURES_CHECKER = 1, TEXTURES_IMAGE = 2, TEXTURES_ALL = 3, N_TEXTURE_LEVELS = 4 };

// Texture::value (texture.rs:184-193) of entry `tex` of `textures` (DScene::shade_textures, or
// its copy in LDS).  Checker nesting is followed iteratively.
// `lds_tables` = the scene's first `n_staged` Perlin tables as the shade kernel staged them
// in LDS (null / 0 when it staged none); other tables are read from global memory.
// A texture of a kind above LEVEL evaluates to black: the host picks LEVEL from the scene.
// have_turbulence: `turbulence` is perlin_turbulence(h.p, 7) from the table of texture `tex` itself, which
// is a noise texture (the caller had the wave compute it, perlin_turbulence_wave).
// NOISE_INLINE: turbulence that is not known is evaluated inline (a kernel whose waves hold one class of
// hits each); otherwise behind a call.
template <int LEVEL, bool NOISE_INLINE = true>
__device__ __forceinline__ V3 texture_value(const DScene& sc, const mirt_texture* textures, uint32_t tex, const Surface& h,
                                            const mirt_perlin* lds_tables, uint32_t n_staged, bool have_turbulence = false,
                                            double turbulence_known = 0.0) {
    for (int guard = 0; guard < 64; ++guard) {
        const TextureEntry entry = load_texture(textures, tex);
        const TextureEntry* t = &entry;
        const uint32_t kind = t->kind;
        if (kind == MIRT_TEX_CONSTANT) return v3(t->color[0], t->color[1], t->color[2]);
        if (LEVEL >= TEXTURES_CHECKER && kind == MIRT_TEX_CHECKER) {  // texture.rs:45-54
            tex = checker_odd(h.p) ? t->odd : t->even;
            continue;
        }
        if (LEVEL >= TEXTURES_ALL && kind == MIRT_TEX_NOISE) {  // texture.rs:82-87
            double turb;
            if (have_turbulence) {
                turb = turbulence_known;
            } else if (!NOISE_INLINE) {
                turb = perlin_turbulence_rare(t->perlin < n_staged ? lds_tables + t->perlin : sc.perlins + t->perlin, h.p.x, h.p.y, h.p.z);
            } else if (t->perlin < n_staged) {
                const mirt_perlin* pl = lds_tables + t->perlin;
                turb = perlin_turbulence<true>(&pl->vecs[0][0], pl->perm_x, pl->perm_y, pl->perm_z, h.p, 7);
            } else {
                const mirt_perlin* __restrict__ pl = sc.perlins + t->perlin;
                turb = perlin_turbulence<false>(&pl->vecs[0][0], pl->perm_x, pl->perm_y, pl->perm_z, h.p, 7);
            }
            have_turbulence = false;
            const double turbulence = t->scale * h.p.z + 10.0 * turb;
            const double c = (1.0 * 0.5) * (1.0 + checker_sin(turbulence));
            return v3(c, c, c);
        }
        if (LEVEL < TEXTURES_IMAGE || kind != MIRT_TEX_IMAGE) break;
        // MIRT_TEX_IMAGE, texture.rs:106-122
        double u, v;
        surface_uv(h, &u, &v);
        const mirt_image* __restrict__ im = sc.images + t->image;
        u = fclamp(u, 0.0, 1.0);
        v = 1.0 - fclamp(v, 0.0, 1.0);
        uint32_t i = f64_as_u32(u * (double)im->width);
        uint32_t j = f64_as_u32(v * (double)im->height);
        if (i > im->width - 1) i = im->width - 1;
        if (j > im->height - 1) j = im->height - 1;
        const double color_scale = 1.0 / 255.0;
        // (the pixels are global memory: said so, the three loads need not go by the generic path)
        const __attribute__((address_space(1))) uint8_t* px =
            (const __attribute__((address_space(1))) uint8_t*)(im->rgb + 3 * ((size_t)j * im->width + i));
        return v3((double)px[0] * color_scale, (double)px[1] * color_scale, (double)px[2] * color_scale);
    }
    return v3(0.0, 0.0, 0.0);
}

// --------------------------------------------------------------------- materials
// material.rs:211-225
__device__ __forceinline__ V3 random_in_unit_sphere(PathRng& rng) {
    V3 p;
    do {
        const double a = rng.random_pm1();
        const double b = rng.random_pm1();
        const double c = rng.random_pm1();
        p = v3(a, b, c);
    } while (square_length(p) >= 1.0);
    return p;
}

// The same for the 64 lanes of a wave at once.  Left to itself the loop above runs for the
// slowest lane -- 6.5 trips for 45 lanes against 1.9 for one (the acceptance probability is
// pi / 6) -- while the lanes that are done look on.  Here, once no more than half of the lanes
// are still drawing, each of them is given h = 2, 4, ... 32 lanes that try its NEXT h candidates
// side by side; it takes the first of them that lies inside the sphere, which is exactly the one
// the loop above would have stopped at (candidate k of a lane is draws 3k, 3k + 1, 3k + 2 of its
// stage, whoever evaluates it).  Must be called by all 64 lanes of the wave together.
//   want     this lane needs a sample;   rng   its sampler, at the first draw of the loop
//   scratch  IN_SPHERE_SCRATCH x 8 bytes of LDS owned by this wave
constexpr int IN_SPHERE_SCRATCH = 32;
__device__ __forceinline__ V3 random_in_unit_sphere_wave(bool want, const PathRng& rng, unsigned long long* scratch) {
    const uint32_t lane = threadIdx.x & 63u;
    V3 p = v3(0.0, 0.0, 0.0);
    bool pending = want;
    uint32_t used = 0;  // candidates every pending lane has gone through (the same for all of them)
    for (;;) {
        const unsigned long long waiting = __ballot(pending);
        const uint32_t n = (uint32_t)__popcll(waiting);
        if (n == 0) break;
        // helpers per pending lane: the largest power of two with n * h <= 64, at most 32
        const uint32_t log_h = min(5u, 31u - (uint32_t)__builtin_clz(64u / n));
        const uint32_t h = 1u << log_h;
        PathRng mine(rng.stream, 0u, 0u);
        bool valid = pending;
        uint32_t group = 0, rank = 0;
        if (log_h == 0) {
            mine.arg = rng.arg + MIRT_GOLDEN * (3ull * used);
        } else {
            // pending lane of rank r (r-th set bit of `waiting`) leaves its argument in scratch[r];
            // lane L works for the pending lane of rank L / h on its candidate used + L % h
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(waiting >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)waiting, 0u));
            if (pending) scratch[rank] = rng.arg;
            group = lane >> log_h;
            valid = group < n;
            const unsigned long long base = scratch[valid ? group : 0u];
            mine.arg = base + MIRT_GOLDEN * (3ull * (used + (lane & (h - 1u))));
        }
        const double a = mine.random_pm1();
        const double b = mine.random_pm1();
        const double c = mine.random_pm1();
        const bool inside = valid && !(a * a + b * b + c * c >= 1.0);
        if (log_h == 0) {
            if (inside) {
                p = v3(a, b, c);
                pending = false;
            }
        } else {
            // the first lane of group `rank` whose candidate lies inside, if any
            const unsigned long long accepted = __ballot(inside);
            const uint32_t mine_first = (rank << log_h) & 63u;
            const uint32_t bits = (uint32_t)(accepted >> mine_first) & (uint32_t)((1ull << h) - 1ull);
            const uint32_t winner = (mine_first + (uint32_t)__builtin_ctz(bits | 0x80000000u)) & 63u;
            // (every lane takes part in the exchange; only a pending lane with a winner keeps what it gets)
            auto pull = [&](double x) -> double {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(winner << 2), (int)(uint32_t)u);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(winner << 2), (int)(uint32_t)(u >> 32));
                return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            };
            const double wa = pull(a), wb = pull(b), wc = pull(c);
            if (pending && bits != 0u) {
                p = v3(wa, wb, wc);
                pending = false;
            }
        }
        used += h;
    }
    return p;
}
// material.rs:227-229
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }
// material.rs:231-237
__device__ __forceinline__ V3 refract(V3 uv, V3 n, double etai_over_etat) {
    const double cos_theta = fmin(dot(-uv, n), 1.0);
    const V3 perpendicular = etai_over_etat * (uv + cos_theta * n);
    const V3 parallel = (-sqrt(fabs(1.0 - square_length(perpendicular)))) * n;
    return perpendicular + parallel;
}
// material.rs:239-243; powi(5) = x * (x*x)*(x*x)
__device__ __forceinline__ double schlick(double cosine, double ref_idx) {
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 *= r0;
    const double x = 1.0 - cosine;
    const double x2 = x * x;
    const double x4 = x2 * x2;
    return r0 + (1.0 - r0) * (x * x4);
}

}  // namespace mirt
