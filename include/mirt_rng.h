/*
 * mirt_rng.h -- the sampler behind `seed` in mirt.h, stated as integer
 * arithmetic so that any implementation (device, host, checker) draws the
 * same numbers.
 *
 * The reference draws every random number from rand 0.7.3's SmallRng seeded
 * from OS entropy (src/renderer.rs:30,47; src/main.rs:64; draws through
 * RandomDouble::random_double = gen_range(0.0, 1.0), src/util.rs:3-11), so no
 * stream of it can be reproduced and none is pinned by a test.  What is kept
 * is the ORDER and NUMBER of draws at each call site; the generator is this
 * counter-based one:
 *
 *   stream(seed, pixel, sample)         one 64-bit key per camera path
 *   draw(stream, stage, k)              k-th draw of stage `stage` of that path
 *
 * stage 0 is the camera: draws 0 and 1 are the jitter du, dv (renderer.rs:79-80),
 * draw 2 the shutter time (camera.rs:73), draws 3, 4, ... the pairs of the lens
 * disk's rejection loop (camera.rs:66,78-88).  The reference draws the time after
 * the pairs; here it has an index of its own so that it does not depend on how
 * many pairs the loop takes -- and a pinhole camera (lens_radius = 0), which
 * multiplies the loop's result by zero, need not run it.  Stage b+1 is the
 * material hit at bounce b (material.rs:53,79,119).  A rejection loop simply keeps
 * counting k.
 *
 * mirt_seq_* is a plain sequential generator over the same mixer for host-side
 * set-up work (scene generation, BVH axis draws, Perlin tables), standing in
 * for the reference's `rng: &mut T` parameters.
 */
#ifndef MIRT_RNG_H
#define MIRT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define MIRT_HD __host__ __device__ static inline
#else
#define MIRT_HD static inline
#endif

#define MIRT_GOLDEN 0x9E3779B97F4A7C15ull

/* SplitMix64 finaliser (Steele, Lea, Flood 2014; constants of Stafford's mix13). */
MIRT_HD uint64_t mirt_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MIRT_HD uint64_t mirt_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint64_t a = mirt_mix64(seed + MIRT_GOLDEN * ((uint64_t)pixel + 1u));
    return mirt_mix64(a + 0xD1B54A32D192ED03ull * ((uint64_t)sample + 1u));
}

MIRT_HD uint64_t mirt_rng_bits(uint64_t stream, uint32_t stage, uint32_t k) {
    uint64_t ctr = ((uint64_t)stage << 32) | (uint64_t)k;
    return mirt_mix64(stream + MIRT_GOLDEN * (ctr + 1u));
}

/* Uniform double in [0, 1) with 53 random bits (host-side set-up draws). */
MIRT_HD double mirt_bits_to_double(uint64_t bits) {
    return (double)(bits >> 11) * 0x1.0p-53;
}

/* Uniform double in [1, 2): 52 random bits as the mantissa under the exponent of 1.0.
 * rand 0.7.3 builds gen_range(0.0, 1.0) the same way (a [1, 2) value from 52 bits, then
 * value * 1.0 + (0.0 - 1.0)); on the device it is one integer instruction. */
MIRT_HD double mirt_bits_to_1_2(uint64_t bits) {
    union {
        uint64_t u;
        double d;
    } v;
    v.u = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    return v.d;
}

/* random_double() of the render path (src/util.rs:8-10): uniform in [0, 1), 52 random bits. */
MIRT_HD double mirt_rng_double(uint64_t stream, uint32_t stage, uint32_t k) {
    return mirt_bits_to_1_2(mirt_rng_bits(stream, stage, k)) - 1.0;
}

/* ---- sequential generator for host-side set-up ------------------------- */
typedef struct mirt_seq {
    uint64_t state;
} mirt_seq;

MIRT_HD mirt_seq mirt_seq_new(uint64_t seed) {
    mirt_seq s;
    s.state = mirt_mix64(seed ^ 0x6A09E667F3BCC909ull);
    return s;
}

MIRT_HD uint64_t mirt_seq_u64(mirt_seq* s) {
    s->state += MIRT_GOLDEN;
    return mirt_mix64(s->state);
}

/* random_double(): uniform in [0, 1). */
MIRT_HD double mirt_seq_double(mirt_seq* s) {
    return mirt_bits_to_double(mirt_seq_u64(s));
}

/* gen_range(lower, upper) for f64 (vec3.rs:27-33). */
MIRT_HD double mirt_seq_range(mirt_seq* s, double lower, double upper) {
    return lower + (upper - lower) * mirt_seq_double(s);
}

/* gen_range(0, n) for integers: unbiased enough for n << 2^32 (multiply-shift). */
MIRT_HD uint32_t mirt_seq_below(mirt_seq* s, uint32_t n) {
    uint64_t hi = mirt_seq_u64(s) >> 32;
    return (uint32_t)((hi * (uint64_t)n) >> 32);
}

#endif /* MIRT_RNG_H */
