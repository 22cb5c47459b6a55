// checker_sin.h -- sin(x) for the checker texture (texture.rs:45-54), written out so that it
// needs few registers where it is inlined: the library sin() carries its large-argument
// reduction into every kernel that calls it (25 more VGPRs in the path kernel, which is the
// difference between 3 and 4 waves per SIMD there).
//
// Method (Sun's fdlibm, s_sin.c / e_rem_pio2.c / k_sin.c / k_cos.c, restated): for
// |x| < 2^20 * pi/2, n = rint(x * 2/pi) and x - n * pi/2 with pi/2 split in three 33-bit
// pieces (each product n * piece is exact), carried as a double-double (r, w); then the
// degree-13 sine or degree-14 cosine polynomial by the parity of n.  Error below 1 ulp over
// that range (checked against the host library in tests/test_checker_sin.py).  Plain C++,
// no FMA contraction assumed, so host and device agree bit for bit.  The caller handles
// |x| >= SIN_SMALL_LIMIT, infinities and NaN some other way.
#pragma once

#if defined(__HIPCC__)
#define CHECKER_SIN_HD __host__ __device__ inline
#else
#define CHECKER_SIN_HD inline
#endif

namespace mirt {

constexpr double SIN_SMALL_LIMIT = 1647099.0;  // 2^20 * pi/2, rounded down

CHECKER_SIN_HD double sin_small(double x) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11;
    const double pio2_3 = 2.02226624871116645580e-21;
    const double pio2_3t = 8.47842766036889956997e-32;
    const double fn = __builtin_rint(x * invpio2);
    const int n = (int)fn;
    // three rounds of e_rem_pio2.c, unconditionally (the reference code skips the later ones
    // when the earlier result is already accurate enough; doing them anyway changes nothing then)
    // Rounds two and three of e_rem_pio2.c, both unconditionally (that code stops after a round
    // whose result is accurate enough; here the rounding error of each subtraction is carried
    // along instead, which gives the same double-double):
    //   x - fn*pio2 = r3 + (e2 + e3 - fn*pio2_3t),  every fn*piece product exact
    const double t = x - fn * pio2_1;
    const double w2 = fn * pio2_2;
    const double r2 = t - w2;
    const double e2 = (t - r2) - w2;
    const double w3 = fn * pio2_3;
    const double r3 = r2 - w3;
    const double e3 = (r2 - r3) - w3;
    const double w = (fn * pio2_3t - e3) - e2;
    const double y0 = r3 - w;
    const double y1 = (r3 - y0) - w;
    const double z = y0 * y0;
    double v;
    if (n & 1) {
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05;
        const double C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        v = 1.0 - (0.5 * z - (z * p - y0 * y1));
    } else {
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04;
        const double S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        const double c = z * y0;
        const double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        v = y0 - ((z * (0.5 * y1 - c * p) - y1) - c * S1);
    }
    return (n & 2) ? -v : v;
}

// The checker texture only asks whether sin(10x) * sin(10y) * sin(10z) is negative
// (texture.rs:45-54), and the sign of a sine needs no polynomial: with n = rint(x * 2/pi) and
// r = x - n * pi/2, sin(x) has the sign of r for n = 0 (mod 4), is positive for n = 1, has the
// opposite sign of r for n = 2 and is negative for n = 3 (|r| <= pi/4 + a little).  Two terms
// of pi/2 give r to within 3e-15 for |x| < 2^20 * pi/2 (n * pio2_1 is exact, the third piece
// of pi/2 is below 2.1e-21), so the sign of r is certain once |r| > 1e-12.
// Returns 1 (negative sine), 0 (positive sine) or -1: not decided here (|x| too large or not
// finite, or x within 1e-12 of a multiple of pi/2 -- there the caller evaluates the sine).
CHECKER_SIN_HD int sin_is_negative_fast(double x) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11;
    if (!(__builtin_fabs(x) < SIN_SMALL_LIMIT)) return -1;
    const double fn = __builtin_rint(x * invpio2);
    const int n = (int)fn;
    const double r = (x - fn * pio2_1) - fn * pio2_2;
    if (!(__builtin_fabs(r) > 1.0e-12)) return -1;
    const int flipped = (n >> 1) & 1;
    return (n & 1) ? flipped : (flipped ^ (r < 0.0 ? 1 : 0));
}

}  // namespace mirt
