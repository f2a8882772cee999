// og_math.h -- deterministic float64 elementary functions shared by host and device.
//
// Why this exists: a forward-difference Jacobian amplifies every last-ulp difference in the
// residual by 1/h = 6.7e7.  To compare the gfx950 kernels against a CPU
// oracle *bit for bit*, both sides must round identically.  IEEE-754 guarantees that for
// + - * / sqrt and fma, but not for exp/sin/cos/log, where ROCm's ocml and glibc/NumPy differ in
// the last place.  Every function here is therefore built only from correctly rounded
// operations (no library calls, no fast-math, compile with -ffp-contract=off) so that hipcc for
// gfx950 and gcc for x86-64 produce the same bits.  Algorithms are the classic Cody-Waite
// reduction + minimax-polynomial constructions (Sun fdlibm lineage, error < 1 ulp); the
// coefficient values are mathematical constants of those published approximations.
//
// The reference calls NumPy ufuncs here (np.exp / np.sin / np.cos / np.sqrt, e.g.
// examples/04_Goddard_0knot.py:35, examples/01_Brachistochrone_Problem.py:26-28); results agree
// with NumPy to <= 1 ulp, which is inside the FD noise floor defined in SURVEY.md section 8(c).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OG_HD __host__ __device__ inline
#define OG_HDI __host__ __device__ __forceinline__
#else
#define OG_HD inline
#define OG_HDI inline __attribute__((always_inline))
#endif

// OG_KEEP(x): the value is materialised here (device code: an empty asm that claims to read and write the register), so
// that a load feeding a select is not sunk into the select's arm and executed under an exec mask
// OG_ANY(c): true in every lane of the wavefront if c holds in any of them (a branch on it is uniform: no exec masks)
#if defined(__HIP_DEVICE_COMPILE__)
#define OG_KEEP(x) asm volatile("" : "+v"(x))
#define OG_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define OG_KEEP(x) ((void)0)
#define OG_ANY(c) (c)
#endif

namespace ogm {

OG_HD uint64_t bits_of(double x) { uint64_t u; memcpy(&u, &x, sizeof u); return u; }
OG_HD double from_bits(uint64_t u) { double x; memcpy(&x, &u, sizeof x); return x; }
OG_HD uint32_t hi_word(double x) { return (uint32_t)(bits_of(x) >> 32); }
OG_HD double fabs_(double x) { return from_bits(bits_of(x) & 0x7fffffffffffffffULL); }
OG_HD bool isnan_(double x) { return x != x; }

// correctly rounded on both sides (v_sqrt_f64 + refinement on gfx950, sqrtsd on x86)
OG_HD double sqrt_(double x) { return __builtin_sqrt(x); }
OG_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 2^k scaling by exponent arithmetic; k in [-2100, 2100].  Exact unless the result is subnormal.
OG_HD double scalb_(double x, int k) {
    // split into at most three exact power-of-two multiplies
    while (k > 1000) { x *= from_bits((uint64_t)(1023 + 1000) << 52); k -= 1000; }
    while (k < -1000) { x *= from_bits((uint64_t)(1023 - 1000) << 52); k += 1000; }
    return x * from_bits((uint64_t)(1023 + k) << 52);
}

// round-half-away nearest integer for |v| < 2^31 (argument reduction only)
OG_HD int nearest_int(double v) { return (int)(v < 0.0 ? v - 0.5 : v + 0.5); }

// ---------------------------------------------------------------- exp
OG_HD double exp_(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;   // 0x3fe62e42fee00000
    const double ln2_lo = 1.90821492927058770002e-10;   // 0x3dea39ef35793c76
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (isnan_(x)) return x;
    if (x > 7.09782712893383973096e+02) return from_bits(0x7ff0000000000000ULL);
    if (x < -7.45133219101941108420e+02) return 0.0;
    double ax = fabs_(x);
    double hi = x, lo = 0.0;
    int k = 0;
    if (ax > 0.34657359027997264) {                     // 0.5*ln2
        k = nearest_int(x * inv_ln2);
        hi = x - (double)k * ln2_hi;
        lo = (double)k * ln2_lo;
    } else if (ax < 3.725290298461914e-09) {            // 2^-28: exp(x) = 1 + x to < 1ulp
        return 1.0 + x;
    }
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y;
    if (k == 0) y = 1.0 - ((r * c) / (c - 2.0) - r);
    else        y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return (k == 0) ? y : scalb_(y, k);
}

// ---------------------------------------------------------------- log
OG_HD double log_(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                 L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    uint64_t u = bits_of(x);
    int k = 0;
    if (isnan_(x)) return x;
    if ((u << 1) == 0) return from_bits(0xfff0000000000000ULL);          // log(+-0) = -inf
    if (u >> 63) return from_bits(0x7ff8000000000000ULL);                // log(<0) = nan
    if ((u >> 52) == 0x7ff) return x;                                    // +inf
    if ((u >> 52) == 0) { x *= 18014398509481984.0; u = bits_of(x); k -= 54; }  // subnormal
    // normalise mantissa into [sqrt(2)/2, sqrt(2))
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffULL);
    double m = from_bits(u);
    double f = m - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (L2 + w * (L4 + w * L6));
    double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// ---------------------------------------------------------------- sin / cos
// reduce x to y0 + y1 in [-pi/4, pi/4], return quadrant (mod 4).  Up to |x| < 2^20 * pi/2: three-stage
// Cody-Waite with 33+33+53-bit pieces of pi/2 (exact there).  From there to 2^45: the multiple of pi/2 is
// subtracted in double-double arithmetic from four 53-bit pieces of pi/2 (212 bits), every product exact
// through an explicit fma (two-product) - an angle that large is already nonsense for a trajectory, but a bad
// line-search step may produce one, and sin/cos must then still be what NumPy returns (<= 1 ulp), not noise
// of the size of the FD step.  Beyond 2^45 (and for inf/nan) the reduction returns NaN, so that the row is
// caught by the engine's non-finite detection instead of carrying |sin| > 1 into the solver.
OG_HD void two_sum(double a, double b, double* s, double* e) {
    const double t = a + b;
    const double bb = t - a;
    *s = t;
    *e = (a - (t - bb)) + (b - bb);
}

// add one double to a non-overlapping expansion e[0..m) (increasing magnitude) exactly; it then has m+1 terms
OG_HD void expansion_grow(double* e, const int m, const double t) {
    double q = t;
    for (int i = 0; i < m; ++i) two_sum(q, e[i], &q, &e[i]);
    e[m] = q;
}

OG_HD int rem_pio2_large(double x, double* y0, double* y1) {
    const double inv_pio2 = 6.36619772367581382433e-01;
    const double pa = 1.5707963267948966e+00, pb = 6.123233995736766e-17,
                 pc = -1.4973849048591698e-33, pd = 5.562271104316826e-50;
    if (!(fabs_(x) < 35184372088832.0)) {                       // 2^45, inf, nan
        *y0 = (x - x) / (x - x);                                // 0/0 or nan: NaN either way
        *y1 = 0.0;
        return 0;
    }
    const double fn = (double)(long long)(x * inv_pio2 + (x < 0.0 ? -0.5 : 0.5));
    // x - fn*pi/2 = (x - ph_a) - pl_a - ph_b - pl_b - ph_c - pl_c - fn*pd with every product split exactly
    // (fma), x - ph_a exact (the two agree to within pi/4).  The leading terms may cancel down to 2^-50, so
    // they are summed as an exact expansion (Shewchuk) and only then rounded to y0 + y1.
    double e[7];
    double ph = fn * pa, pl = __builtin_fma(fn, pa, -ph);
    e[0] = x - ph;
    expansion_grow(e, 1, -pl);
    ph = fn * pb, pl = __builtin_fma(fn, pb, -ph);
    expansion_grow(e, 2, -ph);
    expansion_grow(e, 3, -pl);
    ph = fn * pc, pl = __builtin_fma(fn, pc, -ph);
    expansion_grow(e, 4, -ph);
    expansion_grow(e, 5, -pl);
    expansion_grow(e, 6, -(fn * pd));
    double hi = e[6], lo = 0.0, err;
    for (int i = 5; i >= 0; --i) {
        two_sum(hi, e[i], &hi, &err);
        lo += err;
    }
    two_sum(hi, lo, &hi, &lo);
    *y0 = hi;
    *y1 = lo;
    return (int)((long long)fn & 3);
}

OG_HD int rem_pio2(double x, double* y0, double* y1) {
    if (!(fabs_(x) < 1647099.0)) return rem_pio2_large(x, y0, y1);      // 2^20 * pi/2
    const double inv_pio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11;
    const double p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21;
    const double p3 = 2.02226624871116645580e-21, p3t = 8.47842766036889956997e-32;
    double fn = (double)(long long)(x * inv_pio2 + (x < 0.0 ? -0.5 : 0.5));
    double r = x - fn * p1;
    double w = fn * p1t;
    double y = r - w;
    int ex = (int)((hi_word(x) >> 20) & 0x7ff);
    int ey = (int)((hi_word(y) >> 20) & 0x7ff);
    if (ex - ey > 16) {                       // cancellation: bring in the next 33 bits
        double t = r;
        w = fn * p2;
        r = t - w;
        w = fn * p2t - ((t - r) - w);
        y = r - w;
        ey = (int)((hi_word(y) >> 20) & 0x7ff);
        if (ex - ey > 49) {                   // and the last 53
            t = r;
            w = fn * p3;
            r = t - w;
            w = fn * p3t - ((t - r) - w);
            y = r - w;
        }
    }
    *y0 = y;
    *y1 = (r - y) - w;
    return (int)((long long)fn & 3);
}

OG_HD double ksin(double x, double y, int have_tail) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

OG_HD double kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ax = fabs_(x);
    if (ax < 0.30000001192092896) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx = (ax > 0.78125) ? 0.28125
                               : from_bits((uint64_t)(hi_word(ax) - 0x00200000u) << 32);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

OG_HD double sin_(double x) {
    double ax = fabs_(x);
    if (!(ax < 1.0e300 * 1.0e300)) return x - x;                        // nan / inf -> nan
    if (ax <= 7.85398163397448278999e-01) {
        if (ax < 7.450580596923828e-09) return x;                        // 2^-27
        return ksin(x, 0.0, 0);
    }
    double y0, y1;
    int q = rem_pio2(x, &y0, &y1);
    switch (q) {
        case 0: return ksin(y0, y1, 1);
        case 1: return kcos(y0, y1);
        case 2: return -ksin(y0, y1, 1);
        default: return -kcos(y0, y1);
    }
}

OG_HD double cos_(double x) {
    double ax = fabs_(x);
    if (!(ax < 1.0e300 * 1.0e300)) return x - x;
    if (ax <= 7.85398163397448278999e-01) {
        if (ax < 7.450580596923828e-09) return 1.0;
        return kcos(x, 0.0);
    }
    double y0, y1;
    int q = rem_pio2(x, &y0, &y1);
    switch (q) {
        case 0: return kcos(y0, y1);
        case 1: return -ksin(y0, y1, 1);
        case 2: return -kcos(y0, y1);
        default: return ksin(y0, y1, 1);
    }
}

// tan via the same reduction: sin/cos quotient of the reduced argument (<= 2 ulp; the
// reference's examples only call np.tan in initial-guess code, never inside a callback).
OG_HD double tan_(double x) {
    double ax = fabs_(x);
    if (!(ax < 1.0e300 * 1.0e300)) return x - x;
    if (ax < 7.450580596923828e-09) return x;
    double y0 = x, y1 = 0.0;
    int q = 0;
    if (ax > 7.85398163397448278999e-01) q = rem_pio2(x, &y0, &y1);
    double s = ksin(y0, y1, 1), c = kcos(y0, y1);
    return (q & 1) ? -(c / s) : (s / c);
}

// ---------------------------------------------------------------- atan / atan2 / asin / acos
OG_HD double atan_(double x) {
    const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                          9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                          1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
                 a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
                 a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
                 a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
                 a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
                 a10 = 1.62858201153657823623e-02;
    if (isnan_(x)) return x;
    const bool neg = (bits_of(x) >> 63) != 0;
    const uint32_t ix = hi_word(x) & 0x7fffffffu;
    if (ix >= 0x44100000u) {                          // |x| >= 2^66 (or inf)
        const double r = hi[3] + lo[3];
        return neg ? -r : r;
    }
    int id;
    if (ix < 0x3fdc0000u) {                           // |x| < 0.4375
        if (ix < 0x3e200000u) return x;               // |x| < 2^-29
        id = -1;
    } else {
        x = fabs_(x);
        if (ix < 0x3ff30000u) {                       // |x| < 1.1875
            if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else                  { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else if (ix < 0x40038000u) {                // |x| < 2.4375
            id = 2; x = (x - 1.5) / (1.0 + 1.5 * x);
        } else {
            id = 3; x = -1.0 / x;
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    const double r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
    return neg ? -r : r;
}

OG_HD double atan2_(double y, double x) {
    const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
    const double pi_o_2 = 1.5707963267948965580e+00, pi_o_4 = 7.8539816339744827900e-01;
    if (isnan_(x) || isnan_(y)) return x + y;
    const uint64_t ux = bits_of(x), uy = bits_of(y);
    const uint32_t hx = (uint32_t)(ux >> 32), hy = (uint32_t)(uy >> 32);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);     // 2*sign(x) + sign(y)
    if (x == 1.0) return atan_(y);
    if ((uy << 1) == 0) {                             // y == +-0
        switch (m) { case 0: case 1: return y; case 2: return pi; default: return -pi; }
    }
    if ((ux << 1) == 0) return (hy >> 31) ? -pi_o_2 : pi_o_2;           // x == +-0
    const bool xinf = (ux << 1) == 0xffe0000000000000ULL, yinf = (uy << 1) == 0xffe0000000000000ULL;
    if (xinf) {
        if (yinf) {
            switch (m) { case 0: return pi_o_4; case 1: return -pi_o_4;
                         case 2: return 3.0 * pi_o_4; default: return -3.0 * pi_o_4; }
        }
        switch (m) { case 0: return 0.0; case 1: return -0.0; case 2: return pi; default: return -pi; }
    }
    if (yinf) return (hy >> 31) ? -pi_o_2 : pi_o_2;
    const int k = ((int)iy - (int)ix) >> 20;
    double z;
    int mm = m;
    if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; mm &= 1; }                  // |y/x| > 2^60
    else if ((hx >> 31) && k < -60) z = 0.0;                             // 0 > |y|/x > -2^-60
    else z = atan_(fabs_(y / x));
    switch (mm) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

OG_HD double asin_poly_p(double t) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    return t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
}

OG_HD double asin_poly_q(double t) {
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    return 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
}

OG_HD double asin_(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pio4_hi = 7.85398163397448278999e-01;
    if (isnan_(x)) return x;
    const bool neg = (bits_of(x) >> 63) != 0;
    const uint32_t ix = hi_word(x) & 0x7fffffffu;
    const double ax = fabs_(x);
    if (ax >= 1.0) {
        if (ax == 1.0) return x * pio2_hi + x * pio2_lo;
        return (x - x) / (x - x);                      // |x| > 1: NaN
    }
    if (ix < 0x3fe00000u) {                            // |x| < 0.5
        if (ix < 0x3e500000u) return x;                // |x| < 2^-26
        const double t = x * x;
        return x + x * (asin_poly_p(t) / asin_poly_q(t));
    }
    const double w = 1.0 - ax;
    const double t = w * 0.5;
    const double p = asin_poly_p(t), q = asin_poly_q(t);
    const double s = sqrt_(t);
    double r;
    if (ix >= 0x3fef3333u) {                           // |x| >= 0.975
        r = pio2_hi - (2.0 * (s + s * (p / q)) - pio2_lo);
    } else {
        const double sh = from_bits(bits_of(s) & 0xffffffff00000000ULL);
        const double c = (t - sh * sh) / (s + sh);
        const double rr = p / q;
        const double pp = 2.0 * s * rr - (pio2_lo - 2.0 * c);
        const double qq = pio4_hi - 2.0 * sh;
        r = pio4_hi - (pp - qq);
    }
    return neg ? -r : r;
}

OG_HD double acos_(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pi = 3.14159265358979311600e+00;
    if (isnan_(x)) return x;
    const bool neg = (bits_of(x) >> 63) != 0;
    const uint32_t ix = hi_word(x) & 0x7fffffffu;
    const double ax = fabs_(x);
    if (ax >= 1.0) {
        if (ax == 1.0) return neg ? pi + 2.0 * pio2_lo : 0.0;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000u) {                            // |x| < 0.5
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;      // |x| < 2^-57
        const double z = x * x;
        const double r = asin_poly_p(z) / asin_poly_q(z);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (neg) {                                          // x < -0.5
        const double z = (1.0 + x) * 0.5;
        const double s = sqrt_(z);
        const double r = asin_poly_p(z) / asin_poly_q(z);
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5;                   // x > 0.5
    const double s = sqrt_(z);
    const double df = from_bits(bits_of(s) & 0xffffffff00000000ULL);
    const double c = (z - df * df) / (s + df);
    const double r = asin_poly_p(z) / asin_poly_q(z);
    const double w = r * s + c;
    return 2.0 * (df + w);
}

// ---------------------------------------------------------------- hyperbolic, other logarithms, roots
// Built on exp_ / log_ above with the classical correction steps (W. Kahan's expm1 / log1p through the rounded
// exp / 1 + x, fdlibm's formulae for sinh / cosh / tanh on top of expm1, one Newton step for cbrt): within 2 ulp of
// NumPy's libm over the whole range (tests/test_og_math.py), and - what matters for parity - the SAME bits on the
// host twin and on gfx950, since they are made of the same bit-reproducible pieces.
// sinh(h) / h and cosh(h) by their series in h^2, |h| <= 1 (term 12 is below 1e-23)
OG_HD double sinhc_series(double h2) {
    double p = 1.0 + h2 / 600.0;                                // 24 * 25
    p = 1.0 + h2 / 506.0 * p;                                   // 22 * 23
    p = 1.0 + h2 / 420.0 * p;
    p = 1.0 + h2 / 342.0 * p;
    p = 1.0 + h2 / 272.0 * p;
    p = 1.0 + h2 / 210.0 * p;
    p = 1.0 + h2 / 156.0 * p;
    p = 1.0 + h2 / 110.0 * p;
    p = 1.0 + h2 / 72.0 * p;
    p = 1.0 + h2 / 42.0 * p;
    p = 1.0 + h2 / 20.0 * p;
    return 1.0 + h2 / 6.0 * p;
}
OG_HD double cosh_series(double h2) {
    double p = 1.0 + h2 / 552.0;                                // 23 * 24
    p = 1.0 + h2 / 462.0 * p;                                   // 21 * 22
    p = 1.0 + h2 / 380.0 * p;
    p = 1.0 + h2 / 306.0 * p;
    p = 1.0 + h2 / 240.0 * p;
    p = 1.0 + h2 / 182.0 * p;
    p = 1.0 + h2 / 132.0 * p;
    p = 1.0 + h2 / 90.0 * p;
    p = 1.0 + h2 / 56.0 * p;
    p = 1.0 + h2 / 30.0 * p;
    p = 1.0 + h2 / 12.0 * p;
    return 1.0 + h2 / 2.0 * p;
}
OG_HD double expm1_(double x) {
    if (isnan_(x)) return x;
    if (x > 709.8) return from_bits(0x7ff0000000000000ULL);
    if (x < -40.0) return -1.0;
    if (fabs_(x) < 1.0) {
        // exp(x) - 1 would cancel: the series itself, x (1 + x/2 (1 + x/3 (1 + ... x/24)))
        double p = 1.0 + x / 24.0;
        for (int k = 23; k >= 2; --k) p = 1.0 + x / (double)k * p;
        return x * p;
    }
    return exp_(x) - 1.0;
}
OG_HD double log1p_(double x) {
    if (isnan_(x) || x == from_bits(0x7ff0000000000000ULL)) return x;
    if (x < -1.0) return from_bits(0x7ff8000000000000ULL);
    const double u = 1.0 + x;
    if (u == 1.0) return x;
    // Kahan: log(u) corrected by the rounding of u = 1 + x
    return log_(u) * (x / (u - 1.0));
}
OG_HD double sinh_(double x) {
    if (isnan_(x)) return x;
    const double ax = fabs_(x), sg = x < 0.0 ? -1.0 : 1.0;
    if (ax < 1.0) return x * sinhc_series(x * x);
    if (ax < 22.0) {
        const double e = exp_(ax);
        return sg * (0.5 * e - 0.5 / e);
    }
    if (ax < 709.0) return sg * 0.5 * exp_(ax);
    const double w = exp_(0.5 * ax);
    return sg * (0.5 * w) * w;
}
OG_HD double cosh_(double x) {
    if (isnan_(x)) return x;
    const double ax = fabs_(x);
    if (ax < 1.0) return cosh_series(x * x);
    if (ax < 22.0) {
        const double t = exp_(ax);
        return 0.5 * t + 0.5 / t;
    }
    if (ax < 709.0) return 0.5 * exp_(ax);
    const double w = exp_(0.5 * ax);
    return (0.5 * w) * w;
}
OG_HD double tanh_(double x) {
    if (isnan_(x)) return x;
    const double ax = fabs_(x), sg = x < 0.0 ? -1.0 : 1.0;
    if (ax >= 22.0) return sg;
    if (ax < 1.0) {
        const double x2 = x * x;
        return x * sinhc_series(x2) / cosh_series(x2);
    }
    const double t = exp_(2.0 * ax);
    return sg * (1.0 - 2.0 / (t + 1.0));
}
OG_HD double log2_(double x) {
    // x = m 2^e with m in [1/sqrt 2, sqrt 2): log2 x = e + log(m) / ln 2 - exact for powers of two, and the quotient's
    // error is compensated through an FMA
    if (!(x > 0.0) || x - x != 0.0) return log_(x);            // 0, negative, inf, NaN: log's own answers
    int e = 0;
    if (x < 2.2250738585072014e-308) {
        x *= 18014398509481984.0;                               // 2^54
        e = -54;
    }
    const uint64_t u = bits_of(x);
    e += (int)((u >> 52) & 0x7ff) - 1023;
    double m = from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double l = log_(m);
    const double inv_hi = 1.4426950408889634, inv_lo = 2.0355273740931033e-17;     // 1 / ln 2 = hi + lo
    const double q = l * inv_hi;
    return (double)e + (q + (fma_(l, inv_hi, -q) + l * inv_lo));
}
OG_HD double log10_(double x) {
    const double l = log_(x);
    if (!(l == l) || l - l != 0.0) return l;
    const double inv_hi = 0.4342944819032518, inv_lo = 1.098319650216765e-17;       // 1 / ln 10 = hi + lo
    const double q = l * inv_hi;
    return q + (fma_(l, inv_hi, -q) + l * inv_lo);
}
OG_HD double cbrt_(double x) {
    if (isnan_(x) || x == 0.0 || x - x != 0.0) return x;       // NaN, +-0, +-inf
    const double ax = fabs_(x);
    double t = exp_(log_(ax) * (1.0 / 3.0));
    // two Newton steps on t^3 = |x| in the form t -= t (t^3 - |x|) / (3 t^3): the first removes exp/log's few ulp,
    // the second confirms (a fixed point is within half an ulp of the root or next to it)
    for (int i = 0; i < 2; ++i) {
        const double t2 = t * t, t3 = t2 * t;
        const double r = fma_(t2, t, -t3);                       // t^3 = t3 + r
        t -= t * ((t3 - ax) + r) / (3.0 * t3);
    }
    return x < 0.0 ? -t : t;
}
OG_HD double hypot_(double x, double y) {
    double a = fabs_(x), b = fabs_(y);
    const double inf = from_bits(0x7ff0000000000000ULL);
    if (a == inf || b == inf) return inf;
    if (isnan_(a) || isnan_(b)) return a + b;
    if (a < b) { const double t = a; a = b; b = t; }
    if (a == 0.0) return 0.0;
    // scale by a power of two so that neither square over- or underflows, then correct sqrt(a^2 + b^2) by the
    // rounding errors of the squares (FMA) - within 1 ulp
    const int e = (int)((bits_of(a) >> 52) & 0x7ff) - 1023;
    const double sc = scalb_(1.0, -e), as = a * sc, bs = b * sc;
    const double a2 = as * as, b2 = bs * bs, s2 = a2 + b2;
    const double err = (fma_(as, as, -a2) + fma_(bs, bs, -b2)) + ((a2 - s2) + b2);
    const double h = sqrt_(s2);
    return scalb_(h + err / (2.0 * h), e);
}
// floor / trunc / fmod are exact operations (the result is always representable): the compiler's builtins give the
// same bits on both sides (v_floor_f64 / v_trunc_f64 and ocml's exact fmod on gfx950, libm on x86)
OG_HD double floor_(double x) { return __builtin_floor(x); }
OG_HD double trunc_(double x) { return __builtin_trunc(x); }
OG_HD double fmod_(double a, double b) { return __builtin_fmod(a, b); }
// np.remainder / np.mod / Python's %: npy_remainder of numpy/_core/src/npymath/npy_math_internal.h.src - fmod, moved
// to the divisor's sign; fmod's NaN for a zero divisor
OG_HD double mod_(double a, double b) {
    double m = fmod_(a, b);
    if (b == 0.0) return m;
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = from_bits((bits_of(b) & 0x8000000000000000ULL));       // copysign(0, b)
    }
    return m;
}
// x ** y for a traced exponent: exp(y log x) for x > 0 (NumPy calls libm's pow there: this is |y log x| ulp away from
// it at worst, a few ulp for the exponents models use); pow's special cases for the rest
OG_HD double pow_(double x, double y) {
    if (y == 0.0 || x == 1.0) return 1.0;
    if (isnan_(x) || isnan_(y)) return x + y;
    // the bound first: a float-to-int cast is undefined beyond INT_MAX / for inf, and host and device saturate differently
    if (fabs_(y) <= 64.0 && y == (double)(int)y && x - x == 0.0) {
        // a small whole exponent (at run time): square and multiply, exact where the products are
        int k = (int)fabs_(y);
        double base = x, acc = 1.0;
        while (k) {
            if (k & 1) acc *= base;
            k >>= 1;
            if (k) base *= base;
        }
        return y < 0.0 ? 1.0 / acc : acc;
    }
    if (x > 0.0) return exp_(y * log_(x));
    const double inf = from_bits(0x7ff0000000000000ULL);
    const bool yint = fabs_(y) < 9.0e15 && (y == (double)(long long)y);
    const bool yodd = yint && (((long long)y) & 1LL);
    if (x == 0.0) {
        const bool neg = (bits_of(x) >> 63) != 0 && yodd;
        return y > 0.0 ? (neg ? -0.0 : 0.0) : (neg ? -inf : inf);
    }
    if (!yint) return (fabs_(y) == inf) ? ((fabs_(x) < 1.0) == (y < 0.0) ? inf : (fabs_(x) == 1.0 ? 1.0 : 0.0))
                                        : from_bits(0x7ff8000000000000ULL);
    const double r = exp_(y * log_(-x));
    return yodd ? -r : r;
}

// ---------------------------------------------------------------- linear table lookup
// scipy.interpolate.interp1d(kind="linear") as the reference's example 11 uses it
// (examples/11_Polar_TSTO_Taiki.py:21-27; SciPy 1.15.3 scipy/interpolate/_interpolate.py
// _call_linear + _evaluate): i = clip(searchsorted(xg, x, "left"), 1, n-1);
// y = (y[i]-y[i-1])/(xg[i]-xg[i-1]) * (x - xg[i-1]) + y[i-1]; outside [xg[0], xg[n-1]] the
// fill values replace it unless extrapolating.  mode 0: fill values; 1: extrapolate;
// 2: the reference would raise ValueError (bounds_error=True) - a kernel cannot, it returns NaN.
OG_HD double interp_linear(const double* xg, const double* yg, const int n, const int mode,
                           const double fill_below, const double fill_above, const double x) {
    int lo = 0, hi = n;
    while (lo < hi) {                         // first index with xg[index] >= x
        const int mid = (lo + hi) >> 1;
        if (xg[mid] < x) lo = mid + 1; else hi = mid;
    }
    int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const double x_lo = xg[i - 1], x_hi = xg[i], y_lo = yg[i - 1], y_hi = yg[i];
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    double y = slope * (x - x_lo) + y_lo;
    if (mode != 1) {
        if (x < xg[0]) y = (mode == 0) ? fill_below : from_bits(0x7ff8000000000000ULL);
        if (x > xg[n - 1]) y = (mode == 0) ? fill_above : from_bits(0x7ff8000000000000ULL);
    }
    return y;
}

}  // namespace ogm
