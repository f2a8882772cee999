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
// reduce x to y0 + y1 in [-pi/4, pi/4], return quadrant (mod 4).  Three-stage Cody-Waite with
// 33+33+53-bit pieces of pi/2: exact for |x| < 2^20 * pi/2; beyond that accuracy degrades
// gracefully (trajectory angles never get there).
OG_HD int rem_pio2(double x, double* y0, double* y1) {
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
