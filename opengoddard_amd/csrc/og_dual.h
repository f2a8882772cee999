// og_dual.h - first-order dual numbers for the exact-Jacobian mode.
//
// The generated callback code (codegen.emit_header) is written against `typename X::scalar`: with the
// accessor types of the FD sweep the scalar is `double` and nothing changes, bit for bit; with an
// accessor that returns `ogdual` (value, derivative along one decision variable) the very same
// functions propagate exact first derivatives.  Every rule below is the textbook one; the value
// parts use the same og_math.h functions as the FD path, so F(x) is identical in both modes.
//
// Same source for hipcc (device) and g++ (the CPU twin): GPU and twin derivatives agree bit for
// bit, like the values do.
#ifndef OG_DUAL_H
#define OG_DUAL_H

#include "og_math.h"

struct ogdual {
    double v, d;
    OG_HDI ogdual() : v(0.0), d(0.0) {}
    OG_HDI ogdual(const double value) : v(value), d(0.0) {}
    OG_HDI ogdual(const double value, const double deriv) : v(value), d(deriv) {}
};

OG_HDI ogdual operator-(const ogdual a) { return ogdual(-a.v, -a.d); }
OG_HDI ogdual operator+(const ogdual a, const ogdual b) { return ogdual(a.v + b.v, a.d + b.d); }
OG_HDI ogdual operator-(const ogdual a, const ogdual b) { return ogdual(a.v - b.v, a.d - b.d); }
OG_HDI ogdual operator*(const ogdual a, const ogdual b) { return ogdual(a.v * b.v, a.d * b.v + a.v * b.d); }
OG_HDI ogdual operator/(const ogdual a, const ogdual b) {
    const double q = a.v / b.v;
    return ogdual(q, (a.d == 0.0 && b.d == 0.0) ? 0.0 : (a.d - q * b.d) / b.v);
}
OG_HDI ogdual operator+(const ogdual a, const double b) { return ogdual(a.v + b, a.d); }
OG_HDI ogdual operator+(const double a, const ogdual b) { return ogdual(a + b.v, b.d); }
OG_HDI ogdual operator-(const ogdual a, const double b) { return ogdual(a.v - b, a.d); }
OG_HDI ogdual operator-(const double a, const ogdual b) { return ogdual(a - b.v, -b.d); }
OG_HDI ogdual operator*(const ogdual a, const double b) { return ogdual(a.v * b, a.d * b); }
OG_HDI ogdual operator*(const double a, const ogdual b) { return ogdual(a * b.v, a * b.d); }
OG_HDI ogdual operator/(const ogdual a, const double b) { return ogdual(a.v / b, a.d / b); }
OG_HDI ogdual operator/(const double a, const ogdual b) {
    const double q = a / b.v;
    return ogdual(q, -(q * b.d) / b.v);
}

#define OG_DUAL_CMP(op)                                                          \
    OG_HDI bool operator op(const ogdual a, const ogdual b) { return a.v op b.v; } \
    OG_HDI bool operator op(const ogdual a, const double b) { return a.v op b; }   \
    OG_HDI bool operator op(const double a, const ogdual b) { return a op b.v; }
OG_DUAL_CMP(<)
OG_DUAL_CMP(<=)
OG_DUAL_CMP(>)
OG_DUAL_CMP(>=)
OG_DUAL_CMP(==)
OG_DUAL_CMP(!=)
#undef OG_DUAL_CMP

namespace ogm {

// (a quantity that does not depend on the seeded variable keeps derivative 0 through a singular
// rule - sqrt at 0, log at 0, asin at 1 - instead of 0/0; a forward difference sees 0 there too)
// (at r == 0 the slope is taken as 0 as well: where the dynamics contain V |V|-like terms, V^2 underflows to 0
// for |V| < 1e-154 while dV^2 = 2V is still non-zero, and 2V / (2 * 0) would put an inf - then 0 * inf = NaN
// - into a Jacobian whose composite entry is 0; a bare sqrt(x) at x = 0 has no finite derivative in any mode)
OG_HDI ogdual sqrt_(const ogdual a) {
    const double r = sqrt_(a.v);
    return ogdual(r, (a.d == 0.0 || r == 0.0) ? 0.0 : a.d / (2.0 * r));
}
OG_HDI ogdual exp_(const ogdual a) {
    const double e = exp_(a.v);
    return ogdual(e, a.d * e);
}
OG_HDI ogdual log_(const ogdual a) { return ogdual(log_(a.v), a.d == 0.0 ? 0.0 : a.d / a.v); }
OG_HDI ogdual sin_(const ogdual a) { return ogdual(sin_(a.v), a.d * cos_(a.v)); }
OG_HDI ogdual cos_(const ogdual a) { return ogdual(cos_(a.v), -(a.d * sin_(a.v))); }
OG_HDI ogdual tan_(const ogdual a) {
    const double t = tan_(a.v);
    return ogdual(t, a.d * (1.0 + t * t));
}
OG_HDI ogdual fabs_(const ogdual a) {
    // d|x|/dx = sign(x); 0 at the kink (NumPy's sign(0)), where a one-sided difference sees +-1
    const double s = a.v > 0.0 ? 1.0 : (a.v < 0.0 ? -1.0 : 0.0);
    return ogdual(fabs_(a.v), s * a.d);
}
OG_HDI ogdual atan_(const ogdual a) { return ogdual(atan_(a.v), a.d / (1.0 + a.v * a.v)); }
OG_HDI ogdual asin_(const ogdual a) {
    return ogdual(asin_(a.v), a.d == 0.0 ? 0.0 : a.d / sqrt_(1.0 - a.v * a.v));
}
OG_HDI ogdual acos_(const ogdual a) {
    return ogdual(acos_(a.v), a.d == 0.0 ? 0.0 : -(a.d / sqrt_(1.0 - a.v * a.v)));
}
OG_HDI ogdual atan2_(const ogdual y, const ogdual x) {
    return ogdual(atan2_(y.v, x.v),
                  (y.d == 0.0 && x.d == 0.0) ? 0.0 : (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v));
}
OG_HDI ogdual atan2_(const ogdual y, const double x) { return atan2_(y, ogdual(x)); }
OG_HDI ogdual atan2_(const double y, const ogdual x) { return atan2_(ogdual(y), x); }

OG_HDI ogdual tanh_(const ogdual a) {
    const double t = tanh_(a.v);
    return ogdual(t, a.d * (1.0 - t * t));
}
OG_HDI ogdual sinh_(const ogdual a) { return ogdual(sinh_(a.v), a.d * cosh_(a.v)); }
OG_HDI ogdual cosh_(const ogdual a) { return ogdual(cosh_(a.v), a.d * sinh_(a.v)); }
OG_HDI ogdual expm1_(const ogdual a) { return ogdual(expm1_(a.v), a.d * exp_(a.v)); }
OG_HDI ogdual log1p_(const ogdual a) { return ogdual(log1p_(a.v), a.d == 0.0 ? 0.0 : a.d / (1.0 + a.v)); }
OG_HDI ogdual log2_(const ogdual a) {
    return ogdual(log2_(a.v), a.d == 0.0 ? 0.0 : a.d / (a.v * 0.6931471805599453));
}
OG_HDI ogdual log10_(const ogdual a) {
    return ogdual(log10_(a.v), a.d == 0.0 ? 0.0 : a.d / (a.v * 2.302585092994046));
}
OG_HDI ogdual cbrt_(const ogdual a) {
    const double c = cbrt_(a.v);
    return ogdual(c, (a.d == 0.0 || c == 0.0) ? 0.0 : a.d / (3.0 * c * c));
}
OG_HDI ogdual hypot_(const ogdual x, const ogdual y) {
    const double h = hypot_(x.v, y.v);
    return ogdual(h, ((x.d == 0.0 && y.d == 0.0) || h == 0.0) ? 0.0 : (x.v * x.d + y.v * y.d) / h);
}
OG_HDI ogdual hypot_(const ogdual x, const double y) { return hypot_(x, ogdual(y)); }
OG_HDI ogdual hypot_(const double x, const ogdual y) { return hypot_(ogdual(x), y); }
// remainder (the divisor's sign, Python's %) and fmod (the dividend's sign): a - q b with q constant between jumps
OG_HDI ogdual mod_(const ogdual a, const ogdual b) {
    const double q = floor_(a.v / b.v);
    return ogdual(mod_(a.v, b.v), a.d - ((b.d == 0.0 || q - q != 0.0) ? 0.0 : q * b.d));
}
OG_HDI ogdual mod_(const ogdual a, const double b) { return ogdual(mod_(a.v, b), a.d); }
OG_HDI ogdual mod_(const double a, const ogdual b) { return mod_(ogdual(a), b); }
OG_HDI ogdual fmod_(const ogdual a, const ogdual b) {
    const double q = trunc_(a.v / b.v);
    return ogdual(fmod_(a.v, b.v), a.d - ((b.d == 0.0 || q - q != 0.0) ? 0.0 : q * b.d));
}
OG_HDI ogdual fmod_(const ogdual a, const double b) { return ogdual(fmod_(a.v, b), a.d); }
OG_HDI ogdual fmod_(const double a, const ogdual b) { return fmod_(ogdual(a), b); }
// x ** y: d = x^y (y' log x + y x' / x); a part whose factor does not depend on the seeded variable stays out (0 log 0)
OG_HDI ogdual pow_(const ogdual x, const ogdual y) {
    const double p = pow_(x.v, y.v);
    double d = 0.0;
    if (x.d != 0.0) d += y.v * pow_(x.v, y.v - 1.0) * x.d;
    if (y.d != 0.0) d += p * log_(x.v) * y.d;
    return ogdual(p, d);
}
OG_HDI ogdual pow_(const ogdual x, const double y) { return pow_(x, ogdual(y)); }
OG_HDI ogdual pow_(const double x, const ogdual y) { return pow_(ogdual(x), y); }

// Piecewise-linear table: the derivative is the slope of the segment the value came from (the
// segment to the right at a knot, like the forward difference sees it), 0 on the constant fills.
OG_HD ogdual interp_linear(const double* xg, const double* yg, const int n, const int mode, const double fill_below,
                           const double fill_above, const ogdual x) {
    const double y = interp_linear(xg, yg, n, mode, fill_below, fill_above, x.v);
    int lo = 0, hi = n;
    while (lo < hi) {                         // first index with xg[index] > x  (right-hand segment at a knot)
        const int mid = (lo + hi) >> 1;
        if (xg[mid] <= x.v) lo = mid + 1; else hi = mid;
    }
    const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    double slope = (yg[i] - yg[i - 1]) / (xg[i] - xg[i - 1]);
    if (mode != 1 && (x.v < xg[0] || x.v > xg[n - 1])) slope = 0.0;
    return ogdual(y, slope * x.d);
}

}  // namespace ogm

#endif /* OG_DUAL_H */
