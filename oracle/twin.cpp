// TEST INFRASTRUCTURE - CPU twin of the GPU sweep for one traced problem (plain C++, g++).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call this.
//
// It restates, with ordinary loops and a *materialised* perturbed vector, exactly what the
// reference + SciPy compute per callback evaluation and per FD column:
//   twin_eval   F(x) = [cost | c_eq | c_ineq]            OpenGoddard/optimize.py:670-709
//   twin_sweep  J_T[i] = (F(x + h_i e_i) - F(x)) / dx_i  scipy/optimize/_numdiff.py:584-625
// using the same generated pointwise functions (OgGen) and the same deterministic math
// (og_math.h) as the kernels, compiled with -ffp-contract=off.  The collocation product
// D[k,:] . x~ is accumulated as a left-to-right fma chain over l - the order the gfx950
// v_mfma_f64_16x16x4_f64 path produces - so GPU results can be compared bit for bit.  The
// reference itself uses OpenBLAS dgemv here (optimize.py:682), whose summation order is
// unspecified; that difference is rounding-level and is bounded in tests/ against the goldens.
//
// Nothing here knows about tiles, wavefronts, LDS or MFMA operand layouts: it is the
// independent check of the kernel skeleton's indexing.
#include <cmath>
#include <cstring>
#include <vector>

#include "og_dual.h"
#include OG_GEN_HEADER

namespace {

struct XVec {
    typedef double scalar;
    const double* x;
    double operator()(const int i) const { return x[i]; }
    double ldy(const double* p) const { return *p; }          // base collocation product (scratch of the evaluation)
};

// one decision variable carries a unit derivative: F evaluated on it yields dF/dx_j exactly
struct XDual {
    typedef ogdual scalar;
    const double* x;
    int j;
    ogdual operator()(const int i) const { return ogdual(x[i], i == j ? 1.0 : 0.0); }
    double ldy(const double* p) const { return *p; }
};

inline double fma_chain(const double a, const double b, const double acc) { return __builtin_fma(a, b, acc); }
inline ogdual fma_chain(const ogdual a, const double b, const ogdual acc) {
    return ogdual(__builtin_fma(a.v, b, acc.v), __builtin_fma(a.d, b, acc.d));
}

template <class XA>
void eval_generic(const XA& xa, const double* const* D, const double* cv, typename XA::scalar* F) {
    typedef typename XA::scalar S;
    S out[OgGen::MAX_OUT];
    for (int g = 0; g < OgGen::N_GROUPS; ++g) {
        const int L = OgGen::G_LEN(g), nout = OgGen::G_NOUT(g);
        if (OgGen::G_KIND(g) == 0) {
            for (int k = 0; k < L; ++k) {
                OgGen::group_eval(g, k, xa, (const S*)nullptr, cv, out);
                for (int o = 0; o < nout; ++o) F[OgGen::G_ROW(g, o) + k] = out[o];
            }
            continue;
        }
        const int phase = OgGen::G_PHASE(g), mv0 = OgGen::G_MV0(g), nmv = OgGen::G_NMV(g);
        const double* Dp = D[phase];
        std::vector<S> operand((size_t)nmv * L), y((size_t)nmv * L);
        for (int s = 0; s < nmv; ++s)
            for (int l = 0; l < L; ++l) operand[(size_t)s * L + l] = OgGen::mv_operand(mv0 + s, l, xa, cv);
        for (int s = 0; s < nmv; ++s)
            for (int k = 0; k < L; ++k) {
                S acc = S(0.0);
                for (int l = 0; l < L; ++l) acc = fma_chain(operand[(size_t)s * L + l], Dp[(size_t)k * L + l], acc);
                y[(size_t)s * L + k] = acc;
            }
        S yk[OgGen::MAX_NMV];
        for (int k = 0; k < L; ++k) {
            for (int s = 0; s < nmv; ++s) yk[s] = y[(size_t)s * L + k];
            OgGen::group_eval(g, k, xa, (const S*)yk, cv, out);
            for (int o = 0; o < nout; ++o) F[OgGen::G_ROW(g, o) + k] = out[o];
        }
    }
}

void eval_into(const double* x, const double* const* D, const double* cv, double* F) {
    eval_generic(XVec{x}, D, cv, F);
}

}  // namespace

extern "C" {

void twin_dims(int* n, int* m, int* m_eq, int* m_ineq) {
    *n = OgGen::N_VAR;
    *m = OgGen::M;
    *m_eq = OgGen::M_EQ;
    *m_ineq = OgGen::M_INEQ;
}

void twin_eval(const double* x, const double* const* D, const double* cv, double* F) {
    eval_into(x, D, cv, F);
}

// JT: ncols x m, row r = FD column cols[r].  F0 (m) receives F(x).
void twin_sweep(const double* x, const double* h, const double* const* D, const double* cv,
                const int* cols, int ncols, double* F0, double* JT) {
    const int n = OgGen::N_VAR, m = OgGen::M;
    eval_into(x, D, cv, F0);
    std::vector<double> x1(x, x + n), F1(m);
    for (int r = 0; r < ncols; ++r) {
        const int i = cols[r];
        x1[i] += h[i];
        const double dx = x1[i] - x[i];
        eval_into(x1.data(), D, cv, F1.data());
        for (int q = 0; q < m; ++q) JT[(size_t)r * m + q] = (F1[q] - F0[q]) / dx;
        x1[i] = x[i];
    }
}

// Exact Jacobian (forward-mode derivatives of the same generated code): JT row r = dF/dx_{cols[r]}.
void twin_exact(const double* x, const double* const* D, const double* cv, const int* cols, int ncols,
                double* F0, double* JT) {
    const int m = OgGen::M;
    eval_into(x, D, cv, F0);
    std::vector<ogdual> F1(m);
    for (int r = 0; r < ncols; ++r) {
        eval_generic(XDual{x, cols[r]}, D, cv, F1.data());
        for (int q = 0; q < m; ++q) JT[(size_t)r * m + q] = F1[q].d;
    }
}

}  // extern "C"
