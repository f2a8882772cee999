// og_lgl.h -- Legendre-Gauss-Lobatto nodes / weights / differentiation matrix, float64,
// written once for host and device (same IEEE operations => same bits on x86-64 and gfx950).
//
// Replaces:
//   Problem._nodes_LGL                    reference OpenGoddard/optimize.py:183-187
//   Problem._weight_LGL                   reference OpenGoddard/optimize.py:189-195
//   Problem._differentiation_matrix_LGL   reference OpenGoddard/optimize.py:197-213
// The reference obtains the interior nodes as Gauss-Jacobi(1,1) roots from SciPy (Golub-Welsch
// eigenproblem + one Newton step) and P_{N-1} from scipy.special.lpn.  Here the interior nodes
// are the roots of P'_{N-1} by Newton iteration on the three-term recurrence, started from the
// Chebyshev-Gauss-Lobatto points; both constructions converge to the same numbers (difference
// vs. the reference: tau <= 2e-16 abs, w and D <= 1e-12 relative; tests/test_lgl.py).
// Index order is the reference's: ascending tau, tau[0] = -1, tau[N-1] = +1, D row-major [k][l].
#pragma once
#include "og_math.h"

namespace oglgl {

// P_n(x) and P_{n-1}(x) by the Bonnet recurrence (j+1) P_{j+1} = (2j+1) x P_j - j P_{j-1}.
OG_HD void legendre_pair(int n, double x, double* pn, double* pnm1) {
    double p0 = 1.0, p1 = x;
    if (n == 0) { *pn = 1.0; *pnm1 = 0.0; return; }
    for (int j = 1; j < n; ++j) {
        double p2 = ((double)(2 * j + 1) * x * p1 - (double)j * p0) / (double)(j + 1);
        p0 = p1;
        p1 = p2;
    }
    *pn = p1;
    *pnm1 = p0;
}

OG_HD double legendre(int n, double x) {
    double a, b;
    legendre_pair(n, x, &a, &b);
    return a;
}

// k-th LGL node of an N-point rule (0 <= k < N), ascending.  Antisymmetric by construction:
// node(N-1-k) == -node(k) bit for bit, centre node of an odd rule == 0.
OG_HD double node(int N, int k) {
    if (k == 0) return -1.0;
    if (k == N - 1) return 1.0;
    if (2 * k == N - 1) return 0.0;
    const int n = N - 1;                       // degree whose derivative vanishes at the nodes
    const int kk = (2 * k < N - 1) ? (N - 1 - k) : k;   // solve on the positive half
    const double pi = 3.14159265358979311600e+00;
    // Chebyshev-Gauss-Lobatto start; ascending index kk maps to -cos(pi*kk/n) > 0
    double x = -ogm::cos_(pi * (double)kk / (double)n);
    for (int it = 0; it < 100; ++it) {
        double pn, pnm1;
        legendre_pair(n, x, &pn, &pnm1);
        double om = (1.0 - x) * (1.0 + x);
        double d1 = (double)n * (pnm1 - x * pn) / om;                           // P'_n
        double d2 = (2.0 * x * d1 - (double)n * (double)(n + 1) * pn) / om;     // P''_n
        double dx = d1 / d2;
        x -= dx;
        if (ogm::fabs_(dx) <= 2.0e-16 * ogm::fabs_(x)) break;
    }
    return (kk == k) ? x : -x;
}

// weight, same operation order as the reference: 2 / (N*(N-1) * P_{N-1}(tau_k)^2)
OG_HD double weight(int N, double pk) {
    return 2.0 / ((double)(N * (N - 1)) * (pk * pk));
}

// D[k][l]; pk, pl = P_{N-1}(tau_k), P_{N-1}(tau_l)
OG_HD double dmat(int N, int k, int l, double tk, double tl, double pk, double pl) {
    if (k != l) return pk / pl / (tk - tl);
    if (k == 0) return -(double)(N * (N - 1)) * 0.25;
    if (k == N - 1) return (double)(N * (N - 1)) * 0.25;
    return 0.0;
}

}  // namespace oglgl
