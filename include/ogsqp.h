/* ogsqp.h - C ABI of the MI355X-native SQP core.
 *
 * What it replaces.  The reference hands the NLP to SciPy (`optimize.py:723-749`), whose Fortran
 * `slsqp` (`scipy:_slsqp_py.py:427-432`, Kraft 1988) spends 8 s per major iteration at n = 1442 in
 * its LSQ/LSEI/LSI/LDP/NNLS chain while the callbacks take 0.03 s on the GPU.  These entry points
 * are the part of that chain that touches O(n^2) data; the O(n) book-keeping of the major
 * iteration (merit function, line search, convergence tests) stays on the host
 * (`opengoddard_amd/sqp.py`).  Every buffer that is O(n^2) - the transposed FD Jacobian written
 * by `og_fd_sweep_dev`, the quasi-Newton factor, the QP work matrices - lives in HBM and never
 * crosses PCIe.
 *
 *   slsqp_optmz.f                                   here
 *   ----------------------------------------------  ------------------------------------------
 *   label 110  "reset BFGS matrix" (L = I, D = I)    og_qp_reset            (Z = I)
 *   lsq -> lsei -> lsi -> ldp -> nnls                og_qp_solve_dev        (one QP subproblem)
 *   label 140-150 augmented problem (n+1 vars)       og_qp_solve_dev(augmented=1, rho)
 *   label 260-320 + ldl(): damped BFGS on L D L'     og_qp_bfgs             (product form on Z)
 *   v(i) = g(i) - sum_j a(j,i) r(j)  (label 160)     og_jt_times
 *
 * The quasi-Newton matrix is carried as an inverse factor Z with B^-1 = Z Z' (n x n, row-major,
 * leading dimension n+1); see DESIGN.md section 9 for the method.
 *
 * All functions return 0 on success; otherwise a non-zero code and og_qp_last_error() describes
 * it.  One call in flight per handle.  Host pointers unless the name says `_dev`.
 */
#ifndef OGSQP_H
#define OGSQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGSQP_ABI_VERSION 1

typedef struct og_qp_s* og_qp_handle;

/* QP status, numbered like the `mode` of lsq()/slsqp() */
#define OG_QP_SOLVED 1
#define OG_QP_TOO_MANY_EQ 2      /* more equality constraints than variables */
#define OG_QP_ITERATION_LIMIT 3  /* "More than 3*n iterations in LSQ subproblem" */
#define OG_QP_INCOMPATIBLE 4     /* "Inequality constraints incompatible" */
#define OG_QP_SINGULAR_C 6       /* "Singular matrix C in LSQ subproblem" */

/* n variables, m_eq equality rows, m_ineq inequality rows (fixed for the handle's lifetime). */
int og_qp_create(int32_t abi_version, int32_t device, int32_t n, int32_t m_eq, int32_t m_ineq,
                 og_qp_handle* out);
void og_qp_destroy(og_qp_handle qp);

/* Z = I. */
int og_qp_reset(og_qp_handle qp);

/* Copy the factor to / from the host (n x n, row-major, contiguous); tests and checkpoints. */
int og_qp_get_factor(og_qp_handle qp, double* Z);
int og_qp_set_factor(og_qp_handle qp, const double* Z);

/* One QP subproblem:   min 1/2 d'B d + g'd   s.t.  a_j'd + c_j  = 0   (j <  m_eq)
 *                                                  a_j'd + c_j >= 0   (j >= m_eq)
 *                                                  dl <= d <= du      (non-finite: no bound)
 *
 * d_jt      device, n rows with leading dimension ld: row i holds [dcost/dx_i, da_1/dx_i, ...]
 *           exactly as og_fd_sweep_dev wrote it (transposed Jacobian, column 0 = cost gradient).
 *           Column 0 is not read; the cost gradient comes through g (a user-supplied analytic
 *           cost derivative replaces the FD one, `optimize.py:730-733`).
 * augmented 0: the QP above.  1: the relaxed problem of slsqp label 140-150 in n+1 variables:
 *           a_j'd + c_j (1 - delta) = 0 / >= max(-c_j,0) delta - ..., 0 <= delta <= 1, with
 *           rho (= l(n3), 100 then x10 per retry) on the diagonal of E.  d, bound_mult then
 *           have n+1 entries.  The stored factor is left untouched.
 * d         step (n or n+1), clipped to [dl,du] like lsq() does
 * mult      m multipliers r (equalities: free sign; inequalities >= 0), grad L = g - A'r - bound_mult
 * bound_mult  n (n+1) multipliers of the bounds, > 0 lower active, < 0 upper active
 * status    OG_QP_*;  iterations: active-set iterations spent
 *
 * On OG_QP_SOLVED with augmented == 0 the stored factor is replaced by Z Q (the same B; columns
 * rotated by the elimination of the equalities), which is what og_qp_bfgs expects next.
 */
int og_qp_solve_dev(og_qp_handle qp, const double* d_jt, int64_t ld, const double* g,
                    const double* c, const double* dl, const double* du, int32_t augmented,
                    double rho, double* d, double* mult, double* bound_mult, int32_t* status,
                    int32_t* iterations, void* hip_stream);

/* Same with the Jacobian on the host: A is m x n row-major (a_j' in row j).  Staged through a
 * scratch buffer; used by the tests and by callers without a device-resident Jacobian. */
int og_qp_solve(og_qp_handle qp, const double* A, const double* g, const double* c,
                const double* dl, const double* du, int32_t augmented, double rho, double* d,
                double* mult, double* bound_mult, int32_t* status, int32_t* iterations);

/* The rows of the inequality part that were active (multiplier > 0 or just added) at the solution of the last
 * solved subproblem, in this numbering: j < m_ineq general inequality j; m_ineq + 2 i lower bound of variable i;
 * m_ineq + 2 i + 1 its upper bound (i = n: the relaxation variable of an augmented solve).  The next solve on the
 * handle starts its active-set method from them (rows appended to the LQ sweep of the equalities, multipliers
 * checked, negative ones dropped - DESIGN.md section 9): the solution is the same, late in an SQP run it is reached
 * in a handful of changes instead of one per active row.  og_qp_set_active replaces the list (count = 0: the next
 * solve starts from the empty set, like lsq()'s NNLS does every time); OGSQP_WARM=0 in the environment turns the
 * warm start off for a whole process.  No reference counterpart (slsqp_optmz.f keeps no state between calls of lsq). */
int og_qp_get_active(og_qp_handle qp, int32_t* ids, int32_t capacity, int32_t* count);
int og_qp_set_active(og_qp_handle qp, const int32_t* ids, int32_t count);

/* How many subproblems of this handle were solved twice: the default kernels of the LQ sweep (look-ahead inside one
 * launch) and of the triangular solves (one chained launch) hand data between workgroups of a launch with bounded
 * waits; when a wait gives up - a device shared with other streams or tenants - og_qp_solve(_dev) re-runs the
 * subproblem with the separate-launch forms (same result to rounding) instead of failing.  OGSQP_SPIN_LIMIT=<n>
 * shortens the bound (tests force the path with 1).  No reference counterpart. */
int og_qp_recoveries(og_qp_handle qp, int32_t* count);

/* Round 6.  Where the stack of constraint rows and the inverse of the active triangle fit the chip's LDS (up to 4096
 * rows, one wavefront each, 16 per compute unit: BASELINE.json's C3 and C4), the whole active-set loop of a
 * subproblem is ONE launch (k_rows_resident, csrc/ogsqp_resident.h) instead of two launches per change.
 * *launches = such launches so far, *changes = active-set changes they made (0 / 0: the two-launch form serves this
 * handle - rows too many or too long, or OGSQP_RESIDENT=0 in the environment).  No reference counterpart (SciPy's
 * lsq() is a scalar Fortran loop, scipy/optimize/_slsqp_py.py:427-432 -> slsqp_optmz.f). */
int og_qp_resident_stats(og_qp_handle qp, int64_t* launches, int64_t* changes);

/* Powell-damped BFGS (slsqp label 260-320) on the factor: s = step, eta = change of the
 * Lagrangian gradient, Bs = B s.  *reset_needed = 1 when the update is undefined (s'Bs or the
 * damped s'eta not positive) and the factor was left unchanged. */
int og_qp_bfgs(og_qp_handle qp, const double* s, const double* eta, const double* Bs,
               int32_t* reset_needed);

/* out[i] = sum_k d_jt[i*ld + k] * coef[k], k < 1+m: with coef = [1, -r] this is the gradient of the
 * Lagrangian v = g - A'r (slsqp label 160 / 270); with coef = e_0 it extracts the cost gradient. */
int og_jt_times(og_qp_handle qp, const double* d_jt, int64_t ld, const double* coef, double* out,
                void* hip_stream);

const char* og_qp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OGSQP_H */
