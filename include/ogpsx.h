/* ogpsx.h -- C ABI of libogpsx.so, the MI355X (gfx950) pseudospectral NLP evaluation engine.
 *
 * This is the drop-in boundary for the hot path of istellartech/OpenGoddard (SURVEY.md
 * section 8(b)).  The reference has no FFI: its hot path is Python closures that SciPy's SLSQP
 * calls 3n+2 times per major iteration.  Each entry point below names the reference code it
 * replaces (paths relative to the reference checkout; "scipy:" = SciPy 1.15.3's
 * scipy/optimize).  All functions return 0 on success and a non-zero code otherwise;
 * og_last_error() then describes the failure (thread-local string).  Buffers are caller-owned;
 * the library owns only device scratch tied to a handle.  Calls on one handle must not overlap;
 * host-pointer variants block until results are in the caller's buffers.
 */
#ifndef OGPSX_H
#define OGPSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_ABI_VERSION 1

typedef struct og_problem_s* og_handle;

/* ---- LGL discretisation ------------------------------------------------------------------
 * Replaces Problem._nodes_LGL / _weight_LGL / _differentiation_matrix_LGL
 * (OpenGoddard/optimize.py:183-213).  tau[N], w[N], D[N*N] row-major.  N >= 3.
 * og_lgl runs on the host; og_lgl_dev runs the HIP kernel on the current device and writes to
 * device pointers (stream may be NULL).  Both produce identical bits. */
int og_lgl(int32_t N, double* tau, double* w, double* D);
int og_lgl_dev(int32_t N, double* d_tau, double* d_w, double* d_D, void* hip_stream);

/* ---- finite-difference step rule -----------------------------------------------------------
 * Replaces the step selection SciPy applies before the column loop (scipy:_numdiff.py:500-515
 * with _adjust_scheme_to_bounds '1-sided', scipy:_numdiff.py:44-70; absolute step
 * 1.4901161193847656e-08 from scipy:_slsqp_py.py:33).  lb/ub use -inf/+inf for "no bound".
 * Writes the signed step h[n]. */
int og_fd_step(int32_t n, const double* x, const double* lb, const double* ub, double* h);

/* ---- problem handle ------------------------------------------------------------------------
 * A handle binds one compiled callback module (the traced dynamics / equality / inequality /
 * cost of one Problem, built by opengoddard_amd.codegen) to a device.  Replaces the closure
 * construction in Problem.solve (OpenGoddard/optimize.py:670-733). */
typedef struct og_desc {
    int32_t abi_version;        /* OG_ABI_VERSION */
    int32_t device;             /* HIP device ordinal */
    int32_t n;                  /* decision variables */
    int32_t m_eq;               /* equality rows (user + defects + knots) */
    int32_t m_ineq;             /* inequality rows */
    int32_t n_phase;
    const int32_t* nodes;       /* [n_phase] LGL nodes per phase */
    const double* const* D;     /* [n_phase] row-major N x N matrices, or NULL: use og_lgl */
    const double* cvec;         /* constant table referenced by the module (may be NULL) */
    int32_t n_cvec;
    const char* module_path;    /* path of the compiled callback module (libogk_<hash>.so) */
} og_desc;

int og_problem_create(const og_desc* desc, og_handle* out);
void og_problem_destroy(og_handle h);

/* Sizes of a handle: n, m = 1 + m_eq + m_ineq, m_eq, m_ineq. */
int og_problem_dims(og_handle h, int32_t* n, int32_t* m, int32_t* m_eq, int32_t* m_ineq);

/* How og_fd_sweep(_dev) runs on this handle: 5 = evaluation and structured sweep in ONE launch (ogk_fused),
 * 1 = the same two kernels as two launches, 2 = evaluation + literal dense sweep (validation).  Chosen at
 * og_problem_create: OGPSX_SWEEP=fused|split|dense, default by size (one launch while the Jacobian is at
 * most 100 MB - there the kernel boundary is a third of the step; above, the two launches are faster).
 * No reference counterpart: SciPy's approx_derivative has one way to run (scipy/optimize/_numdiff.py:584-625). */
int og_sweep_mode(og_handle h);

/* ---- single evaluation ---------------------------------------------------------------------
 * F(x) = [cost | c_eq | c_ineq], m doubles.  Replaces one call each of cost_add, equality_add
 * and the user inequality (OpenGoddard/optimize.py:670-709, 723-728). */
int og_eval(og_handle h, const double* x, double* F);

/* ---- forward-difference sweep --------------------------------------------------------------
 * Transposed Jacobian rows for decision-vector columns [col_lo, col_hi):
 *     JT[(j - col_lo) * m + r] = (F_r(x + h_j e_j) - F_r(x)) / ((x_j + h_j) - x_j)
 * i.e. exactly SciPy's J_transposed (scipy:_numdiff.py:584-625) for the stacked function
 * [cost | c_eq | c_ineq]; column 0 of a row is the cost gradient entry, columns 1..m_eq the
 * equality Jacobian, the rest the inequality Jacobian.  Also returns F(x) in F0 (may be NULL).
 * Replaces the 3n+2 Python callback evaluations of one SLSQP major iteration
 * (scipy:_slsqp_py.py:299-313, 438-440). */
int og_fd_sweep(og_handle h, const double* x, const double* hstep,
                int32_t col_lo, int32_t col_hi, double* JT, double* F0);

/* Device-pointer variants: all pointers are device memory on the handle's device (for example
 * torch.Tensor.data_ptr()), hip_stream is a hipStream_t (NULL = default stream).  Asynchronous:
 * they only enqueue work.  d_F0 must hold m doubles, d_JT (col_hi-col_lo)*m doubles. */
int og_eval_dev(og_handle h, const double* d_x, double* d_F, void* hip_stream);
int og_fd_sweep_dev(og_handle h, const double* d_x, const double* d_hstep,
                    int32_t col_lo, int32_t col_hi, double* d_JT, double* d_F0,
                    void* hip_stream);

/* Columns only: like og_fd_sweep_dev but d_F0 is an *input* that must already hold F(x) from
 * the most recent og_eval_dev on this handle at the same x (that call also refreshes the
 * handle's sweep scratch).  One kernel launch; this is the unit bench.py times for the
 * roofline figure.
 *
 * Environment: OGPSX_SWEEP=dense (read when a handle is created) selects the literal dense
 * sweep - every row re-evaluated for every column - instead of the default structured sweep
 * that only re-evaluates rows reading the perturbed variable.  Results are identical. */
int og_fd_columns_dev(og_handle h, const double* d_x, const double* d_hstep,
                      int32_t col_lo, int32_t col_hi, double* d_JT, const double* d_F0,
                      void* hip_stream);

/* ---- persistent-zero output buffers ------------------------------------------------------------
 * Most of J_T is structurally zero (a row of F reads a handful of decision variables; the pattern is
 * fixed by the traced callbacks), and SciPy's dense loop recomputes those zeros as (F0 - F0)/dx for
 * every column of every sweep (scipy:_numdiff.py:592-620).  A caller that keeps ONE output buffer per
 * block of columns can register it: og_jt_register_dev zero-fills d_JT ((col_hi-col_lo)*m doubles,
 * enqueued on hip_stream) and from then on og_fd_sweep_dev / og_fd_columns_dev / og_jacobian_exact_dev
 * called with exactly (d_JT, col_lo, col_hi) write only the entries that can be non-zero.  The result
 * in the buffer is the same matrix, entry for entry, as without registration - including sweeps at
 * points where F(x) has non-finite rows (those rows become NaN in every column, as dense differencing
 * makes them; the library re-cleans the buffer on the next sweep by itself).  Contract: between
 * sweeps the caller only reads the buffer; work on it must be stream-ordered after the registration.
 * At most 63 registrations per handle; the host-pointer entry points register their own staging
 * buffer.  No reference counterpart (the reference allocates a fresh dense array per sweep). */
int og_jt_register_dev(og_handle h, double* d_JT, int32_t col_lo, int32_t col_hi, void* hip_stream);
int og_jt_unregister_dev(og_handle h, double* d_JT);

/* ---- exact Jacobian (opt-in, changes the numbers SLSQP sees) ---
 * Same layout as the sweep: JT[(j - col_lo) * m + r] = dF_r/dx_j, but by forward-mode
 * differentiation of the traced callbacks (no step h, no subtraction, no FD noise; where a callback
 * is not differentiable - np.where / maximum at the switch, a table knot - the derivative of the
 * branch F(x) itself takes).  Replaces the same 3n+2 evaluations of `approx_derivative`
 * (`scipy/optimize/_slsqp_py.py:299-313`) as og_fd_sweep; F0 as there. */
int og_jacobian_exact(og_handle h, const double* x, int32_t col_lo, int32_t col_hi, double* JT, double* F0);
int og_jacobian_exact_dev(og_handle h, const double* d_x, int32_t col_lo, int32_t col_hi, double* d_JT,
                          double* d_F0, void* hip_stream);

/* ---- diagnostics ---------------------------------------------------------------------------*/
const char* og_last_error(void);
/* Developer diagnostics: phase stamps (s_memrealtime ticks) written by kernel modules compiled with
 * -DOGK_TRACE=1 on a handle created with OGPSX_TRACE=1 in the environment; 8 doubles per wavefront,
 * 8 wavefronts per workgroup, in grid order (tools/trace_fused.py).  Reads `count` doubles and clears
 * the buffer.  Fails on an ordinary handle. */
int og_trace_read(og_handle h, double* out, int64_t count);
int og_device_count(void);     /* HIP devices visible to the library (0 without a GPU) */

#ifdef __cplusplus
}
#endif
#endif /* OGPSX_H */
