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
 * og_problem_create: OGPSX_SWEEP=fused|split|dense; the default is 5 at every size.  The one-launch form writes
 * the non-zeros only, so it needs a persistent-zero output (og_jt_register_dev / _host; the host-pointer entry
 * points register their own staging buffer): a sweep into an unregistered device buffer runs as the two launches
 * of mode 1 from the same handle, with identical results.
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

/* The same for a host matrix handed to og_fd_sweep / og_jacobian_exact: og_jt_register_host zero-fills JT
 * ((col_hi-col_lo)*m doubles) and later calls with exactly (JT, col_lo, col_hi) transfer only the packed
 * non-zeros (plus F and the count of non-finite rows, one pinned copy) and scatter them on the host; a sweep
 * with non-finite rows transfers the dense block, and the next call re-zeroes the matrix first.  Contract as
 * above: the caller only reads JT between calls.  This is what Problem.solve uses: SciPy's SLSQP copies the
 * Jacobian it is handed (scipy:_slsqp_py.py:490-511), so one persistent matrix serves every major iteration.
 * Round 6: the registration page-locks JT and maps it into the device's address space (hipHostRegister; undone by
 * og_jt_unregister_host / og_problem_destroy, which must come before the matrix is freed), and og_fd_sweep's one
 * launch then writes the structural non-zeros straight into JT over PCIe - no packed copy, no host scatter (C3:
 * 0.097 -> 0.047 ms per call, C4 0.21 -> 0.088, C5 0.38 -> 0.17; x and h are read in place too, out of a pinned
 * buffer of the handle, instead of being copied first) - under the same persistent-zero protocol as a
 * device buffer (a NaN fill is cleaned by the next sweep).  Where the host refuses the mapping (or OGPSX_HOST=staged)
 * the packed transfer + scatter above is what runs; the matrix in JT is the same either way. */
int og_jt_register_host(og_handle h, double* JT, int32_t col_lo, int32_t col_hi);
/* Which of the two serves og_fd_sweep into the registered host matrix JT: *path = 1 the mapped matrix (the launch writes
 * it over PCIe), 2 the packed copy + host scatter, 0 not decided yet (the first ten sweeps time both - two warm-ups and
 * three timed calls each - and keep the faster: what scattered device writes into host pages cost is the host's doing;
 * on every box of round 6 mapped won, 0.043-0.052 against 0.10-0.12 ms per sweep at C3), -1 JT is not registered.
 * OGPSX_HOST=mapped | staged in the environment decide without the trial.  No reference counterpart. */
int og_jt_host_path(og_handle h, const double* JT, int32_t* path);
/* Page-locked host memory from the HIP runtime (hipHostMalloc) for a matrix that is going to be registered with
 * og_jt_register_host: it is mapped into the device's address space already and lies in the driver's large fragments, so
 * the mapped path needs a handful of address translations per sweep where a malloc'ed matrix of 4 KB pages needs
 * thousands (the engine's own persistent matrix comes from here).  No reference counterpart. */
int og_pinned_alloc(int64_t bytes, void** out);
int og_pinned_free(void* ptr);
int og_jt_unregister_host(og_handle h, double* JT);

/* ---- static pattern and packed non-zeros ---------------------------------------------------------
 * Which entries of J_T can be non-zero is fixed by the traced callbacks: column j reads the collocation
 * block its state slice owns (N consecutive defect rows) and a list of row items.  og_pattern returns that
 * pattern for the columns [col_lo, col_hi) as CSR over columns: *nnz entries, indptr[col_hi-col_lo+1]
 * (relative to the block), rows[*nnz]; any output pointer may be NULL.  og_pack_dev gathers exactly those
 * entries of a dense block (d_JT: (col_hi-col_lo) x m) into d_vals (*nnz doubles, pattern order);
 * og_unpack_dev scatters them back into a dense block whose other entries it leaves alone.  No reference
 * counterpart: the reference's Jacobian is dense (scipy:_numdiff.py:587). */
int og_pattern(og_handle h, int32_t col_lo, int32_t col_hi, int64_t* nnz, int64_t* indptr, int32_t* rows);
int og_pack_dev(og_handle h, const double* d_JT, int32_t col_lo, int32_t col_hi, double* d_vals, void* hip_stream);
int og_unpack_dev(og_handle h, const double* d_vals, int32_t col_lo, int32_t col_hi, double* d_JT, void* hip_stream);

/* ---- column sharding across GPUs ------------------------------------------
 * The n forward-difference columns are independent given x (scipy:_numdiff.py:592-620 has no cross-iteration
 * dependency): rank r of `world` sweeps the block [r*B, min(n, (r+1)*B)), B = ceil(n/world), into ITS rows of
 * a full n x m replica that was zeroed once (register the block: og_jt_register_dev(h, d_JT_full + r*B*m,
 * r*B, ...)), and the ranks exchange only the packed non-zeros: og_shard_plan fixes the layout (*block_vals =
 * doubles per rank's message, padded to the largest block so that one all-gather of equal messages does it);
 * og_shard_pack_dev gathers this rank's non-zeros into d_send (*block_vals doubles); after the all-gather
 * (RCCL ncclAllGather / torch.distributed.all_gather_into_tensor; world * *block_vals doubles in d_recv)
 * og_shard_unpack_dev scatters every other rank's non-zeros into the replica.  When F(x) has non-finite rows,
 * or had in the previous step, the rows of the other ranks are filled from this rank's own F(x) - F(x)
 * first (every rank evaluates the same F(x)), so the replica equals the single-GPU result in that case too.
 * Call order per step on one stream: og_fd_sweep_dev (own block), og_shard_pack_dev, all-gather,
 * og_shard_unpack_dev.  Results are bitwise independent of `world`.  A rank that owns no columns (more ranks
 * than blocks of B columns) evaluates F(x) only (og_shard_sweep_dev / og_eval_dev) and og_shard_unpack_dev keeps
 * the NaN history of its replica in a pair of device words of its own. */
int og_shard_plan(og_handle h, int32_t world, int32_t* block_cols, int64_t* block_vals);
/* The all-gather itself, for one process per GPU, on the caller's stream with no detour through another
 * runtime's streams: rank 0 makes a unique id (og_shard_comm_unique_id: 128 bytes = ncclGetUniqueId), the
 * launcher hands it to every rank (bench.py: torch.distributed.broadcast), every rank calls og_shard_comm_init
 * (ncclCommInitRank on the handle's device; also makes the shard plan for `world`), and per step
 * og_shard_all_gather_dev = ncclAllGather of *block_vals doubles from d_send into d_recv (world * *block_vals).
 * librccl is resolved at run time (the copy already in the process wins).  og_problem_destroy destroys the
 * communicator. */
int og_shard_comm_unique_id(uint8_t* id128);
int og_shard_comm_init(og_handle h, const uint8_t* id128, int32_t rank, int32_t world);
void og_shard_comm_destroy(og_handle h);
/* What RCCL reports about the handle's communicator: ncclCommCount, ncclCommUserRank, ncclCommCuDevice - so that a
 * launcher (bench.py --gpus N) can print that the all-gather really spans N ranks on N devices.  (No reference
 * counterpart: the reference has no multi-device path) */
int og_shard_comm_info(og_handle h, int32_t* nranks, int32_t* rank, int32_t* device);
int og_shard_all_gather_dev(og_handle h, const double* d_send, double* d_recv, void* hip_stream);
int og_shard_pack_dev(og_handle h, int32_t rank, const double* d_JT_block, double* d_send, void* hip_stream);
/* og_fd_sweep_dev of rank's block and og_shard_pack_dev in ONE launch: the sweep kernel stores every non-zero
 * into the block of the replica and into the message (needs the block registered; otherwise it runs the two
 * calls). */
int og_shard_sweep_dev(og_handle h, int32_t rank, const double* d_x, const double* d_hstep, double* d_JT_block,
                       double* d_F0, double* d_send, void* hip_stream);
int og_shard_unpack_dev(og_handle h, int32_t rank, const double* d_recv, double* d_JT_full, void* hip_stream);

/* ---- several GPUs of one node from one process ---------------------------------------------------
 * og_comm_init(G, devs) names the devices the sharded handles use and brings up one RCCL communicator per
 * device (ncclCommInitAll; librccl is loaded at run time - a copy already in the process, e.g. torch's, is
 * reused).  OGPSX_GATHER=peer exchanges the packed blocks with hipMemcpyPeerAsync instead (the reported
 * alternative of SURVEY.md section 8(e); it also allows listing one device several times, which is how the
 * single-GPU test box exercises G > 1), OGPSX_GATHER=rccl makes a missing librccl an error.
 * og_multi_create makes one sub-handle, one stream and one zero-initialised n x m replica of J_T per device
 * (desc->device is ignored).  og_multi_fd_sweep = og_fd_sweep over all n columns, the columns split in G
 * contiguous blocks, the packed non-zeros exchanged by one grouped ncclAllGather over xGMI, the host result
 * (JT: n x m, may be registered with og_multi_jt_register_host; F0 may be NULL) taken from device 0.
 * og_multi_fd_sweep_enqueue does the same without the download and without synchronising: afterwards every
 * device's replica (og_multi_replica_dev: device pointers + the device's stream) holds the whole matrix, for
 * device-side consumers such as the SQP core.  Bitwise the same matrix as a single-device sweep for any G.
 * Replaces the same 3n+2 callback evaluations as og_fd_sweep (scipy:_slsqp_py.py:299-313). */
typedef struct og_multi_s* og_multi;
int og_comm_init(int32_t G, const int32_t* devs);
void og_comm_finalize(void);
int og_comm_size(void);          /* devices named by the last og_comm_init (0: none) */
int og_comm_uses_rccl(void);     /* 1: ncclAllGather, 0: peer copies */
int og_multi_create(const og_desc* desc, og_multi* out);
void og_multi_destroy(og_multi mh);
int og_multi_devices(og_multi mh);
int og_multi_eval(og_multi mh, const double* x, double* F);
int og_multi_fd_sweep(og_multi mh, const double* x, const double* hstep, double* JT, double* F0);
int og_multi_fd_sweep_enqueue(og_multi mh, const double* x, const double* hstep);
int og_multi_synchronize(og_multi mh);
int og_multi_replica_dev(og_multi mh, int32_t g, double** d_JT_full, double** d_F0, void** hip_stream);
int og_multi_jt_register_host(og_multi mh, double* JT);

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
/* Diagnostics: synchronise `device` and copy `bytes` bytes of its memory to the host (tests read the replicas of
 * a multi-device handle with it). */
int og_device_read(int32_t device, const void* d_src, void* dst, int64_t bytes);
/* Developer diagnostics: phase stamps (s_memrealtime ticks) written by kernel modules compiled with
 * -DOGK_TRACE=1 on a handle created with OGPSX_TRACE=1 in the environment; 8 doubles per wavefront,
 * 8 wavefronts per workgroup, in grid order (tools/trace_fused.py).  Reads `count` doubles and clears
 * the buffer.  Fails on an ordinary handle. */
int og_trace_read(og_handle h, double* out, int64_t count);
int og_device_count(void);     /* HIP devices visible to the library (0 without a GPU) */
/* Measurement aid: enqueue `count` launches of a kernel that does nothing, `blocks` x `threads`, on `hip_stream`.
 * Timed between two events it gives the launch floor of this box for a grid of the sweep's geometry - what any
 * one-launch step costs before it computes anything (bench.py: roofline.latency_floor_us). */
int og_probe_launch(int32_t blocks, int32_t threads, int32_t count, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* OGPSX_H */
