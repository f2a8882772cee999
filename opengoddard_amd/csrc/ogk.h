// ogk.h -- internal ABI between libogpsx.so (generic runtime) and a compiled callback module
// libogk_<hash>.so (sweep kernels instantiated for one traced problem).  Not part of the public
// C ABI (include/ogpsx.h); both sides are built from this tree.
#pragma once
#include <stdint.h>

#define OGK_ABI 11
#define OGK_MAX_PHASE 32

// MFMA operand image of a differentiation matrix D (N x N, row-major [k][l]) for
// v_mfma_f64_16x16x4_f64 with the state vectors as A and D^T as B:
//   frag[(nt*KS + ks)*64 + lane] = D[16*nt + (lane & 15)][4*ks + (lane >> 4)]   (0 when padded)
// nt in [0, NT) output-node tiles, ks in [0, KS) steps over the contraction index l.
static inline int ogk_frag_nt(int N) { return (N + 15) / 16; }
static inline int ogk_frag_ks(int N) { return (N + 3) / 4; }
static inline long ogk_frag_size(int N) { return (long)ogk_frag_nt(N) * ogk_frag_ks(N) * 64; }
static inline void ogk_frag_pack(int N, const double* D, double* frag) {
    const int NT = ogk_frag_nt(N), KS = ogk_frag_ks(N);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = 16 * nt + (lane & 15), l = 4 * ks + (lane >> 4);
                frag[((long)nt * KS + ks) * 64 + lane] = (k < N && l < N) ? D[(long)k * N + l] : 0.0;
            }
}

#define OGK_OTHER_PART (-4242)   /* ogk_launch: this mode's kernels are in the other part of a two-part module */

typedef struct ogk_info {
    int32_t abi;
    int32_t n, m, m_eq, m_ineq;
    int32_t n_phase, n_mv, n_groups, n_cvec;
    int32_t n_y0;           // doubles of scratch for the unperturbed collocation products
    int32_t phase_nodes[OGK_MAX_PHASE];
    int32_t n_eval_blocks;  // evaluation workgroups of one launch (what the mode-5 ticket counts)
    int32_t fused_ok;       // mode 5 runs as ONE launch on this module (its LDS fits); else the caller uses modes 0 + 1
} ogk_info;

typedef struct ogk_args {
    const double* x0;       // [n] decision vector (device)
    const double* h;        // [n] signed FD steps (device; unused for a plain evaluation)
    const double* dfrag;    // packed D fragments of all phases (device)
    const double* cvec;     // constant table (device, may be NULL when n_cvec == 0)
    double* f0;             // [m] F(x0): written by mode 0, read by modes 1 and 2
    double* y0;             // [n_y0] base collocation products   (written by mode 0)
    double* xop;            // [n_y0] base collocation operands   (written by mode 0)
    double* t0;             // [m] base dynamics terms of defect rows (written by mode 0)
    double* z;              // [m] F0 - F0: 0, or NaN for non-finite rows (written by mode 0)
    int* nonfinite;         // number of non-finite rows of F(x0): counted by mode 0, read by mode 1
    int* nonfinite_next;    // the slot the *next* evaluation counts into (mode 0 zeroes it)
    unsigned* ready;        // mode 5: ticket the evaluation workgroups count into; the last one resets it
    double* jt;             // [(col_hi-col_lo) * m] transposed Jacobian rows (mode 1)
    // Persistent-zero output (og_jt_register_dev, include/ogpsx.h).  jt_sparse != 0: the structural zeros of
    // `jt` are known to hold zeros already, so the sweep writes ONLY the positions that can be non-zero (row
    // items and collocation tiles).  The exception is kept on the device, because the asynchronous entry
    // points never learn it - and entirely on the device, so that the launch arguments are the same for
    // every sweep into a buffer (a captured hipGraph can be replayed): jt_launches counts the launches into the
    // buffer; a sweep whose F(x0) has non-finite rows fills its rows completely (NaN where dense FD gives NaN)
    // and stores its own launch number into *jt_state; the next sweep finds *jt_state == its number - 1 and
    // fills completely once more (zeros), which cleans the buffer.  Who counts: the last evaluation workgroup
    // of a mode-5 launch (one thread); thread 0 of a mode-0 launch when jt_bump is set (the evaluation that
    // precedes a mode 1 / 2 / 4 launch); a one-thread kernel (mode 10) before a lone mode-1 launch.
    int32_t jt_sparse;
    int32_t jt_bump;        // mode 0: count one launch into *jt_launches; mode 9 (unpack of a rank that owns no
                            // columns): mark a NaN fill in *jt_state when F(x0) had non-finite rows
    uint32_t* jt_launches;  // launches into the registered buffer so far
    uint32_t* jt_state;     // number of the last launch into the buffer that left NaN fill behind
    // mode 5 counts the non-finite rows of F(x0) in *nonfinite (zero between launches: its last evaluation
    // workgroup moves the total to *nonfinite_result and clears the counter)
    int* nonfinite_result;
    int32_t col_lo, col_hi; // FD columns handled by this launch
    // Packed non-zeros (modes 6-9).  The static pattern of J_T is the tracer's: column j can be non-zero in
    // the collocation block its state slice owns (N consecutive rows) and at its row items, in that order.
    const int64_t* poff;    // [n] offset of column j's entries in the packed array (modes 7-9)
    const int64_t* pind;    // [n+1] the pattern's own prefix sums (modes 8, 9): column j has pind[j+1]-pind[j] entries
    const int32_t* prow;    // [nnz] row index of every pattern entry, flat (modes 8, 9): no table walks there
    int32_t* pint;          // mode 6: [n] entries per column (out); mode 7: row index of every packed entry (out)
    double* pvals;          // packed values: mode 8 writes them, mode 9 reads them
    double* ptail;          // mode 8: m + 1 doubles that receive F(x0) and the count of non-finite rows, or NULL
    int32_t ulo, uhi;       // mode 9: columns to scatter; those inside [col_lo, col_hi) are this rank's own: skipped
    double* trace;          // -DOGK_TRACE builds: [workgroup][8 wavefronts][8] phase stamps (else unused)
    int64_t dfrag_off[OGK_MAX_PHASE];
} ogk_args;

#ifdef __cplusplus
extern "C" {
#endif
// exported by every callback module
int ogk_get_info(ogk_info* out);
// mode 0: evaluate F(x0) into f0 (+ scratch y0/t0/z).  mode 1: structured FD sweep over
// [col_lo, col_hi) into jt (needs mode 0's outputs at the same x0).  mode 2: dense FD sweep.
// mode 3 / 4: exact Jacobian, dense / structured (needs mode 0).  mode 5: modes 0 + 1 in one launch.
// mode 6 / 7: the static pattern (entries per column / row indices).  mode 8: gather the pattern entries of
// the columns [col_lo, col_hi) of `jt` into pvals.  mode 9: scatter pvals into the rows [ulo, uhi) of a full
// matrix `jt` (row 0 = column 0), filling those rows from z first when F(x0) has non-finite rows or the
// previous step left such a fill behind.  mode 10: count one launch into *jt_launches.
// Only enqueues kernels on `stream`; returns a hipError_t value (0 = success).
int ogk_launch(const ogk_args* args, int mode, void* stream);
#ifdef __cplusplus
}
#endif
