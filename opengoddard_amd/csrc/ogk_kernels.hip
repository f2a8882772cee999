// ogk_kernels.hip -- hand-written gfx950 sweep kernels, instantiated for one traced problem.
//
// Compiled once per problem as  hipcc --offload-arch=gfx950 -ffp-contract=off -DOG_GEN_HEADER=...
// The generated header only supplies pointwise device functions (OgGen::group_eval,
// OgGen::mv_operand) and small index tables; everything about parallel decomposition, LDS,
// MFMA and memory traffic is in this file.
//
// What the kernels replace:
//   a7  solve.equality_add     OpenGoddard/optimize.py:670-698   collocation defects + user rows
//   a10 solve.cost_add         OpenGoddard/optimize.py:700-709
//   a14 _dense_difference      scipy/optimize/_numdiff.py:584-625 forward-difference column loop
//
// Forward-difference column j evaluates F at x0 + h_j e_j.  The stack of n perturbed vectors
// ((n+1) x n doubles in the reference's formulation) is never materialised: every read of the
// decision vector goes through XCol, which returns x0[i] except at i == j.
//
// One kernel, ogk_sweep, with two kinds of workgroup.
// Collocation workgroups (defect_body): one per (phase, 16-node output tile, 64 FD columns);
//   each of its 4 wavefronts owns 16 columns and all states of the phase.
//   - the D-matrix panel for the node tile is staged in LDS in MFMA operand order (ogk.h),
//     shared by the 4 waves and by all states;
//   - the unperturbed collocation operands x~_s = (p_s*u)/u are computed once per workgroup
//     into LDS; a column changes at most one element of one state, patched in registers;
//   - Y[s][c][k] = sum_l x~_s,c[l] * D[k][l] runs on v_mfma_f64_16x16x4_f64 (A = state
//     vectors of 16 columns, B = D^T panel), a k-ordered fma chain per output, identical to
//     the oracle's loop;
//   - the epilogue evaluates the phase's traced dynamics at each (column, node) the lane
//     holds, forms defect = Y - (tf-t0)/2 * f, and writes the difference quotient straight
//     into the transposed Jacobian (row-major n x m, SciPy's J_transposed).
// Row workgroups (rows_body): cost, user equality / inequality rows and knot rows; one thread per
//   (row item, 8 columns), consecutive lanes = consecutive rows => coalesced J_T stores.
//
// Mode 0 (SWEEP = false) is the same code with no perturbation; it writes F(x0), which mode 1
// subtracts.  Using one code path for base and perturbed values keeps structural zeros of the
// Jacobian exactly 0.0.
#include <hip/hip_runtime.h>
#include "ogk.h"
#include OG_GEN_HEADER

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

struct XCol {
    const double* x0;
    int j;          // perturbed index, -1 for none
    double xj;      // x0[j] + h[j]
    __device__ __forceinline__ double operator()(const int i) const {
        const double v = x0[i];
        return i == j ? xj : v;
    }
};

constexpr int ROWS_COLS_PER_THREAD = 8;

__device__ __forceinline__ int defect_block_to_group(int bx, int* nt_out) {
    for (int g = 0; g < OgGen::N_GROUPS; ++g) {
        if (OgGen::G_KIND(g) != 1) continue;
        const int ntiles = (OgGen::G_LEN(g) + 15) >> 4;
        if (bx < ntiles) { *nt_out = bx; return g; }
        bx -= ntiles;
    }
    *nt_out = 0;
    return -1;
}

template <bool SWEEP>
__device__ __forceinline__ void defect_body(const ogk_args& a, const int bx, const int by,
                                            double* lds) {
    int nt;
    const int g = defect_block_to_group(bx, &nt);
    if (g < 0) return;
    const int N = OgGen::G_LEN(g);
    const int KS = (N + 3) >> 2;
    const int NP = KS * 4;
    const int phase = OgGen::G_PHASE(g);
    const int mv0 = OgGen::G_MV0(g);
    const int nmv = OgGen::G_NMV(g);
    const int nout = OgGen::G_NOUT(g);
    const int tid = (int)threadIdx.x;

    // ---- stage the D panel (MFMA B-operand order) and the base operands in LDS
    double* dpanel = lds;
    double* xt = lds + KS * 64;
    const double* src = a.dfrag + a.dfrag_off[phase] + (long)nt * KS * 64;
    for (int i = tid; i < KS * 64; i += 256) dpanel[i] = src[i];
    const XCol base{a.x0, -1, 0.0};
    for (int i = tid; i < nmv * NP; i += 256) {
        const int s = i / NP, l = i - s * NP;
        xt[i] = (l < N) ? OgGen::mv_operand(mv0 + s, l, base, a.cvec) : 0.0;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    if (!SWEEP && wave != 0) return;
    const int c0 = SWEEP ? a.col_lo + (by * 4 + wave) * 16 : 0;
    if (SWEEP && c0 >= a.col_hi) return;

    // ---- the column this lane feeds into the A operand, and the one operand entry it changes
    int hit_s = -1, hit_l = -1;
    double hit_v = 0.0;
    if (SWEEP) {
        const int ja = c0 + (lane & 15);
        if (ja < a.col_hi) {
            for (int s = 0; s < nmv; ++s) {
                const int leaf = OgGen::MV_LEAF(mv0 + s);
                if (ja >= leaf && ja < leaf + N) {
                    hit_s = s;
                    hit_l = ja - leaf;
                    const XCol xa{a.x0, ja, a.x0[ja] + a.h[ja]};
                    hit_v = OgGen::mv_operand(mv0 + s, hit_l, xa, a.cvec);
                }
            }
        }
    }

    // ---- batched D.X on the matrix cores: acc[s] (16 columns x 16 nodes) += A_s (16x4) * B (4x16)
    v4f64 acc[OgGen::MAX_NMV];
#pragma unroll
    for (int s = 0; s < OgGen::MAX_NMV; ++s) acc[s] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int lk = lane >> 4;
    for (int ks = 0; ks < KS; ++ks) {
        const double b = dpanel[ks * 64 + lane];
        const int l = ks * 4 + lk;
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) {
            if (s < nmv) {
                double av = xt[s * NP + l];
                if (s == hit_s && l == hit_l) av = hit_v;
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[s], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: dynamics + defect + difference quotient.  C/D layout of the f64 MFMA:
    //      column (node) = lane & 15, row (FD column) = (lane >> 4) + 4 * reg.
    const int k = nt * 16 + (lane & 15);
    if (k >= N) return;
    double y[OgGen::MAX_NMV];
    double out[OgGen::MAX_OUT];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = lk + 4 * reg;
        int j = -1;
        double xj = 0.0, dx = 1.0;
        if (SWEEP) {
            j = c0 + c;
            if (j >= a.col_hi) continue;
            const double xb = a.x0[j];
            xj = xb + a.h[j];
            dx = xj - xb;
        } else if (c != 0) {
            continue;
        }
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) y[s] = acc[s][reg];
        const XCol xa{a.x0, j, xj};
        OgGen::group_eval(g, k, xa, y, a.cvec, out);
#pragma unroll
        for (int o = 0; o < OgGen::MAX_OUT; ++o) {     // static index: keeps out[] in registers
            if (o >= nout) break;
            const int row = OgGen::G_ROW(g, o) + k;
            if (SWEEP) {
                a.jt[(long)(j - a.col_lo) * OgGen::M + row] = (out[o] - a.f0[row]) / dx;
            } else {
                a.f0[row] = out[o];
            }
        }
    }
}

template <bool SWEEP>
__device__ __forceinline__ void rows_body(const ogk_args& a, const int bx, const int by) {
    const int ri = bx * 256 + (int)threadIdx.x;
    if (ri >= OgGen::N_ROW_ITEMS) return;
    int g = 0;
    for (; g < OgGen::N_GROUPS; ++g) {
        if (OgGen::G_KIND(g) != 0) continue;
        const int i0 = OgGen::G_ITEM0(g);
        if (ri >= i0 && ri < i0 + OgGen::G_LEN(g)) break;
    }
    if (g >= OgGen::N_GROUPS) return;
    const int k = ri - OgGen::G_ITEM0(g);
    const int nout = OgGen::G_NOUT(g);
    double out[OgGen::MAX_OUT];
    if (!SWEEP) {
        const XCol base{a.x0, -1, 0.0};
        OgGen::group_eval(g, k, base, nullptr, a.cvec, out);
#pragma unroll
        for (int o = 0; o < OgGen::MAX_OUT; ++o)
            if (o < nout) a.f0[OgGen::G_ROW(g, o) + k] = out[o];
        return;
    }
    const int j0 = a.col_lo + by * ROWS_COLS_PER_THREAD;
    for (int c = 0; c < ROWS_COLS_PER_THREAD; ++c) {
        const int j = j0 + c;
        if (j >= a.col_hi) break;
        const double xb = a.x0[j];
        const double xj = xb + a.h[j];
        const double dx = xj - xb;
        const XCol xa{a.x0, j, xj};
        OgGen::group_eval(g, k, xa, nullptr, a.cvec, out);
        double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
#pragma unroll
        for (int o = 0; o < OgGen::MAX_OUT; ++o) {
            if (o >= nout) break;
            const int row = OgGen::G_ROW(g, o) + k;
            jrow[row] = (out[o] - a.f0[row]) / dx;
        }
    }
}

// One launch for the whole stacked function: workgroups [0, ndef*ytiles) run the collocation
// path (heavier, scheduled first), the rest run the row path.
template <bool SWEEP>
__global__ __launch_bounds__(256) void ogk_sweep(const ogk_args a, const int ndef,
                                                 const int defect_total, const int row_blocks) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int id = (int)blockIdx.x;
    if (id < defect_total) {
        defect_body<SWEEP>(a, id % ndef, id / ndef, lds);
    } else {
        const int rid = id - defect_total;
        rows_body<SWEEP>(a, rid % row_blocks, rid / row_blocks);
    }
}

int defect_blocks() {
    int nb = 0;
    for (int g = 0; g < OgGen::N_GROUPS; ++g)
        if (OgGen::G_KIND(g) == 1) nb += (OgGen::G_LEN(g) + 15) >> 4;
    return nb;
}

size_t defect_lds_bytes() {
    size_t worst = 0;
    for (int g = 0; g < OgGen::N_GROUPS; ++g) {
        if (OgGen::G_KIND(g) != 1) continue;
        const int KS = (OgGen::G_LEN(g) + 3) >> 2;
        const size_t need = ((size_t)KS * 64 + (size_t)OgGen::G_NMV(g) * KS * 4) * sizeof(double);
        worst = need > worst ? need : worst;
    }
    return worst;
}

}  // namespace

extern "C" int ogk_get_info(ogk_info* out) {
    out->abi = OGK_ABI;
    out->n = OgGen::N_VAR;
    out->m = OgGen::M;
    out->m_eq = OgGen::M_EQ;
    out->m_ineq = OgGen::M_INEQ;
    out->n_phase = OgGen::N_PHASE;
    out->n_mv = OgGen::N_MV;
    out->n_groups = OgGen::N_GROUPS;
    out->n_cvec = OgGen::N_CVEC;
    for (int i = 0; i < OGK_MAX_PHASE; ++i)
        out->phase_nodes[i] = i < OgGen::N_PHASE ? OgGen::PHASE_NODES(i) : 0;
    return 0;
}

extern "C" int ogk_launch(const ogk_args* args, int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int ndef = defect_blocks();
    const size_t lds = defect_lds_bytes();
    const int row_blocks = (OgGen::N_ROW_ITEMS + 255) / 256;
    if (mode == 0) {
        const int total = ndef + row_blocks;
        if (total > 0)
            hipLaunchKernelGGL(ogk_sweep<false>, dim3(total), dim3(256), lds, stream, *args, ndef,
                               ndef, row_blocks);
        return (int)hipGetLastError();
    }
    const int ncols = args->col_hi - args->col_lo;
    if (ncols <= 0) return 0;
    const int defect_total = ndef * ((ncols + 63) / 64);
    const int rows_total = row_blocks * ((ncols + ROWS_COLS_PER_THREAD - 1) / ROWS_COLS_PER_THREAD);
    if (defect_total + rows_total > 0)
        hipLaunchKernelGGL(ogk_sweep<true>, dim3(defect_total + rows_total), dim3(256), lds, stream,
                           *args, ndef, defect_total, row_blocks);
    return (int)hipGetLastError();
}
