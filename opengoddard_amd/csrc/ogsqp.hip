// ogsqp.hip - the QP subproblem and quasi-Newton update of the SQP driver on gfx950.
//
// Replaces the LSQ/LSEI/LSI/LDP/NNLS chain of SciPy's Fortran slsqp (include/ogsqp.h, DESIGN.md
// section 9).  Everything O(n^2) stays in HBM:
//
//   1. T = [C Z ; Z]  (TN GEMM on the FP64 matrix cores, reading the transposed FD Jacobian as
//      the sweep kernel left it), then one orthogonal LQ sweep from the right in compact-WY
//      panels of 8 reflectors (k_lq_panel / k_lq_apply_reg): C Z Q = [L 0],
//      J = Z Q.  J is again a factor of B^-1, its trailing columns Y span the null space of C
//      and are B-orthonormal, so the equality-constrained minimiser is two triangular solves
//      and the inequality part becomes a least-distance problem  min |y|^2, W y + b >= 0  with
//      W = [G;I] Y.
//   2. Goldfarb-Idnani dual active set on the LDP.  State: an orthonormal basis Q1 of the active
//      normals (classical Gram-Schmidt with the DGKS re-orthogonalisation test), the triangular R
//      and its explicit inverse (no triangular solve on the critical path), Givens chains for
//      removals.  Two kernels: k_gi_iter - one launch per change, every workgroup prices, the last
//      one to arrive does the update out of LDS; k_gi_coop - one cooperative launch per QP, the
//      update itself spread over workgroups that each own a slice of the null space (used from a
//      null space of 512 on; DESIGN.md section 9 has the measurements behind that threshold).
//   3. Product-form BFGS on the factor: Z <- Z - s (v'Z)/alpha.
//
// Reductions run in a fixed order: results are bit-reproducible from run to run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <iterator>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/ogsqp.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define OG_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(100 + (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr double DEPENDENT = 1e-10;   // |projection| / |normal| below this: linearly dependent
constexpr double FEASIBLE = 1e-12;    // rounding level of a normalised constraint value
constexpr double SINGULAR_C = 2.220446049250313e-16;  // lsei's own test: ABS(C(I,I)) < EPMACH -> mode 6.  Absolute,
                                                      // like there: a rank deficiency that rounding leaves at 1e-14
                                                      // passes (and yields the same wild step SciPy takes)
constexpr double REDUNDANT = 1e-13;   // |L_kk| <= this * max |L_jj|: equality k is a combination of the ones before it
constexpr double CONSISTENT = 1e-9;   // ... and redundant if its residual is below this * (1 + max |c|), else mode 6
constexpr int REFINE = 3;             // at most this many re-orthogonalisation passes (one is the rule)
constexpr double REORTH = 1e-8;       // another pass while the last correction exceeds this, relative
constexpr int GI_THREADS = 1024;
constexpr int GI_WAVES = GI_THREADS / 64;
constexpr size_t LDS_LIMIT = 160 * 1024 - 2048;

typedef double d4 __attribute__((ext_vector_type(4)));

// Constraint normals as the sweep kernel stores them: variable-major, constraint j of variable i
// at jt[i*ld + 1 + j]; the extra variable of the relaxed problem has its own contiguous row.
struct AView {
    const double* jt;
    long ld;
    const double* extra;
    int n;
};

__device__ __forceinline__ double aval(const AView& A, int i, int j) {
    return i < A.n ? A.jt[(long)i * A.ld + 1 + j] : A.extra[j];
}

// Cross-lane moves through DPP (4-cycle VALU modifiers) instead of __shfl_xor, which is a pair of
// ds_bpermute round trips through the LDS crossbar per 64-bit value (measured: 7 us for eight
// sums in k_lq_panel).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_f64(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
}

// Sum over the wavefront, the same value in every lane; fixed order.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x124>(v);   // row_ror 4
    v += dpp_f64<0x128>(v);   // row_ror 8   -> every lane holds the total of its row of 16
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}

// Sum over the workgroup, identical in every thread; `red` holds blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += red[w];
    return total;
}

// (value, index) minimum over the workgroup, ties to the lower index.
__device__ __forceinline__ void block_argmin(double& v, int& idx, double* redv, int* redi) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (ov < v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        redv[threadIdx.x >> 6] = v;
        redi[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    v = redv[0];
    idx = redi[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (redv[w] < v || (redv[w] == v && redi[w] < idx)) {
            v = redv[w];
            idx = redi[w];
        }
}

// ------------------------------------------------------------------------------------------
// out[j][k] = sum_i A(i, col0 + j) * Jw[i][k]     (rows x nq, row-major, leading dimension ldw)
// 64 x 64 output tile per workgroup, 4 wavefronts of 2 x 2 MFMA 16x16x4 tiles, K staged through
// LDS 16 rows at a time.  Both operands are read along their contiguous direction.
// The Jacobian of a collocation NLP is mostly zeros (7-25 % of the equality block, around 1 % of the path
// constraints: D blocks, node-diagonal blocks, a few dense columns), and the sweep kernel leaves exact zeros there.
// k_gemm_map marks, per block of 64 constraints, the slabs of 16 variables that hold anything at all; k_gemm_tn then
// walks the marked slabs only - at C3 about one in seven.
__global__ __launch_bounds__(256) void k_gemm_map(AView A, int col0, int rows, int nq, const int* __restrict__ sel,
                                                  unsigned char* __restrict__ map, int kblocks) {
    const int kb = blockIdx.x, mb = blockIdx.y;
    const int tid = threadIdx.x, ii = tid >> 4, jb = (tid & 15) * 4;
    const int i = kb * 16 + ii;
    int nz = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = mb * 64 + jb + e;
        if (i < nq && j < rows) nz |= aval(A, i, col0 + (sel ? sel[j] : j)) != 0.0;
    }
    nz = __syncthreads_or(nz);
    if (tid == 0) map[(long)mb * kblocks + kb] = nz ? 1 : 0;
}

// out[j][k] = sum_i A(i, col0 + j) * Jw[i][k]     (rows x nq, row-major, leading dimension ldw)
// 64 x 64 output tile per workgroup, 4 wavefronts of 2 x 2 MFMA 16x16x4 tiles, K staged through LDS 16 rows at a
// time - the slabs k_gemm_map marked, the next one's operands requested while the current one is multiplied.  Both
// operands are read along their contiguous direction.
// sel (optional): output row j takes the column col0 + sel[j] of A instead of col0 + j (rows of a warm start).
__global__ __launch_bounds__(256) void k_gemm_tn(AView A, int col0, int rows, const double* __restrict__ Jw,
                                                 int ldw, int nq, double* __restrict__ out,
                                                 const int* __restrict__ sel, const unsigned char* __restrict__ map,
                                                 int kblocks) {
    __shared__ double As[16][80];
    __shared__ double Bs[16][80];
    __shared__ short s_list[1024];
    __shared__ int s_count;
    const int j0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wj = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int ii = tid >> 4, jb = (tid & 15) * 4;
    // the marked slabs of my block of constraints, in order (one wavefront compacts the map)
    if (wave == 0) {
        int count = 0;
        const unsigned char* mine = map + (long)blockIdx.y * kblocks;
        for (int base = 0; base < kblocks; base += 64) {
            const int kb = base + lane;
            const bool on = kb < kblocks && mine[kb];
            const unsigned long long bal = __ballot(on);
            if (on) s_list[count + __popcll(bal & ((1ull << lane) - 1ull))] = (short)kb;
            count += __popcll(bal);
        }
        if (lane == 0) s_count = count;
    }
    __syncthreads();
    const int count = s_count;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    int jcol[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + jb + e;
        jcol[e] = j < rows ? col0 + (sel ? sel[j] : j) : -1;
    }
    double ra[4], rb[4];
    auto fetch = [&](int slab) {
        const int i = (int)s_list[slab] * 16 + ii;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + jb + e;
            ra[e] = (i < nq && jcol[e] >= 0) ? aval(A, i, jcol[e]) : 0.0;
            rb[e] = (i < nq && k < nq) ? Jw[(long)i * ldw + k] : 0.0;
        }
    };
    if (count > 0) fetch(0);
    for (int slab = 0; slab < count; ++slab) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[ii][jb + e] = ra[e];
            Bs[ii][jb + e] = rb[e];
        }
        __syncthreads();
        if (slab + 1 < count) fetch(slab + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kr = kk * 4 + (lane >> 4);
            const double a0 = As[kr][wj + (lane & 15)], a1 = As[kr][wj + 16 + (lane & 15)];
            const double b0 = Bs[kr][wk + (lane & 15)], b1 = Bs[kr][wk + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = j0 + wj + tj * 16 + (lane >> 4) + 4 * reg;
                const int k = k0 + wk + tk * 16 + (lane & 15);
                if (j < rows && k < nq) out[(long)j * ldw + k] = acc[tj][tk][reg];
            }
}

// Work copy of the factor; the relaxed problem appends the variable delta with 1/rho on the diagonal.
__global__ void k_copy_factor(const double* __restrict__ Z, double* __restrict__ Jw, int ld, int n, int nq,
                              double inv_rho) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (k >= nq) return;
    double v;
    if (i < n && k < n)
        v = Z[(long)i * ld + k];
    else
        v = (i == k) ? inv_rho : 0.0;
    Jw[(long)i * ld + k] = v;
}

__global__ void k_identity(double* Z, int ld, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (k < n) Z[(long)i * ld + k] = (i == k) ? 1.0 : 0.0;
}

// Coefficients of the relaxation variable: -c_j for equalities, max(-c_j, 0) for inequalities.
__global__ void k_relaxation_row(const double* c, int meq, int m, double* extra) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) extra[j] = j < meq ? -c[j] : fmax(-c[j], 0.0);
}

// ------------------------------------------------------------------------------------------
// LQ sweep, LQ_NB Householder reflectors per trip (compact WY form, row-vector convention):
//
//   k_lq_panel  one workgroup: reflectors of the rows k .. k+nb-1 of Tc, each applied to the panel
//               rows below it before the next one is formed; stores V (nb x L, L = nq - k, zero
//               left of its diagonal), the new diagonal entries, and the upper-triangular T with
//               H_0 H_1 ... H_{nb-1} = I - V' T V.
//   k_lq_apply  every remaining row of Tc and every row of Jw:  row <- row - ((row V') T) V,
//               two passes over the row whatever nb is.
//
// Of the panel rows only the finished entries of L (left of the diagonal) are written back:
// nobody reads their tails again, their new diagonals live in diagL.
#ifndef OGSQP_LQ_NB
#define OGSQP_LQ_NB 8
#endif
constexpr int LQ_NB = OGSQP_LQ_NB;   // reflectors per panel

struct LqPanel {
    double T[LQ_NB][LQ_NB];
    int nb;
    int pad;
    long long tr[8];   // OGSQP_TRACE: s_memtime ticks per section of the last panel kernel
};

// K sums over the workgroup at once (identical in every thread); red holds (blockDim/64) * K doubles.
template <int K>
__device__ __forceinline__ void block_sum_vec(double (&v)[K], double* red) {
#pragma unroll
    for (int e = 0; e < K; ++e) v[e] = wave_sum(v[e]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < K; ++e) red[(threadIdx.x >> 6) * K + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < K; ++e) {
        double total = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += red[w * K + e];
        v[e] = total;
    }
}

#ifndef OGSQP_PANEL_SMALL_PT
#define OGSQP_PANEL_SMALL_PT 128
#endif
constexpr int PANEL_SMALL_PT = OGSQP_PANEL_SMALL_PT;   // threads of the panel kernel for rows that fit their registers
constexpr int LQ_PT_MAX = 512;   // threads of the panel kernel for long rows
constexpr int LQ_CPT_MAX = 16;   // panel columns per thread: rows of up to LQ_PT_MAX * LQ_CPT_MAX entries

// Few threads with many columns each keep the reductions cheap (fewer wavefronts to combine) as long
// as the panel fits in their registers; long rows need the full 512 threads.
template <int LQ_PT, int LQ_CPT>
__global__ __launch_bounds__(LQ_PT) void k_lq_panel(double* __restrict__ Tc, int ld, int meq, int nq, int k,
                                                   double* __restrict__ V, double* __restrict__ diagL,
                                                   LqPanel* __restrict__ panel, double* __restrict__ dmaxbuf) {
    constexpr int NPAIR = LQ_NB * (LQ_NB - 1) / 2;
    __shared__ double red[(LQ_PT / 64) * NPAIR];
    const int tid = threadIdx.x;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memtime();
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PMARK(slot) do { const long long now_ = __builtin_amdgcn_s_memtime(); t_sec[slot] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define PMARK(slot) do { } while (0)
#endif
    const int nb = min(LQ_NB, meq - k), L = nq - k;
    // the panel lives in registers: thread t owns the columns t, t + LQ_PT, ...
    double P[LQ_NB][LQ_CPT];
#pragma unroll
    for (int b = 0; b < LQ_NB; ++b)
#pragma unroll
        for (int e = 0; e < LQ_CPT; ++e) {
            const int j = tid + e * LQ_PT;
            P[b][e] = (b < nb && j < L) ? Tc[(long)(k + b) * ld + k + j] : 0.0;
        }
    __syncthreads();
    PMARK(0);   // load
    __shared__ double s_lower[LQ_NB][LQ_NB];
    __shared__ double s_diag[LQ_NB];
    double dmax = dmaxbuf[0];           // largest pivot of the sweep so far
    double beta[LQ_NB];
#pragma unroll
    for (int b = 0; b < LQ_NB; ++b) {
        beta[b] = 0.0;
        if (b < nb) {
            // one reduction per reflector: |row b|^2 and the raw products of the rows below with row b;
            // column b of the panel (owned by thread b) is passed through LDS alongside
            __shared__ double s_col[LQ_NB];
            // entries of row b left of its diagonal are finished entries of L: back to Tc, and zero
            // in the register copy, so that neither the products nor the updates below need a
            // column predicate (the reflector vector is zero there)
            // (kept in LDS until the end: a global store here would make every barrier below wait
            // for its acknowledgement)
            if (tid < b) {
                s_lower[b][tid] = P[b][0];
                P[b][0] = 0.0;
            }
            double vals[LQ_NB];
#pragma unroll
            for (int r = 0; r < LQ_NB; ++r) {
                vals[r] = 0.0;
                if (r >= b)
#pragma unroll
                    for (int e = 0; e < LQ_CPT; ++e) vals[r] += P[r][e] * P[b][e];
            }
            if (tid == b) {
#pragma unroll
                for (int r = 0; r < LQ_NB; ++r) s_col[r] = P[r][0];
            }
            PMARK(1);   // products
            block_sum_vec<LQ_NB>(vals, red);
            PMARK(2);   // reduction
            const double sigma2 = vals[b];
            const double x0 = s_col[b];
            const double sigma = sqrt(sigma2);
            // what is left of a row that depends on the earlier ones is rounding noise: no reflector is built
            // from it (it would rotate the null-space basis by that noise); its pivot is recorded as exactly 0
            const bool live = sigma > REDUNDANT * dmax && sigma > 0.0;
            dmax = fmax(dmax, sigma);
            const double alpha = !live ? 0.0 : (x0 >= 0.0 ? -sigma : sigma);
            const double v0 = x0 - alpha;
            const double vv = sigma2 - x0 * x0 + v0 * v0;
            const double bt = (live && vv > 0.0) ? 2.0 / vv : 0.0;
            beta[b] = bt;
            if (tid == 0) s_diag[b] = alpha;
            if (tid == b) P[b][0] = v0;
            // H_b on the panel rows below:  row . v_b = (row . row_b) - row[b] alpha
#pragma unroll
            for (int r = 0; r < LQ_NB; ++r)
                if (r > b && r < nb) {
                    const double f = bt * (vals[r] - s_col[r] * alpha);
#pragma unroll
                    for (int e = 0; e < LQ_CPT; ++e) P[r][e] -= f * P[b][e];
                }
            __syncthreads();                      // s_col is rewritten by the next reflector
            PMARK(3);   // update
        }
    }
    // Gram matrix of the reflector vectors: per-thread products, 16-lane sums by DPP, then the
    // (LQ_PT / 16) partial sums of every pair are added by one thread per pair out of LDS
    __shared__ double s_part[NPAIR][LQ_PT / 16 + 1];
    __shared__ double s_gram[LQ_NB][LQ_NB];
    __shared__ double s_T[LQ_NB][LQ_NB];
    {
        int idx = 0;
#pragma unroll
        for (int a = 0; a < LQ_NB; ++a)
#pragma unroll
            for (int b = a + 1; b < LQ_NB; ++b) {
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < LQ_CPT; ++e) acc += P[a][e] * P[b][e];
                acc += dpp_f64<0xB1>(acc);
                acc += dpp_f64<0x4E>(acc);
                acc += dpp_f64<0x124>(acc);
                acc += dpp_f64<0x128>(acc);
                if ((tid & 15) == 0) s_part[idx][tid >> 4] = acc;
                ++idx;
            }
    }
    PMARK(4);   // gram products
    __syncthreads();
    for (int pair = tid; pair < NPAIR; pair += LQ_PT) {
        double total = 0.0;
        for (int w = 0; w < LQ_PT / 16; ++w) total += s_part[pair][w];
        // pair index -> (a, b), a < b
        int a = 0, rem = pair;
        while (rem >= LQ_NB - 1 - a) {
            rem -= LQ_NB - 1 - a;
            ++a;
        }
        s_gram[a][a + 1 + rem] = total;
    }
    for (int e = tid; e < LQ_NB * LQ_NB; e += LQ_PT) s_T[e / LQ_NB][e % LQ_NB] = 0.0;
    __syncthreads();
    PMARK(5);   // gram reduction
#pragma unroll
    for (int b = 0; b < LQ_NB; ++b)
#pragma unroll
        for (int e = 0; e < LQ_CPT; ++e) {
            const int j = tid + e * LQ_PT;
            if (b < nb && j < L) V[(long)b * ld + j] = P[b][e];
        }
    // T by forward accumulation; row a of T depends only on itself: thread a does row a
    __shared__ double s_beta[LQ_NB];
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) s_beta[b] = beta[b];
        panel->nb = nb;
        panel->pad = 0;
        dmaxbuf[0] = dmax;
    }
    __syncthreads();
    if (tid < nb) {
        const int a = tid;
        s_T[a][a] = s_beta[a];
        for (int b = a + 1; b < nb; ++b) {
            double acc = 0.0;
            for (int c = a; c < b; ++c) acc += s_T[a][c] * s_gram[c][b];
            s_T[a][b] = -s_beta[b] * acc;
        }
    }
    __syncthreads();
    for (int e = tid; e < LQ_NB * LQ_NB; e += LQ_PT) {
        const int a = e / LQ_NB, b = e % LQ_NB;
        panel->T[a][b] = s_T[a][b];
        if (a < nb && b < a) Tc[(long)(k + a) * ld + k + b] = s_lower[a][b];   // finished entries of L
        if (a < nb && b == 0) diagL[k + a] = s_diag[a];
    }
    PMARK(6);   // store V, T
#ifdef OGSQP_TRACE
    if (tid == 0)
        for (int e = 0; e < 8; ++e) panel->tr[e] = t_sec[e];
#endif
#undef PMARK
}

// One workgroup = LQ_RW rows; its four wavefronts split the columns, so every load of V serves
// LQ_RW rows and there are enough wavefronts (rows) to fill the chip.
constexpr int LQ_RW = 4;
__global__ __launch_bounds__(256) void k_lq_apply(double* __restrict__ Tc, double* __restrict__ Jw, int ld, int meq,
                                                  int nq, int k, const double* __restrict__ V,
                                                  const LqPanel* __restrict__ panel) {
    __shared__ double part[4][LQ_RW][LQ_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = panel->nb, L = nq - k;
    const int below = meq - k - nb, nrows = below + nq;
    const int r0 = blockIdx.x * LQ_RW;
    double* rows[LQ_RW];
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i) {
        const int r = min(r0 + i, nrows - 1);            // tail: repeat the last row, masked on store
        rows[i] = (r < below) ? Tc + (long)(k + nb + r) * ld + k : Jw + (long)(r - below) * ld + k;
    }
    const int valid = min(LQ_RW, nrows - r0);
    double w[LQ_RW][LQ_NB];
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i)
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) w[i][b] = 0.0;
    for (int j = tid; j < L; j += 256) {
        double v[LQ_NB];
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) v[b] = b < nb ? V[(long)b * ld + j] : 0.0;
#pragma unroll
        for (int i = 0; i < LQ_RW; ++i) {
            const double x = rows[i][j];
#pragma unroll
            for (int b = 0; b < LQ_NB; ++b) w[i][b] += x * v[b];
        }
    }
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i)
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) {
            const double t = wave_sum(w[i][b]);
            if (lane == 0) part[wave][i][b] = t;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i) {
        double t[LQ_NB];
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) t[b] = (part[0][i][b] + part[1][i][b]) + (part[2][i][b] + part[3][i][b]);
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < LQ_NB; ++a)
                if (a <= b) acc += t[a] * panel->T[a][b];
            w[i][b] = acc;
        }
    }
    for (int j = tid; j < L; j += 256) {
        double v[LQ_NB];
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) v[b] = b < nb ? V[(long)b * ld + j] : 0.0;
#pragma unroll
        for (int i = 0; i < LQ_RW; ++i) {
            if (i < valid) {
                double x = rows[i][j];
#pragma unroll
                for (int b = 0; b < LQ_NB; ++b) x -= w[i][b] * v[b];
                rows[i][j] = x;
            }
        }
    }
}

// Same with the row segments and the reflector vectors held in registers between the two passes
// (rows of up to 256 * JT entries): one trip to memory instead of two.
template <int JT>
__global__ __launch_bounds__(256) void k_lq_apply_reg(double* __restrict__ Tc, double* __restrict__ Jw, int ld,
                                                      int meq, int nq, int k, const double* __restrict__ V,
                                                      const LqPanel* __restrict__ panel) {
    __shared__ double part[4][LQ_RW][LQ_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = panel->nb, L = nq - k;
    const int below = meq - k - nb, nrows = below + nq;
    const int r0 = blockIdx.x * LQ_RW;
    double* rows[LQ_RW];
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i) {
        const int r = min(r0 + i, nrows - 1);
        rows[i] = (r < below) ? Tc + (long)(k + nb + r) * ld + k : Jw + (long)(r - below) * ld + k;
    }
    const int valid = min(LQ_RW, nrows - r0);
    double x[LQ_RW][JT], v[LQ_NB][JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        const int j = tid + t * 256;
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) v[b][t] = (b < nb && j < L) ? V[(long)b * ld + j] : 0.0;
#pragma unroll
        for (int i = 0; i < LQ_RW; ++i) x[i][t] = j < L ? rows[i][j] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i)
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < JT; ++t) acc += x[i][t] * v[b][t];
            acc = wave_sum(acc);
            if (lane == 0) part[wave][i][b] = acc;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LQ_RW; ++i) {
        double t8[LQ_NB], wt[LQ_NB];
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) t8[b] = (part[0][i][b] + part[1][i][b]) + (part[2][i][b] + part[3][i][b]);
#pragma unroll
        for (int b = 0; b < LQ_NB; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < LQ_NB; ++a)
                if (a <= b) acc += t8[a] * panel->T[a][b];
            wt[b] = acc;
        }
        if (i < valid) {
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                const int j = tid + t * 256;
                double xv = x[i][t];
#pragma unroll
                for (int b = 0; b < LQ_NB; ++b) xv -= wt[b] * v[b][t];
                if (j < L) rows[i][j] = xv;
            }
        }
    }
}

#include "ogsqp_lq16.h"
#include "ogsqp_lqwide.h"

// max |diag| / min |diag| test of the triangular factor -> flag[0] = 1 when singular
__global__ void k_check_diag(const double* diagL, int meq, int* flag, double* dthresh) {
    __shared__ double red[16];
    __shared__ double red2[16];
    double mx = 0.0, mn = INFINITY;
    for (int i = threadIdx.x; i < meq; i += blockDim.x) {
        const double a = fabs(diagL[i]);
        mx = fmax(mx, a);
        mn = fmin(mn, a);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, off));
        mn = fmin(mn, __shfl_xor(mn, off));
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mx;
        red2[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            mx = fmax(mx, red[w]);
            mn = fmin(mn, red2[w]);
        }
        // a vanishing pivot is not fatal by itself: k_trsv decides (redundant if consistent, mode 6 if not)
        (void)mn;
        flag[0] = 0;
        dthresh[0] = fmax(REDUNDANT * mx, SINGULAR_C);
    }
}

// ------------------------------------------------------------------------------------------
// L x = rhs (transposed = 0) or L' x = rhs (transposed = 1); L = strictly lower part of Tc with
// diagL on the diagonal.  One workgroup; 64 x 64 diagonal blocks are solved by one wavefront out
// of LDS, the panel below (above) is a GEMV spread over all threads.
// value of v in lane `src` (uniform): two v_readlane_b32 instead of the LDS round trip of __shfl
__device__ __forceinline__ double readlane_f64(const double v, const int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(1024) void k_trsv(const double* __restrict__ Tc, int ld, const double* __restrict__ diagL,
                                               int meq, int transposed, double scale_rhs, const double* __restrict__ rhs,
                                               double* __restrict__ x, const double* __restrict__ dthresh,
                                               int* __restrict__ flag) {
    extern __shared__ double lds[];
    __shared__ double red[16];
    double* xs = lds;                 // meq
    double* blk = lds + meq;          // 64 x 65
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double biggest = 0.0;
    for (int i = tid; i < meq; i += 1024) {
        const double v = scale_rhs * rhs[i];
        xs[i] = v;
        biggest = fmax(biggest, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) biggest = fmax(biggest, __shfl_xor(biggest, off));
    if (lane == 0) red[wave] = biggest;
    __syncthreads();
    for (int w2 = 0; w2 < 16; ++w2) biggest = fmax(biggest, red[w2]);
    // a vanishing pivot: the equality is a combination of earlier ones - dropped if its residual vanishes
    // too (its component and its multiplier are zero), "singular matrix C" (flag 0) if it does not
    const double tiny = dthresh[0], tol = CONSISTENT * (1.0 + biggest);
    const int nblk = (meq + 63) / 64;
    for (int bi = 0; bi < nblk; ++bi) {
        const int b = transposed ? nblk - 1 - bi : bi;
        const int i0 = b * 64, bs = min(64, meq - i0);
        for (int e = tid; e < 64 * 64; e += 1024) {
            const int r = e >> 6, c = e & 63;
            blk[r * 65 + c] = (r < bs && c < r) ? Tc[(long)(i0 + r) * ld + i0 + c] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            const double dlane = lane < bs ? diagL[i0 + lane] : 1.0;     // the block's diagonal, one entry per lane
            double xv = lane < bs ? xs[i0 + lane] : 0.0;
            if (!transposed) {
                for (int c = 0; c < bs; ++c) {
                    const double dc = readlane_f64(dlane, c), num = readlane_f64(xv, c);
                    const bool gone = !(fabs(dc) > tiny);
                    if (gone && fabs(num) > tol && lane == 0) flag[0] = 1;
                    const double xc = gone ? 0.0 : num / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane > c)
                        xv -= blk[lane * 65 + c] * xc;
                }
            } else {
                for (int c = bs - 1; c >= 0; --c) {
                    const double dc = readlane_f64(dlane, c);
                    const double xc = !(fabs(dc) > tiny) ? 0.0 : readlane_f64(xv, c) / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane < c)
                        xv -= blk[c * 65 + lane] * xc;
                }
            }
            if (lane < bs) xs[i0 + lane] = xv;
        }
        __syncthreads();
        if (!transposed) {
            for (int r = i0 + bs + tid; r < meq; r += 1024) {
                const double* row = Tc + (long)r * ld + i0;
                double acc0 = 0.0, acc1 = 0.0;
                int c = 0;
                for (; c + 16 <= bs; c += 16) {          // 16 independent loads in flight
                    double l[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) l[e] = row[c + e];
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        acc0 += l[e] * xs[i0 + c + e];
                        acc1 += l[e + 1] * xs[i0 + c + e + 1];
                    }
                }
                for (; c < bs; ++c) acc0 += row[c] * xs[i0 + c];
                xs[r] -= acc0 + acc1;
            }
        } else {
            for (int r = tid; r < i0; r += 1024) {
                const double* col = Tc + (long)i0 * ld + r;
                double acc0 = 0.0, acc1 = 0.0;
                int c = 0;
                for (; c + 16 <= bs; c += 16) {
                    double l[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) l[e] = col[(long)(c + e) * ld];
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        acc0 += l[e] * xs[i0 + c + e];
                        acc1 += l[e + 1] * xs[i0 + c + e + 1];
                    }
                }
                for (; c < bs; ++c) acc0 += col[(long)c * ld] * xs[i0 + c];
                xs[r] -= acc0 + acc1;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < meq; i += 1024) x[i] = xs[i];
}

// The same solve spread over the chip: one launch per 64 x 64 diagonal block.  Every workgroup solves the block
// for itself (64 dependent steps out of LDS: cheap, and it spares a hand-off between workgroups) and then updates
// its own share of the remaining right-hand side with the block's 64 columns - the part that kept the single
// workgroup of k_trsv at one CU's bandwidth (282 us for the 3.8 MB of L at C3).  x holds scale * rhs on entry
// (k_trsv_prepare, which also leaves max |rhs| in scal[0]) and the solution at the end.
__global__ __launch_bounds__(256) void k_trsv_prepare(const double* __restrict__ rhs, int meq, double scale_rhs,
                                                      double* __restrict__ x, double* __restrict__ scal,
                                                      double* __restrict__ pending) {
    __shared__ double red[4];
    double biggest = 0.0;
    for (int i = threadIdx.x; i < meq; i += 256) {
        const double v = scale_rhs * rhs[i];
        x[i] = v;
        if (pending) pending[i] = __longlong_as_double(0x7ff8dead5eedcafell);     // k_trsv_chain: "not solved yet"
        biggest = fmax(biggest, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) biggest = fmax(biggest, __shfl_xor(biggest, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = biggest;
    __syncthreads();
    if (threadIdx.x == 0) scal[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Inverses of the 64 x 64 diagonal blocks of L, once per subproblem (both triangular solves use them): one wavefront
// per block, lane j builds column j of the inverse by forward substitution out of LDS.  A block with a vanishing
// pivot (a redundant equality) is marked instead: its step takes the pivot-by-pivot path, which knows what to do
// with it.
__global__ __launch_bounds__(64) void k_trsv_invert(const double* __restrict__ Tc, int ld, const double* __restrict__ diagL,
                                                    int meq, const double* __restrict__ dthresh,
                                                    double* __restrict__ Linv, int* __restrict__ has_gone) {
    __shared__ double blk[64 * 65];
    __shared__ double dinv[64];
    __shared__ int s_gone;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int i0 = b * 64, bs = min(64, meq - i0);
    const double tiny = dthresh[0];
    if (lane == 0) s_gone = 0;
    __syncthreads();
    for (int r = 0; r < 64; ++r) blk[r * 65 + lane] = (r < bs && lane < r) ? Tc[(long)(i0 + r) * ld + i0 + lane] : 0.0;
    {
        const double d = lane < bs ? diagL[i0 + lane] : 1.0;
        const bool gone = !(fabs(d) > tiny);
        if (gone) s_gone = 1;
        dinv[lane] = gone ? 0.0 : 1.0 / d;
    }
    __syncthreads();
    // column `lane` of the inverse: x_i for i >= lane
    double* out = Linv + (long)b * 64 * 64;
    double x[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int c = 0; c < i; ++c) {
            const double l = blk[i * 65 + c];                     // (the same address in every lane: a broadcast)
            const double xc = c >= lane ? x[c] : 0.0;
            if (c & 1) acc1 = fma(l, xc, acc1);
            else acc0 = fma(l, xc, acc0);
        }
        x[i] = i < lane ? 0.0 : ((i == lane ? 1.0 : 0.0) - (acc0 + acc1)) * dinv[i];
        out[(long)i * 64 + lane] = x[i];                          // row-major: Linv[i][lane]
    }
    if (lane == 0) has_gone[b] = s_gone;
}

constexpr int TRSV_ROWS = 64;        // rows of the remaining right-hand side per workgroup
__global__ __launch_bounds__(256) void k_trsv_block(const double* __restrict__ Tc, int ld,
                                                    const double* __restrict__ diagL, int meq, int transposed, int b,
                                                    double* __restrict__ x, double* __restrict__ sol,
                                                    const double* __restrict__ scal,
                                                    const double* __restrict__ dthresh, int* __restrict__ flag,
                                                    const double* __restrict__ Linv, const int* __restrict__ has_gone) {
    // x: the right-hand side as the blocks before this one left it (read for the block, updated for the rest);
    // sol: the solution (a separate vector: a workgroup that starts late must still find the block's right-hand side)
    __shared__ double blk[64 * 65];
    __shared__ double xb[64];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = b * 64, bs = min(64, meq - i0);
    const double tiny = dthresh[0], tol = CONSISTENT * (1.0 + scal[0]);
    if (!has_gone[b]) {
        // the block's inverse is there (k_trsv_invert): x_b = inv(L_bb) rhs_b, or its transpose, as a 64 x 64 product -
        // every thread a quarter of a row, no dependent chain
        const double* inv = Linv + (long)b * 64 * 64;
        const int row = lane, q = wave;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const int c0 = 16 * q + c;
            const double a0 = transposed ? inv[(long)c0 * 64 + row] : inv[(long)row * 64 + c0];
            const double a1 = transposed ? inv[(long)(c0 + 1) * 64 + row] : inv[(long)row * 64 + c0 + 1];
            acc0 = fma(a0, c0 < bs ? x[i0 + c0] : 0.0, acc0);
            acc1 = fma(a1, c0 + 1 < bs ? x[i0 + c0 + 1] : 0.0, acc1);
        }
        part[q][row] = acc0 + acc1;
        __syncthreads();
        if (wave == 0) xb[lane] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    } else {
        // pivot by pivot (a vanishing pivot: the equality is a combination of earlier ones - dropped if its residual
        // vanishes too, "singular matrix C" (flag 0) if it does not)
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t, r = e >> 6, c = e & 63;
            v[t] = Tc[(long)(i0 + min(r, bs - 1)) * ld + i0 + c];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t, r = e >> 6, c = e & 63;
            blk[r * 65 + c] = (r < bs && c < r) ? v[t] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            const double dlane = lane < bs ? diagL[i0 + lane] : 1.0;     // the block's diagonal, one entry per lane
            double xv = lane < bs ? x[i0 + lane] : 0.0;
            if (!transposed) {
                for (int c = 0; c < bs; ++c) {
                    const double dc = readlane_f64(dlane, c), num = readlane_f64(xv, c);
                    const bool gone = !(fabs(dc) > tiny);
                    if (gone && fabs(num) > tol && lane == 0 && blockIdx.x == 0) flag[0] = 1;
                    const double xc = gone ? 0.0 : num / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane > c)
                        xv -= blk[lane * 65 + c] * xc;
                }
            } else {
                for (int c = bs - 1; c >= 0; --c) {
                    const double dc = readlane_f64(dlane, c);
                    const double xc = !(fabs(dc) > tiny) ? 0.0 : readlane_f64(xv, c) / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane < c)
                        xv -= blk[c * 65 + lane] * xc;
                }
            }
            xb[lane] = xv;
        }
    }
    __syncthreads();
    if (!transposed) {
        // rows below the block: a wavefront per row, its 64 entries of the block's columns in one coalesced load
        const int first = i0 + bs + blockIdx.x * TRSV_ROWS, last = min(first + TRSV_ROWS, meq);
        const double xl = lane < bs ? xb[lane] : 0.0;
        for (int r0 = first + wave * 4; r0 < last; r0 += 16) {         // four rows per trip, loads in flight together
            double v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = Tc[(long)min(r0 + e, meq - 1) * ld + i0 + min(lane, bs - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double acc = wave_sum(v[e] * xl);
                if (lane == 0 && r0 + e < last) x[r0 + e] -= acc;
            }
        }
    } else {
        // rows above the block: a thread per row, the block's rows are contiguous in it
        const int r = blockIdx.x * 256 + tid;
        if (r < i0) {
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            const double* col = Tc + (long)i0 * ld + r;
            int c = 0;
            for (; c + 4 <= bs; c += 4) {
                acc0 += col[(long)c * ld] * xb[c];
                acc1 += col[(long)(c + 1) * ld] * xb[c + 1];
                acc2 += col[(long)(c + 2) * ld] * xb[c + 2];
                acc3 += col[(long)(c + 3) * ld] * xb[c + 3];
            }
            for (; c < bs; ++c) acc0 += col[(long)c * ld] * xb[c];
            x[r] -= (acc0 + acc1) + (acc2 + acc3);
        }
    }
    if (blockIdx.x == 0 && tid < bs) sol[i0 + tid] = xb[tid];
}

// The blocked solve as ONE launch: workgroup b owns block row b of L (64 rows) and there is no launch per block -
// sixteen kernel boundaries and sixteen small grids at C3 (8.6 us each) become a chain of hand-offs through memory.
// Forward (L x = rhs): workgroup b subtracts L(b, c) x_c from its right-hand side for c = 0 .. b - 1 as the x_c
// appear, then x_b = inv(L_bb) r_b (k_trsv_invert; pivot by pivot when the block has a vanishing pivot) and publishes
// it.  Transposed (L' x = rhs): the same from the last block down, with the tiles L(c, b)'.  Publication is the
// solution vector itself: it starts as a NaN with a payload no computation produces (k_trsv_prepare), the owner
// stores the values with agent scope, readers poll them with agent-scope loads.  At most 128 workgroups (n <= 8191):
// all resident, each waits only for workgroups before it in the chain, and a wait that gives up says so (flag[3]).
// Per block the chain pays one trip through memory and two 64 x 64 products (~2.7 us); all other tiles of a block
// row are applied while the chain is still further up.
constexpr unsigned long long TRSV_PENDING = 0x7ff8dead5eedcafeull;
__global__ __launch_bounds__(256) void k_trsv_chain(const double* __restrict__ Tc, int ld, const double* __restrict__ diagL,
                                                    int meq, int transposed, const double* __restrict__ rhs,
                                                    double* __restrict__ sol, const double* __restrict__ scal,
                                                    const double* __restrict__ dthresh, int* __restrict__ flag,
                                                    const double* __restrict__ Linv, const int* __restrict__ has_gone,
                                                    int spin_limit) {
    __shared__ double blk[64 * 65];
    __shared__ double r[64], xs[64], xb[64];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (meq + 63) / 64;
    const int b = transposed ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;     // the chain's order = the grid's
    const int i0 = b * 64, bs = min(64, meq - i0);
    const double tiny = dthresh[0], tol = CONSISTENT * (1.0 + scal[0]);
    if (tid < 64) r[tid] = tid < bs ? rhs[i0 + tid] : 0.0;
    // operands of the block's own step, requested before the chain is waited for
    const bool gone_block = has_gone[b] != 0;
    const double* inv = Linv + (long)b * 64 * 64;
    double iv[16];
    {
        const int row = lane, q = wave;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int c0 = 16 * q + c;
            iv[c] = transposed ? inv[(long)c0 * 64 + row] : inv[(long)row * 64 + c0];
        }
    }
    // tiles of the block row (forward: L(b, c), thread = (row tid / 4, columns 16 (tid % 4) ..); transposed: L(c, b),
    // wavefront = 16 rows of the tile, lane = column - both read 128-byte pieces of rows of L)
    const int steps = transposed ? nblk - 1 - b : b;
    auto load_tile = [&](int c, double (&t)[16]) {
        if (!transposed) {
            const int row = tid >> 2, q = tid & 3;
            const double* src = Tc + (long)(i0 + min(row, bs - 1)) * ld + c * 64 + 16 * q;
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e] = src[e];
        } else {
            const int cs = min(64, meq - c * 64);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 16 * wave + e;
                t[e] = Tc[(long)(c * 64 + min(row, cs - 1)) * ld + i0 + min(lane, bs - 1)];
            }
        }
    };
    double tile[16];
    if (steps > 0) load_tile(transposed ? nblk - 1 : 0, tile);
    __syncthreads();
    for (int sidx = 0; sidx < steps; ++sidx) {
        const int c = transposed ? nblk - 1 - sidx : sidx;
        const int cs = min(64, meq - c * 64);
        double next[16];
        if (sidx + 1 < steps) load_tile(transposed ? c - 1 : c + 1, next);
        // x_c: wavefront 0 polls it
        if (wave == 0) {
            double v = 0.0;
            int spins = 0;
            while (true) {
                v = lane < cs ? __hip_atomic_load(sol + c * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                const bool pending = (unsigned long long)__double_as_longlong(v) == TRSV_PENDING;
                if (!__any(pending)) break;
                if (++spins > spin_limit) {    // (2^25 by default, about a second: a workgroup that comes this late is not coming)
                    if (lane == 0) __hip_atomic_store(flag + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            xs[lane] = v;
        }
        __syncthreads();
        if (!transposed) {
            const int row = tid >> 2, q = tid & 3;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                a0 = fma(tile[e], xs[16 * q + e], a0);
                a1 = fma(tile[e + 1], xs[16 * q + e + 1], a1);
            }
            double acc = a0 + a1;
            acc += dpp_f64<0xB1>(acc);     // the four threads of a row are neighbours: quad sums
            acc += dpp_f64<0x4E>(acc);
            if (q == 0 && row < bs) r[row] -= acc;
        } else {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                a0 = fma(tile[e], 16 * wave + e < cs ? xs[16 * wave + e] : 0.0, a0);
                a1 = fma(tile[e + 1], 16 * wave + e + 1 < cs ? xs[16 * wave + e + 1] : 0.0, a1);
            }
            part[wave][lane] = a0 + a1;
            __syncthreads();
            if (wave == 0 && lane < bs) r[lane] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[e] = next[e];
    }
    // the block's own step
    if (!gone_block) {
        const int row = lane, q = wave;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const int c0 = 16 * q + c;
            acc0 = fma(iv[c], c0 < bs ? r[c0] : 0.0, acc0);
            acc1 = fma(iv[c + 1], c0 + 1 < bs ? r[c0 + 1] : 0.0, acc1);
        }
        part[q][row] = acc0 + acc1;
        __syncthreads();
        if (wave == 0) xb[lane] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    } else {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t, rr = e >> 6, c = e & 63;
            v[t] = Tc[(long)(i0 + min(rr, bs - 1)) * ld + i0 + c];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t, rr = e >> 6, c = e & 63;
            blk[rr * 65 + c] = (rr < bs && c < rr) ? v[t] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            const double dlane = lane < bs ? diagL[i0 + lane] : 1.0;
            double xv = lane < bs ? r[lane] : 0.0;
            if (!transposed) {
                for (int c = 0; c < bs; ++c) {
                    const double dc = readlane_f64(dlane, c), num = readlane_f64(xv, c);
                    const bool gone = !(fabs(dc) > tiny);
                    if (gone && fabs(num) > tol && lane == 0) flag[0] = 1;
                    const double xc = gone ? 0.0 : num / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane > c)
                        xv -= blk[lane * 65 + c] * xc;
                }
            } else {
                for (int c = bs - 1; c >= 0; --c) {
                    const double dc = readlane_f64(dlane, c);
                    const double xc = !(fabs(dc) > tiny) ? 0.0 : readlane_f64(xv, c) / dc;
                    if (lane == c)
                        xv = xc;
                    else if (lane < c)
                        xv -= blk[c * 65 + lane] * xc;
                }
            }
            xb[lane] = xv;
        }
    }
    __syncthreads();
    if (wave == 0 && lane < bs) __hip_atomic_store(sol + i0 + lane, xb[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ------------------------------------------------------------------------------------------
// out[k] = base[k] + sum_i M[i*ld + k] * x[i]   (k < ncols, i < nrows): 64 columns per workgroup,
// 16 wavefronts split the rows, fixed-order combination through LDS.
__global__ __launch_bounds__(1024) void k_gemv_cols(const double* __restrict__ M, long ld, int nrows, int ncols,
                                                    const double* __restrict__ x, const double* __restrict__ base,
                                                    double* __restrict__ out) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (k < ncols)
        for (int i = wave; i < nrows; i += 16) acc += M[(long)i * ld + k] * x[i];
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && k < ncols) {
        double total = base ? base[k] : 0.0;
        for (int w = 0; w < 16; ++w) total += part[w][lane];
        out[k] = total;
    }
}

// out[j] = base[j] + sum_i A(i, col0 + j) * x[i]
__global__ __launch_bounds__(1024) void k_gemv_cols_A(AView A, int col0, int nq, int ncols, const double* __restrict__ x,
                                                      const double* __restrict__ base, double* __restrict__ out) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (j < ncols)
        for (int i = wave; i < nq; i += 16) acc += aval(A, i, col0 + j) * x[i];
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && j < ncols) {
        double total = base ? base[j] : 0.0;
        for (int w = 0; w < 16; ++w) total += part[w][lane];
        out[j] = total;
    }
}

// out[i] = add[i] + alpha * sum_k M[i*ld + k] * x[k]   (one wavefront per row)
__global__ __launch_bounds__(256) void k_gemv_rows(const double* __restrict__ M, long ld, int nrows, int ncols,
                                                   const double* __restrict__ x, double alpha,
                                                   const double* __restrict__ add, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const double* row = M + (long)i * ld;
    double acc = 0.0;
    for (int k = lane; k < ncols; k += 64) acc += row[k] * x[k];
    acc = wave_sum(acc);
    if (lane == 0) out[i] = (add ? add[i] : 0.0) + alpha * acc;
}

// out[i] = sum_k A-row(i)[k] * coef[k] over the 1+m stored columns (Lagrangian gradient)
__global__ __launch_bounds__(256) void k_jt_times(const double* __restrict__ jt, long ld, int n, int width,
                                                  const double* __restrict__ coef, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = jt + (long)i * ld;
    double acc = 0.0;
    for (int k = lane; k < width; k += 64) acc += row[k] * coef[k];
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
}

__global__ void k_concat_neg(const double* a, int na, const double* b, int nb, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na)
        out[i] = a[i];
    else if (i < na + nb)
        out[i] = -b[i - na];
}

// ------------------------------------------------------------------------------------------
// Least-distance problem.  Constraint stack: r < mg general rows (GJ), then nq lower bounds, then
// nq upper bounds (rows of Jw, negated for the upper ones).
struct GiState {
    int phase;  // 0 select, 1 continue with p, 2 solved, 3 iteration limit, 4 incompatible
    int p;
    int q;
    int iters;
    int cur;  // which of the two R / RI buffers is live
    unsigned ticket;
    double up;
    double ynorm;
    int dbg;
    int dbg2;
    int warm_removals;  // k_rows_resident: changes the warm start's pairs in front of it had made
    int arrive;         // rows mode: arrivals at the barrier between the two products of a spread warm-start removal (its own
                        // word since round 6: `cur` is the live-buffer index of the two older kernels)
    long long tr[40];   // OGSQP_TRACE: accumulated s_memtime ticks (10 ns) per section of the update
};

struct GiPartial {
    double value;
    int index;
    int pad;
};

struct GiArgs {
    const double* GJ;
    const double* Jw;
    int ld, meq, nq, mg, nr, qcap;
    const double* bval;   // mg + 2 nq
    const double* scale;  // mg + 2 nq, 0 = not usable
    const double* own;    // mg + 2 nq
    double* u;            // mg + 2 nq
    int* isact;           // mg + 2 nq
    double* y;            // nr
    int* act;             // qcap
    double* R[2];
    double* RI[2];
    double* Q1t;          // qcap x nr
    GiPartial* partials;
    GiState* st;
    int limit;
};

__device__ __forceinline__ const double* stack_row(const GiArgs& g, int r, double& sign) {
    if (r < g.mg) {
        sign = 1.0;
        return g.GJ + (long)r * g.ld + g.meq;
    }
    r -= g.mg;
    if (r < g.nq) {
        sign = 1.0;
        return g.Jw + (long)r * g.ld + g.meq;
    }
    sign = -1.0;
    return g.Jw + (long)(r - g.nq) * g.ld + g.meq;
}

// Row norms over all nq columns and over the null-space columns; b, scale, own of every stack row.
// A row whose projection on the null space vanishes (an inequality or bound that repeats an
// equality) cannot be influenced by y: scale 0 takes it out of the problem, as LSEI's elimination does.
__global__ __launch_bounds__(256) void k_ldp_setup(const double* __restrict__ GJ, const double* __restrict__ Jw, int ld,
                                                   int meq, int nq, int mg, const double* __restrict__ bG,
                                                   const double* __restrict__ cin, const double* __restrict__ deq,
                                                   const double* __restrict__ dl, const double* __restrict__ du,
                                                   double* __restrict__ bval, double* __restrict__ scale,
                                                   double* __restrict__ own, int* flag) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= mg + nq) return;
    const double* row = r < mg ? GJ + (long)r * ld : Jw + (long)(r - mg) * ld;
    double full = 0.0, red = 0.0;
    for (int k = lane; k < nq; k += 64) {
        const double v = row[k];
        full += v * v;
        if (k >= meq) red += v * v;
    }
    full = sqrt(wave_sum(full));
    red = sqrt(wave_sum(red));
    if (lane != 0) return;
    const bool movable = red > DEPENDENT * full;
    if (r < mg) {
        const double b = bG[r];
        bval[r] = b;
        scale[r] = movable ? red : 0.0;
        own[r] = movable ? FEASIBLE * fabs(b) / red : 0.0;
    } else {
        const int i = r - mg;
        const double lo = dl[i], hi = du[i];
        const bool has_lo = isfinite(lo), has_hi = isfinite(hi);
        const double blo = deq[i] - lo, bhi = hi - deq[i];
        bval[mg + i] = has_lo ? blo : 0.0;
        bval[mg + nq + i] = has_hi ? bhi : 0.0;
        scale[mg + i] = (has_lo && movable) ? red : 0.0;
        scale[mg + nq + i] = (has_hi && movable) ? red : 0.0;
        own[mg + i] = (has_lo && movable) ? FEASIBLE * fabs(blo) / red : 0.0;
        own[mg + nq + i] = (has_hi && movable) ? FEASIBLE * fabs(bhi) / red : 0.0;
    }
}

__global__ void k_gi_init(GiArgs g, const int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int mt = g.mg + 2 * g.nq;
    if (i < mt) {
        g.u[i] = 0.0;
        g.isact[i] = 0;
    }
    if (i < g.nr) g.y[i] = 0.0;
    if (i == 0) {
        GiState s;
        s.phase = flag[1] ? 4 : 0;
        s.p = -1;
        s.q = 0;
        s.iters = 0;
        s.cur = 0;
        s.ticket = 0u;
        s.up = 0.0;
        s.ynorm = 0.0;
        s.dbg = 0;
        s.dbg2 = 0;
        s.warm_removals = 0;
        s.arrive = 0;
        for (int e = 0; e < 40; ++e) s.tr[e] = 0;
        *g.st = s;
    }
}

#ifdef OGSQP_TRACE
#define TRACE_MARK(slot)                                        \
    do {                                                        \
        const long long now_ = __builtin_amdgcn_s_memtime();    \
        if (tid == 0) st->tr[slot] += now_ - t_mark;            \
        t_mark = now_;                                          \
    } while (0)
#else
#define TRACE_MARK(slot) do { } while (0)
#endif

// One active-set iteration.
__global__ __launch_bounds__(GI_THREADS) void k_gi_iter(GiArgs g) {
    extern __shared__ double lds[];
    __shared__ double redv[GI_WAVES];
    __shared__ int redi[GI_WAVES];
    __shared__ int s_last;
    GiState* st = g.st;
    const int phase0 = st->phase;
    if (phase0 >= 2) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = g.nr, mg = g.mg, nq = g.nq;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memtime();
#endif

    // ---- all workgroups: constraint values and the most violated usable row -----------------
    {
        const double slack = FEASIBLE * st->ynorm;
        double best = INFINITY;
        int besti = 0x7fffffff;
        const int r = blockIdx.x * GI_WAVES + wave;
        if (r < mg + nq) {
            const double* row = (r < mg ? g.GJ + (long)r * g.ld : g.Jw + (long)(r - mg) * g.ld) + g.meq;
            double dot = 0.0;
            for (int k = lane; k < nr; k += 64) dot += row[k] * g.y[k];
            dot = wave_sum(dot);
            if (r < mg) {
                if (g.scale[r] > 0.0 && !g.isact[r]) {
                    best = (g.bval[r] + dot) / g.scale[r] + g.own[r] + slack;
                    besti = r;
                }
            } else {
                const int lo = r, hi = r + nq;
                if (g.scale[lo] > 0.0 && !g.isact[lo]) {
                    best = (g.bval[lo] + dot) / g.scale[lo] + g.own[lo] + slack;
                    besti = lo;
                }
                if (g.scale[hi] > 0.0 && !g.isact[hi]) {
                    const double v = (g.bval[hi] - dot) / g.scale[hi] + g.own[hi] + slack;
                    if (v < best) {
                        best = v;
                        besti = hi;
                    }
                }
            }
        }
        block_argmin(best, besti, redv, redi);
        if (tid == 0) {
            // publish at agent scope (write-through, no L2 flush), then take a ticket; the stores are
            // acknowledged before the ticket is issued, so whoever draws the last ticket sees them all
            __hip_atomic_store(&g.partials[blockIdx.x].value, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g.partials[blockIdx.x].index, besti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == gridDim.x - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
    }
    TRACE_MARK(0);   // phase A of the last workgroup + wait for the others

    // ---- last workgroup: the update ------------------------------------------------------------
    const int qcap = g.qcap;
    double* nv = lds;                 // normal of p in the null-space coordinates
    double* zv = nv + nr;             // its component orthogonal to the active normals
    double* av = zv + nr;             // Q1' n, accumulated over the passes  (new column of R)
    double* ainc = av + qcap;         // Q1' vec of the current pass
    double* rv = ainc + qcap;         // R^-1 Q1' n                         (dual step direction)
    double* cs = rv + qcap;           // cosine / sine of the rotations of a removal (2 qcap)
    double* carried = cs + 2 * qcap;  // row being rotated downwards, indexed by old column
    double* red = carried + qcap;

    if (tid == 0) st->ticket = 0u;
    int p;
    if (phase0 == 0) {
        double v = INFINITY;
        int idx = 0x7fffffff;
        for (int b = tid; b < (int)gridDim.x; b += GI_THREADS) {
            const double pv = __hip_atomic_load(&g.partials[b].value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int pi = __hip_atomic_load(&g.partials[b].index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pv < v || (pv == v && pi < idx)) {
                v = pv;
                idx = pi;
            }
        }
        block_argmin(v, idx, redv, redi);
        if (!(v < 0.0)) {
            if (tid == 0) st->phase = 2;
            return;
        }
        p = idx;
        if (tid == 0) {
            st->p = p;
            st->up = 0.0;
        }
    } else {
        p = st->p;
    }
    TRACE_MARK(1);   // election
    const int iters = st->iters + 1;
    if (iters > g.limit) {
        if (tid == 0) st->phase = 3;
        return;
    }
    const int q = st->q, cur = st->cur;
    if (p < 0 || p >= mg + 2 * nq || q < 0 || q > nr || q > qcap) {   // cannot happen; never index with it
        if (tid == 0) {
            st->phase = 3;
            st->dbg = 100 + q;
            st->dbg2 = p;
        }
        return;
    }
    const double up_old = (phase0 == 0) ? 0.0 : st->up;
    double* R = g.R[cur];
    double* RI = g.RI[cur];
    double* Q1t = g.Q1t;              // q x nr, row j = j-th orthonormal direction of the active normals

    double psign;
    const double* prow = stack_row(g, p, psign);
    double part_y = 0.0, part_nn = 0.0;
    for (int i = tid; i < nr; i += GI_THREADS) {
        const double v = psign * prow[i];
        nv[i] = v;
        part_y += v * g.y[i];
        part_nn += v * v;
    }
    for (int j = tid; j < q; j += GI_THREADS) av[j] = 0.0;
    const double sp = g.bval[p] + block_sum(part_y, red);
    const double nn = block_sum(part_nn, red);
    TRACE_MARK(2);   // load normal

    // z = (I - Q1 Q1') n by classical Gram-Schmidt, repeated ("twice is enough")
    for (int pass = 0; pass <= REFINE; ++pass) {
        const double* vec = pass == 0 ? nv : zv;
        for (int j0 = wave * 4; j0 < q; j0 += GI_WAVES * 4) {
            // four directions per trip: 4 x nr/64 independent loads in flight per lane
            const int cnt = min(4, q - j0);
            const double* row0 = Q1t + (long)j0 * nr;
            const double* row1 = cnt > 1 ? row0 + nr : row0;
            const double* row2 = cnt > 2 ? row0 + 2 * (long)nr : row0;
            const double* row3 = cnt > 3 ? row0 + 3 * (long)nr : row0;
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
            for (int i = lane; i < nr; i += 64) {
                const double x = vec[i];
                d0 += row0[i] * x;
                d1 += row1[i] * x;
                d2 += row2[i] * x;
                d3 += row3[i] * x;
            }
            d0 = wave_sum(d0);
            d1 = wave_sum(d1);
            d2 = wave_sum(d2);
            d3 = wave_sum(d3);
            if (lane == 0) {
                ainc[j0] = d0;
                if (cnt > 1) ainc[j0 + 1] = d1;
                if (cnt > 2) ainc[j0 + 2] = d2;
                if (cnt > 3) ainc[j0 + 3] = d3;
            }
        }
        __syncthreads();
        TRACE_MARK(3);   // projections Q1' vec
        double part_corr = 0.0;
        for (int j = tid; j < q; j += GI_THREADS) {
            av[j] += ainc[j];
            part_corr += ainc[j] * ainc[j];
        }
        const double corr = pass == 0 ? INFINITY : block_sum(part_corr, red);
        for (int i = tid; i < nr; i += GI_THREADS) {
            double acc0 = vec[i], acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            const double* col = Q1t + i;
            int j = 0;
            for (; j + 32 <= q; j += 32) {
                double l[32];
#pragma unroll
                for (int e = 0; e < 32; ++e) l[e] = col[(long)(j + e) * nr];
#pragma unroll
                for (int e = 0; e < 32; e += 4) {
                    acc0 -= ainc[j + e] * l[e];
                    acc1 -= ainc[j + e + 1] * l[e + 1];
                    acc2 -= ainc[j + e + 2] * l[e + 2];
                    acc3 -= ainc[j + e + 3] * l[e + 3];
                }
            }
            for (; j + 4 <= q; j += 4) {
                const double l0 = col[(long)j * nr], l1 = col[(long)(j + 1) * nr], l2 = col[(long)(j + 2) * nr],
                             l3 = col[(long)(j + 3) * nr];
                acc0 -= ainc[j] * l0;
                acc1 -= ainc[j + 1] * l1;
                acc2 -= ainc[j + 2] * l2;
                acc3 -= ainc[j + 3] * l3;
            }
            for (; j < q; ++j) acc0 -= ainc[j] * col[(long)j * nr];
            zv[i] = (acc0 + acc1) + (acc2 + acc3);
        }
        __syncthreads();
        TRACE_MARK(4);   // z update
        if (tid == 0) st->tr[10] += 1;
        if (pass >= 1 && !(corr > (REORTH * REORTH) * nn)) break;
        if (pass == 0) {
            // Daniel-Gragg-Kaufman-Stewart: no second pass when the first one cancelled little
            double part_z0 = 0.0;
            for (int i = tid; i < nr; i += GI_THREADS) part_z0 += zv[i] * zv[i];
            if (block_sum(part_z0, red) > 0.5 * nn) break;
        }
    }
    // r = R^-1 a with the explicit inverse (column sweep, contiguous in i)
    for (int i = tid; i < q; i += GI_THREADS) {
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        const double* col = RI + i;
        int j = i;
        for (; j + 32 <= q; j += 32) {
            double l[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) l[e] = col[(long)(j + e) * qcap];
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                acc0 += l[e] * av[j + e];
                acc1 += l[e + 1] * av[j + e + 1];
                acc2 += l[e + 2] * av[j + e + 2];
                acc3 += l[e + 3] * av[j + e + 3];
            }
        }
        for (; j < q; ++j) acc0 += col[(long)j * qcap] * av[j];
        rv[i] = (acc0 + acc1) + (acc2 + acc3);
    }
    double part_zz = 0.0;
    for (int i = tid; i < nr; i += GI_THREADS) part_zz += zv[i] * zv[i];
    const double zz = block_sum(part_zz, red);
    // q == nr: the active normals already span the null space, nothing is independent of them
    const bool dependent = (q >= nr) || !(zz > (DEPENDENT * DEPENDENT) * nn);

    // ratio test over the active rows
    double t1 = INFINITY;
    int kdrop = 0x7fffffff;
    for (int j = tid; j < q; j += GI_THREADS)
        if (rv[j] > 0.0) {
            const double cand = g.u[g.act[j]] / rv[j];
            if (cand < t1) {
                t1 = cand;
                kdrop = j;
            }
        }
    block_argmin(t1, kdrop, redv, redi);
    TRACE_MARK(5);   // r, |z|, ratio test
    const double t2 = dependent ? INFINITY : -sp / zz;
    const double t = fmin(t1, t2);
    if (!(t < INFINITY)) {
        if (tid == 0) {
            st->phase = 4;
            st->iters = iters;
        }
        return;
    }
    for (int j = tid; j < q; j += GI_THREADS) g.u[g.act[j]] -= t * rv[j];
    const double up = up_old + t;
    double part_yy = 0.0;
    for (int i = tid; i < nr; i += GI_THREADS) {
        double yi = g.y[i];
        if (!dependent) {
            yi += t * zv[i];
            g.y[i] = yi;
        }
        part_yy += yi * yi;
    }
    const double ynorm = sqrt(block_sum(part_yy, red));
    const bool full_step = (t2 < INFINITY) && (t2 <= t1);
    TRACE_MARK(6);   // u, y update
    if (full_step) {
        // p joins the active set: Q1 gets z/|z|, R the column [a; |z|], RI the column [-r/|z|; 1/|z|]
        const double delta = sqrt(zz), inv = 1.0 / delta;
        for (int i = tid; i < q; i += GI_THREADS) {
            R[(long)q * qcap + i] = av[i];
            RI[(long)q * qcap + i] = -rv[i] * inv;
        }
        for (int i = tid; i < nr; i += GI_THREADS) Q1t[(long)q * nr + i] = zv[i] * inv;
        if (tid == 0) {
            R[(long)q * qcap + q] = delta;
            RI[(long)q * qcap + q] = inv;
            g.act[q] = p;
            g.u[p] = up;
            g.isact[p] = 1;
            st->q = q + 1;
            st->phase = 0;
            st->iters = iters;
            st->ynorm = ynorm;
        }
        TRACE_MARK(7);   // append
        return;
    }

    // ---- partial step: active row k leaves ------------------------------------------------------
    const int k = kdrop;
    if (k < 0 || k >= q) {
        if (tid == 0) {
            st->phase = 3;
            st->dbg = -7;
            st->dbg2 = k;
        }
        return;
    }
    double* Rn = g.R[cur ^ 1];
    double* RIn = g.RI[cur ^ 1];
    // unchanged parts: columns before k; rows above k of the columns after k (shifted left)
    for (int c = wave; c < q; c += GI_WAVES) {
        if (c == k) continue;
        const int cn = c < k ? c : c - 1;
        const int top = c < k ? c + 1 : k;
        for (int i = lane; i < top; i += 64) Rn[(long)cn * qcap + i] = R[(long)c * qcap + i];
        if (c < k)
            for (int i = lane; i <= c; i += 64) RIn[(long)c * qcap + i] = RI[(long)c * qcap + i];
    }
    for (int c = k + 1 + tid; c < q; c += GI_THREADS) carried[c] = R[(long)c * qcap + k];
    __syncthreads();
    if (wave == 0) {
        // Givens chain restoring the triangle of R without column k; one wavefront, lanes own the
        // columns congruent to them modulo 64
        for (int j = k; j < q - 1; ++j) {
            const int cp = j + 1;  // old column holding the pivot pair
            const double a = ((volatile double*)carried)[cp];
            const double b = R[(long)cp * qcap + j + 1];
            const double hyp = sqrt(a * a + b * b);
            const double co = hyp > 0.0 ? a / hyp : 1.0;
            const double si = hyp > 0.0 ? b / hyp : 0.0;
            if (lane == 0) {
                cs[2 * j] = co;
                cs[2 * j + 1] = si;
                Rn[(long)j * qcap + j] = hyp;
            }
            int c = cp + 1 + ((lane - (cp + 1)) % 64 + 64) % 64;
            for (; c < q; c += 64) {
                const double x = carried[c], yv = R[(long)c * qcap + j + 1];
                Rn[(long)(c - 1) * qcap + j] = co * x + si * yv;
                carried[c] = -si * x + co * yv;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    __syncthreads();
    // the same rotations on the columns of RI (row k and the last column drop out) ...
    for (int i = tid; i < q; i += GI_THREADS) {
        if (i == k) continue;
        const int in = i < k ? i : i - 1;
        double x = i < k ? RI[(long)k * qcap + i] : 0.0;
        for (int j = (i - 1 > k ? i - 1 : k); j < q - 1; ++j) {
            const double yv = RI[(long)(j + 1) * qcap + i];
            const double co = cs[2 * j], si = cs[2 * j + 1];
            RIn[(long)j * qcap + in] = co * x + si * yv;
            x = -si * x + co * yv;
        }
    }
    // ... and on the orthonormal directions (in place: every thread owns one coordinate)
    for (int i = tid; i < nr; i += GI_THREADS) {
        double* col = Q1t + i;
        double x = col[(long)k * nr];
        for (int j = k; j < q - 1; ++j) {
            const double yv = col[(long)(j + 1) * nr];
            const double co = cs[2 * j], si = cs[2 * j + 1];
            col[(long)j * nr] = co * x + si * yv;
            x = -si * x + co * yv;
        }
    }
    __syncthreads();
    int* shifted = (int*)carried;
    const int leaving = g.act[k];
    for (int j = k + tid; j < q - 1; j += GI_THREADS) shifted[j] = g.act[j + 1];
    __syncthreads();
    for (int j = k + tid; j < q - 1; j += GI_THREADS) g.act[j] = shifted[j];
    if (tid == 0) {
        g.u[leaving] = 0.0;
        g.isact[leaving] = 0;
        st->q = q - 1;
        st->cur = cur ^ 1;
        st->phase = 1;
        st->up = up;
        st->iters = iters;
        st->ynorm = ynorm;
        st->tr[11] += 1;
    }
    TRACE_MARK(8);   // removal
}

// ------------------------------------------------------------------------------------------
// The same active-set method spread over G workgroups (cooperative launch, resident for the whole
// LDP).  One CU streams the orthonormal basis at ~50 GB/s, which is what bounds k_gi_iter; here every
// workgroup owns a slice of `width` null-space coordinates - its columns of Q1, its part of the
// incoming normal, of z and of y - plus the rows of R^-1 whose storage slot is congruent to it, and a
// share of the pricing.  What has to be global (the projections Q1'n, |z|^2, the election of the
// incoming and of the leaving row) goes through small per-workgroup partial vectors in HBM and a
// grid barrier; the O(q) book-keeping (active list, slot map, free list) is replicated, every
// workgroup executing the same deterministic updates.  Only the Givens chain on R of a removal is
// serial (workgroup 0); the rotations are then applied by everybody to their own slices.
constexpr int COOP_THREADS = 512;   // 256 VGPRs per thread: the batched loads below need them
constexpr int COOP_WAVES = COOP_THREADS / 64;

struct CoopPartial {
    double metric;
    int index;
    int k;
    double ratio;
    double pad;
};

struct CoopArgs {
    GiArgs g;
    int G, nslices, width, astride;   // G workgroups, the first nslices of them own a slice
    double* Q1s;        // G x qcap x width
    double* RIr;        // qcap slots x qcap, row-major: row slot[i] = row i of the upper-triangular R^-1
    double* apart;      // 2 x G x astride (one set per Gram-Schmidt pass): [a_0 .. a_{q-1} | nn, sp, zz, -]
    double* zg;         // z, all slices
    double* rg;         // dual direction r, logical order
    double* cs;         // rotations of a removal
    CoopPartial* cpart; // G
    unsigned* bar;      // barrier counter
    int* abort_flag;
};

constexpr long COOP_SPIN_LIMIT = 200L * 1000 * 1000;

// Barrier over the cooperative grid: monotone counter, bounded spin (a lost workgroup aborts the
// solve instead of hanging the GPU).
__device__ __forceinline__ bool grid_barrier(const CoopArgs& c, unsigned& epoch) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // my workgroup's stores leave this XCD's L2
        __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)c.G;
        long spins = 0;
        int ok = 1;
        while (__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > COOP_SPIN_LIMIT ||
                ((spins & 1023) == 0 && __hip_atomic_load(c.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(c.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // and the others' become visible here
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

// Everything the workgroups of k_gi_coop tell each other goes through agent-scope (write-through /
// L2-bypassing) loads and stores, and the grid barrier carries no fence: an acquire fence would
// invalidate this XCD's whole L2 and with it the workgroup's private slice of Q1 and rows of R^-1,
// which is exactly what has to stay close.
__device__ __forceinline__ double ld_shared(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_shared(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_shared(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_shared(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Slice kernels of k_gi_coop.  `width` is 16 or 64: with 16 coordinates per workgroup a wavefront
// works on four rows of the slice at a time (one per 16-lane DPP row), with 64 on one.
__device__ __forceinline__ double row16_sum(double v) {       // sum over each aligned group of 16 lanes
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x128>(v);
    return v;
}

// out[j] = Q1s[j][:] . vec  for j < q  (my columns only)
__device__ __forceinline__ void slice_project(const double* __restrict__ Q1s, int width, int cw, int q,
                                              const double* vec, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (width == 16) {
        const int grp = lane >> 4, col = lane & 15;
        const double x = col < cw ? vec[col] : 0.0;
        for (int j0 = wave * 4; j0 < q; j0 += COOP_WAVES * 4) {
            const int j = j0 + grp;
            double v = (j < q && col < cw) ? Q1s[(long)j * 16 + col] * x : 0.0;
            v = row16_sum(v);
            if (col == 0 && j < q) st_shared(out + j, v);
        }
    } else {
        const double x = lane < cw ? vec[lane] : 0.0;
        for (int j = wave; j < q; j += COOP_WAVES) {
            double v = lane < cw ? Q1s[(long)j * width + lane] * x : 0.0;
            v = wave_sum(v);
            if (lane == 0) st_shared(out + j, v);
        }
    }
}

// zsl[i] = base[i] - sum_j coef[j] Q1s[j][i]  on my columns; `scratch` holds COOP_THREADS doubles
__device__ __forceinline__ void slice_subtract(const double* __restrict__ Q1s, int width, int cw, int q,
                                               const double* coef, const double* base, double* zsl,
                                               double* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc = 0.0;
    if (width == 16) {
        const int grp = lane >> 4, col = lane & 15;
        if (col < cw)
            for (int j = wave * 4 + grp; j < q; j += COOP_WAVES * 4) acc += coef[j] * Q1s[(long)j * 16 + col];
        scratch[(wave * 4 + grp) * 16 + col] = acc;       // 64 partial sums per column
        __syncthreads();
        if (tid < 16) {
            double sum = 0.0;
            for (int part = 0; part < COOP_WAVES * 4; ++part) sum += scratch[part * 16 + tid];
            zsl[tid] = tid < cw ? base[tid] - sum : 0.0;
        }
    } else {
        if (lane < cw)
            for (int j = wave; j < q; j += COOP_WAVES) acc += coef[j] * Q1s[(long)j * width + lane];
        scratch[wave * 64 + lane] = acc;
        __syncthreads();
        if (tid < 64) {
            double sum = 0.0;
            for (int part = 0; part < COOP_WAVES; ++part) sum += scratch[part * 64 + tid];
            zsl[tid] = tid < cw ? base[tid] - sum : 0.0;
        }
    }
    __syncthreads();
}

// sum over the G workgroups of entry j of their partial vectors, eight loads in flight
__device__ __forceinline__ double sum_partials(const double* __restrict__ apart, int astride, int G, int j) {
    // 32 loads issued before the first is consumed: one memory round trip per 32 workgroups, fixed order
    double acc = 0.0;
    for (int b0 = 0; b0 < G; b0 += 32) {
        double l[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) l[e] = b0 + e < G ? ld_shared(apart + (long)(b0 + e) * astride + j) : 0.0;
#pragma unroll
        for (int e = 0; e < 32; ++e) acc += l[e];
    }
    return acc;
}

// Entries j < q of the summed partial vectors into vec[], the scalar slots (qcap + 0 .. 3) into sc[]:
// one pass, every entry by its own thread.
__device__ __forceinline__ void gather_partials(const double* __restrict__ apart, int astride, int G, int q, int qcap,
                                                double* vec, double* sc) {
    for (int j = threadIdx.x; j < q + 4; j += COOP_THREADS) {
        const int idx = j < q ? j : qcap + (j - q);
        const double v = sum_partials(apart, astride, G, idx);
        if (j < q)
            vec[j] = v;
        else
            sc[j - q] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(COOP_THREADS) void k_gi_coop(CoopArgs c) {
    extern __shared__ double lds[];
    __shared__ double redv[COOP_WAVES];
    __shared__ int redi[COOP_WAVES];
    __shared__ double red[COOP_WAVES];
    __shared__ double scal[4];
    const GiArgs& g = c.g;
    GiState* st = g.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x, G = c.G, NS = c.nslices, width = c.width;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap;
    const int mt = mg + 2 * nq;
    const int c0 = w * width;
    const int cw = max(0, min(width, nr - c0));          // my null-space coordinates: c0 .. c0+cw-1
    // LDS.  Replicated in every workgroup (identical, updated by identical code): y, the
    // multipliers and the list of the active rows, their storage slots, the free slots, the
    // active-row bitmap.  Private: my slices of the incoming normal and of z.
    double* av = lds;                   // accumulated projections (new column of R); Givens carry (workgroup 0)
    double* a2 = av + qcap;             // projections of the second pass | dual direction r; rotations (2 qcap)
    double* uact = a2 + 2 * qcap;       // multipliers of the active rows, logical order
    double* nsl = uact + qcap;          // slices: incoming normal, z
    double* zsl = nsl + width;
    double* yfull = zsl + width;        // y
    double* zfull = yfull + nr;         // z gathered from all slices; diagonal of R (workgroup 0, removal)
    double* scratch = zfull + nr;       // COOP_THREADS partial sums of the slice kernels
    int* act = (int*)(scratch + COOP_THREADS);   // active rows, logical order
    int* slot = act + qcap;             // storage slot of every active row (rows of RIr)
    int* freel = slot + qcap;           // free slots, a stack
    unsigned* abits = (unsigned*)(freel + qcap);   // bit r: stack row r is active
    double* rv = a2 + qcap;
    double* carried = av;
    double* diagc = zfull;
    double* Q1s = c.Q1s + (long)(w < c.nslices ? w : 0) * qcap * width;
    // partial vectors exist for the slice owners only (the others own no column of Q1)
    const int wp = w < NS ? w : 0;
    double* part[2] = {c.apart + (long)wp * c.astride, c.apart + (long)(NS + wp) * c.astride};
    const double* parts[2] = {c.apart, c.apart + (long)NS * c.astride};

#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memtime();
    long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CMARK(slot) do { const long long now_ = __builtin_amdgcn_s_memtime(); t_acc[slot] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define CMARK(slot) do { } while (0)
#endif
    unsigned epoch = 0;
    int q = 0, iters = 0, cur = 0, nfree = qcap, phase = st->phase;
    double ynorm = 0.0;
    for (int i = tid; i < qcap; i += COOP_THREADS) freel[i] = qcap - 1 - i;      // pop order 0, 1, 2, ...
    for (int i = tid; i < nr; i += COOP_THREADS) yfull[i] = 0.0;
    for (int i = tid; i < (mt + 31) / 32; i += COOP_THREADS) abits[i] = 0u;
    __syncthreads();
    if (phase >= 2) return;

    for (;;) {
        // ---- pricing: my share of the rows at the current y ------------------------------------
        int p;
        {
            const double slack = FEASIBLE * ynorm;
            double best = INFINITY;
            int besti = 0x7fffffff;
            // my rows in groups of PG: the row data and the per-row constants of a whole group are
            // requested together, so a group costs one memory round trip, not one per row and field
            constexpr int PG = 4;
            const int rstride = G * COOP_WAVES;
            for (int r0 = (w * COOP_WAVES) + wave; r0 < mg + nq; r0 += PG * rstride) {
                double dots[PG], sc_lo[PG], sc_hi[PG], bv_lo[PG], bv_hi[PG], ow_lo[PG], ow_hi[PG];
#pragma unroll
                for (int e = 0; e < PG; ++e) {
                    const int r = r0 + e * rstride;
                    const bool live = r < mg + nq;
                    const int lo = live ? r : 0, hi = (live && r >= mg) ? r + nq : lo;
                    sc_lo[e] = live ? g.scale[lo] : 0.0;
                    bv_lo[e] = g.bval[lo];
                    ow_lo[e] = g.own[lo];
                    sc_hi[e] = (live && r >= mg) ? g.scale[hi] : 0.0;
                    bv_hi[e] = g.bval[hi];
                    ow_hi[e] = g.own[hi];
                    dots[e] = 0.0;
                }
                if (nr <= 512) {
                    double x[PG][8];
#pragma unroll
                    for (int e = 0; e < PG; ++e) {
                        const int r = min(r0 + e * rstride, mg + nq - 1);
                        const double* row = (r < mg ? g.GJ + (long)r * g.ld : g.Jw + (long)(r - mg) * g.ld) + g.meq;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int k = lane + 64 * t;
                            x[e][t] = k < nr ? row[k] : 0.0;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < PG; ++e)
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int k = lane + 64 * t;
                            if (k < nr) dots[e] += x[e][t] * yfull[k];
                        }
                } else {
#pragma unroll
                    for (int e = 0; e < PG; ++e) {
                        const int r = min(r0 + e * rstride, mg + nq - 1);
                        const double* row = (r < mg ? g.GJ + (long)r * g.ld : g.Jw + (long)(r - mg) * g.ld) + g.meq;
                        double acc = 0.0;
                        int k = lane;
                        for (; k + 64 * 15 < nr; k += 64 * 16) {
                            double l[16];
#pragma unroll
                            for (int t = 0; t < 16; ++t) l[t] = row[k + 64 * t];
#pragma unroll
                            for (int t = 0; t < 16; ++t) acc += l[t] * yfull[k + 64 * t];
                        }
                        for (; k < nr; k += 64) acc += row[k] * yfull[k];
                        dots[e] = acc;
                    }
                }
#pragma unroll
                for (int e = 0; e < PG; ++e) {
                    const int r = r0 + e * rstride;
                    const double dot = wave_sum(dots[e]);
                    if (r >= mg + nq) continue;
                    if (sc_lo[e] > 0.0 && !((abits[r >> 5] >> (r & 31)) & 1u)) {
                        const double v = (bv_lo[e] + dot) / sc_lo[e] + ow_lo[e] + slack;
                        if (v < best || (v == best && r < besti)) {
                            best = v;
                            besti = r;
                        }
                    }
                    const int hi = r + nq;
                    if (r >= mg && sc_hi[e] > 0.0 && !((abits[hi >> 5] >> (hi & 31)) & 1u)) {
                        const double v = (bv_hi[e] - dot) / sc_hi[e] + ow_hi[e] + slack;
                        if (v < best || (v == best && hi < besti)) {
                            best = v;
                            besti = hi;
                        }
                    }
                }
            }
            block_argmin(best, besti, redv, redi);
            if (tid == 0) {
                st_shared(&c.cpart[w].metric, best);
                st_shared(&c.cpart[w].index, besti);
            }
            CMARK(0);   // pricing
            if (!grid_barrier(c, epoch)) return;
            CMARK(1);   // barrier 1
            double v = INFINITY;
            int idx = 0x7fffffff;
            if (tid < G) {                          // one partial per thread, then the usual reduction:
                v = ld_shared(&c.cpart[tid].metric);            // the same answer in every workgroup
                idx = ld_shared(&c.cpart[tid].index);
            }
            block_argmin(v, idx, redv, redi);
            if (!(v < 0.0)) {
                phase = 2;
                break;
            }
            p = idx;
        }
        double up = 0.0;
        bool leave = false;
        // ---- steps with p until it joins the active set ------------------------------------------
        for (;;) {
            ++iters;
            if (iters > g.limit || q > nr || q > qcap) {
                phase = 3;
                leave = true;
                break;
            }
            double psign;
            const double* prow = stack_row(g, p, psign) + c0;
            // my slice of the normal; partial projections on my columns of Q1, |n|^2 and n.y
            if (tid < width) nsl[tid] = tid < cw ? psign * prow[tid] : 0.0;
            __syncthreads();
            if (w < NS) slice_project(Q1s, width, cw, q, nsl, part[0]);
            if (w < NS) {
                double pn = 0.0, py = 0.0;
                if (tid < cw) {
                    pn = nsl[tid] * nsl[tid];
                    py = nsl[tid] * yfull[c0 + tid];
                }
                pn = block_sum(pn, red);
                py = block_sum(py, red);
                if (tid == 0) {
                    st_shared(part[0] + qcap, pn);
                    st_shared(part[0] + qcap + 1, py);
                }
            }
            CMARK(2);   // projections 1
            if (!grid_barrier(c, epoch)) return;
            CMARK(3);   // barrier 2
            gather_partials(parts[0], c.astride, NS, q, qcap, av, scal);
            const double nn = scal[0];
            const double sp = g.bval[p] + scal[1];
            // z = n - Q1 a on my coordinates; |z|^2 = |n|^2 - |a|^2 needs no further reduction
            slice_subtract(Q1s, width, cw, q, av, nsl, zsl, scratch);
            double aa_part = 0.0;
            for (int j = tid; j < q; j += COOP_THREADS) aa_part += av[j] * av[j];
            double zz = fmax(nn - block_sum(aa_part, red), 0.0);
            CMARK(4);   // gather a, z1
            // Daniel-Gragg-Kaufman-Stewart: a second Gram-Schmidt pass only when the first one cancelled
            // more than half of |n|^2 (every workgroup takes the same branch: nn and a are the same bits)
            if (!(zz > 0.5 * nn)) {
                if (w < NS) slice_project(Q1s, width, cw, q, zsl, part[1]);
                if (w < NS) {
                    double pz = tid < cw ? zsl[tid] * zsl[tid] : 0.0;
                    pz = block_sum(pz, red);
                    if (tid == 0) st_shared(part[1] + qcap + 2, pz);
                }
                CMARK(6);   // projections 2
                if (!grid_barrier(c, epoch)) return;
                CMARK(7);   // barrier 3
                gather_partials(parts[1], c.astride, NS, q, qcap, a2, scal);
                double corr_part = 0.0;
                for (int j = tid; j < q; j += COOP_THREADS) {
                    av[j] += a2[j];
                    corr_part += a2[j] * a2[j];
                }
                zz = fmax(scal[2] - block_sum(corr_part, red), 0.0);      // |z - Q1 a2|^2 = |z|^2 - |a2|^2
                slice_subtract(Q1s, width, cw, q, a2, zsl, zsl, scratch);
            }
            const bool dependent = (q >= nr) || !(zz > (DEPENDENT * DEPENDENT) * nn);
            // my part of z and my entries of the dual direction r = R^-1 a go out for everybody
            if (tid < cw) st_shared(c.zg + c0 + tid, zsl[tid]);
            double t1 = INFINITY;
            int kdrop = 0x7fffffff;
            for (int i = wave; i < q; i += COOP_WAVES) {
                if (slot[i] % G != w) continue;
                const double* row = c.RIr + (long)slot[i] * qcap;
                double dot = 0.0;
                for (int j = i + lane; j < q; j += 64) dot += row[j] * av[j];
                dot = wave_sum(dot);
                if (lane == 0) {
                    st_shared(c.rg + i, dot);
                    if (dot > 0.0) {
                        const double cand = uact[i] / dot;
                        if (cand < t1 || (cand == t1 && i < kdrop)) {
                            t1 = cand;
                            kdrop = i;
                        }
                    }
                }
            }
            if (lane != 0) {
                t1 = INFINITY;
                kdrop = 0x7fffffff;
            }
            block_argmin(t1, kdrop, redv, redi);
            if (tid == 0) {
                st_shared(&c.cpart[w].ratio, t1);
                st_shared(&c.cpart[w].k, kdrop);
            }
            CMARK(8);   // z2, dual direction
            if (!grid_barrier(c, epoch)) return;
            CMARK(9);   // barrier 4
            t1 = INFINITY;
            kdrop = 0x7fffffff;
            if (tid < G) {
                t1 = ld_shared(&c.cpart[tid].ratio);
                kdrop = ld_shared(&c.cpart[tid].k);
            }
            block_argmin(t1, kdrop, redv, redi);
            const double t2 = dependent ? INFINITY : -sp / zz;
            const double t = fmin(t1, t2);
            if (!(t < INFINITY)) {
                phase = 4;
                leave = true;
                break;
            }
            // everybody updates its copy of the multipliers and of y from the gathered r and z
            for (int i = tid; i < q; i += COOP_THREADS) {
                const double ri = ld_shared(c.rg + i);
                rv[i] = ri;
                uact[i] -= t * ri;
            }
            up += t;
            double pyy = 0.0;
            for (int i = tid; i < nr; i += COOP_THREADS) {
                double yi = yfull[i];
                if (!dependent) {
                    yi += t * ld_shared(c.zg + i);
                    yfull[i] = yi;
                }
                pyy += yi * yi;
            }
            ynorm = sqrt(block_sum(pyy, red));
            const bool full_step = (t2 < INFINITY) && (t2 <= t1);
            if (full_step) {
                // p joins: Q1 gets z/|z| (my columns), R the column [a; |z|] (workgroup 0), R^-1 the
                // column [-r/|z|; 1/|z|] (row owners), in the first free slot
                const double delta = sqrt(zz), inv = 1.0 / delta;
                const int sl = freel[nfree - 1];
                if (tid < cw) Q1s[(long)q * width + tid] = zsl[tid] * inv;
                for (int i = tid; i < q; i += COOP_THREADS)
                    if (slot[i] % G == w) c.RIr[(long)slot[i] * qcap + q] = -rv[i] * inv;
                if (sl % G == w && tid == 0) c.RIr[(long)sl * qcap + q] = inv;
                if (w == 0) {
                    double* R = g.R[cur];
                    for (int i = tid; i < q; i += COOP_THREADS) R[(long)q * qcap + i] = av[i];
                    if (tid == 0) R[(long)q * qcap + q] = delta;
                }
                __syncthreads();
                if (tid == 0) {
                    act[q] = p;
                    slot[q] = sl;
                    uact[q] = up;
                    abits[p >> 5] |= 1u << (p & 31);
                }
                --nfree;
                ++q;
                __syncthreads();
                CMARK(10);  // updates, append
                break;
            }
            // ---- partial step: active row k leaves --------------------------------------------------
            const int k = kdrop;
            if (w == 0) {
                double* R = g.R[cur];
                double* Rn = g.R[cur ^ 1];
                for (int cc = wave; cc < q; cc += COOP_WAVES) {
                    if (cc == k) continue;
                    const int cn = cc < k ? cc : cc - 1;
                    const int top = cc < k ? cc + 1 : k;
                    for (int i = lane; i < top; i += 64) Rn[(long)cn * qcap + i] = R[(long)cc * qcap + i];
                }
                for (int cc = k + 1 + tid; cc < q; cc += COOP_THREADS) {
                    carried[cc] = R[(long)cc * qcap + k];
                    diagc[cc] = R[(long)cc * qcap + cc];
                }
                __syncthreads();
                if (wave == 0) {
                    for (int j = k; j < q - 1; ++j) {
                        const int cp = j + 1;
                        const double a = ((volatile double*)carried)[cp];
                        const double b = diagc[cp];
                        const double hyp = sqrt(a * a + b * b);
                        const double co = hyp > 0.0 ? a / hyp : 1.0;
                        const double si = hyp > 0.0 ? b / hyp : 0.0;
                        if (lane == 0) {
                            st_shared(c.cs + 2 * j, co);
                            st_shared(c.cs + 2 * j + 1, si);
                            Rn[(long)j * qcap + j] = hyp;
                        }
                        int cc = cp + 1 + ((lane - (cp + 1)) % 64 + 64) % 64;
                        for (; cc < q; cc += 64) {
                            const double x = carried[cc], yv = R[(long)cc * qcap + j + 1];
                            Rn[(long)(cc - 1) * qcap + j] = co * x + si * yv;
                            carried[cc] = -si * x + co * yv;
                        }
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_s_waitcnt(0);
                    }
                }
                __syncthreads();
            }
            if (!grid_barrier(c, epoch)) return;         // rotations published
            double* cs = a2;                              // (second-pass projections and r are spent)
            for (int j = k + tid; j < q - 1; j += COOP_THREADS) {
                cs[2 * j] = ld_shared(c.cs + 2 * j);
                cs[2 * j + 1] = ld_shared(c.cs + 2 * j + 1);
            }
            __syncthreads();
            // my columns of Q1: rotate rows k .. q-1, the last one drops out
            if (tid < cw) {
                double* col = Q1s + tid;
                double x = col[(long)k * width];
                for (int j = k; j < q - 1; ++j) {
                    const double yv = col[(long)(j + 1) * width];
                    const double co = cs[2 * j], si = cs[2 * j + 1];
                    col[(long)j * width] = co * x + si * yv;
                    x = -si * x + co * yv;
                }
            }
            // my rows of R^-1: the same rotations on their columns, shifted left by one
            for (int i = wave; i < q; i += COOP_WAVES) {
                if (i == k || slot[i] % G != w || lane != 0) continue;
                double* row = c.RIr + (long)slot[i] * qcap;
                double x = i < k ? row[k] : 0.0;
                for (int j = (i - 1 > k ? i - 1 : k); j < q - 1; ++j) {
                    const double yv = row[j + 1];
                    const double co = cs[2 * j], si = cs[2 * j + 1];
                    row[j] = co * x + si * yv;
                    x = -si * x + co * yv;
                }
            }
            __syncthreads();
            // replicated book-keeping: row k leaves the lists, its slot is free again
            {
                const int freed = slot[k], gone = act[k];
                int a_keep = 0, s_keep = 0;
                double u_keep = 0.0;
                for (int base = k; base < q - 1; base += COOP_THREADS) {
                    const int j = base + tid;
                    if (j < q - 1) {
                        a_keep = act[j + 1];
                        s_keep = slot[j + 1];
                        u_keep = uact[j + 1];
                    }
                    __syncthreads();
                    if (j < q - 1) {
                        act[j] = a_keep;
                        slot[j] = s_keep;
                        uact[j] = u_keep;
                    }
                    __syncthreads();
                }
                if (tid == 0) {
                    freel[nfree] = freed;
                    abits[gone >> 5] &= ~(1u << (gone & 31));
                }
                ++nfree;
                --q;
                cur ^= 1;
                __syncthreads();
            }
        }
        if (leave) break;
    }
    // ---- results: y and the multipliers back to the per-constraint arrays, state for the host -----
    if (w == 0) {
        for (int i = tid; i < nr; i += COOP_THREADS) g.y[i] = yfull[i];
        for (int j = tid; j < q; j += COOP_THREADS) g.u[act[j]] = uact[j];
        if (tid == 0) {
            st->phase = phase;
            st->q = q;
            st->iters = iters;
            st->cur = cur;
            st->ynorm = ynorm;
#ifdef OGSQP_TRACE
            for (int e = 0; e < 12; ++e) st->tr[e] = t_acc[e];
#endif
        }
    }
#undef CMARK
}

#include "ogsqp_rows.h"
#include "ogsqp_resident.h"

// d = clip(deq + Y y), multipliers of the general inequalities and of the bounds
__global__ __launch_bounds__(256) void k_finish_step(const double* __restrict__ Jw, int ld, int meq, int nq, int nr,
                                                     int mg, const double* __restrict__ y, const double* __restrict__ deq,
                                                     const double* __restrict__ dl, const double* __restrict__ du,
                                                     const double* __restrict__ u, double* __restrict__ d,
                                                     double* __restrict__ bm) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    const double* row = Jw + (long)i * ld + meq;
    double acc = 0.0;
    for (int k = lane; k < nr; k += 64) acc += row[k] * y[k];
    acc = wave_sum(acc);
    if (lane == 0) {
        double v = deq[i] + acc;
        if (isfinite(dl[i])) v = fmax(v, dl[i]);
        if (isfinite(du[i])) v = fmin(v, du[i]);
        d[i] = v;
        bm[i] = u[mg + i] - u[mg + nq + i];
    }
}

// tvec[i] = g[i] - sum_j A(i, meq + j) mu[j] - bm[i]
__global__ __launch_bounds__(256) void k_dual_residual(AView A, int meq, int mg, int nq, const double* __restrict__ gvec,
                                                       const double* __restrict__ mu, const double* __restrict__ bm,
                                                       double* __restrict__ tvec) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    double acc = 0.0;
    for (int j = lane; j < mg; j += 64) acc += aval(A, i, meq + j) * mu[j];
    acc = wave_sum(acc);
    if (lane == 0) tvec[i] = gvec[i] - acc - bm[i];
}

// Z[i][k] -= s[i] * vz[k] * inv_alpha
__global__ void k_rank1(double* Z, int ld, int n, const double* s, const double* vz, double inv_alpha) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (k < n) Z[(long)i * ld + k] -= s[i] * (vz[k] * inv_alpha);
}

}  // namespace

struct og_qp_s {
    int device = 0;
    int n = 0, n1 = 0, ldw = 0, meq = 0, mg = 0, m = 0, qcap = 0;   // ldw: row pitch of the n1-wide matrices (128-byte rows)
    hipStream_t stream = nullptr;
    double *Z = nullptr, *Jw = nullptr, *Tc = nullptr, *GJ = nullptr, *diagL = nullptr, *Vp = nullptr;
    LqPanel* panel = nullptr;
    double *extra = nullptr, *g = nullptr, *c = nullptr, *dl = nullptr, *du = nullptr;
    double *w1 = nullptr, *t1 = nullptr, *xcat = nullptr, *deq = nullptr, *bG = nullptr;
    double *bval = nullptr, *scale = nullptr, *own = nullptr, *u = nullptr, *y = nullptr;
    double *R[2] = {nullptr, nullptr}, *RI[2] = {nullptr, nullptr}, *Q1t = nullptr;
    double *apart = nullptr, *uact = nullptr, *csbuf = nullptr, *zg = nullptr, *dthresh = nullptr;
    CoopPartial* cpart = nullptr;
    unsigned* bar = nullptr;
    int* abort_flag = nullptr;
    int gi_mode = 0;                   // 0 rows (k_rows_decide / k_rows_apply, the default), 1 the two older kernels
    int rows_r4 = 0;                   // rounds 3-4's register form of the pass (k_rows_apply_r4): 0 for rows of up to 512
                                       // coordinates, where it measured faster (C3: 6.8-7.0 against 8.0 us; C4's 589 coordinates:
                                       // 10.9 against 9.0 us for this round's form), 1 (OGSQP_ROWS=r4) up to 1024, -1 (=lds) never
    int rows_stage = 0;                // ... their second pass out of LDS instead of the caches: only when forced (1)
    bool rows_stream = true;           // rows of more than 1024 null-space coordinates are streamed (k_rows_apply_stream);
                                       // OGSQP_ROWS=reg: the register kernels k_rows_apply<TAIL> for every length
    bool warm_enabled = true;          // start the active-set method from the previous subproblem's active rows
    std::vector<int> warm;             // ... in the canonical numbering of og_qp_get_active
    bool warm_use = true;              // ... when the last two solutions shared most of their active rows (early in an
                                       // SQP run they do not: taking the stale rows out again costs more than it saves)
    double *dots = nullptr, *dvec = nullptr, *rvec = nullptr, *uval = nullptr;
    bool warm_spread = true;           // removals of the warm start: both products with the inverse spread over the grid of
                                       // k_rows_decide (OGSQP_WARM_SPREAD=0: by one workgroup, rounds 3-4)
    GiPartial *price = nullptr, *ratio = nullptr;
    RowsDecision* rec = nullptr;
    bool resident = true;              // round 6: the active-set loop as ONE launch with every row of W and of the inverse in
                                       // registers (k_rows_resident, ogsqp_resident.h) where they fit the chip - up to 4096
                                       // rows of up to 1024 null-space coordinates: C3, C4; OGSQP_RESIDENT=0: the two-launch form
    unsigned long long* res_mail = nullptr;   // its mailbox (self-validating records) ...
    unsigned* res_seq = nullptr;              // ... and the exchange counters that go on counting from launch to launch
    int warm_pairs_hint = 4;           // two-launch pairs enqueued in front of it for the warm start's removals
    long resident_launches = 0, resident_changes = 0;
    int *d_warm = nullptr, *d_slot = nullptr;
    unsigned char* gemm_map = nullptr; // per block of 64 constraints: which slabs of 16 variables hold a non-zero
    double* trsv_work = nullptr;       // right-hand side of a triangular solve while the blocks are eliminated
    double* Linv = nullptr;            // inverses of the 64 x 64 diagonal blocks of L
    int* has_gone = nullptr;           // ... per block: it holds a vanishing pivot (no inverse)
    double* V16 = nullptr;             // reflector vectors of a 16-wide panel
    Lq16Panel* panel16 = nullptr;
    double* V16b = nullptr;            // ... of the panel the look-ahead factors during the trailing update
    Lq16Panel* panel16b = nullptr;
    // rows longer than one workgroup's registers (n + 1 > 2048): column-split panels, 64-reflector blocks, GEMMs (ogsqp_lqwide.h)
    bool lq_wide = false;              // (on from og_qp_create when n + 1 > LQW_SLAB; OGSQP_LQ=8 / 16 and OGSQP_WIDE=0 turn it off)
    double* Vall = nullptr;            // reflector vectors of the whole sweep, row k = reflector k (zero left of its panel)
    Lq16Panel* panelw = nullptr;       // T of the panel being applied inside a block
    LqWideMail* wide_mail = nullptr;
    unsigned* wide_count = nullptr;    // monotone count of the panel workgroups' steps, ever
    unsigned wide_token = 0u;
    bool wide_inblock = false;         // OGSQP_WIDE_INBLOCK=1: the later rows of a block get a panel's reflectors in ONE
    unsigned wide_in_token = 0u;       // launch (k_wy_inblock) instead of three (product over slices, finish, update) - same
                                       // bits; measured at C5: 7.00 instead of 7.07 s over 121 subproblems, not worth a
                                       // kernel that waits for its neighbours: off by default
    double* wy_part = nullptr;         // column slices of a product (k_wy_w with blockIdx.y > 0), summed by k_wy_sum
    size_t wy_part_cap = 0;
    double *wy_w = nullptr, *wy_m = nullptr, *wy_t = nullptr, *wy_small = nullptr;   // 2 x (rows x 64) coefficients, M, T = M^-1 (one per block of a sweep), 2 x (64 x 16)
    // The block reflectors are applied to the rest of C Z and to Z on two streams of their own while the caller's stream
    // goes on with the next block's panels (OGSQP_WIDE_AHEAD=0: everything on the caller's stream, one after the other)
    struct WyLane {
        hipStream_t s = nullptr;       // lane 0: the stream of the call
        double *w = nullptr, *part = nullptr;
    } lane[3];
    bool wide_ahead = false;
    std::vector<hipEvent_t> ev_t, ev_tc;   // per block: T is there (caller's stream) / the rest of C Z has it applied (lane 1)
    hipEvent_t ev_join[2] = {nullptr, nullptr};
    int spin_limit = 1 << 19;          // bound of the inter-workgroup waits: polls of ~1.5 us each, i.e. about a second (2^25 - a
                                       // minute per lost wait - until round 5); OGSQP_SPIN_LIMIT: tests force a loss with 1
    int recoveries = 0;                // subproblems re-run with the separate-launch forms after a wait gave up
    bool lq_ahead = true;              // OGSQP_LQ=16: panel and trailing update as separate launches
    int trsv_mode = 0;                 // OGSQP_TRSV: 0 one chained launch, 1 ("block") a launch per block, 2 ("single")
    unsigned* lq_go = nullptr;         // look-ahead: head workgroups that have finished the next panel's rows, ever
    unsigned lq_token = 0u;            // ... and what the count will be after the launch being enqueued
    double* lq_wpart = nullptr;        // the head workgroups' partial products (LQ_HEADS x 64 x 4)
    bool lq16 = true;                  // OGSQP_LQ=8: the sweep of 8-reflector panels only
    int coop_mode = 1;                 // 0 never, 1 by size, 2 always (when it fits)
    int last_iters = 1000;             // active-set changes of the previous subproblem on this handle (a solve starts with many)
    double *d = nullptr, *bm = nullptr, *tvec = nullptr, *rhs = nullptr, *lam = nullptr, *vz = nullptr;
    double *svec = nullptr, *vvec = nullptr, *coef = nullptr, *outn = nullptr;
    int *isact = nullptr, *act = nullptr, *flag = nullptr;
    GiPartial* partials = nullptr;
    GiState* st = nullptr;
    double* jt_stage = nullptr;
    std::vector<double> host_stage;
    std::vector<void*> owned;
};

namespace {

template <typename T>
int dev_alloc(og_qp_s* qp, T** ptr, size_t count) {
    void* raw = nullptr;
    OG_HIP(hipMalloc(&raw, (count ? count : 1) * sizeof(T)));
    qp->owned.push_back(raw);
    *ptr = (T*)raw;
    return 0;
}

// OGSQP_DEBUG=1: announce every stage on stderr and synchronise after it, so that a device fault
// can be attributed to a kernel.
bool debug_stages() {
    static const bool on = getenv("OGSQP_DEBUG") != nullptr;
    return on;
}

#define OG_STAGE(name)                                                             \
    do {                                                                           \
        if (debug_stages()) {                                                      \
            OG_HIP(hipStreamSynchronize(s));                                       \
            fprintf(stderr, "[ogsqp] stage done; next: %s\n", name);               \
            fflush(stderr);                                                        \
        }                                                                          \
    } while (0)

#define OG_TRY(expr)          \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) return rc_;  \
    } while (0)

// L x = scale * rhs (or L' x): block by block over the whole chip (k_trsv_block); OGSQP_TRSV=single: the one-workgroup
// kernel.  `scal` is a device scalar of scratch.
int launch_trsv(og_qp_s* qp, int ldw, int meq, int transposed, double scale_rhs, const double* rhs, double* x,
                hipStream_t s);
int launch_gemm(og_qp_s* qp, const AView& A, int col0, int rows, int nq, int ldw, double* out, const int* sel,
                hipStream_t s);

size_t gi_lds_bytes(int nr, int qcap) {
    return (size_t)(2 * nr + 6 * qcap + 64) * sizeof(double);
}

size_t rows_lds_bytes(int nr, int qcap) {          // k_rows_decide: the incoming normal, the dual direction, a reflector
    return (size_t)(nr + 2 * qcap + 16) * sizeof(double);
}

}  // namespace

namespace {
// column slices of a product with `rows` rows of length L: enough workgroups for every SIMD of the chip (a workgroup of
// k_wy_w is two wavefronts that issue 16 MFMAs per 2 KB of A); few rows: at most 32 slices - the kernels that add the
// slices up walk them one after the other.  (k_wy_inblock uses the SAME slices: its sums are these sums.)
void wy_split(const og_qp_s* qp, int rows, int L, int* nsplit_out, int* kb_per_out) {
    const int tiles = (rows + 16 * WYW_WAVES - 1) / (16 * WYW_WAVES), nblk = (L + 15) / 16;
    // (1024 workgroups aimed at: 2048 / 4096 fill the SIMDs of the side streams better and cost the sums more - 7.04 against
    // 6.85 s over 121 subproblems of C5, profiles/r05_wyw_grid.jsonl)
    int nsplit = std::max(1, std::min(std::min(tiles <= 4 ? 32 : WYW_SPLIT_MAX, 1024 / tiles), nblk / 4));
    nsplit = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsplit, qp->wy_part_cap / ((size_t)rows * LQW_BLOCK)));
    const int kb_per = (nblk + nsplit - 1) / nsplit;
    *nsplit_out = (nblk + kb_per - 1) / kb_per;
    *kb_per_out = kb_per;
}

// W (rows x 64) = A V' by k_wy_w; few rows are split over the chip by columns (partials in wy_part, summed in order)
void launch_wy_w(og_qp_s* qp, const double* A, int ld, int rows, int L, const double* V, int ldv, int nb, double* W,
                 int* nsplit_out, hipStream_t s, double* part = nullptr) {
    if (!part) part = qp->wy_part;
    const int tiles = (rows + 16 * WYW_WAVES - 1) / (16 * WYW_WAVES);
    int nsplit = 1, kb_per = 1;
    wy_split(qp, rows, L, &nsplit, &kb_per);
    double* dst = nsplit > 1 ? part : W;
    hipLaunchKernelGGL(k_wy_w, dim3(tiles, nsplit), dim3(64 * WYW_WAVES), 0, s, A, ld, rows, L, V, ldv, nb, dst, kb_per);
    if (nsplit_out) {
        *nsplit_out = nsplit;                         // the caller sums the slices itself (k_wy_small_finish)
        return;
    }
    if (nsplit > 1) {
        const long count = (long)rows * LQW_BLOCK;
        hipLaunchKernelGGL(k_wy_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (const double*)part,
                           nsplit, count, W);
    }
}

// A <- A - C V with the coefficients C = W T (T != nullptr: 64 reflectors, W from launch_wy_w) or C = W (16 reflectors,
// coefficients ready): k_wy_update, the rows in tiles of 64, the columns in as many slices as fill the chip
void launch_wy_update(double* A, int ld, int rows, int L, const double* V, int ldv, int nb, const double* W, int ldc,
                      const double* T, hipStream_t s) {
    if (rows <= 0 || L <= 0) return;
    const int tiles = (rows + 16 * WYU_WAVES - 1) / (16 * WYU_WAVES), nblk = (L + 15) / 16;
    int nsplit = std::max(1, std::min(2048 / tiles, nblk / 4));
    const int kb_per = (nblk + nsplit - 1) / nsplit;
    nsplit = (nblk + kb_per - 1) / kb_per;
    if (T)
        hipLaunchKernelGGL((k_wy_update<LQW_BLOCK / 16, true>), dim3(tiles, nsplit), dim3(64 * WYU_WAVES), 0, s, A, ld, rows, L,
                           V, ldv, nb, W, ldc, T, kb_per);
    else
        hipLaunchKernelGGL((k_wy_update<1, false>), dim3(tiles, nsplit), dim3(64 * WYU_WAVES), 0, s, A, ld, rows, L, V, ldv,
                           nb, W, ldc, T, kb_per);
}

// rows <- rows - ((rows V') M^-1) V for `rows` rows of length L starting at A (leading dimension ld), V: nb reflectors
// (row-major, leading dimension ldv) over the same L columns, T = M^-1 from k_wy_make_m / k_wy_invert: two passes over
// the rows, both hand-written for the FP64 matrix cores (k_wy_w, k_wy_update)
void wy_apply_block(og_qp_s* qp, const og_qp_s::WyLane& ln, double* A, int ld, int rows, int L, const double* V, int ldv,
                    int nb, const double* T) {
    if (rows <= 0) return;
    double* W = ln.w;                                             // rows x LQW_BLOCK
    launch_wy_w(qp, A, ld, rows, L, V, ldv, nb, W, nullptr, ln.s, ln.part);                                    // W = A V'
    launch_wy_update(A, ld, rows, L, V, ldv, nb, W, LQW_BLOCK, T, ln.s);                                      // A -= (W T) V
}

// The sweep over rows longer than LQW_SLAB entries, from reflector k on, in blocks of LQW_BLOCK reflectors: returns the
// first reflector it did not handle (rows short enough for the look-ahead kernels, or msweep).
int lq_sweep_wide(og_qp_s* qp, int msweep, int nq, int ldw, int k, hipStream_t s, int* done) {
    og_qp_s::WyLane& l0 = qp->lane[0];
    l0.s = s, l0.w = qp->wy_w, l0.part = qp->wy_part;
    const bool ahead = qp->wide_ahead;
    int rc = 0, b = 0;
    // (a failing runtime call ends the loop through rc instead of returning: the exit below joins the lanes first)
#define WIDE_HIP(expr)                                                                                   \
    do {                                                                                                 \
        const hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess && !rc)                                                                     \
            rc = fail(5, std::string("og_qp_solve_dev: ") + #expr + " failed: " + hipGetErrorString(e_)); \
    } while (0)
    while (!rc && k < msweep && nq - k > LQW_SLAB) {
        const int k0 = k, nbk = std::min(LQW_BLOCK, msweep - k0), L0 = nq - k0;
        for (int sub = 0; sub < nbk && !rc; sub += LQ16) {
            const int kk = k0 + sub, nb16 = std::min(LQ16, msweep - kk), len = nq - kk;
            const int nwg = (len + LQW_SLAB - 1) / LQW_SLAB;
            double* V = qp->Vall + (size_t)kk * ldw + kk;
            hipLaunchKernelGGL(k_lq_panel16_wide, dim3(nwg), dim3(P16_THREADS), (size_t)2 * LQW_SLAB * sizeof(double), s, qp->Tc,
                               ldw, msweep, nq, kk, V, ldw, qp->diagL, qp->panelw, qp->dthresh + 1, qp->wide_mail,
                               (int)(qp->wide_token & 1u), qp->flag + 2, qp->spin_limit);
            ++qp->wide_token;                                  // the mailbox's generation alternates from launch to launch
            // the later rows of this block: rows <- rows - ((rows V16') T16) V16, T16 from the panel kernel
            const int rest = k0 + nbk - (kk + nb16);
            if (rest > 0) {
                double* A = qp->Tc + (size_t)(kk + nb16) * ldw + kk;
                int nsplit = 1, kb_per = 1;
                wy_split(qp, rest, len, &nsplit, &kb_per);
                if (qp->wide_inblock && rest <= 16 * WIB_WAVES && kb_per <= WIB_KB) {
                    // one launch: the slices' workgroups exchange their shares of the products (k_wy_inblock)
                    qp->wide_in_token += (unsigned)nsplit;
                    hipLaunchKernelGGL(k_wy_inblock, dim3(nsplit), dim3(64 * WIB_WAVES), 0, s, A, ldw, rest, len,
                                       (const double*)V, ldw, nb16, (const Lq16Panel*)qp->panelw, qp->wy_part, kb_per,
                                       qp->wide_count + 1, qp->wide_in_token, qp->flag + 2, qp->spin_limit);
                } else {
                    double* W2 = qp->wy_small;                              // rest x 16
                    launch_wy_w(qp, A, ldw, rest, len, V, ldw, LQ16, qp->wy_part, &nsplit, s);
                    hipLaunchKernelGGL(k_wy_small_finish, dim3(1), dim3(1024), 0, s, (const double*)qp->wy_part, nsplit, rest,
                                       (const Lq16Panel*)qp->panelw, W2);
                    launch_wy_update(A, ldw, rest, len, V, ldw, LQ16, W2, LQ16, nullptr, s);
                }
            }
        }
        if (rc) break;
        // the block reflector: M = T^-1 from the Gram matrix of its nbk reflector vectors (columns k0 .. nq); every block
        // of a sweep has its own T (the lanes that apply it may be a few blocks behind)
        const double* Vb = qp->Vall + (size_t)k0 * ldw + k0;
        double* T = qp->wy_t + (size_t)b * LQW_BLOCK * LQW_BLOCK;
        launch_wy_w(qp, Vb, ldw, nbk, L0, Vb, ldw, nbk, qp->wy_m, nullptr, s);                    // S = V V' (nbk x 64)
        hipLaunchKernelGGL(k_wy_make_m, dim3(1), dim3(256), 0, s, qp->wy_m, nbk);
        hipLaunchKernelGGL(k_wy_invert, dim3(1), dim3(LQW_BLOCK), 0, s, (const double*)qp->wy_m, nbk, T);
        // ... applied to what is left of C Z (and of the warm-start rows) and to Z
        double* Arest = qp->Tc + (size_t)(k0 + nbk) * ldw + k0;
        const int rows_rest = msweep - k0 - nbk;
        if (!ahead) {
            // (in the same two pieces as below: the column slices of a product - hence its rounding - depend on the
            // number of rows, and the two orders of execution are to give the same bits)
            const int nxt = std::min(LQW_BLOCK, rows_rest);
            wy_apply_block(qp, l0, Arest, ldw, nxt, L0, Vb, ldw, nbk, T);
            wy_apply_block(qp, l0, Arest + (size_t)nxt * ldw, ldw, rows_rest - nxt, L0, Vb, ldw, nbk, T);
            wy_apply_block(qp, l0, qp->Jw + k0, ldw, nq, L0, Vb, ldw, nbk, T);
        } else {
            // Only the next block's rows are needed before its panels can start: they are done here, on this stream; the
            // other rows of C Z (lane 1) and Z (lane 2) get the block on their own streams - the panels are one
            // latency chain on four workgroups, the applications stream through the whole matrix: side by side they use
            // what the other leaves idle.  Same arithmetic on disjoint rows: the results do not depend on the overlap.
            const og_qp_s::WyLane &l1 = qp->lane[1], &l2 = qp->lane[2];
            const int nxt = std::min(LQW_BLOCK, rows_rest);
            WIDE_HIP(hipEventRecord(qp->ev_t[b], s));
            if (b > 0) WIDE_HIP(hipStreamWaitEvent(s, qp->ev_tc[b - 1], 0));   // (lane 1 had these rows for block b - 1)
            wy_apply_block(qp, l0, Arest, ldw, nxt, L0, Vb, ldw, nbk, T);
            WIDE_HIP(hipStreamWaitEvent(l1.s, qp->ev_t[b], 0));
            wy_apply_block(qp, l1, Arest + (size_t)nxt * ldw, ldw, rows_rest - nxt, L0, Vb, ldw, nbk, T);
            WIDE_HIP(hipEventRecord(qp->ev_tc[b], l1.s));
            WIDE_HIP(hipStreamWaitEvent(l2.s, qp->ev_t[b], 0));
            wy_apply_block(qp, l2, qp->Jw + k0, ldw, nq, L0, Vb, ldw, nbk, T);
        }
        k = k0 + nbk;
        ++b;
    }
    if (ahead && b > 0 && !rc) {                                  // the caller's stream goes on when both lanes are through
        for (int l = 0; l < 2 && !rc; ++l) {
            WIDE_HIP(hipEventRecord(qp->ev_join[l], qp->lane[1 + l].s));
            WIDE_HIP(hipStreamWaitEvent(s, qp->ev_join[l], 0));
        }
    }
    if (rc && ahead) {
        // an error on the way out: what is already queued on the lanes must not go on writing C Z and Z under
        // the caller's feet - or under the next attempt's: both lanes are drained before the error is handed up
        for (int l = 1; l < 3; ++l) (void)hipStreamSynchronize(qp->lane[l].s);
    }
#undef WIDE_HIP
    *done = k;
    return rc;
}

// out = (the `rows` columns of A from col0 on, or the ones sel names)' Jw: the map of non-empty slabs, then the product
int launch_gemm(og_qp_s* qp, const AView& A, int col0, int rows, int nq, int ldw, double* out, const int* sel,
                hipStream_t s) {
    const int kblocks = (nq + 15) / 16, mblocks = (rows + 63) / 64;
    if (kblocks > 1024) return fail(4, "og_qp_solve_dev: more than 16384 variables");
    hipLaunchKernelGGL(k_gemm_map, dim3(kblocks, mblocks), dim3(256), 0, s, A, col0, rows, nq, sel, qp->gemm_map, kblocks);
    hipLaunchKernelGGL(k_gemm_tn, dim3((nq + 63) / 64, mblocks), dim3(256), 0, s, A, col0, rows, qp->Jw, ldw, nq, out, sel,
                       (const unsigned char*)qp->gemm_map, kblocks);
    return 0;
}

int launch_trsv(og_qp_s* qp, int ldw, int meq, int transposed, double scale_rhs, const double* rhs, double* x,
                hipStream_t s) {
    if (qp->trsv_mode == 2 || meq <= 128) {
        const size_t trsv_lds = (size_t)(meq + 64 * 65) * sizeof(double);
        hipLaunchKernelGGL(k_trsv, dim3(1), dim3(1024), trsv_lds, s, qp->Tc, ldw, qp->diagL, meq, transposed, scale_rhs,
                           rhs, x, qp->dthresh, qp->flag);
        return 0;
    }
    if (!transposed)       // (the forward solve comes first in a subproblem: the inverses serve the transposed one too)
        hipLaunchKernelGGL(k_trsv_invert, dim3((meq + 63) / 64), dim3(64), 0, s, qp->Tc, ldw, qp->diagL, meq, qp->dthresh,
                           qp->Linv, qp->has_gone);
    const int nblk = (meq + 63) / 64;
    if (qp->trsv_mode == 0 && nblk <= 128 && x != rhs) {
        // one launch: a chain of hand-offs between the block rows' workgroups (k_trsv_chain)
        hipLaunchKernelGGL(k_trsv_prepare, dim3(1), dim3(256), 0, s, rhs, meq, scale_rhs, qp->trsv_work, qp->dthresh + 2, x);
        hipLaunchKernelGGL(k_trsv_chain, dim3(nblk), dim3(256), 0, s, qp->Tc, ldw, qp->diagL, meq, transposed,
                           (const double*)qp->trsv_work, x, qp->dthresh + 2, qp->dthresh, qp->flag, (const double*)qp->Linv,
                           (const int*)qp->has_gone, qp->spin_limit);
        return 0;
    }
    hipLaunchKernelGGL(k_trsv_prepare, dim3(1), dim3(256), 0, s, rhs, meq, scale_rhs, qp->trsv_work, qp->dthresh + 2,
                       (double*)nullptr);
    for (int bi = 0; bi < nblk; ++bi) {
        const int b = transposed ? nblk - 1 - bi : bi;
        const int i0 = b * 64, bs = std::min(64, meq - i0);
        const int rest = transposed ? i0 : meq - i0 - bs;
        const int grid = std::max(1, transposed ? (rest + 255) / 256 : (rest + TRSV_ROWS - 1) / TRSV_ROWS);
        hipLaunchKernelGGL(k_trsv_block, dim3(grid), dim3(256), 0, s, qp->Tc, ldw, qp->diagL, meq, transposed, b,
                           qp->trsv_work, x, qp->dthresh + 2, qp->dthresh, qp->flag, (const double*)qp->Linv,
                           (const int*)qp->has_gone);
    }
    return 0;
}
}  // namespace

extern "C" {

const char* og_qp_last_error(void) { return g_error.c_str(); }

int og_qp_create(int32_t abi_version, int32_t device, int32_t n, int32_t m_eq, int32_t m_ineq, og_qp_handle* out) {
    if (abi_version != OGSQP_ABI_VERSION) return fail(1, "og_qp_create: ABI version mismatch");
    if (!out || n < 1 || m_eq < 0 || m_ineq < 0) return fail(2, "og_qp_create: bad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(3, "og_qp_create: no HIP device visible (the SQP core has no CPU fallback)");
    if (device < 0 || device >= count) return fail(3, "og_qp_create: no such device");
    OG_HIP(hipSetDevice(device));
    og_qp_s* qp = new og_qp_s();
    qp->device = device;
    qp->n = n;
    qp->n1 = n + 1;
    qp->ldw = (n + 1 + 15) / 16 * 16;
    qp->meq = m_eq;
    qp->mg = m_ineq;
    qp->m = m_eq + m_ineq;
    qp->qcap = qp->n1 - (m_eq < qp->n1 ? m_eq : qp->n1);
    if (qp->qcap < 1) qp->qcap = 1;
    const size_t n1 = qp->n1, ldw = qp->ldw, mt = (size_t)qp->mg + 2 * n1, qc = qp->qcap;
    if (rows_lds_bytes((int)qc, (int)qc) > LDS_LIMIT) {
        delete qp;
        return fail(4, "og_qp_create: null space of the equalities too large for the LDS-resident part of the "
                       "active-set update (n + 1 - m_eq = " + std::to_string(qc) + ")");
    }
    // rows of up to 8192 entries can go through round 2's one-workgroup panels (the forms every other sweep falls back
    // to); longer ones - up to 16384 - exist only for the column-split panels of the wide sweep (ogsqp_lqwide.h)
    if (n1 > (size_t)LQW_SLAB * LQW_MAX) {
        delete qp;
        return fail(4, "og_qp_create: more than " + std::to_string(LQW_SLAB * LQW_MAX - 1) + " variables");
    }
    const bool beyond_fallback = n1 > (size_t)LQ_PT_MAX * LQ_CPT_MAX;
    int rc = 0;
    auto A = [&](auto** p, size_t cnt) { if (!rc) rc = dev_alloc(qp, p, cnt); };
    // Tc and diagL: the equalities, then the rows of a warm start (at most qcap of them) behind them in the sweep
    A(&qp->Z, n1 * ldw); A(&qp->Jw, n1 * ldw); A(&qp->Tc, ((size_t)qp->meq + qc) * ldw); A(&qp->GJ, (size_t)qp->mg * ldw);
    A(&qp->diagL, qp->meq + qc); A(&qp->dots, (size_t)qp->mg + n1); A(&qp->dvec, n1); A(&qp->rvec, qc);
    A(&qp->price, 2048); A(&qp->ratio, 256); A(&qp->rec, 1); A(&qp->d_warm, qc); A(&qp->d_slot, qc); A(&qp->uval, qc);
    A(&qp->V16, (size_t)LQ16 * ldw); A(&qp->panel16, 1); A(&qp->V16b, (size_t)LQ16 * ldw); A(&qp->panel16b, 1); A(&qp->lq_go, 4); A(&qp->lq_wpart, (size_t)LQ_HEADS * 64 * 4); A(&qp->trsv_work, qp->meq); A(&qp->gemm_map, ((size_t)std::max(qp->meq, qp->mg) + 63 + qc) / 64 * ((n1 + 15) / 16) + 64); A(&qp->Linv, ((size_t)qp->meq + 63) / 64 * 4096); A(&qp->has_gone, ((size_t)qp->meq + 63) / 64); A(&qp->Vp, (size_t)LQ_NB * ldw); A(&qp->panel, 1); A(&qp->extra, qp->m); A(&qp->g, n1); A(&qp->c, qp->m); A(&qp->dl, n1); A(&qp->du, n1);
    A(&qp->w1, qp->meq); A(&qp->t1, n1); A(&qp->xcat, n1); A(&qp->deq, n1); A(&qp->bG, qp->mg);
    A(&qp->bval, mt); A(&qp->scale, mt); A(&qp->own, mt); A(&qp->u, mt); A(&qp->y, n1);
    A(&qp->Q1t, qc * (qc + 64)); A(&qp->apart, 2 * 64 * (qc + 8)); A(&qp->uact, qc); A(&qp->zg, n1); A(&qp->dthresh, 4); A(&qp->csbuf, 2 * qc);
    A(&qp->cpart, 256); A(&qp->bar, 1); A(&qp->abort_flag, 1); A(&qp->R[0], qc * qc); A(&qp->R[1], qc * qc); A(&qp->RI[0], qc * qc); A(&qp->RI[1], qc * qc);
    A(&qp->d, n1); A(&qp->bm, n1); A(&qp->tvec, n1); A(&qp->rhs, qp->meq); A(&qp->lam, qp->meq); A(&qp->vz, n1);
    A(&qp->svec, n1); A(&qp->vvec, n1); A(&qp->coef, qp->m + 1); A(&qp->outn, n1);
    A(&qp->isact, mt); A(&qp->act, qc); A(&qp->flag, 4);
    // the mailbox of the resident active-set launch (ogsqp_resident.h): sized for this handle's rows when they fit the chip
    const size_t res_wg_cap = ((size_t)qp->mg + n1 + qc + RES_ROWS - 1) / RES_ROWS;
    const size_t res_records = res_mail_records(res_wg_cap <= (size_t)RES_MAX_WG ? (int)res_wg_cap : 1, (int)qc);
    A(&qp->res_mail, 2 * res_records); A(&qp->res_seq, 4);
    if (!rc && (hipMemset(qp->res_mail, 0, 2 * res_records * sizeof(unsigned long long)) != hipSuccess ||
                hipMemset(qp->res_seq, 0, 4 * sizeof(unsigned)) != hipSuccess))
        rc = fail(5, "og_qp_create: hipMemset failed");
    A(&qp->partials, ((size_t)qp->mg + n1) / GI_WAVES + 2); A(&qp->st, 1);
    if (!rc && hipStreamCreate(&qp->stream) != hipSuccess) rc = fail(5, "og_qp_create: hipStreamCreate failed");
    if (!rc && hipMemset(qp->lq_go, 0, 4 * sizeof(unsigned)) != hipSuccess) rc = fail(5, "og_qp_create: hipMemset failed");
    if (!rc && hipFuncSetAttribute((const void*)k_gi_iter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_LIMIT) != hipSuccess)
        rc = fail(5, "og_qp_create: cannot raise the dynamic LDS limit");
    if (!rc && hipFuncSetAttribute((const void*)k_gi_coop, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_LIMIT) != hipSuccess)
        rc = fail(5, "og_qp_create: cannot raise the dynamic LDS limit");
    if (!rc && hipFuncSetAttribute((const void*)k_trsv, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_LIMIT) != hipSuccess)
        rc = fail(5, "og_qp_create: cannot raise the dynamic LDS limit");
    if (!rc && (hipFuncSetAttribute((const void*)k_rows_decide, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_LIMIT) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_rows_invert, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_LIMIT) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_rows_resident, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_LIMIT) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_rows_apply_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_LIMIT) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_rows_apply_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_LIMIT) != hipSuccess))
        rc = fail(5, "og_qp_create: cannot raise the dynamic LDS limit");
    {
        const void* steps[13] = {(const void*)k_lq_step16<2, 1>, (const void*)k_lq_step16<2, 2>, (const void*)k_lq_step16<4, 1>,
                                 (const void*)k_lq_step16<4, 2>, (const void*)k_lq_step16<8, 2>, (const void*)k_lq_step16<8, 3>,
                                 (const void*)k_lq_step16<8, 4>, (const void*)k_lq_step16<12, 4>, (const void*)k_lq_step16<12, 5>,
                                 (const void*)k_lq_step16<12, 6>, (const void*)k_lq_step16<16, 6>, (const void*)k_lq_step16<16, 7>,
                                 (const void*)k_lq_step16<16, 8>};
        for (const void* f : steps) {
            const hipError_t e = rc ? hipSuccess : hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess)
                rc = fail(5, std::string("og_qp_create: cannot raise the dynamic LDS limit of the look-ahead kernel: ") +
                                 hipGetErrorString(e));
        }
    }
    if (rc) {
        og_qp_destroy(qp);
        return rc;
    }
    {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCooperativeLaunch, device);
        // OGSQP_GI = "single": one-workgroup active-set update (k_gi_iter), "coop": the cooperative
        // multi-workgroup one (k_gi_coop); default: by size (the barriers of the cooperative kernel cost
        // 12 us per change, its bandwidth pays from a null space of about 500 on)
        const char* mode = getenv("OGSQP_GI");
        qp->coop_mode = !can ? 0 : (mode && std::string(mode) == "single") ? 0 : (mode && std::string(mode) == "coop") ? 2 : 1;
        // the default is the row-parallel method in rotated coordinates (ogsqp_rows.h); "single" / "coop" / "old"
        // select the two older kernels (kept for comparison; they need their own, smaller, LDS budget)
        qp->gi_mode = (mode && (std::string(mode) == "single" || std::string(mode) == "coop" || std::string(mode) == "old")) ? 1 : 0;
        if (qp->gi_mode == 1 && gi_lds_bytes((int)qc, (int)qc) > LDS_LIMIT) qp->gi_mode = 0;
        const char* resk = getenv("OGSQP_RESIDENT");
        qp->resident = !(resk && std::string(resk) == "0");
        const char* rowsk = getenv("OGSQP_ROWS");
        qp->rows_stream = !(rowsk && std::string(rowsk) == "reg");
        qp->rows_r4 = (rowsk && std::string(rowsk) == "r4") ? 1 : (rowsk && std::string(rowsk) == "lds") ? -1 : 0;
        qp->rows_stage = (rowsk && std::string(rowsk) == "stage") ? 1 : (rowsk && std::string(rowsk) == "nostage") ? -1 : 0;
        const char* lq = getenv("OGSQP_LQ");
        qp->lq16 = !(lq && std::string(lq) == "8");
        qp->lq_ahead = !(lq && std::string(lq) == "16");
        const char* tr = getenv("OGSQP_TRSV");
        qp->trsv_mode = (tr && std::string(tr) == "block") ? 1 : (tr && std::string(tr) == "single") ? 2 : 0;
        // rows longer than one workgroup holds: the wide sweep (ogsqp_lqwide.h) when the default kernels are selected
        // (OGSQP_WIDE=0: round 2's 8-reflector kernels serve those rows, as before round 4)
        const char* wide = getenv("OGSQP_WIDE");
        if (qp->lq16 && qp->lq_ahead && n1 > (size_t)LQW_SLAB && n1 <= (size_t)LQW_SLAB * LQW_MAX &&
            !(wide && std::string(wide) == "0")) {
            const size_t vrows = ((size_t)qp->meq + qc + LQW_BLOCK - 1) / LQW_BLOCK * LQW_BLOCK + LQW_BLOCK;
            A(&qp->Vall, vrows * ldw); A(&qp->panelw, 1); A(&qp->wide_mail, 1); A(&qp->wide_count, 4);
            A(&qp->wy_w, 2 * (n1 + vrows) * LQW_BLOCK); A(&qp->wy_m, (size_t)LQW_BLOCK * LQW_BLOCK);
            const size_t nblocks = vrows / LQW_BLOCK + 1;
            A(&qp->wy_t, nblocks * LQW_BLOCK * LQW_BLOCK);
            qp->wy_part_cap = std::max((size_t)WYW_SPLIT_MAX * 2 * LQW_BLOCK * LQW_BLOCK, 4 * (n1 + vrows) * LQW_BLOCK);
            A(&qp->wy_part, qp->wy_part_cap);
            A(&qp->wy_small, (size_t)2 * LQ16 * LQW_BLOCK);
            if (!rc && hipMemset(qp->wide_count, 0, 4 * sizeof(unsigned)) != hipSuccess)
                rc = fail(5, "og_qp_create: hipMemset failed");
            const char* inblock = getenv("OGSQP_WIDE_INBLOCK");
            qp->wide_inblock = inblock && std::string(inblock) == "1";
            if (!rc && hipMemset(qp->Vall, 0, vrows * ldw * sizeof(double)) != hipSuccess)
                rc = fail(5, "og_qp_create: hipMemset failed");
            if (!rc) {
                const int cells = (int)(sizeof(LqWideMail) / sizeof(double));
                hipLaunchKernelGGL(k_lq_wide_arm, dim3((cells + 255) / 256), dim3(256), 0, 0, qp->wide_mail);
                if (hipDeviceSynchronize() != hipSuccess) rc = fail(5, "og_qp_create: arming the panel mailbox failed");
            }
            const char* wahead = getenv("OGSQP_WIDE_AHEAD");
            if (!rc && !(wahead && std::string(wahead) == "0")) {
                for (int l = 1; l < 3 && !rc; ++l) {
                    og_qp_s::WyLane& ln = qp->lane[l];
                    A(&ln.w, 2 * (n1 + vrows) * LQW_BLOCK);
                    A(&ln.part, qp->wy_part_cap);
                    if (!rc && hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking) != hipSuccess)
                        rc = fail(5, "og_qp_create: hipStreamCreate failed");
                    if (!rc && hipEventCreateWithFlags(&qp->ev_join[l - 1], hipEventDisableTiming) != hipSuccess)
                        rc = fail(5, "og_qp_create: hipEventCreate failed");
                }
                qp->ev_t.assign(nblocks, nullptr);
                qp->ev_tc.assign(nblocks, nullptr);
                for (size_t e = 0; e < nblocks && !rc; ++e)
                    if (hipEventCreateWithFlags(&qp->ev_t[e], hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&qp->ev_tc[e], hipEventDisableTiming) != hipSuccess)
                        rc = fail(5, "og_qp_create: hipEventCreate failed");
                qp->wide_ahead = !rc;
            }
            if (rc) {
                og_qp_destroy(qp);
                return rc;
            }
            qp->lq_wide = true;
        }
        if (beyond_fallback && !qp->lq_wide) {
            og_qp_destroy(qp);
            return fail(4, "og_qp_create: rows of more than " + std::to_string(LQ_PT_MAX * LQ_CPT_MAX) +
                               " entries need the wide sweep (OGSQP_LQ / OGSQP_WIDE must not turn it off)");
        }
        // (beyond 8192 variables a lost wait cannot fall back to a form without one: the long bound of rounds 3-4 there)
        if (beyond_fallback) qp->spin_limit = 1 << 25;
        const char* spin = getenv("OGSQP_SPIN_LIMIT");
        if (spin && atoi(spin) > 0) qp->spin_limit = atoi(spin);
        const char* wspread = getenv("OGSQP_WARM_SPREAD");
        qp->warm_spread = !(wspread && std::string(wspread) == "0");
        const char* warm = getenv("OGSQP_WARM");
        qp->warm_enabled = !(warm && std::string(warm) == "0");
    }
    *out = qp;
    return og_qp_reset(qp);
}

void og_qp_destroy(og_qp_handle qp) {
    if (!qp) return;
    (void)hipSetDevice(qp->device);
    for (void* p : qp->owned) (void)hipFree(p);
    for (int l = 1; l < 3; ++l) {
        if (qp->lane[l].s) (void)hipStreamDestroy(qp->lane[l].s);
        if (qp->ev_join[l - 1]) (void)hipEventDestroy(qp->ev_join[l - 1]);
    }
    for (hipEvent_t e : qp->ev_t) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : qp->ev_tc) if (e) (void)hipEventDestroy(e);
    if (qp->jt_stage) (void)hipFree(qp->jt_stage);
    if (qp->stream) (void)hipStreamDestroy(qp->stream);
    delete qp;
}

int og_qp_reset(og_qp_handle qp) {
    if (!qp) return fail(2, "og_qp_reset: null handle");
    OG_HIP(hipSetDevice(qp->device));
    dim3 grid((qp->n1 + 255) / 256, qp->n1);
    hipLaunchKernelGGL(k_identity, grid, dim3(256), 0, qp->stream, qp->Z, qp->ldw, qp->n1);
    OG_HIP(hipGetLastError());
    OG_HIP(hipStreamSynchronize(qp->stream));
    return 0;
}

int og_qp_get_factor(og_qp_handle qp, double* Z) {
    if (!qp || !Z) return fail(2, "og_qp_get_factor: null argument");
    OG_HIP(hipSetDevice(qp->device));
    OG_HIP(hipMemcpy2D(Z, (size_t)qp->n * sizeof(double), qp->Z, (size_t)qp->ldw * sizeof(double),
                       (size_t)qp->n * sizeof(double), qp->n, hipMemcpyDeviceToHost));
    return 0;
}

int og_qp_set_factor(og_qp_handle qp, const double* Z) {
    if (!qp || !Z) return fail(2, "og_qp_set_factor: null argument");
    OG_HIP(hipSetDevice(qp->device));
    OG_HIP(hipMemcpy2D(qp->Z, (size_t)qp->ldw * sizeof(double), Z, (size_t)qp->n * sizeof(double),
                       (size_t)qp->n * sizeof(double), qp->n, hipMemcpyHostToDevice));
    return 0;
}

// One attempt at the subproblem.  *lost = 1 (and nothing of the handle's state changed: the factor, the warm-start
// list) when an inter-workgroup wait of the look-ahead sweep or of the chained triangular solves gave up.
static int qp_solve_attempt(og_qp_handle qp, const double* d_jt, int64_t ld, const double* g, const double* c,
                            const double* dl, const double* du, int32_t augmented, double rho, double* d, double* mult,
                            double* bound_mult, int32_t* status, int32_t* iterations, void* hip_stream, int* lost) {
    if (!qp || !d_jt || !g || !dl || !du || !d || !mult || !bound_mult || !status)
        return fail(2, "og_qp_solve_dev: null argument");
    if (qp->m > 0 && !c) return fail(2, "og_qp_solve_dev: null constraint values");
    if (ld < 1 + qp->m) return fail(2, "og_qp_solve_dev: leading dimension smaller than 1 + m");
    if (augmented && !(rho > 0.0)) return fail(2, "og_qp_solve_dev: rho must be positive");
    OG_HIP(hipSetDevice(qp->device));
    hipStream_t s = qp->stream;
    if (hip_stream) OG_HIP(hipStreamSynchronize((hipStream_t)hip_stream));   // producer of d_jt
    const int n = qp->n, n1 = qp->n1, ldw = qp->ldw, meq = qp->meq, mg = qp->mg, m = qp->m;
    const int nq = augmented ? n + 1 : n;
    const int nr = nq - meq;
    if (iterations) *iterations = 0;
    if (meq > nq) {
        *status = OG_QP_TOO_MANY_EQ;
        return 0;
    }
    // ---- inputs
    std::vector<double>& hs = qp->host_stage;
    hs.assign((size_t)3 * n1 + m, 0.0);
    double* hg = hs.data();
    double* hdl = hg + n1;
    double* hdu = hdl + n1;
    double* hc = hdu + n1;
    memcpy(hg, g, sizeof(double) * n);
    memcpy(hdl, dl, sizeof(double) * nq);
    memcpy(hdu, du, sizeof(double) * nq);
    if (m) memcpy(hc, c, sizeof(double) * m);
    OG_HIP(hipMemcpyAsync(qp->g, hg, sizeof(double) * n1, hipMemcpyHostToDevice, s));
    OG_HIP(hipMemcpyAsync(qp->dl, hdl, sizeof(double) * n1, hipMemcpyHostToDevice, s));
    OG_HIP(hipMemcpyAsync(qp->du, hdu, sizeof(double) * n1, hipMemcpyHostToDevice, s));
    if (m) OG_HIP(hipMemcpyAsync(qp->c, hc, sizeof(double) * m, hipMemcpyHostToDevice, s));
    OG_HIP(hipMemsetAsync(qp->flag, 0, 4 * sizeof(int), s));     // [2]: a look-ahead workgroup of the LQ sweep gave up waiting
    if (augmented && m)
        hipLaunchKernelGGL(k_relaxation_row, dim3((m + 255) / 256), dim3(256), 0, s, qp->c, meq, m, qp->extra);
    AView A{d_jt, (long)ld, qp->extra, n};
    // ---- work factor, C Z, LQ sweep
    OG_STAGE("copy_factor");
    hipLaunchKernelGGL(k_copy_factor, dim3((nq + 255) / 256, nq), dim3(256), 0, s, qp->Z, qp->Jw, ldw, n, nq,
                       augmented ? 1.0 / rho : 0.0);
    // ---- rows of a warm start: active at the solution of the previous subproblem, appended to the sweep
    int nwarm = 0;
    if (qp->gi_mode == 0 && qp->warm_enabled && qp->warm_use && nr > 0 && !qp->warm.empty()) {
        std::vector<int> general, bound;
        for (int id : qp->warm) {
            if (id < mg) {
                general.push_back(id);
            } else {
                const int i = (id - mg) >> 1, upper = (id - mg) & 1;
                if (i < nq) bound.push_back(mg + (upper ? nq : 0) + i);
            }
        }
        std::vector<int> ids(general);
        ids.insert(ids.end(), bound.begin(), bound.end());
        if ((int)ids.size() > nr) ids.resize(nr);
        nwarm = (int)ids.size();
        const int ng = std::min((int)general.size(), nwarm);
        if (nwarm) {
            OG_HIP(hipMemcpyAsync(qp->d_warm, ids.data(), sizeof(int) * nwarm, hipMemcpyHostToDevice, s));
            OG_HIP(hipStreamSynchronize(s));                   // ids is a local
            double* Text = qp->Tc + (size_t)meq * ldw;
            OG_STAGE("warm rows");
            if (ng)
                OG_TRY(launch_gemm(qp, A, meq, ng, nq, ldw, Text, (const int*)qp->d_warm, s));
            if (nwarm > ng)
                hipLaunchKernelGGL(k_rows_gather_bounds, dim3((nq + 255) / 256, nwarm - ng), dim3(256), 0, s, qp->Jw, ldw,
                                   nq, mg, (const int*)qp->d_warm, ng, nwarm, Text);
        }
    }
    const int msweep = meq + nwarm;                            // rows the sweep makes triangular
    if (msweep) {
        OG_STAGE("gemm C Z");
        if (meq)
            OG_TRY(launch_gemm(qp, A, 0, meq, nq, ldw, qp->Tc, (const int*)nullptr, s));
        OG_STAGE("lq sweep");
        OG_HIP(hipMemsetAsync(qp->dthresh, 0, 2 * sizeof(double), s));
        double* Vcur = qp->V16;
        double* Vnxt = qp->V16b;
        Lq16Panel* pcur = qp->panel16;
        Lq16Panel* pnxt = qp->panel16b;
        int factored = -1;                                  // the panel the previous launch factored on the side
        OG_HIP(hipMemsetAsync(qp->lq_go, 0, 2 * sizeof(unsigned), s)); // counts of the head workgroups, this sweep
        qp->lq_token = 0u;
        int kstart = 0;
        if (qp->lq_wide && nq > LQW_SLAB) {
            // long rows: column-split panels and 64-reflector blocks until the rows fit one workgroup (ogsqp_lqwide.h)
            OG_STAGE("lq sweep, wide blocks");
            OG_TRY(lq_sweep_wide(qp, msweep, nq, ldw, 0, s, &kstart));
        }
        for (int k = kstart; k < msweep;) {
            if (qp->lq16 && nq - k <= 2048 && k % LQ16 == 0) {
                // 16 reflectors per trip: row-distributed panel kernel, MFMA trailing update (ogsqp_lq16.h)
                const int nb16 = std::min(LQ16, msweep - k), len16 = nq - k;
                const int nrows16 = (msweep - k - nb16) + nq;
 const int eg = (len16 + 255) / 256, ub = (len16 + 127) / 128;
                // rows per workgroup of the trailing update: 16, or as few as fill the chip (not below 8: every
                // workgroup reads all of V)
                const int rpg = std::max(8, std::min(LQ16, (nrows16 + 239) / 240));
#define OG_PANEL16(E)                                                                                              \
    hipLaunchKernelGGL(k_lq_panel16<E>, dim3(1), dim3(P16_THREADS), (size_t)((E) <= 6 ? P16_RING : 2) * 256 * (E) * sizeof(double), s, qp->Tc, ldw, \
                       msweep, nq, k, Vcur, ldw, qp->diagL, pcur, qp->dthresh + 1)
                if (factored != k) {
                    // (E = groups of 256 columns, exactly: the panel's steps are bound by the multiply-adds it issues,
                    // padding included)
                    switch (eg) {
                        case 1: OG_PANEL16(1); break;
                        case 2: OG_PANEL16(2); break;
                        case 3: OG_PANEL16(3); break;
                        case 4: OG_PANEL16(4); break;
                        case 5: OG_PANEL16(5); break;
                        case 6: OG_PANEL16(6); break;
                        case 7: OG_PANEL16(7); break;
                        default: OG_PANEL16(8); break;
                    }
                }
#undef OG_PANEL16
#ifdef OGSQP_TRACE
                if ((k == 0 || k == 512) && factored != k) {
                    Lq16Panel hp;
                    OG_HIP(hipMemcpyAsync(&hp, pcur, sizeof(Lq16Panel), hipMemcpyDeviceToHost, s));
                    OG_HIP(hipStreamSynchronize(s));
                    fprintf(stderr, "[ogsqp trace] panel16 at k = %d (len %d) ticks (wavefront 0): load %lld first reflector %lld waits for the flag %lld vector %lld row in line %lld other row %lld store %lld\n",
                            k, len16, hp.tr[0], hp.tr[5], hp.tr[1], hp.tr[2], hp.tr[3], hp.tr[4], hp.tr[6]);
                }
#endif
#ifdef OGSQP_TRACE
#define OG_APPLY16_TRACE()                                                                                               \
    if (k == 0) {                                                                                                        \
        long long tr[8];                                                                                                 \
        OG_HIP(hipStreamSynchronize(s));                                                                                 \
        OG_HIP(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_apply16_trace), sizeof tr));                                         \
        fprintf(stderr, "[ogsqp trace] apply16 at k = 0 (len %d, %d rows per workgroup), 10 ns ticks: header %lld loads %lld " \
                        "product-1 %lld barrier %lld products-2,3 %lld stores %lld\n", len16, rpg, tr[0], tr[1], tr[2],  \
                tr[3], tr[4], tr[5]);                                                                                    \
        if (qp->lq_ahead) {                                                                                              \
            Lq16Panel hp;                                                                                                \
            OG_HIP(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_head16_trace), sizeof tr));                                      \
            OG_HIP(hipMemcpy(&hp, pnxt, sizeof(Lq16Panel), hipMemcpyDeviceToHost));                                      \
            fprintf(stderr, "[ogsqp trace] head workgroup 0, 10 ns ticks: header %lld loads %lld product-1 %lld barrier %lld " \
                            "exchange %lld products-2,3 + stores issued %lld drain %lld; its panel (shader clocks, wavefront 0): load %lld " \
                            "first reflector %lld waits for the flag %lld vector %lld row in line %lld other row %lld store %lld\n", tr[0], tr[1],  \
                    tr[2], tr[3], tr[6], tr[4], tr[5], hp.tr[0], hp.tr[5], hp.tr[1], hp.tr[2], hp.tr[3], hp.tr[4], hp.tr[6]); \
        }                                                                                                                \
    }
#else
#define OG_APPLY16_TRACE() do { } while (0)
#endif
                if (qp->lq_ahead && k + LQ16 < msweep) {
                    // the launch of the trailing update factors the next panel on the side (k_lq_step16)
#define OG_STEP16(U, E)                                                                                           \
    hipLaunchKernelGGL((k_lq_step16<U, E>), dim3(1 + LQ_HEADS + (std::max(nrows16 - LQ16, 0) + rpg - 1) / rpg),        \
                       dim3(64 * A16_WAVES), (size_t)((E) <= 6 ? P16_RING : 2) * 256 * (E) * sizeof(double), s, qp->Tc, qp->Jw, ldw, msweep, nq, k,   \
                       (const double*)Vcur, ldw, (const Lq16Panel*)pcur, Vnxt, pnxt, qp->diagL, qp->dthresh + 1, rpg,      \
                       qp->lq_go, (qp->lq_token += LQ_HEADS), qp->lq_wpart, qp->flag + 2, qp->spin_limit)
                    // U by the length of the rows now, E by the length of the NEXT panel's rows (exact)
                    const int en = (len16 - LQ16 + 255) / 256;
                    if (ub <= 2) { if (en <= 1) OG_STEP16(2, 1); else OG_STEP16(2, 2); }
                    else if (ub <= 4) { if (en <= 1) OG_STEP16(4, 1); else OG_STEP16(4, 2); }
                    else if (ub <= 8) { if (en <= 2) OG_STEP16(8, 2); else if (en == 3) OG_STEP16(8, 3); else OG_STEP16(8, 4); }
                    else if (ub <= 12) { if (en <= 4) OG_STEP16(12, 4); else if (en == 5) OG_STEP16(12, 5); else OG_STEP16(12, 6); }
                    else { if (en <= 6) OG_STEP16(16, 6); else if (en == 7) OG_STEP16(16, 7); else OG_STEP16(16, 8); }
#undef OG_STEP16
                    OG_APPLY16_TRACE();
                    factored = k + LQ16;
                    std::swap(Vcur, Vnxt);
                    std::swap(pcur, pnxt);
                    k += LQ16;
                    continue;
                }
#define OG_APPLY16(U)                                                                                            \
    hipLaunchKernelGGL(k_lq_apply16<U>, dim3((nrows16 + rpg - 1) / rpg), dim3(64 * A16_WAVES), 0, s, qp->Tc, qp->Jw, ldw, msweep, \
                       nq, k, (const double*)Vcur, ldw, (const Lq16Panel*)pcur, rpg)
                if (ub <= 2) OG_APPLY16(2);
                else if (ub <= 4) OG_APPLY16(4);
                else if (ub <= 8) OG_APPLY16(8);
                else if (ub <= 12) OG_APPLY16(12);
                else OG_APPLY16(16);
#undef OG_APPLY16
                OG_APPLY16_TRACE();
                k += LQ16;
                continue;
            }
            const int nb = std::min(LQ_NB, msweep - k);
            const int nrows = (msweep - k - nb) + nq;
            const int len = nq - k;                                // length of the panel rows
#define OG_PANEL(PT, CPT)                                                                                      \
    hipLaunchKernelGGL((k_lq_panel<PT, CPT>), dim3(1), dim3(PT), 0, s, qp->Tc, ldw, msweep, nq, k, qp->Vp, qp->diagL, \
                       qp->panel, qp->dthresh + 1)
            if (len <= PANEL_SMALL_PT * 4) OG_PANEL(PANEL_SMALL_PT, 4);
            else if (len <= PANEL_SMALL_PT * 8) OG_PANEL(PANEL_SMALL_PT, 8);
            else if (len <= PANEL_SMALL_PT * 12) OG_PANEL(PANEL_SMALL_PT, 12);
            else if (len <= LQ_PT_MAX * 4) OG_PANEL(LQ_PT_MAX, 4);
            else if (len <= LQ_PT_MAX * 8) OG_PANEL(LQ_PT_MAX, 8);
            else OG_PANEL(LQ_PT_MAX, LQ_CPT_MAX);
#ifdef OGSQP_TRACE
            if (k == 0) {
                LqPanel hp;
                OG_HIP(hipMemcpyAsync(&hp, qp->panel, sizeof(LqPanel), hipMemcpyDeviceToHost, s));
                OG_HIP(hipStreamSynchronize(s));
                fprintf(stderr, "[ogsqp trace] first panel kernel (len %d) ticks: load %lld products %lld reduction %lld update %lld gram %lld gram-red %lld store %lld\n",
                        len, hp.tr[0], hp.tr[1], hp.tr[2], hp.tr[3], hp.tr[4], hp.tr[5], hp.tr[6]);
            }
#endif
#undef OG_PANEL
            {
                const dim3 grid((nrows + LQ_RW - 1) / LQ_RW);
                const int jt = (len + 255) / 256;
#define OG_APPLY(KERNEL) \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, qp->Tc, qp->Jw, ldw, msweep, nq, k, qp->Vp, qp->panel)
                if (jt <= 2) OG_APPLY(k_lq_apply_reg<2>);
                else if (jt <= 4) OG_APPLY(k_lq_apply_reg<4>);
                else if (jt <= 6) OG_APPLY(k_lq_apply_reg<6>);
                else OG_APPLY(k_lq_apply);
#undef OG_APPLY
            }
            k += LQ_NB;
        }
#ifdef OGSQP_TRACE
        {
            LqPanel hp;
            OG_HIP(hipMemcpyAsync(&hp, qp->panel, sizeof(LqPanel), hipMemcpyDeviceToHost, s));
            OG_HIP(hipStreamSynchronize(s));
            fprintf(stderr, "[ogsqp trace] last panel kernel ticks: load %lld products %lld reduction %lld update %lld gram %lld gram-red %lld store %lld\n",
                    hp.tr[0], hp.tr[1], hp.tr[2], hp.tr[3], hp.tr[4], hp.tr[5], hp.tr[6]);
        }
#endif
        OG_STAGE("check diag");
        if (meq) hipLaunchKernelGGL(k_check_diag, dim3(1), dim3(1024), 0, s, qp->diagL, meq, qp->flag, qp->dthresh);
    }
    OG_HIP(hipGetLastError());
    // ---- equality-constrained minimiser: L w1 = -c,  deq = J1 w1 - Y (Y'g)
    OG_STAGE("trsv w1");
    if (meq) launch_trsv(qp, ldw, meq, 0, -1.0, qp->c, qp->w1, s);
    int hflag[4] = {0, 0, 0, 0};
    OG_HIP(hipMemcpyAsync(hflag, qp->flag, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    OG_HIP(hipStreamSynchronize(s));
    if (hflag[2] || hflag[3]) {          // a look-ahead workgroup of the sweep / a block of the chained solve gave up waiting
        *lost = 1;
        return 0;
    }
    if (hflag[0]) {                       // a dependent equality row that contradicts the others
        *status = OG_QP_SINGULAR_C;
        return 0;
    }
    OG_STAGE("deq");
    if (nr > 0)
        hipLaunchKernelGGL(k_gemv_cols, dim3((nr + 63) / 64), dim3(1024), 0, s, qp->Jw + meq, (long)ldw, nq, nr, qp->g,
                           (const double*)nullptr, qp->t1);
    hipLaunchKernelGGL(k_concat_neg, dim3((nq + 255) / 256), dim3(256), 0, s, qp->w1, meq, qp->t1, nr, qp->xcat);
    hipLaunchKernelGGL(k_gemv_rows, dim3((nq + 3) / 4), dim3(256), 0, s, qp->Jw, (long)ldw, nq, nq, qp->xcat, 1.0,
                       (const double*)nullptr, qp->deq);
    // ---- least-distance problem in the null space
    OG_STAGE("gemm G J");
    if (mg) {
        OG_TRY(launch_gemm(qp, A, meq, mg, nq, ldw, qp->GJ, (const int*)nullptr, s));
        hipLaunchKernelGGL(k_gemv_cols_A, dim3((mg + 63) / 64), dim3(1024), 0, s, A, meq, nq, mg, qp->deq,
                           qp->c + meq, qp->bG);
    }
    OG_STAGE("ldp setup");
    hipLaunchKernelGGL(k_ldp_setup, dim3((mg + nq + 3) / 4), dim3(256), 0, s, qp->GJ, qp->Jw, ldw, meq, nq, mg, qp->bG,
                       qp->c + meq, qp->deq, qp->dl, qp->du, qp->bval, qp->scale, qp->own, qp->flag);
    GiArgs ga;
    ga.GJ = qp->GJ; ga.Jw = qp->Jw; ga.ld = ldw; ga.meq = meq; ga.nq = nq; ga.mg = mg; ga.nr = nr;
    ga.qcap = qp->qcap; ga.bval = qp->bval; ga.scale = qp->scale; ga.own = qp->own; ga.u = qp->u;
    ga.isact = qp->isact; ga.y = qp->y; ga.act = qp->act; ga.R[0] = qp->R[0]; ga.R[1] = qp->R[1];
    ga.RI[0] = qp->RI[0]; ga.RI[1] = qp->RI[1]; ga.Q1t = qp->Q1t; ga.partials = qp->partials; ga.st = qp->st;
    const int mt = mg + 2 * nq;
    ga.limit = 10 * (mt + nr) + 100;
    GiState hst;
    memset(&hst, 0, sizeof(hst));
    const bool rows_mode = qp->gi_mode == 0;
    if (rows_mode && nr > 0) {
        // ---- rotated coordinates, one pass over the rows per change (ogsqp_rows.h)
        RowsArgs ra;
        ra.g = ga;
        ra.g.RI[0] = qp->RI[0];
        ra.g.RI[1] = qp->RI[1];
        ra.dots = qp->dots;
        ra.dvec = qp->dvec;
        ra.rvec = qp->rvec;
        ra.vvec = qp->csbuf;
        ra.slot = qp->d_slot;
        ra.price = qp->price;
        ra.ratio = qp->ratio;
        ra.rec = qp->rec;
        const int nrows = mg + nq;
        ra.G1 = std::max(8, std::min(128, (qp->qcap + 15) / 16));
        // a wavefront per row (2048 workgroups at most: k_rows_decide reads that many partial prices in one trip)
        ra.G2 = std::max(1, std::min(2048, (nrows + 1 + ROWS_WAVES - 1) / ROWS_WAVES));
        // the warm start's removals spread over the grid where the tile fits next to the three vectors (ogsqp_rows.h)
        ra.uval = qp->uval;
        ra.lost = qp->flag + 2;
        ra.spin_limit = qp->spin_limit;
        ra.warm_spread = (qp->warm_spread && rows_lds_bytes(nr, qp->qcap) + rows_spread_lds_bytes(qp->qcap) <= LDS_LIMIT) ? 1 : 0;
        ra.only_warm = 0;
        const size_t lds1 = rows_lds_bytes(nr, qp->qcap) + (ra.warm_spread ? rows_spread_lds_bytes(qp->qcap) : 0);
        OG_STAGE("rows init");
        hipLaunchKernelGGL(k_rows_init, dim3((mt + n1 + 255) / 256 + 1), dim3(ROWS_THREADS), 0, s, ra, qp->diagL,
                           (const int*)qp->d_warm, nwarm, qp->dthresh, qp->flag);
        if (nwarm) {
            hipLaunchKernelGGL(k_rows_mark, dim3((nwarm + 255) / 256), dim3(256), 0, s, ra, (const int*)qp->d_warm, nwarm);
            hipLaunchKernelGGL(k_rows_invert, dim3(nwarm), dim3(ROWS_THREADS), (size_t)(nwarm + 1) * sizeof(double), s,
                               ra, (const double*)qp->Tc, (const double*)qp->diagL);
        }
        const int tail_lanes = (nr + 63) / 64;
        // long rows are streamed, the row staged in LDS when five vectors of the null space fit a workgroup's share
        const size_t nrp = (size_t)tail_lanes * 64;
        const bool stream = qp->rows_stream && tail_lanes > 16;
        // (the second pass re-reads the row from the caches by default: staging it in LDS - OGSQP_ROWS=stage, where five
        // vectors fit - costs the occupancy the streamed form lives on: 111 instead of 47 us per change at C5)
        const bool stage = stream && qp->rows_stage > 0 && 5 * nrp * sizeof(double) <= LDS_LIMIT;
        const size_t lds2 = (stage ? 5 : 1) * nrp * sizeof(double);
#define OG_ROWS_APPLY(ra)                                                                                      \
    do {                                                                                                     \
        if (stream && stage) hipLaunchKernelGGL(k_rows_apply_stream<true>, dim3(ra.G2), dim3(ROWS_THREADS), lds2, s, ra); \
        else if (stream) hipLaunchKernelGGL(k_rows_apply_stream<false>, dim3(ra.G2), dim3(ROWS_THREADS), lds2, s, ra); \
        else if (qp->rows_r4 >= 0 && tail_lanes <= 8) hipLaunchKernelGGL(k_rows_apply_r4<8>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra); \
        else if (qp->rows_r4 > 0 && tail_lanes <= 16) hipLaunchKernelGGL(k_rows_apply_r4<16>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra); \
        else if (tail_lanes <= 8) hipLaunchKernelGGL(k_rows_apply<8>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra);   \
        else if (tail_lanes <= 16) hipLaunchKernelGGL(k_rows_apply<16>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra); \
        else if (tail_lanes <= 32) hipLaunchKernelGGL(k_rows_apply<32>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra); \
        else hipLaunchKernelGGL(k_rows_apply<80>, dim3(ra.G2), dim3(ROWS_THREADS), 0, s, ra);                  \
    } while (0)
        OG_ROWS_APPLY(ra);                                     // values and pricing at y = 0
        OG_HIP(hipGetLastError());
        int batch = debug_stages() ? 1 : 8;
        long launched = 0;
        OG_STAGE("rows changes");
        static const bool timing = getenv("OGSQP_TIMING") != nullptr;
        double t_front = 0.0, t_enqueue = 0.0, t_wait = 0.0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        if (timing) {
            const double t0 = now();
            OG_HIP(hipStreamSynchronize(s));
            t_front = now() - t0;
        }
        // ---- round 6: the whole loop as ONE launch with the rows in registers, where they fit (ogsqp_resident.h).  In
        // front of it as many two-launch pairs as the warm start's removals are expected to take (in `only_warm` form:
        // with the warm start over they do nothing); the launch returns at once while the warm start is not over.
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, qp->device);
        const int res_len = std::max(nr, qp->qcap);
        const int res_wg = (nrows + qp->qcap + RES_ROWS - 1) / RES_ROWS;
        bool resident = qp->resident && !debug_stages() && res_len <= RES_MAX_LEN && res_wg <= std::min(RES_MAX_WG, cus) &&
                        res_lds_bytes(nr, qp->qcap) <= LDS_LIMIT;
        if (resident) {
            ResArgs rs;
            rs.r = ra;
            rs.mail = qp->res_mail;
            rs.seq = qp->res_seq;
            rs.NW = res_wg;
            RowsArgs rw = ra;
            rw.only_warm = 1;
            const size_t ldsr = res_lds_bytes(nr, qp->qcap);
            int pairs = nwarm ? std::max(1, qp->warm_pairs_hint) : 0;
            long warm_launched = 0;
            while (true) {
                for (int it = 0; it < pairs; ++it) {
                    hipLaunchKernelGGL(k_rows_decide, dim3(rw.G1), dim3(ROWS_THREADS), lds1, s, rw);
                    OG_ROWS_APPLY(rw);
                }
                warm_launched += pairs;
                hipLaunchKernelGGL(k_rows_resident, dim3(res_wg), dim3(RES_THREADS), ldsr, s, rs);
                ++qp->resident_launches;
                OG_HIP(hipGetLastError());
                OG_HIP(hipMemcpyAsync(&hst, qp->st, sizeof(GiState), hipMemcpyDeviceToHost, s));
                OG_HIP(hipMemcpyAsync(hflag, qp->flag, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
                OG_HIP(hipStreamSynchronize(s));
                if (hst.phase >= 2) break;
                if (hflag[2] || hflag[3] || hst.phase >= 0) {
                    // a workgroup of the resident launch was not there to answer (or a wait gave up earlier in this
                    // attempt): nothing was written back; the attempt is run again with the forms that wait for nothing
                    *lost = 1;
                    return 0;
                }
                if (warm_launched > (long)nwarm + 64)
                    return fail(8, "og_qp_solve_dev: the warm start's removals made no progress (internal error)");
                pairs = std::min(128, std::max(4, 4 * pairs));
            }
            // next time: as many pairs as this warm start's removals took, and a few (a pair that has nothing to do costs
            // ~5 us, a second round trip to the host 50)
            if (nwarm) qp->warm_pairs_hint = std::min(96, hst.warm_removals + 4);
            qp->resident_changes += hst.iters;
        }
        while (!resident) {
            const double te = timing ? now() : 0.0;
            for (int it = 0; it < batch; ++it) {
                hipLaunchKernelGGL(k_rows_decide, dim3(ra.G1), dim3(ROWS_THREADS), lds1, s, ra);
                OG_ROWS_APPLY(ra);
            }
            launched += batch;
            OG_HIP(hipGetLastError());
            const double tw = timing ? now() : 0.0;
            OG_HIP(hipMemcpyAsync(&hst, qp->st, sizeof(GiState), hipMemcpyDeviceToHost, s));
            OG_HIP(hipStreamSynchronize(s));
            if (timing) {
                t_enqueue += tw - te;
                t_wait += now() - tw;
            }
            if (hst.phase >= 2) {
                if (timing)
                    fprintf(stderr, "[ogsqp timing] front end drained in %.3f ms; active set: %d changes, %ld pairs launched, "
                                    "enqueue %.3f ms, wait %.3f ms\n", 1e3 * t_front, hst.iters, launched, 1e3 * t_enqueue,
                            1e3 * t_wait);
                break;
            }
            // every pair of launches is one change (or the end of the warm start): the device's own limit ends the loop
            if (launched > (long)ga.limit + nwarm + 64) {
                // (a wait that gave up earlier in this attempt - sweep, chained solve - leaves garbage the active-set
                // kernels cannot make progress on: that is a lost attempt, to be re-run with the forms that wait for
                // nothing, not an internal error)
                OG_HIP(hipMemcpyAsync(hflag, qp->flag, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
                OG_HIP(hipStreamSynchronize(s));
                if (hflag[2] || hflag[3]) {
                    *lost = 1;
                    return 0;
                }
                return fail(8, "og_qp_solve_dev: the active-set kernels made no progress (internal error)");
            }
            if (batch < 128) batch *= 2;
        }
#undef OG_ROWS_APPLY
    } else if (!rows_mode) {
    OG_STAGE("gi init");
    hipLaunchKernelGGL(k_gi_init, dim3((mt + n1 + 255) / 256), dim3(256), 0, s, ga, qp->flag);
    OG_HIP(hipGetLastError());
    }
    if (debug_stages()) {
        OG_HIP(hipMemcpyAsync(hflag, qp->flag, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        OG_HIP(hipStreamSynchronize(s));
    }
    const int coop_width = nr <= 1024 ? 16 : 64;
    // workgroups beyond the ceil(nr / width) that own a slice only take part in the pricing
    const int coop_slices = nr > 0 ? (nr + coop_width - 1) / coop_width : 0;
    // as many as it takes to bring the pricing (all of W, every change) down to ~256 KB per workgroup,
    // between 64 and one per CU
    const size_t pricing_bytes = (size_t)(mg + nq) * (size_t)nr * sizeof(double);
    const int coop_G = std::max(coop_slices, (int)std::min<size_t>(256, std::max<size_t>(64, pricing_bytes / (256 * 1024))));
    const size_t coop_lds = (size_t)(4 * qp->qcap + 2 * coop_width + 2 * nr + COOP_THREADS) * sizeof(double) +
                            (size_t)3 * qp->qcap * sizeof(int) + (size_t)((mg + 2 * nq + 31) / 32 + 1) * sizeof(unsigned) + 64;
    // measured (tests/perf/solve_timing.py, OGSQP_GI=single|coop): per active-set change the cooperative kernel
    // wins from C2's size on (C3: 40 vs 72 us; the first 25 major iterations of C3, 300-400 changes per
    // subproblem, take 0.49 instead of 0.67 s), but its launch costs 0.15-0.3 ms more than the first batch of
    // single-workgroup launches and late in a solve a subproblem moves a handful of rows: over whole solves the
    // time per subproblem is the same within 4 % below 512 free directions (C2 1.60 / 1.77, C3' 1.50 / 1.65, C3
    // 7.96 / 8.28 ms, single / cooperative).  The single-workgroup kernel is also the one whose sums run in the
    // restatement's order (with exact Jacobians it walks SciPy's path iteration for iteration), so it stays the
    // default there; from 512 on (C4, C5) the cooperative kernel is 1.5-3x faster per subproblem.
    // Between 256 and 512 free directions the choice follows the previous subproblem: early in a solve (or after a
    // restart) hundreds of rows move per subproblem and the cooperative kernel's 40 us per change beat the 72-83 us
    // of single launches; later a handful move and the single-workgroup kernel's cheaper start wins.
    const bool use_coop = qp->coop_mode == 2 ||
                          (qp->coop_mode == 1 && (nr >= 512 || (nr >= 256 && qp->last_iters > 96)));
    bool coop_done = false;
    if (!rows_mode && nr > 0 && use_coop && coop_slices <= 64 && coop_lds <= LDS_LIMIT) {
        // the whole active-set loop in one cooperative launch
        CoopArgs ca;
        ca.g = ga;
        ca.g.Q1t = nullptr;
        ca.G = coop_G;
        ca.nslices = coop_slices;
        ca.width = coop_width;
        ca.astride = qp->qcap + 8;
        ca.Q1s = qp->Q1t;
        ca.RIr = qp->RI[0];
        ca.apart = qp->apart;
        ca.zg = qp->zg;
        ca.rg = qp->uact;
        ca.cs = qp->csbuf;
        ca.cpart = qp->cpart;
        ca.bar = qp->bar;
        ca.abort_flag = qp->abort_flag;
        OG_HIP(hipMemsetAsync(qp->bar, 0, sizeof(unsigned), s));
        OG_HIP(hipMemsetAsync(qp->abort_flag, 0, sizeof(int), s));
        void* kargs[] = {(void*)&ca};
        OG_STAGE("gi cooperative");
        const hipError_t launched = hipLaunchCooperativeKernel((const void*)k_gi_coop, dim3(coop_G),
                                                               dim3(COOP_THREADS), kargs, (unsigned)coop_lds, s);
        if (launched != hipSuccess) {
            // the runtime cannot keep 64 workgroups of this size resident: the one-workgroup kernel
            // below does the same job (both run on the GPU; nothing leaves it)
            (void)hipGetLastError();
            qp->coop_mode = 0;
        } else {
            coop_done = true;
        }
    }
    if (coop_done) {
        int habort = 0;
        OG_HIP(hipMemcpyAsync(&hst, qp->st, sizeof(GiState), hipMemcpyDeviceToHost, s));
        OG_HIP(hipMemcpyAsync(&habort, qp->abort_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        OG_HIP(hipStreamSynchronize(s));
#ifdef OGSQP_TRACE
        {
            static const char* nm[12] = {"pricing", "barrier 1", "projections 1", "barrier 2", "gather a, z1", "barrier 2b",
                                         "projections 2", "barrier 3", "gather a2, z2, r", "barrier 4", "updates",
                                         "barrier 5"};
            fprintf(stderr, "[ogsqp trace] cooperative: %d iterations, G = %d\n", hst.iters, coop_G);
            for (int e = 0; e < 12; ++e)
                fprintf(stderr, "[ogsqp trace]   %-18s %8.0f ticks per iteration\n", nm[e],
                        hst.iters ? (double)hst.tr[e] / hst.iters : 0.0);
        }
#endif
        if (habort) {
            OG_HIP(hipMemcpy(hflag, qp->flag, 4 * sizeof(int), hipMemcpyDeviceToHost));
            if (hflag[2] || hflag[3]) {              // (as above: the input of the loop came from a lost wait)
                *lost = 1;
                return 0;
            }
            return fail(7, "og_qp_solve_dev: the cooperative active-set kernel lost a workgroup at a barrier");
        }
    } else if (rows_mode && nr > 0) {
        // done above
    } else if (nr > 0) {
        const int blocks = (mg + nq + GI_WAVES - 1) / GI_WAVES;
        const size_t lds = gi_lds_bytes(nr, qp->qcap);
        int batch = debug_stages() ? 1 : 8;
        OG_STAGE("gi iterations");
        while (true) {
            for (int it = 0; it < batch; ++it)
                hipLaunchKernelGGL(k_gi_iter, dim3(blocks), dim3(GI_THREADS), lds, s, ga);
            OG_HIP(hipGetLastError());
            OG_HIP(hipMemcpyAsync(&hst, qp->st, sizeof(GiState), hipMemcpyDeviceToHost, s));
            OG_HIP(hipStreamSynchronize(s));
            if (hst.phase >= 2) break;
            if (batch < 64) batch *= 2;
        }
    } else {
        hipLaunchKernelGGL(k_gi_init, dim3((mt + n1 + 255) / 256), dim3(256), 0, s, ga, qp->flag);
        OG_HIP(hipMemcpyAsync(&hst, qp->st, sizeof(GiState), hipMemcpyDeviceToHost, s));
        OG_HIP(hipStreamSynchronize(s));
        if (hst.phase < 2) hst.phase = 2;   // nothing to move: feasibility was settled by k_ldp_setup
    }
    if (iterations) *iterations = hst.iters;
    qp->last_iters = hst.iters;
    if (debug_stages())
        fprintf(stderr, "[ogsqp] LDP finished: phase %d after %d iterations, %d active, unfixable-row flag %d\n",
                hst.phase, hst.iters, hst.q, hflag[1]);
#ifdef OGSQP_TRACE
    {
        static const char* names_iter[9] = {"phase A + wait", "election", "load normal", "projections", "z update",
                                            "r + ratio test", "u,y update", "append", "removal"};
        static const char* names_rows[9] = {"state word", "who comes in", "normal in LDS", "norms", "inverse rows x d1",
                                            "stores + ticket", "others' r, ratio", "u, y", "reflector, lists"};
        static const char* names_res[16] = {"price, workgroup's best", "publish + poll + election", "winner's row", "(2) row from its owner", "norms",
                                            "r: product, publish", "r: poll", "ratio test", "step, u, y, |y|", "row joins", "row leaves",
                                            "pass over rows", "closing barrier", "(wave 15) pass over y", "-", "-"};
        const bool res_trace = rows_mode && qp->resident_launches > 0;
        const char* const* names = rows_mode ? names_rows : names_iter;
        fprintf(stderr, "[ogsqp trace] %d iterations, %lld passes, %lld removals, %d active at the end\n", hst.iters,
                hst.tr[10], hst.tr[11], hst.q);
        for (int e = 0; e < 9; ++e)
            fprintf(stderr, "[ogsqp trace]   %-16s %8.2f us per iteration\n", names[e],
                    hst.iters ? 0.01 * (double)hst.tr[e] / hst.iters : 0.0);
        if (res_trace)
            for (int e = 0; e < 14; ++e)
                fprintf(stderr, "[ogsqp trace]   resident: %-22s %8.2f us per change (%lld changes, %lld partial)\n", names_res[e],
                        hst.tr[38] ? 0.01 * (double)hst.tr[16 + e] / hst.tr[38] : 0.0, hst.tr[38], hst.tr[39]);
    }
#endif
    if (hst.dbg != 0) fprintf(stderr, "[ogsqp] internal check failed: code %d aux %d (q %d, p %d)\n", hst.dbg, hst.dbg2, hst.q, hst.p);
    if (hst.phase != 2) {
        *status = hst.phase == 3 ? OG_QP_ITERATION_LIMIT : OG_QP_INCOMPATIBLE;
        return 0;
    }
    // ---- step and multipliers
    OG_STAGE("finish");
    hipLaunchKernelGGL(k_finish_step, dim3((nq + 3) / 4), dim3(256), 0, s, qp->Jw, ldw, meq, nq, nr, mg, qp->y, qp->deq,
                       qp->dl, qp->du, qp->u, qp->d, qp->bm);
    hipLaunchKernelGGL(k_dual_residual, dim3((nq + 3) / 4), dim3(256), 0, s, A, meq, mg, nq, qp->g, qp->u, qp->bm,
                       qp->tvec);
    if (meq) {
        hipLaunchKernelGGL(k_gemv_cols, dim3((meq + 63) / 64), dim3(1024), 0, s, qp->Jw, (long)ldw, nq, meq, qp->tvec,
                           qp->w1, qp->rhs);
        launch_trsv(qp, ldw, meq, 1, 1.0, qp->rhs, qp->lam, s);
    }
    OG_HIP(hipGetLastError());
    OG_HIP(hipMemcpyAsync(d, qp->d, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
    OG_HIP(hipMemcpyAsync(bound_mult, qp->bm, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
    if (meq) OG_HIP(hipMemcpyAsync(mult, qp->lam, sizeof(double) * meq, hipMemcpyDeviceToHost, s));
    if (mg) OG_HIP(hipMemcpyAsync(mult + meq, qp->u, sizeof(double) * mg, hipMemcpyDeviceToHost, s));
    // (the transposed chained solve for the multipliers ran after the first look at the flags)
    OG_HIP(hipMemcpyAsync(hflag, qp->flag, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    OG_HIP(hipStreamSynchronize(s));
    if (hflag[2] || hflag[3]) {
        *lost = 1;
        return 0;
    }
    if (rows_mode && nr > 0) {
        // the rows active at this solution, in the numbering of og_qp_get_active: where the next subproblem starts
        std::vector<int> act((size_t)std::max(hst.q, 1));
        if (hst.q > 0) OG_HIP(hipMemcpy(act.data(), qp->act, sizeof(int) * hst.q, hipMemcpyDeviceToHost));
        std::vector<int> before(qp->warm);
        qp->warm.clear();
        for (int j = 0; j < hst.q; ++j) {
            const int c = act[j];
            if (c < mg)
                qp->warm.push_back(c);
            else if (c < mg + nq)
                qp->warm.push_back(mg + 2 * (c - mg));
            else
                qp->warm.push_back(mg + 2 * (c - mg - nq) + 1);
        }
        // warm-start the next solve only if this solution kept most of the previous one's rows
        std::vector<int> now(qp->warm);
        std::sort(now.begin(), now.end());
        std::sort(before.begin(), before.end());
        std::vector<int> common;
        std::set_intersection(now.begin(), now.end(), before.begin(), before.end(), std::back_inserter(common));
        const size_t larger = std::max(now.size(), before.size());
        qp->warm_use = before.empty() || 10 * common.size() >= 7 * larger;
    }
    if (!augmented) std::swap(qp->Z, qp->Jw);   // Z Q: same B, what og_qp_bfgs updates next
    *status = OG_QP_SOLVED;
    return 0;
}

int og_qp_solve_dev(og_qp_handle qp, const double* d_jt, int64_t ld, const double* g, const double* c,
                    const double* dl, const double* du, int32_t augmented, double rho, double* d, double* mult,
                    double* bound_mult, int32_t* status, int32_t* iterations, void* hip_stream) {
    int lost = 0;
    int rc = qp_solve_attempt(qp, d_jt, ld, g, c, dl, du, augmented, rho, d, mult, bound_mult, status, iterations,
                              hip_stream, &lost);
    if (rc || !lost) return rc;
    if (qp->n1 > LQ_PT_MAX * LQ_CPT_MAX) {
        // rows of this length have no form that waits for nothing: the same attempt is run again, with the lanes drained -
        // a wait gives up because a workgroup was not resident at that moment (another stream, rank or tenant), which the
        // next attempt need not meet (nothing was committed by the lost one)
        for (int again = 0; again < 2 && lost; ++again) {
            (void)hipDeviceSynchronize();
            ++qp->recoveries;
            lost = 0;
            rc = qp_solve_attempt(qp, d_jt, ld, g, c, dl, du, augmented, rho, d, mult, bound_mult, status, iterations, hip_stream,
                                  &lost);
            if (rc) return rc;
        }
        if (lost)
            return fail(7, "og_qp_solve_dev: an inter-workgroup wait gave up three times and rows of this length have no form without one");
        return 0;
    }
    // The look-ahead sweep and the chained triangular solves hand data between workgroups of ONE launch and assume the
    // workgroups they wait for are resident; on a device shared with other streams, ranks or tenants that may not hold,
    // and a bounded wait gives up.  Nothing was committed: the subproblem is solved again with the forms that wait for
    // nothing (panel and update as separate launches, a launch per block of the triangular solves) - SciPy's core has
    // no such failure mode, so neither does this one.
    const bool ahead = qp->lq_ahead, wide_on = qp->lq_wide, spread_on = qp->warm_spread;
    const int trsv = qp->trsv_mode;
    const bool resident_on = qp->resident;
    qp->resident = false;              // (the resident active-set launch waits for every one of its workgroups)
    qp->warm_spread = false;           // (its wait between the two products is one of the waits that can give up)
    qp->lq_ahead = false;
    qp->lq_wide = false;               // (its column-split panel waits too; round 2's kernels serve the long rows)
    if (qp->trsv_mode == 0) qp->trsv_mode = 1;
    ++qp->recoveries;
    lost = 0;
    rc = qp_solve_attempt(qp, d_jt, ld, g, c, dl, du, augmented, rho, d, mult, bound_mult, status, iterations, hip_stream,
                          &lost);
    qp->lq_ahead = ahead;
    qp->resident = resident_on;
    qp->warm_spread = spread_on;
    qp->lq_wide = wide_on;
    qp->trsv_mode = trsv;
    if (!rc && lost) return fail(7, "og_qp_solve_dev: a wait gave up in the forms that have none (internal error)");
    return rc;
}

int og_qp_recoveries(og_qp_handle qp, int32_t* count) {
    if (!qp || !count) return fail(2, "og_qp_recoveries: null argument");
    *count = qp->recoveries;
    return 0;
}

int og_qp_resident_stats(og_qp_handle qp, int64_t* launches, int64_t* changes) {
    if (!qp || !launches || !changes) return fail(2, "og_qp_resident_stats: null argument");
    *launches = qp->resident_launches;
    *changes = qp->resident_changes;
    return 0;
}

int og_qp_get_active(og_qp_handle qp, int32_t* ids, int32_t capacity, int32_t* count) {
    if (!qp || !count) return fail(2, "og_qp_get_active: null argument");
    *count = (int32_t)qp->warm.size();
    if (ids)
        for (int j = 0; j < (int)qp->warm.size() && j < capacity; ++j) ids[j] = qp->warm[j];
    return 0;
}

int og_qp_set_active(og_qp_handle qp, const int32_t* ids, int32_t count) {
    if (!qp || count < 0 || (count > 0 && !ids)) return fail(2, "og_qp_set_active: bad arguments");
    std::vector<int> next;
    std::vector<char> seen((size_t)qp->mg + 2 * (size_t)qp->n1, 0);
    for (int j = 0; j < count; ++j) {
        if (ids[j] < 0 || ids[j] >= qp->mg + 2 * qp->n1) return fail(2, "og_qp_set_active: no such constraint");
        if (seen[ids[j]]) return fail(2, "og_qp_set_active: a constraint is listed twice");
        seen[ids[j]] = 1;
        next.push_back(ids[j]);
    }
    qp->warm.swap(next);
    qp->warm_use = true;
    return 0;
}

int og_qp_solve(og_qp_handle qp, const double* A, const double* g, const double* c, const double* dl,
                const double* du, int32_t augmented, double rho, double* d, double* mult, double* bound_mult,
                int32_t* status, int32_t* iterations) {
    if (!qp || (qp->m > 0 && !A)) return fail(2, "og_qp_solve: null argument");
    OG_HIP(hipSetDevice(qp->device));
    const size_t ld = (size_t)qp->m + 1, n = qp->n;
    if (!qp->jt_stage) OG_HIP(hipMalloc((void**)&qp->jt_stage, sizeof(double) * ld * n));
    std::vector<double> jt(ld * n, 0.0);
    for (int j = 0; j < qp->m; ++j)
        for (size_t i = 0; i < n; ++i) jt[i * ld + 1 + j] = A[(size_t)j * n + i];
    OG_HIP(hipMemcpy(qp->jt_stage, jt.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice));
    return og_qp_solve_dev(qp, qp->jt_stage, (int64_t)ld, g, c, dl, du, augmented, rho, d, mult, bound_mult, status,
                           iterations, nullptr);
}

int og_qp_bfgs(og_qp_handle qp, const double* s, const double* eta, const double* Bs, int32_t* reset_needed) {
    if (!qp || !s || !eta || !Bs || !reset_needed) return fail(2, "og_qp_bfgs: null argument");
    OG_HIP(hipSetDevice(qp->device));
    const int n = qp->n, n1 = qp->n1, ldw = qp->ldw;
    double h1 = 0.0, h2 = 0.0;
    for (int i = 0; i < n; ++i) {
        h1 += s[i] * eta[i];
        h2 += s[i] * Bs[i];
    }
    const double h3 = 0.2 * h2;
    double theta = 1.0;
    if (h1 < h3) {
        theta = (h2 - h3) / (h2 - h1);
        h1 = h3;
    }
    if (!(h1 > 0.0 && h2 > 0.0) || !std::isfinite(h1) || !std::isfinite(h2)) {
        *reset_needed = 1;
        return 0;
    }
    *reset_needed = 0;
    const double alpha = std::sqrt(h1 / h2);
    std::vector<double>& hs = qp->host_stage;
    hs.assign((size_t)2 * n1, 0.0);
    for (int i = 0; i < n; ++i) {
        const double r = theta * eta[i] + (1.0 - theta) * Bs[i];
        hs[i] = s[i];
        hs[n1 + i] = (r - alpha * Bs[i]) / (alpha * h2);
    }
    hipStream_t st = qp->stream;
    OG_HIP(hipMemcpyAsync(qp->svec, hs.data(), sizeof(double) * n1, hipMemcpyHostToDevice, st));
    OG_HIP(hipMemcpyAsync(qp->vvec, hs.data() + n1, sizeof(double) * n1, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gemv_cols, dim3((n + 63) / 64), dim3(1024), 0, st, qp->Z, (long)ldw, n, n, qp->vvec,
                       (const double*)nullptr, qp->vz);
    hipLaunchKernelGGL(k_rank1, dim3((n + 255) / 256, n), dim3(256), 0, st, qp->Z, ldw, n, qp->svec, qp->vz,
                       1.0 / alpha);
    OG_HIP(hipGetLastError());
    OG_HIP(hipStreamSynchronize(st));
    return 0;
}

int og_jt_times(og_qp_handle qp, const double* d_jt, int64_t ld, const double* coef, double* out, void* hip_stream) {
    if (!qp || !d_jt || !coef || !out) return fail(2, "og_jt_times: null argument");
    OG_HIP(hipSetDevice(qp->device));
    if (hip_stream) OG_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    hipStream_t s = qp->stream;
    const int width = qp->m + 1;
    OG_HIP(hipMemcpyAsync(qp->coef, coef, sizeof(double) * width, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_jt_times, dim3((qp->n + 3) / 4), dim3(256), 0, s, d_jt, (long)ld, qp->n, width, qp->coef,
                       qp->outn);
    OG_HIP(hipGetLastError());
    OG_HIP(hipMemcpyAsync(out, qp->outn, sizeof(double) * qp->n, hipMemcpyDeviceToHost, s));
    OG_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
