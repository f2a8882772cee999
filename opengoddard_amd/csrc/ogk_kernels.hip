// ogk_kernels.hip -- hand-written gfx950 sweep kernels, instantiated for one traced problem.
//
// Compiled once per problem as  hipcc --offload-arch=gfx950 -ffp-contract=off -DOG_GEN_HEADER=...
// The generated header only supplies pointwise device functions (OgGen::group_eval,
// OgGen::defect_tail, OgGen::mv_operand) and small index tables; everything about parallel
// decomposition, LDS, MFMA and memory traffic is in this file.
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
// Launch modes (ogk_launch; ogk.h lists all):
//   0  ogk_eval          F(x0) -> f0, plus scratch the two-launch sweep reuses: the unperturbed collocation
//                        products y0, the dynamics terms t0 = (tf-t0)/2 f, and z = F0 - F0
//                        (0, or NaN where a row is not finite: what dense FD would produce).
//   1  ogk_sweep         structured forward-difference sweep, second of two launches.  Dense FD evaluates
//                        every row for every column although a row changes only when it reads
//                        the perturbed variable; because base and perturbed values come from
//                        the same device functions, every other difference quotient is exactly
//                        (F0-F0)/dx.  This kernel therefore
//                          - lets a workgroup own a few J_T rows (= FD columns): it streams
//                            zeros into them (unless the buffer is a registered persistent-zero one) and
//                            re-evaluates only the (group, output, element)
//                            items whose traced leaves include p[j] (tables OGT_COL/OGT_ELEM);
//                          - runs the collocation product for the N perturbed vectors of each
//                            state slice on v_mfma_f64_16x16x4_f64 (A = 16 perturbed state
//                            vectors, B = the D^T operand image), writing the dense N x N
//                            block d(defect_s)/d(state_s) directly.
//   5  ogk_fused         modes 0 + 1 as ONE launch writing only the non-zeros (the default; see below).
//                        The results of modes 1, 5 and 2 are identical (tests compare them and the CPU twin).
//
// D is kept in HBM/L2 in MFMA B-operand order (ogk.h).  Kernels that reuse a panel across
// wavefronts or states stage it in LDS; the MFMA tiles use each panel once
// per wavefront and read it straight from L2 (measured: no LDS round trip, no barrier, same speed).
//   2  ogk_dense         the literal dense sweep: all rows for all columns (validation, and the
//                        shape SURVEY.md section 7.2 describes).
//
// The MFMA accumulation is a k-ordered fma chain per output (verified on hardware by
// tools/gpu_probe.hip), identical to oracle/twin.cpp's loop.
#include <hip/hip_runtime.h>
#include "ogk.h"
#include "og_dual.h"
#include OG_GEN_HEADER

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Timing experiments only (tools/kstats.sh): -DOGK_EXP=<mask> removes pieces of ogk_sweep so that
// rocprof durations attribute time to phases: 8 = no item evaluation, 16 = no fill, 32 = no MFMA
// tiles.  Results are wrong with any bit set.
#ifndef OGK_EXP
#define OGK_EXP 0
#endif

// -DOGK_TRACE=1 (tools/trace_sweep.py): workgroups overwrite the start of one of their J_T rows
// with shader-clock stamps of their phases; results are garbage, timing is the point.
#ifndef OGK_TRACE
#define OGK_TRACE 0
#endif

namespace {

struct XCol {
    typedef double scalar;
    const double* x0;
    int j;          // perturbed index, -1 for none
    double xj;      // x0[j] + h[j]
    __device__ __forceinline__ double operator()(const int i) const {
        const double v = x0[i];
        return i == j ? xj : v;
    }
    // base collocation product y0[slot offset + node]: from the scratch an evaluation left in memory
    __device__ __forceinline__ double ldy(const double* p) const { return *p; }
};

// the same, for a workgroup of the fused launch that keeps the base products of its node tile in LDS
// (an LDS read must not be a flat load through the y0 pointer: a flat load waits for every outstanding
// store of the wavefront, i.e. for the fill that is supposed to drain underneath the item chains)
struct XColL {
    typedef double scalar;
    const double* x0;
    int j;
    double xj;
    const double* y0_first;     // &y0[offset of the group's first slot]
    const double* yl;           // LDS: [state][N]
    __device__ __forceinline__ double operator()(const int i) const {
        const double v = x0[i];
        return i == j ? xj : v;
    }
    __device__ __forceinline__ double ldy(const double* p) const { return yl[(int)(p - y0_first)]; }
};

// XCol with the base terms of the program's sequential sums (a running cost summed node by node like Python's
// sum()) at hand: the workgroup computed them once, cooperatively, into LDS; a lane re-evaluates only the term
// that reads its own perturbed variable.  The additions stay with the caller, in order - same bits, but a
// chain of LDS reads and adds instead of N evaluations with their global loads per lane.
constexpr int TB_SLOTS = OgGen::N_TBLK > 0 ? OgGen::N_TBLK : 1;
struct XColT {
    typedef double scalar;
    const double* x0;
    int j;
    double xj;
    const double* tc;           // [N_TERMS] base terms in LDS, or NULL: evaluate in place
    int qd[TB_SLOTS];           // per term block: the term that reads p[j] (-1 none, -2 several: in place)
    double td[TB_SLOTS];        // ... and its value at x0 + h e_j
    __device__ __forceinline__ double operator()(const int i) const {
        const double v = x0[i];
        return i == j ? xj : v;
    }
    __device__ __forceinline__ double ldy(const double* p) const { return *p; }
    __device__ __forceinline__ const double* term_cache(const int tb) const {
        return (tc && qd[tb] != -2) ? tc + OgGen::TERM_OFF(tb) : nullptr;
    }
    __device__ __forceinline__ int term_q(const int tb) const { return qd[tb]; }
    __device__ __forceinline__ double term_v(const int tb) const { return td[tb]; }
};
__device__ __forceinline__ XColT make_xcolt(const ogk_args& a, const int j, const double xj, const double* tc) {
    XColT x;
    x.x0 = a.x0, x.j = j, x.xj = xj, x.tc = tc;
    const XCol plain{a.x0, j, xj};
#pragma unroll
    for (int tb = 0; tb < TB_SLOTS; ++tb) {
        x.qd[tb] = -1, x.td[tb] = 0.0;
        if (tc && tb < OgGen::N_TBLK && j >= 0) {
            x.qd[tb] = OgGen::sum_term_q(tb, j);
            if (x.qd[tb] >= 0) x.td[tb] = OgGen::sum_term(tb, x.qd[tb], plain, a.cvec);
        }
    }
    return x;
}
constexpr bool TERM_CACHE = OgGen::N_TERMS > 0 && OgGen::N_TERMS <= 2048;      // LDS doubles a workgroup spends on it
constexpr int TERM_DOUBLES = TERM_CACHE ? OgGen::N_TERMS + 16 : 0;     // (+16: the generated sum loops read whole groups of eight, one group ahead)

// all threads of a workgroup: the base terms into LDS (the caller's barrier publishes them)
template <int THREADS>
__device__ __forceinline__ void fill_terms(const ogk_args& a, double* tc) {
    struct XBase {
        typedef double scalar;
        const double* x0;
        __device__ __forceinline__ double operator()(const int i) const { return x0[i]; }
    };
    const XBase xb{a.x0};
#pragma unroll
    for (int tb = 0; tb < OgGen::N_TBLK; ++tb)
        for (int q = (int)threadIdx.x; q < OgGen::TERM_LEN(tb); q += THREADS)
            tc[OgGen::TERM_OFF(tb) + q] = OgGen::sum_term(tb, q, xb, a.cvec);
}

// Persistent-zero output (ogk.h: jt_sparse / jt_launches / jt_state).  A buffer that is known to hold zeros at
// its structural zeros is only written where something can be non-zero; the fill is needed when the
// buffer is not registered, or when the previous launch into it left a NaN fill behind.
__device__ __forceinline__ bool jt_needs_fill(const ogk_args& a) {
    // both written by earlier kernels (agent scope): plain loads.  *jt_launches already counts this launch
    const unsigned st = *a.jt_state, gen = *a.jt_launches;
    return !a.jt_sparse | (st == gen - 1u);
}
// this launch fills with NaN: the next launch into the buffer must clean up
__device__ __forceinline__ void jt_mark_nan_fill(const ogk_args& a) {
    if (a.jt_sparse) __hip_atomic_store(a.jt_state, *a.jt_launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// The module may be built in two parts (OGK_PART; build.py): part 0 holds what a solve needs from its first sweep
// on (evaluation, the structured and the one-launch sweep, pattern / pack / unpack), part 1 the validation sweep
// and the exact-Jacobian kernels, which the runtime loads when they are first asked for.  Each part instantiates
// the generated callbacks only for its own kernels: the first solve of a new problem shape waits for part 0 only.
#if !defined(OGK_PART) || OGK_PART == 1
#define OGK_HAS_AUX 1        // modes 2, 3, 4: validation sweep, exact Jacobian
#endif
#if !defined(OGK_PART) || OGK_PART == 0
#define OGK_HAS_MAIN 1       // modes 0, 6 - 10: evaluation, pattern, pack, unpack
#endif
#if !defined(OGK_PART) || OGK_PART == 2
#define OGK_HAS_FUSED 1      // mode 5: evaluation + structured sweep in one launch
#endif
#if !defined(OGK_PART) || OGK_PART == 3
#define OGK_HAS_SWEEP 1      // mode 1: the structured sweep as a launch of its own
#endif
// (the split is at the kernels: a __global__ function is what costs code generation; the device functions below
// them are templates or forced-inline and cost nothing where no kernel of the part uses them)
// ------------------------------------------------------------------------------------------
// Mode 2, dense sweep (validation): collocation workgroup = (phase, 16-node tile, 64 FD
// columns), all states; row workgroup = one thread per (row item, 8 columns).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_defect_body(const ogk_args& a, const int bx, const int by,
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
    const int tid = (int)threadIdx.x;

    double* dpanel = lds;
    double* xt = lds + KS * 64;
    const double* src = a.dfrag + a.dfrag_off[phase] + (long)nt * KS * 64;
    for (int i = tid; i < KS * 64; i += 256) dpanel[i] = src[i];
    const XCol base{a.x0, -1, 0.0};
    {
        const int wave_ = tid >> 6, lane_ = tid & 63;
        for (int s = wave_; s < OgGen::MAX_NMV; s += 4)
            for (int l = lane_; l < NP; l += 64)
                xt[s * NP + l] = (s < nmv && l < N) ? OgGen::mv_operand(mv0 + s, l, base, a.cvec) : 0.0;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int c0 = a.col_lo + (by * 4 + wave) * 16;
    if (c0 >= a.col_hi) return;
    const int lk = lane >> 4;
    const int k = nt * 16 + (lane & 15);

    // the column this lane feeds into the A operand, and the one operand entry it changes
    int hit_s = -1, hit_l = -1;
    double hit_v = 0.0;
    {
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

    // batched D.X on the matrix cores: acc[s] (16 columns x 16 nodes) += A_s (16x4) * B (4x16)
    // (branch-free over MAX_NMV: unused slots multiply zeros)
    v4f64 acc[OgGen::MAX_NMV];
#pragma unroll
    for (int s = 0; s < OgGen::MAX_NMV; ++s) acc[s] = (v4f64){0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < KS; ++ks) {
        const double b = dpanel[ks * 64 + lane];
        const int l = ks * 4 + lk;
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) {
            double av = xt[s * NP + l];
            av = (s == hit_s && l == hit_l) ? hit_v : av;
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[s], 0, 0, 0);
        }
    }

    // epilogue.  C/D layout of the f64 MFMA: column (node) = lane & 15,
    // row (FD column) = (lane >> 4) + 4 * reg.
    if (k >= N) return;
    double y[OgGen::MAX_NMV];
    double T[OgGen::MAX_NMV];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + lk + 4 * reg;
        if (j >= a.col_hi) continue;
        const double xb = a.x0[j];
        const double xj = xb + a.h[j];
        const double dx = xj - xb;
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) y[s] = acc[s][reg];
        const XCol xa{a.x0, j, xj};
        OgGen::defect_tail(g, k, xa, a.cvec, T);
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) {     // static index: keeps y/T in registers
            if (s >= nmv) break;
            const int row = OgGen::G_ROW(g, s) + k;
            const double val = y[s] - T[s];
            a.jt[(long)(j - a.col_lo) * OgGen::M + row] = (val - a.f0[row]) / dx;
        }
    }
}

__device__ __forceinline__ void dense_rows_body(const ogk_args& a, const int bx, const int by) {
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

#ifdef OGK_HAS_AUX
__global__ __launch_bounds__(256) void ogk_dense(const ogk_args a, const int ndef,
                                                 const int defect_total, const int row_blocks) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int id = (int)blockIdx.x;
    // every entry is written: a registered buffer only has to learn about a NaN fill
    if (id == 0 && threadIdx.x == 0 && *a.nonfinite != 0) jt_mark_nan_fill(a);
    if (id < defect_total) {
        dense_defect_body(a, id % ndef, id / ndef, lds);
    } else {
        const int rid = id - defect_total;
        dense_rows_body(a, rid % row_blocks, rid / row_blocks);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Mode 3, exact Jacobian: the generated callback code instantiated on
// first-order dual numbers (og_dual.h) with a unit seed on decision variable j gives dF/dx_j without
// a step, a subtraction or FD noise.  One thread per (column j, row item): a row-group element, or a
// collocation node of a defect group, where the derivative of the product D.x~ is one column of D
// times the derivative of the operand the seeded variable feeds (if it is a state of that phase).
// F(x0) and the base products come from mode 0 (same stream, before).  Dense and simple: exactness
// changes the numbers SLSQP sees (it removes the 1e-8 noise of the reference's differences), so this
// is an opt-in mode next to the reference-faithful sweep, not a replacement for it.
struct XDual {
    typedef ogdual scalar;
    const double* x0;
    int j;
    __device__ __forceinline__ ogdual operator()(const int i) const { return ogdual(x0[i], i == j ? 1.0 : 0.0); }
    __device__ __forceinline__ double ldy(const double* p) const { return *p; }
};

__device__ __forceinline__ double dfrag_entry(const ogk_args& a, const int phase, const int N, const int k,
                                              const int l) {
    const int KS = (N + 3) >> 2;
    return a.dfrag[a.dfrag_off[phase] + ((long)(k >> 4) * KS + (l >> 2)) * 64 + (((l & 3) << 4) | (k & 15))];
}

#ifdef OGK_HAS_AUX
__global__ __launch_bounds__(256) void ogk_exact(const ogk_args a, const int n_items) {
    const int item = (int)(blockIdx.x * 256 + threadIdx.x);
    const int j = a.col_lo + (int)blockIdx.y;
    if (item >= n_items || j >= a.col_hi) return;
    // item -> (group, element): row-group items first (G_ITEM0 order), then the defect nodes
    int g = 0, k = -1;
    if (item < OgGen::N_ROW_ITEMS) {
        for (; g < OgGen::N_GROUPS; ++g) {
            if (OgGen::G_KIND(g) != 0) continue;
            const int i0 = OgGen::G_ITEM0(g);
            if (item >= i0 && item < i0 + OgGen::G_LEN(g)) {
                k = item - i0;
                break;
            }
        }
    } else {
        int rest = item - OgGen::N_ROW_ITEMS;
        for (; g < OgGen::N_GROUPS; ++g) {
            if (OgGen::G_KIND(g) != 1) continue;
            if (rest < OgGen::G_LEN(g)) {
                k = rest;
                break;
            }
            rest -= OgGen::G_LEN(g);
        }
    }
    if (k < 0) return;
    const XDual xd{a.x0, j};
    double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
    ogdual out[OgGen::MAX_OUT > OgGen::MAX_NMV ? OgGen::MAX_OUT : OgGen::MAX_NMV];
    const int nout = OgGen::G_NOUT(g);
    if (OgGen::G_KIND(g) == 0) {
        OgGen::group_eval(g, k, xd, (const ogdual*)nullptr, a.cvec, out);
    } else {
        const int N = OgGen::G_LEN(g), phase = OgGen::G_PHASE(g), mv0 = OgGen::G_MV0(g), nmv = OgGen::G_NMV(g);
        ogdual y[OgGen::MAX_NMV];
#pragma unroll
        for (int s = 0; s < OgGen::MAX_NMV; ++s) {
            y[s] = ogdual(0.0);
            if (s < nmv) {
                double dy = 0.0;
                const int leaf = OgGen::MV_LEAF(mv0 + s);
                if (j >= leaf && j < leaf + N) {        // x_j is node l of the state behind this product
                    const int l = j - leaf;
                    const ogdual op = OgGen::mv_operand(mv0 + s, l, xd, a.cvec);
                    dy = __builtin_fma(op.d, dfrag_entry(a, phase, N, k, l), 0.0);
                }
                y[s] = ogdual(a.y0[OgGen::MV_Y0(mv0 + s) + k], dy);
            }
        }
        OgGen::group_eval(g, k, xd, (const ogdual*)y, a.cvec, out);
    }
#pragma unroll
    for (int o = 0; o < (OgGen::MAX_OUT > OgGen::MAX_NMV ? OgGen::MAX_OUT : OgGen::MAX_NMV); ++o) {
        if (o >= nout) break;
        jrow[OgGen::G_ROW(g, o) + k] = out[o].d;
    }
}
#endif

// Mode 4, the same derivatives with the work lists of the structured sweep: a workgroup per column
// zero-fills its row of J_T, then evaluates only the row items that read x_j (OGT_COL / OGT_ELEM) and,
// when x_j is a collocated state, that state's defect rows: column l of D times the operand's derivative,
// minus the dynamics term's derivative on the diagonal.  Same numbers as mode 3 (tests compare both with
// the CPU twin), a fraction of the evaluations.
#ifdef OGK_HAS_AUX
__global__ __launch_bounds__(256) void ogk_exact_struct(const ogk_args a) {
    const int j = a.col_lo + (int)blockIdx.x;
    if (j >= a.col_hi) return;
    const int tid = (int)threadIdx.x;
    double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
    if (jt_needs_fill(a)) {                     // (a registered buffer keeps its structural zeros)
        for (int r = tid; r < OgGen::M; r += 256) jrow[r] = 0.0;
        __syncthreads();
    }
    const XDual xd{a.x0, j};
    const int4 rec = OGT_COL[j];
    for (int e = rec.x + tid; e < rec.y; e += 256) {
        const int4 it = OGT_ELEM[e];
        int row;
        const ogdual v = OgGen::item_value(it.x, it.y, it.z, xd, a.y0, a.cvec, &row);
        jrow[row] = v.d;
    }
    const int own_lo = rec.z, own_hi = rec.w & 0x3fffffff;
    if (own_hi > own_lo) {
        // x_j is node l of the state behind one collocation product
        int si = -1, l = 0;
        for (int s = 0; s < OgGen::N_MV; ++s) {
            const int leaf = OGT_SLOT[s].v[2], len = OGT_SLOT[s].v[0];
            if (j >= leaf && j < leaf + len) {
                si = s;
                l = j - leaf;
            }
        }
        if (si >= 0) {
            const int N = OGT_SLOT[si].v[0], phase = OGT_SLOT[si].v[5], row0 = OGT_SLOT[si].v[3];
            // bit 0: the dynamics term at node k reads the state at node k; bit 1: it reads the
            // state's slice in some other way (every node then)
            const int reads = OGT_SLOT[si].v[6];
            const ogdual op = OgGen::mv_operand(si, l, xd, a.cvec);
            for (int k = tid; k < N; k += 256) {
                double v = __builtin_fma(op.d, dfrag_entry(a, phase, N, k, l), 0.0);
                if ((reads & 2) || ((reads & 1) && k == l)) v = v - OgGen::tail_one(si, k, xd, a.cvec).d;
                jrow[row0 + k] = v;
            }
        }
    }
}
#endif

constexpr int SWEEP_THREADS = 512;   // ogk_sweep / ogk_eval workgroup: 8 wavefronts
constexpr int SWEEP_WAVES = SWEEP_THREADS / 64;

// ------------------------------------------------------------------------------------------
// Mode 0: F(x0) and the scratch the structured sweep reuses.  Collocation workgroup = (phase,
// 16-node tile): the states are the 16 rows of the A operand, so ONE MFMA chain yields every
// state's collocation product for the tile; afterwards each wavefront evaluates one state's
// dynamics term (a short dependent chain each, run concurrently).  Row workgroups: one
// wavefront per 64 elements of one traced output.
// ------------------------------------------------------------------------------------------
template <bool FUSED>
__device__ __forceinline__ void publish_row(const ogk_args& a, const int row, const double val) {
    const double zz = val - val;
    a.f0[row] = val;
    // fused launch: z may be read by other workgroups of the same kernel (non-finite rows only)
    if (FUSED) __hip_atomic_store(&a.z[row], zz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.z[row] = zz;
    if (zz != zz) atomicAdd(a.nonfinite, 1);
}

template <bool FUSED>
__device__ __forceinline__ void eval_defect_body(const ogk_args& a, const int bx, double* lds) {
    const int4 blk = OGT_EVALBLK[bx];                   // {group, node tile, first slot, #slots}
    const int nt = blk.y, mv0 = blk.z, nmv = blk.w;
    const ogt_int8 rec0 = OGT_SLOT[mv0];
    const int N = rec0.v[0];
    const int KS = (N + 3) >> 2;
    const int NP = KS * 4;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lk = lane >> 4;

    double* dpanel = lds;
    double* xt = lds + KS * 64;
    double* ybuf = xt + OgGen::MAX_NMV * NP;            // [state][16 nodes]
    const double* src = a.dfrag + a.dfrag_off[rec0.v[5]] + (long)nt * KS * 64;
    constexpr int UNR = 4;
    double pv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
        pv[u] = (tid + SWEEP_THREADS * u < KS * 64) ? src[tid + SWEEP_THREADS * u] : 0.0;
    // operands: one collocation slot per wavefront at a time (no divergence in mv_operand)
    const XCol base{a.x0, -1, 0.0};
    for (int s = wave; s < OgGen::MAX_NMV; s += SWEEP_WAVES)
        for (int l = lane; l < NP; l += 64) {
            double v = 0.0;
            if (s < nmv && l < N) {
                v = OgGen::mv_operand(mv0 + s, l, base, a.cvec);
                if (nt == 0) a.xop[OGT_SLOT[mv0 + s].v[4] + l] = v;
            }
            xt[s * NP + l] = v;
        }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
        if (tid + SWEEP_THREADS * u < KS * 64) dpanel[tid + SWEEP_THREADS * u] = pv[u];
    for (int i = tid + SWEEP_THREADS * UNR; i < KS * 64; i += SWEEP_THREADS) dpanel[i] = src[i];
    __syncthreads();

    // The last wavefront runs the MFMA chain; meanwhile the others evaluate the dynamics terms (one state per
    // wavefront at the tile's 16 nodes): those do not depend on the products, only the final subtraction does.
    constexpr int TR = (OgGen::MAX_NMV + SWEEP_WAVES - 1) / SWEEP_WAVES;
    if (wave == SWEEP_WAVES - 1) {
        const int srow = lane & 15;
        v4f64 acc = {0.0, 0.0, 0.0, 0.0};
        const double* xrow = xt + (srow < OgGen::MAX_NMV ? srow : 0) * NP;
        const bool live = srow < nmv;
        // LDS operands of the next chunk are requested before this chunk's MFMAs issue
        constexpr int CH = 8;
        double bv[CH], av[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            bv[u] = u < KS ? dpanel[u * 64 + lane] : 0.0;
            av[u] = (u < KS && live) ? xrow[u * 4 + lk] : 0.0;
        }
        for (int ks0 = 0; ks0 < KS; ks0 += CH) {
            double bn[CH], an[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int ks = ks0 + CH + u;
                bn[u] = ks < KS ? dpanel[ks * 64 + lane] : 0.0;
                an[u] = (ks < KS && live) ? xrow[ks * 4 + lk] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (ks0 + u < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < CH; ++u) { bv[u] = bn[u]; av[u] = an[u]; }
        }
        // C/D layout: node = lane & 15, state = (lane >> 4) + 4 * reg
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ybuf[(lk + 4 * reg) * 16 + (lane & 15)] = acc[reg];
    }
    const int k = nt * 16 + lane;
    const bool node_on = lane < 16 && k < N;
    double T[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const int s = wave + r * SWEEP_WAVES;
        T[r] = (node_on && s < nmv) ? OgGen::tail_one(mv0 + s, k, base, a.cvec) : 0.0;
    }
    __syncthreads();
    if (!node_on) return;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const int s = wave + r * SWEEP_WAVES;
        if (s >= nmv) break;
        const ogt_int8 rec = OGT_SLOT[mv0 + s];
        const double y = ybuf[s * 16 + lane];
        const int row = rec.v[3] + k;
        publish_row<FUSED>(a, row, y - T[r]);
        a.t0[row] = T[r];
        a.y0[rec.v[4] + k] = y;
    }
}

template <bool FUSED>
__device__ __forceinline__ void eval_rows_body(const ogk_args& a, const int bx, double* lds) {
    // a row whose code contains a sequential sum (the cost with its running-cost quadrature) would be one lane
    // evaluating every term with its loads: the workgroup computes the terms into LDS first
    bool terms = false;
    if (TERM_CACHE) {
#pragma unroll
        for (int u = 0; u < SWEEP_WAVES; ++u) {
            const int w2 = bx * SWEEP_WAVES + u;
            if (w2 < OGT_N_ROWWAVES) terms = terms || OGT_ROWWAVE[w2].w != 0;
        }
        if (terms) {                                      // workgroup-uniform
            fill_terms<SWEEP_THREADS>(a, lds);
            __syncthreads();
        }
    }
    const int w = bx * SWEEP_WAVES + ((int)threadIdx.x >> 6);
    if (w >= OGT_N_ROWWAVES) return;
    const int4 rw = OGT_ROWWAVE[w];                       // {group, first element, group length, has a sum}
    const int k = rw.y + ((int)threadIdx.x & 63);
    if (k >= rw.z) return;
    const XColT base = make_xcolt(a, -1, 0.0, terms ? lds : nullptr);
    int row;
    const double v = OgGen::item_value(rw.x, 0, k, base, a.y0, a.cvec, &row);
    publish_row<FUSED>(a, row, v);
}

#ifdef OGK_HAS_MAIN
__global__ __launch_bounds__(SWEEP_THREADS) void ogk_eval(const ogk_args a, const int ndef) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int id = (int)blockIdx.x;
    // the evaluation after this one counts into the other slot: clear it now (stream order
    // makes this visible to the next launch)
    if (id == 0 && threadIdx.x == 0) {
        *a.nonfinite_next = 0;
        if (a.jt_bump) __hip_atomic_store(a.jt_launches, *a.jt_launches + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (id < ndef) eval_defect_body<false>(a, id, lds);
    else eval_rows_body<false>(a, id - ndef, lds);
}
#endif

// ------------------------------------------------------------------------------------------
// Mode 1: the structured sweep.  At the sizes this engine sees (n ~ 10^2..10^4) a sweep is a
// chain of dependent global loads and dependent f64 operations (~13 ns each on gfx950), so the
// layout below is about latency: every workgroup learns its job from one wide record
// (OGT_COL / OGT_ELEM / OGT_TILE / OGT_SLOT, generated by the tracer), issues all its loads up
// front, and the long arithmetic chains (one traced output each) run in different wavefronts.
//
// Part A - J_T rows (= FD columns).  The owner of a row streams the "no dependency" value into
//   it and then re-evaluates the (group, output, element) items that read p[j].
//     heavy_column_body   a column with many items (phase final times): one workgroup, lanes =
//                         consecutive items (consecutive nodes of one output: same code);
//     light_columns_body  LIGHT_COLS neighbouring columns per workgroup: lane = column,
//                         wavefront = item slot.
// Part B - tile_body: d(defect_s)/d(state_s) for one collocation slot on the matrix cores.
// ------------------------------------------------------------------------------------------
#ifndef OGK_LIGHT_COLS
#define OGK_LIGHT_COLS 4
#endif
constexpr int LIGHT_COLS = OGK_LIGHT_COLS;   // columns per workgroup in light_columns_body
constexpr int HEAVY_FLAG = 1 << 30;  // in OGT_COL[j].w

// Rows are filled with the "no dependency" value EXCEPT at the positions the items (and the
// MFMA tiles) write: a per-row bitmap in LDS marks those, so no ordering between the fill stores
// and the item stores is needed and the fill drains in the background while the wavefronts are
// already in their (latency-bound) item chains.
__device__ __forceinline__ bool marked(const unsigned* bits, const int r) {
    return (bits[r >> 5] >> (r & 31)) & 1u;
}

__device__ __forceinline__ void fill_row(const ogk_args& a, double* jrow, const unsigned* bits,
                                         const int own_lo, const int own_hi, const int first,
                                         const int stride, const bool all_finite,
                                         const bool z_from_this_launch = false) {
    if (OGK_EXP & 16) return;
    if (all_finite) {
        // every row of F(x0) is finite: (F0-F0)/dx is plain zero.  16-byte stores on the aligned
        // pairs that are wholly free, 8-byte stores for the rest.
        const int a0 = (int)((reinterpret_cast<unsigned long long>(jrow) >> 3) & 1);
        const double2 zero2 = make_double2(0.0, 0.0);
        for (int q = first; a0 + 2 * q < OgGen::M; q += stride) {
            const int r = a0 + 2 * q;
            const bool skip0 = (r >= own_lo && r < own_hi) || marked(bits, r);
            const bool has1 = r + 1 < OgGen::M;
            const bool skip1 = !has1 || (r + 1 >= own_lo && r + 1 < own_hi) || marked(bits, r + 1);
            if (!skip0 && !skip1) {
                *reinterpret_cast<double2*>(jrow + r) = zero2;
            } else {
                if (!skip0) jrow[r] = 0.0;
                if (!skip1) jrow[r + 1] = 0.0;
            }
        }
        if (a0 && first == 0 && !(0 >= own_lo && 0 < own_hi) && !marked(bits, 0)) jrow[0] = 0.0;
    } else {                        // z carries NaN for the non-finite rows
        for (int r = first; r < OgGen::M; r += stride)
            if ((r < own_lo || r >= own_hi) && !marked(bits, r))
                jrow[r] = z_from_this_launch ? __hip_atomic_load(&a.z[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : a.z[r];
    }
}

__device__ __forceinline__ void eval_item(const ogk_args& a, const int4 item, const XCol& xa,
                                          const double dx, double* jrow) {
    int row;
    const double v = OgGen::item_value(item.x, item.y, item.z, xa, a.y0, a.cvec, &row);
    jrow[row] = (v - a.f0[row]) / dx;         // row == item.w (the position marked in the bitmap)
}

constexpr int ROW_WORDS = (OgGen::M + 31) / 32;      // bitmap words per J_T row

__device__ __forceinline__ void heavy_column_body(const ogk_args& a, const int j, unsigned* bits) {
    const int tid = (int)threadIdx.x;
    const int4 col = OGT_COL[j];
    const int own_lo = col.z, own_hi = col.w & ~HEAVY_FLAG;
    const double xb = a.x0[j];
    const double xj = xb + a.h[j];
    const double dx = xj - xb;
    const bool all_finite = *a.nonfinite == 0;
    const bool fill = !all_finite | jt_needs_fill(a);      // workgroup-uniform
    double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
    if (fill) {
        for (int w = tid; w < ROW_WORDS; w += SWEEP_THREADS) bits[w] = 0u;
        __syncthreads();
        for (int e = col.x + tid; e < col.y; e += SWEEP_THREADS) {
            const int r = OGT_ELEM[e].w;
            atomicOr(&bits[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        fill_row(a, jrow, bits, own_lo, own_hi, tid, SWEEP_THREADS, all_finite);
        if (!all_finite && tid == 0) jt_mark_nan_fill(a);
    }
    const XCol xa{a.x0, j, xj};
    for (int e = col.x + tid; e < ((OGK_EXP & 8) ? 0 : col.y); e += SWEEP_THREADS)
        eval_item(a, OGT_ELEM[e], xa, dx, jrow);
}

__device__ __forceinline__ void light_columns_body(const ogk_args& a, const int first_j,
                                                   unsigned* bits) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#if OGK_TRACE
    const long long t_begin = __builtin_amdgcn_s_memtime();
    const long long t_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- everything this thread will need, requested before anything is waited for
    constexpr int WPR = SWEEP_WAVES / LIGHT_COLS;               // wavefronts sharing one row
    const int cf = wave / WPR;                                  // the row this wavefront fills
    const int jf = first_j + cf;
    const int ji = first_j + lane;                              // the column this lane evaluates
    const bool fill_on = jf < a.col_hi;
    const bool item_on = lane < LIGHT_COLS && ji < a.col_hi;
    const int4 colf = OGT_COL[fill_on ? jf : a.col_lo];
    const int4 coli = OGT_COL[item_on ? ji : a.col_lo];
    const double xb = a.x0[item_on ? ji : a.col_lo];
    const double hh = a.h[item_on ? ji : a.col_lo];
    const bool all_finite = *a.nonfinite == 0;
    int4 item = make_int4(0, 0, 0, 0);
    const bool col_live = item_on && !(coli.w & HEAVY_FLAG);
    const bool has_item = col_live && coli.x + wave < coli.y;
    if (has_item) item = OGT_ELEM[coli.x + wave];

    // ---- a registered (persistent-zero) buffer needs no fill while F(x0) is finite: straight to the items
    if (!all_finite | jt_needs_fill(a)) {                        // workgroup-uniform
        // ---- mark the positions the items will write
        for (int w = tid; w < LIGHT_COLS * ROW_WORDS; w += SWEEP_THREADS) bits[w] = 0u;
        __syncthreads();
        if (has_item) {
            unsigned* mine = bits + lane * ROW_WORDS;
            atomicOr(&mine[item.w >> 5], 1u << (item.w & 31));
            for (int e = coli.x + wave + SWEEP_WAVES; e < coli.y; e += SWEEP_WAVES) {
                const int r = OGT_ELEM[e].w;
                atomicOr(&mine[r >> 5], 1u << (r & 31));
            }
        }
        __syncthreads();

        // ---- fill around them (heavy columns are filled by their own workgroup); the stores drain
        //      while the item chains below run
        if (fill_on && !(colf.w & HEAVY_FLAG))
            fill_row(a, a.jt + (long)(jf - a.col_lo) * OgGen::M, bits + cf * ROW_WORDS, colf.z,
                     colf.w, (wave % WPR) * 64 + lane, 64 * WPR, all_finite);
        if (!all_finite && tid == 0) jt_mark_nan_fill(a);
    }
#if OGK_TRACE
    const long long t_filled = __builtin_amdgcn_s_memtime();
#endif
    if (!has_item) return;
    // ---- items: lane = column, wavefront = item slot
    const double xj = xb + hh;
    const double dx = xj - xb;
    double* jrow = a.jt + (long)(ji - a.col_lo) * OgGen::M;
    const XCol xa{a.x0, ji, xj};
#if OGK_TRACE
    const long long t_ready = __builtin_amdgcn_s_memtime();
#endif
    if (!(OGK_EXP & 8)) {
        eval_item(a, item, xa, dx, jrow);
        for (int e = coli.x + wave + SWEEP_WAVES; e < coli.y; e += SWEEP_WAVES)
            eval_item(a, OGT_ELEM[e], xa, dx, jrow);
    }
#if OGK_TRACE
    if (lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        double* t = jrow + 8 * wave;
        t[0] = 1.0e6 + wave; t[1] = (double)t_begin; t[2] = (double)t_filled; t[3] = (double)t_ready;
        t[4] = (double)__builtin_amdgcn_s_memtime(); t[5] = (double)(coli.y - coli.x);
        t[6] = (double)t_real0; t[7] = (double)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

// Part B.  One wavefront = one (16 perturbed columns) x (16 nodes) tile of one collocation slot.
// Every wavefront of the sweep uses its D panel exactly once, so the B operands are read straight
// from the L2-resident operand image (512 contiguous bytes per step) into registers, a chunk of
// k-steps ahead of the MFMAs - no LDS round trip and no workgroup barrier on this path (the
// evaluation and dense kernels, which reuse a panel across wavefronts/states, stage it in LDS).
__device__ __forceinline__ void tile_body(const ogk_args& a, const int bx) {
#if OGK_TRACE
    const long long t_begin = __builtin_amdgcn_s_memtime();
    long long t_staged = 0, t_mfma = 0;
#endif
    const int4 tile = OGT_TILE[bx];                    // {slot, tile group, node tile}
    const int slot = tile.x, nt = tile.z;
    const ogt_int8 rec = OGT_SLOT[slot];
    const int N = rec.v[0], g = rec.v[1], leaf = rec.v[2], row0 = rec.v[3], y0off = rec.v[4];
    const bool diag = rec.v[6] & 1, generic = rec.v[6] & 2;
    const int dep0 = rec.v[7] >> 12, ndep = rec.v[7] & 0xfff;
    const int KS = (N + 3) >> 2;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lk = lane >> 4;
    const int l0 = (tile.y * SWEEP_WAVES + wave) * 16;  // first slice offset of this wave's tile
    if (l0 >= N || leaf + l0 >= a.col_hi || leaf + l0 + 16 <= a.col_lo) return;
    const int k = nt * 16 + (lane & 15);                // output node of this lane
    const int la = l0 + (lane & 15);                    // A-operand row of this lane
    (void)g;

    // ---- requests first: operands of the first chunk, this lane's perturbation, epilogue inputs
    const double* bsrc = a.dfrag + a.dfrag_off[rec.v[5]] + (long)nt * KS * 64 + lane;
    const double* xop = a.xop + y0off;                  // base operands, written by mode 0
    constexpr int CH = 10;                              // k-steps per chunk
    double bv[CH], av[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int l = u * 4 + lk;
        bv[u] = (u < KS) ? bsrc[u * 64] : 0.0;
        av[u] = (u < KS && l < N) ? xop[l] : 0.0;
    }
    const bool a_on = la < N;
    const double xa_b = a.x0[a_on ? leaf + la : leaf];
    const double xa_h = a.h[a_on ? leaf + la : leaf];
    const bool k_on = k < N;
    const int row = row0 + (k_on ? k : 0);
    const double t_base = a.t0[row];
    const double f_base = a.f0[row];
    double xbv[4], hv[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int lc = l0 + lk + 4 * reg;
        const int jj = leaf + ((k_on && lc < N) ? lc : 0);
        xbv[reg] = a.x0[jj];
        hv[reg] = a.h[jj];
    }

    // ---- the dynamics term on the diagonal (k == own perturbed node) needs neither the operands
    //      nor the MFMA result: its chain runs while the loads above are in flight
    double t_diag = t_base;
    bool have_diag = false;
    {
        const int lc = k;                                // column whose perturbed node is k
        const int reg = (lc - l0 - lk) >> 2;
        have_diag = diag && k_on && lc >= l0 + lk && ((lc - l0 - lk) & 3) == 0 && reg < 4 && lc < N &&
                    leaf + lc >= a.col_lo && leaf + lc < a.col_hi;
        if (have_diag) {
            const int jd = leaf + lc;
            const double xbd = a.x0[jd];
            const XCol xd{a.x0, jd, xbd + a.h[jd]};
            t_diag = OgGen::tail_one(slot, k, xd, a.cvec);
        }
    }
#if OGK_TRACE
    t_staged = __builtin_amdgcn_s_memtime();
#endif

    // A operand: row (lane & 15) is the state vector with its own element perturbed
    double hit_v = 0.0;
    if (a_on) {
        const XCol xa{a.x0, leaf + la, xa_b + xa_h};
        hit_v = OgGen::mv_operand(slot, la, xa, a.cvec);
    }
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int ks0 = 0; ks0 < KS; ks0 += CH) {
        // next chunk's operands are requested before this chunk's MFMAs issue
        double bn[CH], an[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ks = ks0 + CH + u;
            const int l = ks * 4 + lk;
            bn[u] = (ks < KS) ? bsrc[ks * 64] : 0.0;
            an[u] = (ks < KS && l < N) ? xop[l] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ks = ks0 + u;
            if (ks < KS) {
                const double aop = (ks * 4 + lk == la) ? hit_v : av[u];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bv[u], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) { bv[u] = bn[u]; av[u] = an[u]; }
    }
#if OGK_TRACE
    t_mfma = __builtin_amdgcn_s_memtime();
#endif
    if (!k_on) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int lc = l0 + lk + 4 * reg;
        const int j = leaf + lc;
        if (lc >= N || j < a.col_lo || j >= a.col_hi) continue;
        const double xj = xbv[reg] + hv[reg];
        const double dx = xj - xbv[reg];
        double t = (have_diag && lc == k) ? t_diag : t_base;
        if (generic) {
            // rare: the dynamics term of node k reads p[j] through something other than the
            // state's own sample at k - take the dependency-table scan
            bool reads = diag && k == lc;
            for (int d = dep0; d < dep0 + ndep; ++d) {
                const int kind = OgGen::DEP_KIND(d), base_d = OgGen::DEP_BASE(d);
                reads = reads || (kind == 1 ? (base_d + k == j)
                                            : (j >= base_d && j < base_d + OgGen::DEP_CNT(d)));
            }
            if (reads) {
                const XCol xg{a.x0, j, xj};
                t = OgGen::tail_one(slot, k, xg, a.cvec);
            }
        }
        const double val = acc[reg] - t;
        a.jt[(long)(j - a.col_lo) * OgGen::M + row] = (val - f_base) / dx;
    }
#if OGK_TRACE
    if (lane == 0 && leaf + l0 >= a.col_lo && leaf + l0 < a.col_hi) {
        __builtin_amdgcn_s_waitcnt(0);
        double* t = a.jt + (long)(leaf + l0 - a.col_lo) * OgGen::M + row0 + nt * 16;
        t[0] = 2.0e6 + wave; t[1] = (double)t_begin; t[2] = (double)t_staged; t[3] = (double)t_mfma;
        t[4] = (double)__builtin_amdgcn_s_memtime(); t[5] = (double)bx;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Mode 5 (ogk_fused): modes 0 and 1 as ONE launch into a registered persistent-zero buffer.  The first
// workgroups of the grid are the evaluation workgroups (they produce F(x0) for the caller, as in mode 0); the
// sweep workgroups that follow do not consume anything from them: what a difference quotient needs of the
// base point - the base value of a row item, the base collocation products of a node tile - is recomputed by
// the workgroup that uses it, with the same device functions and the same MFMA chain, hence the same bits.  A
// kernel boundary (drain, cache write-back / invalidate, dispatch) between the evaluation and the sweep costs a
// third of a step at the sizes of this engine; loads from another workgroup's results inside one kernel would
// have to bypass the (per-XCD, mutually incoherent) L2s, which is slower still (both measured, round 1).
//   fz_light_body   a run of <= 16 neighbouring columns whose defect items lie in one (defect group, 16-node
//                   tile): the whole workgroup stages the tile's D^T panel and the group's operands in LDS,
//                   one wavefront runs the tile's MFMA chain, the others evaluate the items' long parts
//                   meanwhile (lane = column, perturbed | base; wavefront = item slot).
//   fz_heavy_part   one (defect group, node tile) of a column with many items, staged the same way; slots of
//                   items that share their code, one wavefront each (lanes = items, perturbed | base).
//   fz_tile_body    d(defect_s)/d(state_s): <= 7 column tiles per workgroup share ONE base product / base
//                   dynamics term / diagonal term, computed by otherwise idle wavefronts.
// Nothing is filled and nobody waits: the sweep workgroups write the positions that can be non-zero; when F(x0)
// has non-finite rows (or the previous launch left a NaN fill) the LAST evaluation workgroup to finish fills
// the rows from z around those positions (finish_eval).  The launch arguments are pointers only.
// ------------------------------------------------------------------------------------------
// Timing experiments only: -DOGK_FZ=<mask> removes pieces of ogk_fused - 2 = no base-product chain in the light
// workgroups, 4 = no operand staging there, 16 = evaluation workgroups do nothing, 32 = heavy columns skipped,
// 2048 = no light items.  Results are wrong with any bit set.
#ifndef OGK_FZ
#define OGK_FZ 0
#endif


// barrier that orders LDS traffic only: the global stores of the fill keep draining underneath
// (__syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// What is left of the dependence on the evaluation in the fused launch.  A registered (persistent-zero)
// buffer needs a fill only when F(x0) of THIS launch has non-finite rows (NaN in those rows of every column)
// or when the previous launch into the buffer left such a fill behind.  Nobody waits for that verdict:
// every evaluation workgroup takes a ticket when its results are out, and the one that draws the last
// ticket - by then the count of non-finite rows is complete - looks at it and, in the rare case, fills
// every row of this launch's block from z (0, or NaN) around the positions the sweep workgroups write
// (bitmap per row from the column's item list, as in mode 1).  It also resets the ticket, so that the
// launch arguments do not depend on how many launches went before.
__device__ __forceinline__ void fz_fill_from_z(const ogk_args& a, unsigned* bits) {
    const int tid = (int)threadIdx.x;
    for (int j = a.col_lo; j < a.col_hi; ++j) {
        const int4 col = OGT_COL[j];
        const int own_lo = col.z, own_hi = col.w & ~HEAVY_FLAG;
        for (int w = tid; w < ROW_WORDS; w += SWEEP_THREADS) bits[w] = 0u;
        __syncthreads();
        for (int e = col.x + tid; e < col.y; e += SWEEP_THREADS) {
            const int r = OGT_ELEM[e].w;
            atomicOr(&bits[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        fill_row(a, a.jt + (long)(j - a.col_lo) * OgGen::M, bits, own_lo, own_hi, tid, SWEEP_THREADS, false, true);
        __syncthreads();
    }
}

__device__ __forceinline__ void finish_eval(const ogk_args& a, const unsigned n_eval, unsigned* bits) {
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);          // this wavefront's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int last = 0;
        if (t + 1u == n_eval) {
            // the counts of non-finite rows were performed before their workgroups' tickets, at the memory side.
            // Two independent loads, one round trip; the ticket's reset needs no answer.
            const int bad = __hip_atomic_load(a.nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned st = __hip_atomic_load(a.jt_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned gen = __hip_atomic_load(a.jt_launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            __hip_atomic_store(a.ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // this launch's number; the count of non-finite rows moves to where later kernels and the host read
            // it, the counter is zero again for the next launch (nothing of this is a launch argument)
            __hip_atomic_store(a.jt_launches, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.nonfinite_result, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.ptail) a.ptail[OgGen::M] = (double)bad;      // host transfer: the count travels with F(x0)
            __hip_atomic_store(a.nonfinite, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bad != 0) __hip_atomic_store(a.jt_state, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (bad != 0 || st == gen - 1u) ? 1 : 0;
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last) fz_fill_from_z(a, bits);
}

// base collocation products of one (defect group, node tile) on one wavefront: states are the rows of the
// A operand (xt[state][NP] in LDS or global), D^T panel straight from the operand image.  The k-ordered
// chain of ogk_eval.  store(state, node, value) for the live entries.
template <class Store>
__device__ __forceinline__ void base_products_tile(const double* panel, const int N, const int nt,
                                                   const int nmv, const double* xt, const int xstride,
                                                   const Store& store) {
    const int lane = (int)threadIdx.x & 63, lk = lane >> 4, srow = lane & 15;
    const int KS = (N + 3) >> 2;
    const double* bsrc = panel + lane;                  // the tile's [KS][64] operand panel
    const bool live = srow < nmv;
    const double* xrow = xt + (live ? srow : 0) * xstride;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    constexpr int CH = 8;
    double bv[CH], av[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int l = u * 4 + lk;
        bv[u] = u < KS ? bsrc[u * 64] : 0.0;
        av[u] = (u < KS && live && l < N) ? xrow[l] : 0.0;
    }
    for (int ks0 = 0; ks0 < KS; ks0 += CH) {
        // the next chunk's operands are requested before this chunk's MFMAs issue
        double bn[CH], an[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ks = ks0 + CH + u;
            const int l = ks * 4 + lk;
            bn[u] = ks < KS ? bsrc[ks * 64] : 0.0;
            an[u] = (ks < KS && live && l < N) ? xrow[l] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (ks0 + u < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < CH; ++u) { bv[u] = bn[u]; av[u] = an[u]; }
    }
    // C/D layout: node = lane & 15, state = (lane >> 4) + 4 * reg
    const int k = nt * 16 + (lane & 15);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int st = lk + 4 * reg;
        if (st < nmv && k < N) store(st, k, acc[reg]);
    }
}

constexpr int FZ_COLS = OGT_LGRP_COLS;          // columns of a light workgroup of the fused launch (tracer)
// (F(x0 + h e_j) - F(x0)) / dx for one row item, with the base value from the partner lane (FZ_COLS further on, same item, unperturbed x)
template <class XA>
__device__ __forceinline__ void eval_item_paired(const ogk_args& a, const int4 item, const XA& xa, const bool base_role,
                                                 const double dx, double* jrow, double* packed) {
    int row;
    const double v = OgGen::item_value(item.x, item.y, item.z, xa, a.y0, a.cvec, &row);
    const double v0 = __shfl_down(v, FZ_COLS);
    if (!base_role) {
        const double q = (v - v0) / dx;
        jrow[row] = q;
        if (packed) *packed = q;
    }
}

constexpr int FZ_MAXN = OgGen::MAX_NODES;                      // longest phase
constexpr int FZ_NP = ((FZ_MAXN + 3) / 4) * 4;
constexpr int FZ_ROUNDS = 2;                                    // items per column and wavefront evaluated ahead of the base products
constexpr int FZ_ITEM_WAVES = SWEEP_WAVES - 1;                 // item slots of a light workgroup (the last wavefront
                                                               // runs the MFMA chain instead)
// LDS of a light workgroup: D panel of the tile [KS][64] | operands [state][NP] | base products [state][N]
constexpr size_t FZ_TILE_DOUBLES = (size_t)(FZ_NP / 4) * 64 + (size_t)OgGen::MAX_NMV * (FZ_NP + FZ_MAXN);
// ... | base terms of the program's sequential sums [N_TERMS] (workgroups with such items)
constexpr size_t FZ_LDS_BYTES = (FZ_TILE_DOUBLES + TERM_DOUBLES) * sizeof(double);

// flag the service wavefront raises in LDS for the other wavefronts of its workgroup
__device__ __forceinline__ void lds_flag_raise(int* flag, const int value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    *reinterpret_cast<volatile int*>(flag) = value;
}
__device__ __forceinline__ int lds_flag_wait(int* flag) {
    int v;
    while ((v = *reinterpret_cast<volatile int*>(flag)) == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}

#if OGK_TRACE                           // tools/trace_fused.py: phase stamps of every wavefront, next to the results
#define FZ_TRACE_DECL(kind_) long long tr_[6] = {(long long)__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0}; \
    const int tr_kind_ = (kind_)
#define FZ_STAMP(i) tr_[i] = (long long)__builtin_amdgcn_s_memrealtime()
#define FZ_TRACE_OUT(a_)                                                                                   \
    do {                                                                                                   \
        if (((int)threadIdx.x & 63) == 0 && (a_).trace) {                                                  \
            double* t_ = (a_).trace + ((long)blockIdx.x * SWEEP_WAVES + ((int)threadIdx.x >> 6)) * 8;       \
            t_[0] = (double)tr_kind_; t_[1] = (double)tr_[0]; t_[2] = (double)tr_[1]; t_[3] = (double)tr_[2]; \
            t_[4] = (double)tr_[3]; t_[5] = (double)tr_[4]; t_[6] = (double)tr_[5];                          \
            t_[7] = (double)__builtin_amdgcn_s_memrealtime();                                              \
        }                                                                                                  \
    } while (0)
#else
#define FZ_TRACE_DECL(kind_) do { } while (0)
#define FZ_STAMP(i) do { } while (0)
#define FZ_TRACE_OUT(a_) do { } while (0)
#endif

// LDS of a workgroup that owns one (defect group, node tile): what its service wavefront needs
struct FzTile {
    double* dpanel;     // [KS][64] D^T panel of the tile in operand order
    double* xt;         // [state][NP] operands of the group
    double* yb;         // [state][N] base products (the tile's nodes)
};
__device__ __forceinline__ FzTile fz_tile_lds(double* lds) {
    FzTile t;
    t.dpanel = lds;
    t.xt = lds + (FZ_NP / 4) * 64;
    t.yb = t.xt + OgGen::MAX_NMV * FZ_NP;
    return t;
}
// everybody requests the tile's D^T panel and the group's operands and parks them in LDS; the caller's barrier
// publishes them to the service wavefront
__device__ __forceinline__ void fz_stage_tile(const ogk_args& a, const FzTile& t, const int nt, const int mv0,
                                              const int nmv, const int N, const int phase) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KS = (N + 3) >> 2, NP = KS << 2;
    constexpr int ST_D = (FZ_NP / 4 * 64 + SWEEP_THREADS - 1) / SWEEP_THREADS;
    constexpr int ST_S = (OgGen::MAX_NMV + SWEEP_WAVES - 1) / SWEEP_WAVES, ST_L = (FZ_NP + 63) / 64;
    const XCol xbase{a.x0, -1, 0.0};
    double st_d[ST_D];
    const double* src = a.dfrag + a.dfrag_off[phase] + (long)nt * KS * 64;
#pragma unroll
    for (int u = 0; u < ST_D; ++u) {
        const int i = tid + u * SWEEP_THREADS;
        st_d[u] = i < KS * 64 ? src[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ST_S; ++u)                             // a slot per wavefront: no divergence
#pragma unroll
        for (int q = 0; q < ST_L; ++q) {
            const int sl = wave + u * SWEEP_WAVES, l = lane + q * 64;
            if (sl < nmv && l < NP) t.xt[sl * NP + l] = l < N ? OgGen::mv_operand(mv0 + sl, l, xbase, a.cvec) : 0.0;
        }
#pragma unroll
    for (int u = 0; u < ST_D; ++u) {
        const int i = tid + u * SWEEP_THREADS;
        if (i < KS * 64) t.dpanel[i] = st_d[u];
    }
}

__device__ __forceinline__ void fz_light_body(const ogk_args& a, const int b, double* lds) {
    __shared__ int s_flag;              // base products of the tile are in LDS
    FZ_TRACE_DECL(1);
    // {first column, columns, y0 offset of the tile's first slot, node tile, first slot, slots (0: no defect
    //  items), nodes, phase}
    const ogt_int8 grp = OGT_LGRP[b];
    const int first_j = grp.v[0], cnt = grp.v[1], y0_first = grp.v[2], nt = grp.v[3];
    const int mv0 = grp.v[4], nmv = grp.v[5], N = grp.v[6], phase = grp.v[7] & 0xffff;
    const bool terms = TERM_CACHE && (grp.v[7] >> 16) != 0;     // some item contains a sequential sum
    double* tc = lds + FZ_TILE_DOUBLES;
    const bool has_tile = nmv > 0;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool service = wave == SWEEP_WAVES - 1;   // no items: the tile's MFMA chain
    // items: lanes 0..FZ_COLS-1 evaluate at x0 + h e_j, the next FZ_COLS lanes the same items at x0 -
    // one instruction stream, so the base value of a row item costs no time; wavefront = item slot
    const int cl = lane % FZ_COLS;
    const bool base_role = lane >= FZ_COLS;
    const int ji = first_j + cl;
    const bool item_on = !service && lane < 2 * FZ_COLS && cl < cnt && ji >= a.col_lo && ji < a.col_hi;
    const int4 coli = OGT_LRNG[b * FZ_COLS + cl];   // {items begin, end} of this lane's column (no OGT_COL round trip)
    const double xb = a.x0[item_on ? ji : first_j];
    const double hh = a.h[item_on ? ji : first_j];
    int4 item = make_int4(0, 0, 0, 0);
    const bool has_item = item_on && coli.x + wave < coli.y;
    if (has_item) item = OGT_ELEM[coli.x + wave];

    const FzTile t = fz_tile_lds(lds);
    if (tid == 0) s_flag = 0;
    if (has_tile && !(OGK_FZ & 4)) fz_stage_tile(a, t, nt, mv0, nmv, N, phase);
    if (terms) fill_terms<SWEEP_THREADS>(a, tc);
    lds_barrier();            // (only LDS data crosses it: global loads in flight stay in flight)
    FZ_STAMP(1);
    if (service) {
        if (has_tile && !(OGK_FZ & 2))
            base_products_tile(t.dpanel, N, nt, nmv, t.xt, ((N + 3) >> 2) << 2,
                               [&](const int st, const int k, const double v) { t.yb[st * N + k] = v; });
        FZ_STAMP(2);
        if (lane == 0) lds_flag_raise(&s_flag, 1);
        FZ_TRACE_OUT(a);
        return;
    }
    if (has_item && !(OGK_FZ & 2048)) {
        const double xj = xb + hh;
        const double dx = xj - xb;
        double* jrow = a.jt + (long)(ji - a.col_lo) * OgGen::M;
        // sharded sweeps: every value also goes straight into this rank's message (the column's packed entries:
        // its collocation block first, then its items in work-list order) - no separate gather kernel
        double* pcol = a.pvals ? a.pvals + a.poff[ji] + (coli.z - coli.x) : nullptr;
        const XColT xa = make_xcolt(a, base_role ? -1 : ji, xj, terms ? tc : nullptr);
        // The long part of an item - its dynamics term, or the whole value of a row item - does not depend on
        // the base products: the first FZ_ROUNDS items of every column are evaluated while the service
        // wavefront is still in its chain; only the subtraction from the product waits for the flag.
        double tv[FZ_ROUNDS];
        int trow[FZ_ROUNDS], tyo[FZ_ROUNDS];
#pragma unroll
        for (int r = 0; r < FZ_ROUNDS; ++r) {
            const int e = coli.x + wave + r * FZ_ITEM_WAVES;
            tv[r] = 0.0, trow[r] = 0, tyo[r] = -1;
            if (e < coli.y) {
                const int4 it = r == 0 ? item : OGT_ELEM[e];
                tv[r] = OgGen::item_tail(it.x, it.y, it.z, xa, a.cvec, &trow[r], &tyo[r]);
            }
        }
        lds_flag_wait(&s_flag);
        FZ_STAMP(3);
#pragma unroll
        for (int r = 0; r < FZ_ROUNDS; ++r) {
            const double v = tyo[r] >= 0 ? t.yb[tyo[r] - y0_first] - tv[r] : tv[r];
            const double v0 = __shfl_down(v, FZ_COLS);
            const int e = coli.x + wave + r * FZ_ITEM_WAVES;
            if (!base_role && e < coli.y) {
                const double q = (v - v0) / dx;
                jrow[trow[r]] = q;
                if (pcol) pcol[e] = q;
            }
        }
        // item_value indexes y0 by slot offset + node; the accessor turns that into the LDS tile
        const XColL xl{a.x0, base_role ? -1 : ji, xj, a.y0 + y0_first, t.yb};
        for (int e = coli.x + wave + FZ_ROUNDS * FZ_ITEM_WAVES; e < coli.y; e += FZ_ITEM_WAVES)
            eval_item_paired(a, OGT_ELEM[e], xl, base_role, dx, jrow, pcol ? pcol + e : nullptr);
    }
    FZ_STAMP(4);
    FZ_TRACE_OUT(a);
}

// One part of a heavy column (a phase's final time, say: it moves every defect row of the phase): the column's
// items in ONE (defect group, node tile), staged exactly like a light workgroup, or its row items (no tile).
// A slot = up to 32 items with the same code (one output over the tile's nodes / one row group): lanes 0..31 at
// x0 + h e_j, lanes 32..63 the same items at x0, one wavefront per slot.
__device__ __forceinline__ void fz_heavy_part(const ogk_args& a, const int pidx, double* lds) {
    __shared__ int s_flag;
    FZ_TRACE_DECL(2);
    const ogt_int8 rec = OGT_HPART[pidx];      // {column, slots begin, end, y0 offset, node tile, first slot, slots, nodes | phase << 20}
    const int j = rec.v[0];
    if (j < a.col_lo || j >= a.col_hi) return;
    const int y0_first = rec.v[3], nt = rec.v[4], mv0 = rec.v[5], nmv = rec.v[6];
    const int N = rec.v[7] & 0xfffff, phase = (rec.v[7] >> 20) & 0x3ff;
    const bool terms = TERM_CACHE && ((rec.v[7] >> 30) & 1) != 0;
    double* tc = lds + FZ_TILE_DOUBLES;
    const bool has_tile = nmv > 0;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool service = wave == SWEEP_WAVES - 1;
    const double xb = a.x0[j];
    const double hh = a.h[j];
    const int il = lane & 31;
    const bool base_role = lane >= 32;
    int4 slot = make_int4(0, 0, 0, 0), item = make_int4(0, 0, 0, 0);
    const bool has_slot = !service && rec.v[1] + wave < rec.v[2];
    if (has_slot) {
        slot = OGT_HSLOT[rec.v[1] + wave];
        item = OGT_HELEM[slot.x + (il < slot.y ? il : 0)];
    }
    const FzTile t = fz_tile_lds(lds);
    if (tid == 0) s_flag = 0;
    if (has_tile) fz_stage_tile(a, t, nt, mv0, nmv, N, phase);
    if (terms) fill_terms<SWEEP_THREADS>(a, tc);
    lds_barrier();
    FZ_STAMP(1);
    if (service) {
        if (has_tile)
            base_products_tile(t.dpanel, N, nt, nmv, t.xt, ((N + 3) >> 2) << 2,
                               [&](const int st, const int k, const double v) { t.yb[st * N + k] = v; });
        FZ_STAMP(2);
        if (lane == 0) lds_flag_raise(&s_flag, 1);
        FZ_TRACE_OUT(a);
        return;
    }
    if (has_slot) {
        const double xj = xb + hh;
        const double dx = xj - xb;
        double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
        double* pcol = a.pvals ? a.pvals + a.poff[j] : nullptr;
        const XColT xa = make_xcolt(a, base_role ? -1 : j, xj, terms ? tc : nullptr);
        // as in the light workgroups: the items' long chains run while the service wavefront is in its own
        for (int s0 = rec.v[1] + wave; s0 < rec.v[2]; s0 += FZ_ROUNDS * FZ_ITEM_WAVES) {
            double tv[FZ_ROUNDS];
            int trow[FZ_ROUNDS], tyo[FZ_ROUNDS], cnt[FZ_ROUNDS], tpos[FZ_ROUNDS];
#pragma unroll
            for (int r = 0; r < FZ_ROUNDS; ++r) {
                const int s = s0 + r * FZ_ITEM_WAVES;
                tv[r] = 0.0, trow[r] = 0, tyo[r] = -1, cnt[r] = 0, tpos[r] = 0;
                if (s < rec.v[2]) {
                    const int4 sl = (r == 0 && s0 == rec.v[1] + wave) ? slot : OGT_HSLOT[s];
                    const int4 it = (r == 0 && s0 == rec.v[1] + wave) ? item : OGT_HELEM[sl.x + (il < sl.y ? il : 0)];
                    cnt[r] = sl.y;
                    tpos[r] = it.w;                     // the item's place among the column's packed entries
                    tv[r] = OgGen::item_tail(it.x, it.y, it.z, xa, a.cvec, &trow[r], &tyo[r]);
                }
            }
            if (s0 == rec.v[1] + wave) {
                lds_flag_wait(&s_flag);
                FZ_STAMP(3);
            }
#pragma unroll
            for (int r = 0; r < FZ_ROUNDS; ++r) {
                const double v = tyo[r] >= 0 ? t.yb[tyo[r] - y0_first] - tv[r] : tv[r];
                const double v0 = __shfl_down(v, 32);
                if (!base_role && il < cnt[r]) {
                    const double q = (v - v0) / dx;
                    jrow[trow[r]] = q;
                    if (pcol) pcol[tpos[r]] = q;
                }
            }
        }
    }
    FZ_STAMP(4);
    FZ_TRACE_OUT(a);
}

// MFMA tiles of the fused launch.  Workgroup = (collocation slot, node tile, up to SWEEP_WAVES - 1 column
// tiles): wavefront w < nct owns the (16 perturbed columns) x (16 nodes) tile of column tile ct0 + w - one
// accumulator, its MFMA chain and, on the diagonal tile only, the perturbed dynamics term.  What every column
// tile of the node tile shares - the base product and the base dynamics term of the 16 nodes - is computed
// ONCE per workgroup by the last wavefront (and the one before it when it is free), in parallel, and handed
// over through LDS; the column wavefronts meet it only at their epilogue.
__device__ __forceinline__ void fz_tile_body(const ogk_args& a, const int bx, double* lds) {
    __shared__ int s_flags[3];          // [0] base products of the node tile, [1] base dynamics terms, [2] diagonal terms
    FZ_TRACE_DECL(3);
    const int4 tile = OGT_FTILE[bx];                   // {slot, first column tile, node tile, column tiles}
    const int slot = tile.x, ct0 = tile.y, nt = tile.z, nct = tile.w;
    const ogt_int8 rec = OGT_SLOT[slot];
    const int N = rec.v[0], leaf = rec.v[2], row0 = rec.v[3];
    const bool diag = rec.v[6] & 1, generic = rec.v[6] & 2;
    const int dep0 = rec.v[7] >> 12, ndep = rec.v[7] & 0xfff;
    const int KS = (N + 3) >> 2;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lk = lane >> 4, kk = lane & 15;
    double* xt = lds;                                   // [NP] the slot's operand vector
    double* yb = lds + FZ_NP;                           // [16] base products of the node tile
    double* tb = yb + 16;                               // [16] base dynamics terms
    double* td = tb + 16;                               // [16] dynamics terms with the node's own sample perturbed
    const XCol xbase{a.x0, -1, 0.0};
    for (int l = tid; l < KS * 4; l += SWEEP_THREADS)
        xt[l] = l < N ? OgGen::mv_operand(slot, l, xbase, a.cvec) : 0.0;
    if (tid < 3) s_flags[tid] = 0;
    const int k = nt * 16 + kk;                         // output node of this lane
    const bool k_on = k < N;
    const double* bsrc = a.dfrag + a.dfrag_off[rec.v[5]] + (long)nt * KS * 64 + lane;
    constexpr int CH = 10;                              // k-steps per chunk
    const int prod_wave = SWEEP_WAVES - 1;
    const int term_wave = nct < SWEEP_WAVES - 1 ? SWEEP_WAVES - 2 : SWEEP_WAVES - 1;
    // the diagonal column tile (perturbed node = output node) also needs the dynamics term at x0 + h e_j for its
    // 16 (node, own sample) pairs: a third free wavefront takes that chain off the column wavefront's path
    const bool diag_here = diag && nt >= ct0 && nt < ct0 + nct;
    const int diag_wave = nct < SWEEP_WAVES - 2 ? SWEEP_WAVES - 3 : -1;
    const bool col_wave = wave < nct;
    double bv[CH];
    if (col_wave || wave == prod_wave) {
#pragma unroll
        for (int u = 0; u < CH; ++u) bv[u] = (u < KS) ? bsrc[u * 64] : 0.0;
    }
    lds_barrier();
    FZ_STAMP(1);
    if (!col_wave) {
        if (wave == diag_wave) {
            if (diag_here && lk == 0 && k_on && leaf + k >= a.col_lo && leaf + k < a.col_hi) {
                const int jd = leaf + k;
                const double xbd = a.x0[jd];
                const XCol xd{a.x0, jd, xbd + a.h[jd]};
                td[kk] = OgGen::tail_one(slot, k, xd, a.cvec);
            }
            if (lane == 0) lds_flag_raise(&s_flags[2], 1);
        }
        if (wave == term_wave && lk == 0) tb[kk] = k_on ? OgGen::tail_one(slot, k, xbase, a.cvec) : 0.0;
        if (wave == term_wave && lane == 0) lds_flag_raise(&s_flags[1], 1);
        if (wave == prod_wave) {
                v4f64 accb = {0.0, 0.0, 0.0, 0.0};
            for (int ks0 = 0; ks0 < KS; ks0 += CH) {
                double bn[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int ks = ks0 + CH + u;
                    bn[u] = (ks < KS) ? bsrc[ks * 64] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int ks = ks0 + u;
                    if (ks < KS) accb = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[ks * 4 + lk], bv[u], accb, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) bv[u] = bn[u];
            }
            if (lk == 0) yb[kk] = accb[0];              // every row of accb holds the base product of node k
            FZ_STAMP(2);
            if (lane == 0) lds_flag_raise(&s_flags[0], 1);
        }
        FZ_TRACE_OUT(a);
        return;
    }
    const int l0 = (ct0 + wave) * 16;                   // first slice offset of this wave's tile
    if (l0 >= N || leaf + l0 >= a.col_hi || leaf + l0 + 16 <= a.col_lo) return;
    const int la = l0 + kk;                             // A-operand row of this lane
    const bool a_on = la < N;
    const double xa_b = a.x0[a_on ? leaf + la : leaf];
    const double xa_h = a.h[a_on ? leaf + la : leaf];
    const int row = row0 + (k_on ? k : 0);
    double xbv[4], hv[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int lc = l0 + lk + 4 * reg;
        const int jj = leaf + ((k_on && lc < N) ? lc : 0);
        xbv[reg] = a.x0[jj];
        hv[reg] = a.h[jj];
    }
    // the perturbed dynamics term on the diagonal: its chain runs while the loads are in flight
    double t_diag = 0.0;
    bool have_diag = false;
    {
        const int lc = k;                                // column whose perturbed node is k
        const int reg = (lc - l0 - lk) >> 2;
        have_diag = diag && k_on && lc >= l0 + lk && ((lc - l0 - lk) & 3) == 0 && reg < 4 && lc < N &&
                    leaf + lc >= a.col_lo && leaf + lc < a.col_hi;
        if (have_diag && diag_wave < 0) {               // no free wavefront in this workgroup: own chain
            const int jd = leaf + lc;
            const double xbd = a.x0[jd];
            const XCol xd{a.x0, jd, xbd + a.h[jd]};
            t_diag = OgGen::tail_one(slot, k, xd, a.cvec);
        }
    }
    double hit_v = 0.0;
    if (a_on) {
        const XCol xa{a.x0, leaf + la, xa_b + xa_h};
        hit_v = OgGen::mv_operand(slot, la, xa, a.cvec);
    }
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int ks0 = 0; ks0 < KS; ks0 += CH) {
        double bn[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ks = ks0 + CH + u;
            bn[u] = (ks < KS) ? bsrc[ks * 64] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ks = ks0 + u;
            if (ks < KS) {
                const double aop = (ks * 4 + lk == la) ? hit_v : xt[ks * 4 + lk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bv[u], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) bv[u] = bn[u];
    }
    FZ_STAMP(2);
    lds_flag_wait(&s_flags[1]);
    lds_flag_wait(&s_flags[0]);
    if (diag_wave >= 0 && l0 == nt * 16) {              // (the diagonal tile: wavefront-uniform)
        lds_flag_wait(&s_flags[2]);
        if (have_diag) t_diag = td[kk];
    }
    FZ_STAMP(3);
    if (!k_on) return;
    const double t_base = tb[kk];
    const double f_base = yb[kk] - t_base;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int lc = l0 + lk + 4 * reg;
        const int j = leaf + lc;
        if (lc >= N || j < a.col_lo || j >= a.col_hi) continue;
        const double xj = xbv[reg] + hv[reg];
        const double dx = xj - xbv[reg];
        double t = (have_diag && lc == k) ? t_diag : t_base;
        if (generic) {
            bool reads = diag && k == lc;
            for (int d = dep0; d < dep0 + ndep; ++d) {
                const int kind = OgGen::DEP_KIND(d), base_d = OgGen::DEP_BASE(d);
                reads = reads || (kind == 1 ? (base_d + k == j)
                                            : (j >= base_d && j < base_d + OgGen::DEP_CNT(d)));
            }
            if (reads) {
                const XCol xg{a.x0, j, xj};
                t = OgGen::tail_one(slot, k, xg, a.cvec);
            }
        }
        const double val = acc[reg] - t;
        const double q = (val - f_base) / dx;
        a.jt[(long)(j - a.col_lo) * OgGen::M + row] = q;
        if (a.pvals) a.pvals[a.poff[j] + k] = q;        // the column's own block leads its packed entries
    }
    FZ_STAMP(4);
    FZ_TRACE_OUT(a);
}

// At least four wavefronts per SIMD = two workgroups per compute unit = 512 resident workgroups: a problem whose
// callbacks bring the kernel to 129 registers (C5 since round 4) would otherwise run ONE workgroup per compute unit and
// start half of its workgroups microseconds late.  Kernels that need fewer registers (84 at C3) are not affected: the
// attribute is a floor on the occupancy the register allocator must leave possible.
#ifndef OGK_FUSED_ATTR
#define OGK_FUSED_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#endif
#ifdef OGK_HAS_FUSED
__global__ __launch_bounds__(SWEEP_THREADS) OGK_FUSED_ATTR void ogk_fused(const ogk_args a, const int ndef, const int n_eval,
                                                           const int group_lo, const int n_light, const int sum_lo,
                                                           const int n_sum) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int id = (int)blockIdx.x;
    if (id < n_eval) {
        FZ_TRACE_DECL(0);
        if (!(OGK_FZ & 16)) {
            if (id < ndef) eval_defect_body<true>(a, id, lds);
            else eval_rows_body<true>(a, id - ndef, lds);
        }
        FZ_STAMP(1);
        finish_eval(a, (unsigned)n_eval, reinterpret_cast<unsigned*>(lds));
        FZ_STAMP(4);
        FZ_TRACE_OUT(a);
        return;
    }
    id -= n_eval;
    // nothing in the sweep workgroups waits for the evaluation workgroups (finish_eval).  Grid order (OGK_ORDER,
    // timing experiments): 0 = light, heavy parts, MFMA tiles; 1 = MFMA tiles, heavy parts, light; 2 = tiles and
    // light workgroups interleaved, heavy parts first
#ifndef OGK_ORDER
#define OGK_ORDER 0
#endif
    const int n_tile = OGT_N_FTILES, n_heavy = OGT_N_HPART;
    int kind, idx;                                     // 0 light, 1 heavy, 2 tile
    if (OGK_ORDER == 0) {
        if (id < n_light) kind = 0, idx = id;
        else if (id < n_light + n_heavy) kind = 1, idx = id - n_light;
        else kind = 2, idx = id - n_light - n_heavy;
    } else if (OGK_ORDER == 1) {
        if (id < n_tile) kind = 2, idx = id;
        else if (id < n_tile + n_heavy) kind = 1, idx = id - n_tile;
        else kind = 0, idx = id - n_tile - n_heavy;
    } else {
        if (id < n_heavy) kind = 1, idx = id;
        else {
            const int r = id - n_heavy, pairs = n_light < n_tile ? n_light : n_tile;
            if (r < 2 * pairs) kind = (r & 1) ? 0 : 2, idx = r >> 1;
            else if (n_light > n_tile) kind = 0, idx = r - pairs;
            else kind = 2, idx = r - pairs;
        }
    }
    // light workgroups: the ones that carry a sequential sum first (OGT_LSUM / OGT_LPLAIN: both in column order;
    // sum_lo of the first list and group_lo - sum_lo of the second lie below this launch's column range)
    // (n_sum < 0: column order - the launch fits one round of residency and the order only decides who shares a compute unit)
    if (kind == 0)
        fz_light_body(a, n_sum < 0 ? group_lo + idx
                                   : idx < n_sum ? OGT_LSUM[sum_lo + idx] : OGT_LPLAIN[group_lo - sum_lo + idx - n_sum], lds);
    else if (kind == 1) { if (!(OGK_FZ & 32)) fz_heavy_part(a, idx, lds); }
    else fz_tile_body(a, idx, lds);
}
#endif

#ifdef OGK_HAS_SWEEP
__global__ __launch_bounds__(SWEEP_THREADS) void ogk_sweep(const ogk_args a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int id = (int)blockIdx.x;
    if (id < OGT_N_TILES) {
        // MFMA tiles have the longest dependent chain: dispatch them first
        if (!(OGK_EXP & 32)) tile_body(a, id);
    } else if (id < OGT_N_TILES + OgGen::N_HEAVY) {
        // then the columns with many dependent items (e.g. phase final times): a workgroup each
        const int j = OGT_HEAVY[id - OGT_N_TILES];
        if (j >= a.col_lo && j < a.col_hi) heavy_column_body(a, j, reinterpret_cast<unsigned*>(lds));
    } else {
        // all columns of this launch, LIGHT_COLS per workgroup
        light_columns_body(a, a.col_lo + (id - OGT_N_TILES - OgGen::N_HEAVY) * LIGHT_COLS,
                           reinterpret_cast<unsigned*>(lds));
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Modes 6-9: the packed non-zeros.  One wavefront per column; entry i of column j is row own_lo + i for
// i < own_hi - own_lo (the collocation block the MFMA tiles write) and the row of item i - (own_hi - own_lo)
// after that (OGT_ELEM order) - the order codegen.sparsity() documents.
// ------------------------------------------------------------------------------------------
constexpr int PACK_THREADS = 256;
__device__ __forceinline__ int pattern_row(const int4 col, const int own, const int i) {
    return i < own ? col.z + i : OGT_ELEM[col.x + (i - own)].w;
}

#ifdef OGK_HAS_MAIN
__global__ void ogk_count_launch(const ogk_args a) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store(a.jt_launches, *a.jt_launches + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#ifdef OGK_HAS_MAIN
__global__ __launch_bounds__(PACK_THREADS) void ogk_pattern(const ogk_args a, const int rows_pass) {
    const int j = (int)(blockIdx.x * (PACK_THREADS / 64) + (threadIdx.x >> 6));
    const int lane = (int)threadIdx.x & 63;
    if (j >= OgGen::N_VAR) return;
    const int4 col = OGT_COL[j];
    const int own = (col.w & ~HEAVY_FLAG) - col.z, cnt = own + (col.y - col.x);
    if (!rows_pass) {
        if (lane == 0) a.pint[j] = cnt;
        return;
    }
    for (int i = lane; i < cnt; i += 64) a.pint[a.poff[j] + i] = pattern_row(col, own, i);
}
#endif

// Pack / unpack: one workgroup per column, its entries taken from the flat row-index array of the pattern
// (coalesced; the only dependent access is the J_T entry itself).  A phase's final time has a thousand entries,
// an ordinary column a few dozen: a workgroup per column keeps the long ones from becoming the kernel's tail.
#ifdef OGK_HAS_MAIN
__global__ __launch_bounds__(PACK_THREADS) void ogk_pack(const ogk_args a) {
    const int j = a.col_lo + (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (blockIdx.x == 0 && a.ptail) {
        for (int r = tid; r < OgGen::M; r += PACK_THREADS) a.ptail[r] = a.f0[r];
        if (tid == 0) a.ptail[OgGen::M] = (double)*a.nonfinite;
    }
    if (j >= a.col_hi) return;
    const long base = a.pind[j];
    const int cnt = (int)(a.pind[j + 1] - base);
    const double* jrow = a.jt + (long)(j - a.col_lo) * OgGen::M;
    double* out = a.pvals + a.poff[j];
    const int32_t* rows = a.prow + base;
    for (int i = tid; i < cnt; i += PACK_THREADS) out[i] = jrow[rows[i]];
}
#endif

#ifdef OGK_HAS_MAIN
__global__ __launch_bounds__(PACK_THREADS) void ogk_unpack(const ogk_args a) {
    // rows [ulo, uhi) of the full matrix except this rank's own block
    int j = a.ulo + (int)blockIdx.x;
    if (j >= a.col_lo) j += a.col_hi - a.col_lo;
    const int tid = (int)threadIdx.x;
    // a rank without columns of its own (jt_bump): nobody else records that this step fills with NaN.  The
    // other workgroups' decision below does not depend on this store (they see the same non-zero count)
    if (a.jt_bump && blockIdx.x == 0 && tid == 0 && *a.nonfinite != 0) jt_mark_nan_fill(a);
    if (j >= a.uhi) return;
    const long base = a.pind[j];
    const int cnt = (int)(a.pind[j + 1] - base);
    double* jrow = a.jt + (long)j * OgGen::M;
    // every rank evaluated the same F(x0): its z says which rows are NaN in every column; a fill (NaN now, or
    // zeros to clean up after one) goes first, then the barrier, then the column's entries
    const bool fill = a.jt_sparse && (*a.nonfinite != 0 || *a.jt_state == *a.jt_launches - 1u);
    if (fill) {                                         // workgroup-uniform
        for (int r = tid; r < OgGen::M; r += PACK_THREADS) jrow[r] = a.z[r];
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    const double* in = a.pvals + a.poff[j];
    const int32_t* rows = a.prow + base;
    for (int i = tid; i < cnt; i += PACK_THREADS) jrow[rows[i]] = in[i];
}
#endif

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
        const size_t need = ((size_t)KS * 64 + (size_t)OgGen::MAX_NMV * KS * 4 + 256) * sizeof(double);
        worst = need > worst ? need : worst;
    }
    const size_t terms = (size_t)TERM_DOUBLES * sizeof(double);      // evaluation row blocks: cached sum terms
    return worst > terms ? worst : terms;
}

size_t sweep_lds_bytes() { return (size_t)LIGHT_COLS * ROW_WORDS * sizeof(unsigned); }

}  // namespace

extern "C" int ogk_get_info(ogk_info* out) {
    out->abi = OGK_ABI;
    out->n_eval_blocks = defect_blocks() + (OGT_N_ROWWAVES + SWEEP_WAVES - 1) / SWEEP_WAVES;
    {
        size_t lds_bytes = defect_lds_bytes() > FZ_LDS_BYTES ? defect_lds_bytes() : FZ_LDS_BYTES;
        const size_t fill_lds = (size_t)ROW_WORDS * sizeof(unsigned);
        if (fill_lds > lds_bytes) lds_bytes = fill_lds;
        out->fused_ok = (lds_bytes <= 64 * 1024 && out->n_eval_blocks > 0) ? 1 : 0;
    }
    out->n = OgGen::N_VAR;
    out->m = OgGen::M;
    out->m_eq = OgGen::M_EQ;
    out->m_ineq = OgGen::M_INEQ;
    out->n_phase = OgGen::N_PHASE;
    out->n_mv = OgGen::N_MV;
    out->n_groups = OgGen::N_GROUPS;
    out->n_cvec = OgGen::N_CVEC;
    out->n_y0 = OgGen::N_Y0;
    for (int i = 0; i < OGK_MAX_PHASE; ++i)
        out->phase_nodes[i] = i < OgGen::N_PHASE ? OgGen::PHASE_NODES(i) : 0;
    return 0;
}

// A module may be built in parts (OGK_PART, build.py): a part answers OGK_OTHER_PART for a mode whose kernel it
// does not hold and the runtime turns to the part that does.
extern "C" int ogk_launch(const ogk_args* args, int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int ndef = defect_blocks();
    const int ncols = args->col_hi - args->col_lo;
    (void)stream;
    (void)ndef;
    (void)ncols;
#ifdef OGK_HAS_MAIN
    if (mode == 0) {
        const int eval_row_blocks = (OGT_N_ROWWAVES + SWEEP_WAVES - 1) / SWEEP_WAVES;
        if (ndef + eval_row_blocks > 0)
            hipLaunchKernelGGL(ogk_eval, dim3(ndef + eval_row_blocks), dim3(SWEEP_THREADS),
                               defect_lds_bytes(), stream, *args, ndef);
        return (int)hipGetLastError();
    }
    if (mode == 10) {
        hipLaunchKernelGGL(ogk_count_launch, dim3(1), dim3(64), 0, stream, *args);
        return (int)hipGetLastError();
    }
    if (mode == 6 || mode == 7) {
        hipLaunchKernelGGL(ogk_pattern, dim3((OgGen::N_VAR + PACK_THREADS / 64 - 1) / (PACK_THREADS / 64)),
                           dim3(PACK_THREADS), 0, stream, *args, mode == 7 ? 1 : 0);
        return (int)hipGetLastError();
    }
    if (mode == 8) {
        if (ncols <= 0 && !args->ptail) return 0;
        hipLaunchKernelGGL(ogk_pack, dim3(ncols > 0 ? ncols : 1), dim3(PACK_THREADS), 0, stream, *args);
        return (int)hipGetLastError();
    }
    if (mode == 9) {
        const int others = (args->uhi - args->ulo) - ncols;
        if (others > 0)
            hipLaunchKernelGGL(ogk_unpack, dim3(others), dim3(PACK_THREADS), 0, stream, *args);
        return (int)hipGetLastError();
    }
#endif
#ifdef OGK_HAS_SWEEP
    if (mode == 1) {
        if (ncols <= 0) return 0;
        const int light_blocks = (ncols + LIGHT_COLS - 1) / LIGHT_COLS;
        hipLaunchKernelGGL(ogk_sweep, dim3(OGT_N_TILES + OgGen::N_HEAVY + light_blocks),
                           dim3(SWEEP_THREADS), sweep_lds_bytes(), stream, *args);
        return (int)hipGetLastError();
    }
#endif
#ifdef OGK_HAS_FUSED
    if (mode == 5) {
        if (ncols <= 0) return 0;
        const int eval_row_blocks = (OGT_N_ROWWAVES + SWEEP_WAVES - 1) / SWEEP_WAVES;
        // light workgroups whose columns touch [col_lo, col_hi)
        int glo = 0, ghi = OGT_N_LGRP;
        while (glo < ghi && OGH_LGRP_J[glo + 1] <= args->col_lo) ++glo;
        while (ghi > glo && OGH_LGRP_J[ghi - 1] >= args->col_hi) --ghi;
        size_t lds_bytes = defect_lds_bytes() > FZ_LDS_BYTES ? defect_lds_bytes() : FZ_LDS_BYTES;
        const size_t fill_lds = (size_t)ROW_WORDS * sizeof(unsigned);      // finish_eval's bitmap of one row
        if (fill_lds > lds_bytes) lds_bytes = fill_lds;
        // the one-launch form writes the non-zeros only: it needs a registered (persistent-zero) output buffer
        // (og_jt_register_dev) and its LDS window (ogk_info.fused_ok); the caller runs modes 0 + 1 otherwise
        if (!args->jt_sparse || lds_bytes > 64 * 1024 || ndef + eval_row_blocks == 0) return (int)hipErrorInvalidValue;
        // A grid beyond one round of residency (two workgroups per compute unit, 512 on this part) starts its later
        // workgroups microseconds late: the light workgroups with a sequential sum - a chain of as many dependent
        // additions as the sum has terms, the longest of the launch - then go first (C5: 29.8 -> 24.6 us per launch;
        // a grid that fits one round is left in column order: C4 lost 1.1 us to the other order)
        int sum_lo = 0, n_sum = 0;
        for (int gidx = 0; gidx < ghi; ++gidx) {
            if (gidx < glo) sum_lo += OGH_LGRP_SUM[gidx];
            else n_sum += OGH_LGRP_SUM[gidx];
        }
        if (ndef + eval_row_blocks + OGT_N_FTILES + OGT_N_HPART + (ghi - glo) <= 512) n_sum = -1;
        hipLaunchKernelGGL(ogk_fused, dim3(ndef + eval_row_blocks + OGT_N_FTILES + OGT_N_HPART + (ghi - glo)),
                           dim3(SWEEP_THREADS), lds_bytes, stream, *args, ndef, ndef + eval_row_blocks, glo, ghi - glo,
                           sum_lo, n_sum);
        return (int)hipGetLastError();
    }
#endif
#ifdef OGK_HAS_AUX
    if (mode == 4) {
        if (ncols <= 0) return 0;
        hipLaunchKernelGGL(ogk_exact_struct, dim3(ncols), dim3(256), 0, stream, *args);
        return (int)hipGetLastError();
    }
    if (mode == 3) {
        if (ncols <= 0) return 0;
        int n_items = OgGen::N_ROW_ITEMS;
        for (int g = 0; g < OgGen::N_GROUPS; ++g)
            if (OgGen::G_KIND(g) == 1) n_items += OgGen::G_LEN(g);
        if (n_items > 0)
            hipLaunchKernelGGL(ogk_exact, dim3((n_items + 255) / 256, ncols), dim3(256), 0, stream, *args, n_items);
        return (int)hipGetLastError();
    }
    if (mode == 2) {
        if (ncols <= 0) return 0;
        const int row_blocks = (OgGen::N_ROW_ITEMS + 255) / 256;
        const int defect_total = ndef * ((ncols + 63) / 64);
        const int rows_total = row_blocks * ((ncols + ROWS_COLS_PER_THREAD - 1) / ROWS_COLS_PER_THREAD);
        if (defect_total + rows_total > 0)
            hipLaunchKernelGGL(ogk_dense, dim3(defect_total + rows_total), dim3(256),
                               defect_lds_bytes(), stream, *args, ndef, defect_total, row_blocks);
        return (int)hipGetLastError();
    }
#endif
    return OGK_OTHER_PART;
}
