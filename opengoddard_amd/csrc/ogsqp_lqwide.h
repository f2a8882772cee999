// ogsqp_lqwide.h - the LQ sweep for rows LONGER than the 2048 entries one workgroup's registers hold (C5: n + 1 = 6149):
// a 16-reflector panel kernel whose COLUMNS are split over several workgroups, and the block-reflector kernels
// (64 reflectors per block: k_wy_w, k_wy_make_m, k_wy_invert, k_wy_update) that stream the rest of the matrix twice per
// 64 reflectors on the FP64 matrix cores.  Included by ogsqp.hip
// inside its anonymous namespace, after ogsqp_lq16.h; DESIGN.md section 9.
//
// Until round 4 such rows went through round 2's 8-reflector kernels until the sweep had shortened them: one
// workgroup holding all 8 rows of its columns (174 us per panel at C5) and a trailing update that streams the whole of
// [C Z; Z] once per 8 reflectors (242 us) - 59 % of a C5 subproblem.  Here:
//
//   k_lq_panel16_wide   16 rows x L columns; workgroup w of ceil(L / 2048) holds the slab of columns
//                       [2048 w, 2048 (w + 1)) of all 16 rows in registers exactly like k_lq_panel16 holds its rows.
//                       A reflector step needs the products of every row with the current row over ALL columns: each
//                       workgroup adds up its slab's share, the 16 partial sums (and, from workgroup 0, the rows'
//                       entries in the pivot column) meet in a mailbox in HBM - agent-scope stores of values that are
//                       their own flags, bounded polling: the workgroups are the whole grid of the launch and wait
//                       only for each other - and every workgroup derives the same reflector scalars from the same totals and
//                       updates its own slab.  16 exchanges per panel instead of 16 x 8 workgroup-wide reductions over
//                       rows that do not fit.
//   the rest            with V of a 64-row block (four panels) and M = T^-1 = diag(1 / beta) + striu(V V')
//                       (k_wy_make_m, inverted by k_wy_invert), the update of every other row is
//                       A <- A - ((A V') M^-1) V: W = A V' by k_wy_w, then k_wy_update (coefficients W M^-1 and the
//                       rank-64 update in one pass over A; round 5 - rocBLAS dgemm before) - for the rows of the
//                       block's later panels (16 reflectors at a time, T of the panel kernel), for the rest of C Z and
//                       for Z.  The matrix is read twice and written once per 64 reflectors instead of once per 8.
//
// A wait that gives up raises flag[2] like the look-ahead's: the subproblem is then solved again by the old kernels.

constexpr int LQW_E = 8;                         // groups of 256 columns per workgroup: slabs of 2048 columns
constexpr int LQW_SLAB = 256 * LQW_E;
constexpr int LQW_MAX = 8;                       // workgroups per panel: rows of up to 16384 entries (round 5; 4 / 8192 before)
constexpr int LQW_BLOCK = 64;                    // reflectors per block reflector

// The mailbox of the column-split panel.  Every value is its own flag: a slot holds LQW_PENDING - a NaN with a payload no
// computation produces (as in k_trsv_chain) - until its workgroup stores the partial sum there, readers poll the VALUES
// (agent-scope loads) until none of them is pending: one trip through memory per exchange instead of three (store
// acknowledged, counter incremented, counter polled, data loaded).  Two generations: a launch uses generation
// `gen` and re-arms the other one for the next launch (launches of one stream run one after the other, so nobody
// still reads what is being re-armed).
constexpr unsigned long long LQW_PENDING = 0x7ff8dead5eedbeefull;
struct LqWideMail {
    double v[2][LQ16][LQW_MAX][32];              // [generation][step][workgroup]: 16 partial products | 16 pivot-column entries
};

__global__ void k_lq_wide_arm(LqWideMail* mail) {       // once, when the handle is made: everything pending
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int)(sizeof(LqWideMail) / sizeof(double))) (&mail->v[0][0][0][0])[e] = __longlong_as_double((long long)LQW_PENDING);
}

__global__ __launch_bounds__(P16_THREADS) void k_lq_panel16_wide(double* __restrict__ Tc, int ld, int mrows, int nq, int k,
                                                                double* __restrict__ V, int ldv, double* __restrict__ diagL,
                                                                Lq16Panel* __restrict__ panel, double* __restrict__ dmaxbuf,
                                                                LqWideMail* __restrict__ mail, int gen,
                                                                int* __restrict__ lost, int spin_limit) {
    constexpr int E = LQW_E;
    extern __shared__ __attribute__((aligned(32))) double lds[];
    __shared__ double s_part[2][P16_WAVES][LQ16];
    __shared__ double s_xpc[2][LQ16];
    __shared__ double s_tot[LQ16], s_pc[LQ16];       // totals over all workgroups; the rows' pivot-column entries
    __shared__ double s_lower[LQ16][LQ16];
    __shared__ double s_S[LQ16][LQ16];
    __shared__ double s_T[LQ16][LQ16];
    __shared__ double s_beta[LQ16], s_diag[LQ16];
    double* vrow = lds;                              // 2 x LQW_SLAB
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = lane >> 3, slot = wv * 8 + (lane & 7);
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int nb = min(LQ16, mrows - k);
    const int c0 = wg * LQW_SLAB;                    // first column of my slab, from the panel's first column
    const int L = min(LQW_SLAB, nq - k - c0);        // columns of my slab (> 0: the grid is ceil((nq - k) / 2048))
    const bool first = wg == 0;
    double x[2][E][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 2 * rp + h;
        const double* row = Tc + (long)(k + min(r, nb - 1)) * ld + k + c0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = 4 * (64 * e + slot);
            const double* src = row + min(j, 4 * ((L - 1) / 4));
            const dbl4 v = *(const dbl4*)src;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[h][e][i] = (r < nb && j + i < L) ? v[i] : 0.0;
        }
    }
    for (int e = tid; e < LQ16 * LQ16; e += P16_THREADS) {
        (&s_S[0][0])[e] = 0.0;
        (&s_T[0][0])[e] = 0.0;
        (&s_lower[0][0])[e] = 0.0;
    }
    double dmax = dmaxbuf[0];
    if (first) {
        // re-arm the WHOLE other generation (every step, every workgroup slot - the next launch may have more workgroups
        // or steps than this one): nobody reads it during this launch
        double* other = &mail->v[gen ^ 1][0][0][0];
        for (int e = tid; e < LQ16 * LQW_MAX * 32; e += P16_THREADS)
            __hip_atomic_store(other + e, __longlong_as_double((long long)LQW_PENDING), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int b = 0; b < LQ16; ++b) {
        if (b < nb) {                                // (uniform)
            double* vr = vrow + (b & 1) * LQW_SLAB;
            if (rp == (b >> 1)) {
                if (first) {
                    // (only the first group of four columns of the first slab can lie left of the pivot: b < 16)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * slot + i;
                        const bool left = j < b;
                        if (left) s_lower[b][j] = x[b & 1][0][i];
                        x[b & 1][0][i] = left ? 0.0 : x[b & 1][0][i];
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    dbl4 q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i] = x[b & 1][e][i];
                    *(dbl4*)(vr + 4 * (64 * e + slot)) = q;
                }
            }
            if (first && slot == (b >> 2)) {
                s_xpc[b & 1][2 * rp] = x[0][0][b & 3];
                s_xpc[b & 1][2 * rp + 1] = x[1][0][b & 3];
            }
            __syncthreads();
            double v[E][4];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const dbl4 q = *(const dbl4*)(vr + 4 * (64 * e + slot));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[e][i] = q[i];
            }
            double pa[4] = {0.0, 0.0, 0.0, 0.0}, pb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pa[i] = fma(x[0][e][i], v[e][i], pa[i]);
                    pb[i] = fma(x[1][e][i], v[e][i], pb[i]);
                }
            const double acc0 = oct_sum((pa[0] + pa[1]) + (pa[2] + pa[3]));
            const double acc1 = oct_sum((pb[0] + pb[1]) + (pb[2] + pb[3]));
            if ((lane & 7) == 0) {
                s_part[b & 1][wv][2 * rp] = acc0;
                s_part[b & 1][wv][2 * rp + 1] = acc1;
            }
            __syncthreads();
            // ---- the exchange: wavefront 0 posts this workgroup's share and collects everybody's
            if (wv == 0) {
                double* mine = &mail->v[gen][b][wg][0];
                if (lane < LQ16) {
                    double sum = 0.0;
#pragma unroll
                    for (int w8 = 0; w8 < P16_WAVES; ++w8) sum += s_part[b & 1][w8][lane];
                    __hip_atomic_store(mine + lane, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (lane < 2 * LQ16 && first) {
                    __hip_atomic_store(mine + lane, s_xpc[b & 1][lane - LQ16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // lanes 0-15: row `lane`'s partial products of all workgroups; lanes 16-31: workgroup 0's pivot-column entries.
                // (round 5: the slots of ALL workgroups are requested together and polled together - one trip through memory
                // per exchange.  Until round 4 a lane waited for workgroup 0's slot, then asked for workgroup 1's, ...: nwg
                // dependent trips of ~0.7 us, 2-3 us of every reflector step at C5.  The total is added up in the order of
                // the workgroups as before: same bits.)
                if (lane < 2 * LQ16) {
                    const int w_hi = lane < LQ16 ? nwg : 1;
                    double val[LQW_MAX];
#pragma unroll
                    for (int w = 0; w < LQW_MAX; ++w)
                        val[w] = w < w_hi ? __hip_atomic_load(&mail->v[gen][b][w][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0.0;
                    int spins = 0;
                    while (true) {
                        bool pending = false;
#pragma unroll
                        for (int w = 0; w < LQW_MAX; ++w)
                            pending = pending || (w < w_hi && (unsigned long long)__double_as_longlong(val[w]) == LQW_PENDING);
                        if (!pending) break;
                        if (++spins > spin_limit) {
                            __hip_atomic_store(lost, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int w = 0; w < LQW_MAX; ++w)
                            if (w < w_hi && (unsigned long long)__double_as_longlong(val[w]) == LQW_PENDING)
                                val[w] = __hip_atomic_load(&mail->v[gen][b][w][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    double total = 0.0;
#pragma unroll
                    for (int w = 0; w < LQW_MAX; ++w)
                        if (w < w_hi) total += val[w];
                    if (lane < LQ16) s_tot[lane] = total;
                    else s_pc[lane - LQ16] = total;
                }
            }
            __syncthreads();
            const double Db = s_tot[b], D0 = s_tot[2 * rp], D1 = s_tot[2 * rp + 1];
            const double x0 = s_pc[b];
            const double xr0 = s_pc[2 * rp], xr1 = s_pc[2 * rp + 1];
            const double sigma = sqrt(Db);
            const bool live = sigma > REDUNDANT * dmax && sigma > 0.0;
            dmax = fmax(dmax, sigma);
            const double alpha = !live ? 0.0 : (x0 >= 0.0 ? -sigma : sigma);
            const double v0 = x0 - alpha;
            const double vv = Db - x0 * x0 + v0 * v0;
            const double bt = (live && vv > 0.0) ? 2.0 / vv : 0.0;
            const double rv0 = D0 - xr0 * alpha, rv1 = D1 - xr1 * alpha;     // row . v_b
            if (tid == 0) {
                s_beta[b] = bt;
                s_diag[b] = alpha;
            }
            if (wv == 0 && (lane & 7) == 0) {
                if (2 * rp < b) s_S[2 * rp][b] = rv0;
                if (2 * rp + 1 < b) s_S[2 * rp + 1][b] = rv1;
            }
            const double f0 = (2 * rp > b && 2 * rp < nb) ? bt * rv0 : 0.0;
            const double f1 = (2 * rp + 1 > b && 2 * rp + 1 < nb) ? bt * rv1 : 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[0][e][i] = fma(-f0, v[e][i], x[0][e][i]);
                    x[1][e][i] = fma(-f1, v[e][i], x[1][e][i]);
                }
            if (first && slot == (b >> 2)) {
                x[0][0][b & 3] = fma(f0, alpha, x[0][0][b & 3]);
                x[1][0][b & 3] = fma(f1, alpha, x[1][0][b & 3]);
                if (rp == (b >> 1)) x[b & 1][0][b & 3] = v0;     // row b becomes its reflector vector
            }
        }
    }
    __syncthreads();                                              // s_beta of the last step
    if (first && tid < LQ16) {
        const int a = tid;
        double Tr[LQ16];
#pragma unroll
        for (int c = 0; c < LQ16; ++c) Tr[c] = (c == a && a < nb) ? s_beta[a] : 0.0;
#pragma unroll
        for (int b = 1; b < LQ16; ++b) {
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int c = 0; c < b; ++c) {
                if (c & 1) acc1 = fma(Tr[c], s_S[c][b], acc1);
                else acc0 = fma(Tr[c], s_S[c][b], acc0);
            }
            if (b > a && b < nb) Tr[b] = -s_beta[b] * (acc0 + acc1);
        }
#pragma unroll
        for (int c = 0; c < LQ16; ++c) s_T[a][c] = Tr[c];
    }
    // V: my slab of every row; a row whose pivot was dropped (beta = 0: no reflector) is stored as zeros, so that the
    // block reflector's Gram matrix sees none
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 2 * rp + h;
        const bool keep = r < nb && s_beta[min(r, LQ16 - 1)] != 0.0;
        double* vout = V + (long)r * ldv + c0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = 4 * (64 * e + slot);
            if (j + 3 < L) {
                dbl4 q;
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = keep ? x[h][e][i] : 0.0;
                *(dbl4*)(vout + j) = q;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j + i < L) vout[j + i] = keep ? x[h][e][i] : 0.0;
            }
        }
    }
    __syncthreads();
    if (first) {
        for (int e = tid; e < LQ16 * LQ16; e += P16_THREADS) {
            const int a = e / LQ16, b = e % LQ16;
            panel->T[a][b] = s_T[a][b];
            if (a < nb && b < a) Tc[(long)(k + a) * ld + k + b] = s_lower[a][b];
            if (a < nb && b == 0) diagL[k + a] = s_diag[a];
        }
        if (tid == 0) {
            panel->nb = nb;
            panel->pad = 0;
            dmaxbuf[0] = dmax;
        }
    }
}

// M = T^-1 of a block of nb reflectors from their Gram matrix S = V V' (nb x nb, leading dimension LQW_BLOCK; both
// triangles hold the products): strictly upper triangle as it is, 1 / beta_i = |v_i|^2 / 2 on the diagonal; a row of
// zeros (dropped pivot) gets 1 there - its coefficients are zero either way
__global__ __launch_bounds__(256) void k_wy_make_m(double* __restrict__ S, int nb) {
    for (int e = threadIdx.x; e < LQW_BLOCK * LQW_BLOCK; e += blockDim.x) {
        const int i = e / LQW_BLOCK, j = e % LQW_BLOCK;
        double v = S[(long)i * LQW_BLOCK + j];
        if (i >= nb || j >= nb || j < i) v = 0.0;
        if (i == j) v = (i < nb && v > 0.0) ? 0.5 * v : 1.0;
        S[(long)i * LQW_BLOCK + j] = v;              // (every thread rewrites only what it read)
    }
}

// T = M^-1 (upper triangular, nb x nb inside a LQW_BLOCK x LQW_BLOCK array, row-major) - the block reflector's T factor:
// thread j solves M t_j = e_j by back substitution (column j of the inverse of an upper triangle has nothing below
// row j); entries outside the nb x nb corner are zero.  One workgroup of LQW_BLOCK threads, M out of LDS.
__global__ __launch_bounds__(LQW_BLOCK) void k_wy_invert(const double* __restrict__ M, int nb, double* __restrict__ T) {
    __shared__ double s_m[LQW_BLOCK][LQW_BLOCK + 1];
    const int j = threadIdx.x;
    for (int e = j; e < LQW_BLOCK * LQW_BLOCK; e += LQW_BLOCK) s_m[e / LQW_BLOCK][e % LQW_BLOCK] = M[e];
    __syncthreads();
    // column j in registers, every loop unrolled: the reads of M are broadcasts with static addresses that go out ahead
    // of the multiply-adds; entries below row j stay zero, so the sums need no bound on c (51 -> 6 us per block against
    // the loop that kept the column in LDS)
    double t[LQW_BLOCK];
#pragma unroll
    for (int i = LQW_BLOCK - 1; i >= 0; --i) {
        double acc0 = (i == j) ? 1.0 : 0.0, acc1 = 0.0;
#pragma unroll
        for (int c = i + 1; c < LQW_BLOCK; ++c) {
            if (c & 1) acc1 = fma(-s_m[i][c], t[c], acc1);
            else acc0 = fma(-s_m[i][c], t[c], acc0);
        }
        t[i] = (i <= j && j < nb) ? (acc0 + acc1) / s_m[i][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < LQW_BLOCK; ++i) T[(long)i * LQW_BLOCK + j] = t[i];
}

// W (rows x LQW_BLOCK, row-major) = A V' for `rows` rows of length L at A (leading dimension ld) and the nb <= 64
// reflector vectors V (row-major, leading dimension ldv) over the same columns - the tall-skinny product of the block
// reflector's application, hand-written because the library runs it at 0.7 TB/s (12 TF/s; tools/gemm_shapes.py) where
// it is a plain stream of A: a workgroup = 2 wavefronts x 16 rows walks the columns in blocks of 16; the block of V
// (64 x 16) goes through LDS once per workgroup (double-buffered, one barrier per block) and is the A operand of 16
// v_mfma_f64_16x16x4 per wavefront and block, the rows are the B operand straight from their 32-byte loads (lane
// (n, g) owns the columns 16 b + 4 g + i of row n - K slot g of MFMA i, as in k_lq_apply16); the accumulator layout
// (register i of lane (n, g) = reflector 16 q + 4 i + g, row n) is written out as it is.
constexpr int WYW_WAVES = 2;
// blockIdx.y = a slice of the column blocks (kb_per blocks each): few rows (the later panels of a block, the Gram matrix
// of the block's own vectors) are spread over the chip by columns instead; slice y writes its partial products to
// W + y * rows * LQW_BLOCK and k_wy_sum adds the slices up in order.
constexpr int WYW_SPLIT_MAX = 128;
__global__ __launch_bounds__(64 * WYW_WAVES) void k_wy_w(const double* __restrict__ A, int ld, int rows, int L,
                                                        const double* __restrict__ V, int ldv, int nb,
                                                        double* __restrict__ W, int kb_per) {
    __shared__ __attribute__((aligned(32))) double s_v[2][LQW_BLOCK][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int r = (int)blockIdx.x * 16 * WYW_WAVES + 16 * wv + n;
    const double* row = A + (long)min(r, rows - 1) * ld;
    const int nblk_all = (L + 15) / 16;
    const int b_lo = (int)blockIdx.y * kb_per, nblk = min(nblk_all, b_lo + kb_per);
    W += (long)blockIdx.y * rows * LQW_BLOCK;
    // V tile: thread t brings reflector rows t / 4 and 32 + t / 4, columns 4 (t % 4) .. + 3 of the block
    const int vr = tid >> 2, vc = 4 * (tid & 3);
    const double* v0 = V + (long)vr * ldv + vc;
    const double* v1 = V + (long)(vr + 32) * ldv + vc;
    const bool on0 = vr < nb, on1 = vr + 32 < nb;
    auto load_v = [&](int b, dbl4& a0, dbl4& a1) {
        const int j = 16 * b + vc;
        const dbl4 z = {0.0, 0.0, 0.0, 0.0};
        a0 = on0 ? *(const dbl4*)(v0 + 16 * b) : z;
        a1 = on1 ? *(const dbl4*)(v1 + 16 * b) : z;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (j + i >= L) a0[i] = a1[i] = 0.0;
    };
    auto load_x = [&](int b) {
        dbl4 x = *(const dbl4*)(row + min(16 * b + 4 * g, 4 * ((L - 1) / 4)));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (16 * b + 4 * g + i >= L) x[i] = 0.0;
        return x;
    };
    d4 acc[4] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    dbl4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, x = a0;
    if (b_lo < nblk) {
        load_v(b_lo, a0, a1);
        x = load_x(b_lo);
    }
    *(dbl4*)&s_v[b_lo & 1][vr][vc] = a0;
    *(dbl4*)&s_v[b_lo & 1][vr + 32][vc] = a1;
    __syncthreads();
    for (int b = b_lo; b < nblk; ++b) {
        dbl4 xn = x;
        if (b + 1 < nblk) {
            load_v(b + 1, a0, a1);
            xn = load_x(b + 1);
        }
        const double (*vt)[16] = s_v[b & 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const dbl4 vq = *(const dbl4*)&vt[16 * q + n][4 * g];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vq[i], x[i], acc[q], 0, 0, 0);
        }
        if (b + 1 < nblk) {
            *(dbl4*)&s_v[(b + 1) & 1][vr][vc] = a0;
            *(dbl4*)&s_v[(b + 1) & 1][vr + 32][vc] = a1;
        }
        x = xn;
        __syncthreads();
    }
    if (r < rows) {
        double* out = W + (long)r * LQW_BLOCK;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[16 * q + 4 * i + g] = acc[q][i];
    }
}

// A <- A - (W T) V  (HAS_T: the block reflector, W = A V' from k_wy_w, T = M^-1 from k_wy_invert, 64 reflectors)  or
// A <- A - W2 V      (the 16 reflectors of one panel on the later rows of its block, W2 from k_wy_small_finish)
// for `rows` rows of length L at A - hand-written for the FP64 matrix cores (round 5; rounds 3-4 posed both products as
// library GEMMs and resolved rocblas_dgemm at run time).  A workgroup = 4 wavefronts x 16 rows walks its slice of the
// columns (blockIdx.y, kb_per blocks of 16) once: everything is in the transposed form of k_lq_apply16 -
//   coefficients  Z' = T' W'   A operand T[4 s + g][16 q + n], B operand W[row n][4 s + g]; the accumulator (register i
//                              of lane (n, g) = reflector 16 q + 4 i + g, row n) IS the B operand of the update
//   update        X' -= V' Z'  per 16-column block: A operand V[16 q + 4 i + g][16 b + pm(n)] out of the block's tile
//                              of V in LDS (64 x 16, double-buffered, one barrier per block; pm(m) = 4 (m & 3) + (m >> 2)
//                              permutes the columns so that register i of lane (n, g) is column 16 b + 4 g + i of row n -
//                              the 32 bytes the lane loaded and stores), 16 v_mfma_f64_16x16x4 per wavefront and block
// - so A is read once and written once per block reflector (2 KB in, 2 KB out per 16 MFMAs: the matrix cores are half
// busy at the HBM rate), V comes from the L2 once per 64 rows.  Every row's result depends on its own data only: the
// bits do not depend on the grid, the slices or the stream the launch runs on.
constexpr int WYU_WAVES = 4;
template <int NBQ, bool HAS_T>
__global__ __launch_bounds__(64 * WYU_WAVES) void k_wy_update(double* __restrict__ A, int ld, int rows, int L,
                                                             const double* __restrict__ V, int ldv, int nb,
                                                             const double* __restrict__ W, int ldc,
                                                             const double* __restrict__ T, int kb_per) {
    constexpr int NBV = 16 * NBQ;
    __shared__ __attribute__((aligned(32))) double s_v[2][NBV][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int r = (int)blockIdx.x * 16 * WYU_WAVES + 16 * wv + n;
    const bool valid = r < rows;
    const int rc = min(r, rows - 1);
    double* row = A + (long)rc * ld;
    const int nblk_all = (L + 15) / 16;
    const int b_lo = (int)blockIdx.y * kb_per, b_hi = min(nblk_all, b_lo + kb_per);
    if (b_lo >= b_hi) return;                                   // (uniform over the workgroup)
    // ---- coefficients
    d4 z[NBQ];
    const double* crow = W + (long)rc * ldc;
    if (HAS_T) {
#pragma unroll
        for (int q = 0; q < NBQ; ++q) z[q] = d4{0.0, 0.0, 0.0, 0.0};
        double wk[4 * NBQ];
#pragma unroll
        for (int st = 0; st < 4 * NBQ; ++st) wk[st] = 4 * st + g < nb ? crow[4 * st + g] : 0.0;
#pragma unroll
        for (int st = 0; st < 4 * NBQ; ++st) {
            const double* trow = T + (long)(4 * st + g) * LQW_BLOCK + n;
#pragma unroll
            for (int q = 0; q < NBQ; ++q)
                if (16 * q + 15 >= 4 * st)                      // (T is upper triangular: blocks left of the diagonal are zero)
                    z[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(trow[16 * q], wk[st], z[q], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < NBQ; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) z[q][i] = 16 * q + 4 * i + g < nb ? crow[16 * q + 4 * i + g] : 0.0;
    }
    // ---- the walk over my slice of the columns
    const int pm = 4 * (n & 3) + (n >> 2);
    const int vr = tid >> 2, vc = 4 * (tid & 3);               // my dbl4 of the V tile (NBV * 4 of them, one per thread at most)
    const bool loads_v = tid < NBV * 4;
    const double* vsrc = V + (long)min(vr, NBV - 1) * ldv + vc;
    const bool v_on = loads_v && vr < nb;
    auto load_v = [&](int b) {
        dbl4 a = {0.0, 0.0, 0.0, 0.0};
        if (v_on) {
            a = *(const dbl4*)(vsrc + 16 * b);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * b + vc + i >= L) a[i] = 0.0;
        }
        return a;
    };
    auto load_x = [&](int b) { return *(const dbl4*)(row + min(16 * b + 4 * g, 4 * ((L - 1) / 4))); };
    dbl4 x = load_x(b_lo);
    {
        const dbl4 a = load_v(b_lo);
        if (loads_v) *(dbl4*)&s_v[b_lo & 1][vr][vc] = a;
    }
    __syncthreads();
    for (int b = b_lo; b < b_hi; ++b) {
        dbl4 xn = x, an = {0.0, 0.0, 0.0, 0.0};
        if (b + 1 < b_hi) {
            an = load_v(b + 1);
            xn = load_x(b + 1);
        }
        const double (*vt)[16] = s_v[b & 1];
        d4 o0 = d4{0.0, 0.0, 0.0, 0.0}, o1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            double a4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = vt[16 * q + 4 * i + g][pm];
            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], z[q][0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], z[q][1], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], z[q][2], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], z[q][3], o1, 0, 0, 0);
        }
        const int j0 = 16 * b + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] -= o0[i] + o1[i];
        if (valid) {
            if (j0 + 3 < L) {
                *(dbl4*)(row + j0) = x;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j0 + i < L) row[j0 + i] = x[i];
            }
        }
        if (b + 1 < b_hi && loads_v) *(dbl4*)&s_v[(b + 1) & 1][vr][vc] = an;
        x = xn;
        __syncthreads();
    }
}

// The later rows of a block after one of its 16-reflector panels, in ONE launch (round 5; k_wy_w over column slices +
// k_wy_small_finish + the rank-16 update were three launches and ~75 us of the sweep's chain per panel, 195 times per
// C5 subproblem):  rows <- rows - ((rows V16') T16) V16  for `rows` <= 48 rows of length L.  Workgroup y of the grid owns
// the column slice [16 kb_per y, 16 kb_per (y + 1)): it loads its slice of the rows into registers and of the 16
// vectors into LDS ONCE, forms its share of the products (the MFMA chain of k_wy_w), posts it, and waits for the
// others' shares - the workgroups are the whole grid of the launch, at most 32, and wait only for each other (counter
// + bounded spin as in the look-ahead sweep: a wait that gives up raises the flag the host turns into a re-run with the
// separate launches); every workgroup then adds the shares up in the order of the slices, multiplies by the panel's T
// (the sums of k_wy_small_finish, term for term) and updates the slice it still holds (the MFMA chain of
// k_wy_update<1, false>).  Same slices, same sums, same chains: the BITS of the three launches.
constexpr int WIB_WAVES = 3;                     // 48 rows
constexpr int WIB_KB = 16;                       // 16-column blocks per slice at most
__global__ __launch_bounds__(64 * WIB_WAVES) void k_wy_inblock(double* __restrict__ A, int ld, int rows, int L,
                                                              const double* __restrict__ V, int ldv, int nb,
                                                              const Lq16Panel* __restrict__ panel, double* part,
                                                              int kb_per, unsigned* counter, unsigned expect,
                                                              int* __restrict__ lost, int spin_limit) {
    constexpr int VS = 16 * WIB_KB + 16;
    constexpr int NT = 64 * WIB_WAVES;
    __shared__ __attribute__((aligned(32))) double s_v[LQ16][VS];
    __shared__ double s_w[16 * WIB_WAVES][LQ16 + 1];
    __shared__ double s_w2[16 * WIB_WAVES][LQ16 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int y = blockIdx.x, nsl = gridDim.x;
    const int nblk_all = (L + 15) / 16;
    const int b_lo = y * kb_per, nbk = min(nblk_all, b_lo + kb_per) - b_lo;      // (>= 1: the grid is ceil(nblk / kb_per))
    const int r = 16 * wv + n;
    const bool valid = r < rows;
    const int rc = min(r, rows - 1);
    double* row = A + (long)rc * ld;
    // ---- my slice: the rows into registers, the vectors into LDS
    dbl4 x[WIB_KB];
#pragma unroll
    for (int u = 0; u < WIB_KB; ++u) {
        x[u] = dbl4{0.0, 0.0, 0.0, 0.0};
        if (u < nbk) {
            const int j0 = 16 * (b_lo + u) + 4 * g;
            x[u] = *(const dbl4*)(row + min(j0, 4 * ((L - 1) / 4)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (j0 + i >= L) x[u][i] = 0.0;
        }
    }
    for (int e = tid; e < LQ16 * nbk * 4; e += NT) {
        const int vr = e / (nbk * 4), c4 = e % (nbk * 4);
        const int col = 16 * b_lo + 4 * c4;
        dbl4 a = {0.0, 0.0, 0.0, 0.0};
        if (vr < nb) {
            a = *(const dbl4*)(V + (long)vr * ldv + col);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (col + i >= L) a[i] = 0.0;
        }
        *(dbl4*)&s_v[vr][4 * c4] = a;
    }
    __syncthreads();
    // ---- my share of W' = V X'
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < WIB_KB; ++u) {
        if (u < nbk) {
            const dbl4 vq = *(const dbl4*)&s_v[n][16 * u + 4 * g];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vq[i], x[u][i], acc, 0, 0, 0);
        }
    }
    if (valid) {
        double* mine = part + ((long)y * (16 * WIB_WAVES) + r) * LQ16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __hip_atomic_store(mine + 4 * i + g, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lq_wait_for(counter, expect, lost, spin_limit);
    }
    __syncthreads();
    // ---- everybody's shares, in the order of the slices; then W2 = W1 T16 (upper triangular)
    for (int e = tid; e < rows * LQ16; e += NT) {
        const int rr = e / LQ16, j = e % LQ16;
        double sum = 0.0;
        for (int y0 = 0; y0 < nsl; y0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = y0 + u < nsl ? __hip_atomic_load(part + ((long)(y0 + u) * (16 * WIB_WAVES) + rr) * LQ16 + j,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (y0 + u < nsl) sum += v[u];
        }
        s_w[rr][j] = sum;
    }
    __syncthreads();
    for (int e = tid; e < rows * LQ16; e += NT) {
        const int rr = e / LQ16, j = e % LQ16;
        double w2 = 0.0;
        for (int i = 0; i <= j; ++i) w2 = fma(s_w[rr][i], panel->T[i][j], w2);
        s_w2[rr][j] = w2;
    }
    __syncthreads();
    d4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = 4 * i + g < nb ? s_w2[rc][4 * i + g] : 0.0;
    // ---- X' -= V' Z' on the slice I hold
    const int pm = 4 * (n & 3) + (n >> 2);
#pragma unroll
    for (int u = 0; u < WIB_KB; ++u) {
        if (u < nbk) {
            double a4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = s_v[4 * i + g][16 * u + pm];
            d4 o0 = d4{0.0, 0.0, 0.0, 0.0}, o1 = d4{0.0, 0.0, 0.0, 0.0};
            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], z[0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], z[1], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], z[2], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], z[3], o1, 0, 0, 0);
            const int j0 = 16 * (b_lo + u) + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[u][i] -= o0[i] + o1[i];
            if (valid) {
                if (j0 + 3 < L) {
                    *(dbl4*)(row + j0) = x[u];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (j0 + i < L) row[j0 + i] = x[u][i];
                }
            }
        }
    }
}

// out[e] = sum over the slices y < nsplit of part[y * count + e], in order (count = rows * LQW_BLOCK entries)
__global__ __launch_bounds__(256) void k_wy_sum(const double* __restrict__ part, int nsplit, long count,
                                                double* __restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double acc = 0.0;
    for (int y0 = 0; y0 < nsplit; y0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = y0 + u < nsplit ? part[(long)(y0 + u) * count + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (y0 + u < nsplit) acc += v[u];
    }
    out[e] = acc;
}

// The later rows of a block after one of its 16-reflector panels: W1 = sum of the slices (rows x LQW_BLOCK, the first 16
// entries of a row are its products with the panel's vectors), W2 = W1 T16 (T of the panel kernel, upper triangular):
// out (rows x 16, row-major) for the GEMM that subtracts W2 V16.  A thread per (row, reflector).
__global__ __launch_bounds__(1024) void k_wy_small_finish(const double* __restrict__ part, int nsplit, int rows,
                                                          const Lq16Panel* __restrict__ panel, double* __restrict__ out) {
    __shared__ double s_w[64][LQ16 + 1];
    __shared__ double s_t[LQ16][LQ16 + 1];
    const int tid = threadIdx.x, r = tid >> 4, j = tid & 15;          // (rows <= 48: ONE pass of 64 x 16 threads)
    // (round 5: T comes in with the slices' first trip; read from global inside the 16-term sum below it was 16 dependent
    // loads per thread, a third of this kernel's 26 us on the sweep's chain)
    if (tid < LQ16 * LQ16) s_t[tid >> 4][tid & 15] = panel->T[tid >> 4][tid & 15];
    for (int r0 = 0; r0 < rows; r0 += 64) {
        const int rr = r0 + r;
        double acc = 0.0;
        if (rr < rows) {
            // (the kernel is as long as its round trips to the slices: their loads go out sixteen at a time; the
            // additions stay in order)
            for (int y0 = 0; y0 < nsplit; y0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    v[u] = y0 + u < nsplit ? part[((long)(y0 + u) * rows + rr) * LQW_BLOCK + j] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (y0 + u < nsplit) acc += v[u];
            }
        }
        s_w[r][j] = acc;
        __syncthreads();
        if (rr < rows) {
            double w2 = 0.0;
            for (int i = 0; i <= j; ++i) w2 = fma(s_w[r][i], s_t[i][j], w2);
            out[(long)rr * LQ16 + j] = w2;
        }
        __syncthreads();
    }
}
