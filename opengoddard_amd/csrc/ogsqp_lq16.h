// ogsqp_lq16.h - the LQ sweep in panels of 16 reflectors: a panel kernel whose chain per reflector is one row's work
// in one wavefront, and an MFMA trailing update (included by ogsqp.hip inside its anonymous namespace; DESIGN.md
// section 8).
//
// The sweep of 8-reflector panels (k_lq_panel / k_lq_apply_reg) is bound by its panel kernel: every thread holds
// all 8 rows of its columns, so each reflector costs 8 workgroup-wide reductions (25-29 us per panel, one
// workgroup, the chip idle), and the trailing update streams the whole matrix once per 8 reflectors (17-25 us).
//
//   k_lq_panel16   16 rows x L columns, 512 threads.  Rows of up to 1536 entries: a row PAIR per wavefront, no barrier
//                  between reflectors (lq_panel16_waves: 1.25 us per reflector).  Longer rows: eight lanes per row
//                  pair and wavefront, two barriers per reflector (lq_panel16_lanes: 2.1 us).  Rows that are already
//                  reflector vectors take part like the others: their products with the new vector are the
//                  off-diagonal entries of V V' that the T factor needs, for free.
//   k_lq_apply16   row <- row - ((row V') T) V on the FP64 matrix cores, 16 rows per workgroup, the columns split
//                  over its four wavefronts, the row segments held in registers between the two products (one read
//                  and one write of the matrix per 16 reflectors).  All three products run in the transposed form
//                  (reflector index or column index on the M side, matrix row on the N side): the accumulator
//                  layout of one product is then the B operand of the next without a shuffle, given that the K index
//                  of an MFMA may be permuted freely as long as both operands agree on it.
//
// Numerically this is the same Householder sweep as before (and as the restatement's loop), sums in another order.

constexpr int LQ16 = 16;            // reflectors per panel
constexpr int P16_THREADS = 512;
constexpr int P16_WAVES = P16_THREADS / 64;
constexpr int P16_SLOTS = 8 * P16_WAVES;        // column slots per row pair: slot s owns the columns 4 (64 e + s) + i

typedef double dbl4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));

struct Lq16Panel {
    double T[LQ16][LQ16];
    int nb;
    int pad;
    long long tr[8];     // -DOGSQP_TRACE: s_memtime ticks per section of the last panel kernel (wavefront 0)
};

// sum over the eight lanes that share a pair of rows in one wavefront, in every one of them (the column-split panel
// of ogsqp_lqwide.h keeps its rows that way)
__device__ __forceinline__ double oct_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror: the other quad of the eight
    return v;
}

// sqrt(d) to rounding for a normal positive d, without the range and special-case handling of sqrt(): one reciprocal
// square root and five dependent multiply-adds where sqrt() is some twelve operations and the division behind it
// another ten (the panel's chain pays 13 ns for each).  d = 0 gives NaN, which every comparison below turns into
// "no reflector".
__device__ __forceinline__ double fast_sqrt(const double d) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, d);
    return fma(e, h, g);
}
// 1 / s to rounding for a normal positive s (two Newton steps on the hardware's estimate)
__device__ __forceinline__ double fast_inverse(const double s) {
    double y = __builtin_amdgcn_rcp(s);
    double e = fma(-s, y, 1.0);
    y = fma(y, e, y);
    e = fma(-s, y, 1.0);
    return fma(y, e, y);
}

constexpr int P16_RING = 4;     // reflector vectors in LDS at a time
constexpr int P16_SPINS = 1 << 22;

// E: groups of 256 columns (rows of up to 256 E entries from the panel's first column on).  Wavefront w holds the rows
// w and w + 8, lane l the columns 4 (64 e + l) + i of both: a row's products and its norm are sums over ONE wavefront
// (DPP, no exchange through LDS), the reflector vector is read from LDS once per lane and step and serves both rows.
// NO barrier inside the sweep: the owner of row b + 1 takes that row first - product with v_b, update, then at once
// the entries left of the new pivot aside, the norm, the reflector's scalars, vector and scalars into the next slot of
// a ring of LDS buffers, and a flag (LDS operations of one wavefront are performed in order) - and only then its second
// row; the others poll the flag.  What the chain of a step consists of is then ONE row's work and the scalars
// (measured with a barrier per step: the owner's second row and the wait for the slowest of eight wavefronts were on
// it too, 2.0 us per reflector; the two-rows-per-lane-group form before that: 2.1).  A slot of the ring is written
// again four reflectors later, after every wavefront has counted itself done with reading it.  Rows that are already
// reflector vectors still form their product with the new vector: the off-diagonal entries of V V' that the T factor
// needs.
// COHERENT: the rows were written earlier in the SAME launch by other workgroups (k_lq_step16) with agent-scope
// (write-through) stores and the caller has seen their count: an agent-scope acquire fence drops whatever this
// compute unit and its XCD's L2 may still hold of them, then they are read like any rows (48 eight-byte
// agent-scope loads per lane instead took 4 us where the plain 32-byte loads take 1.3).
template <int E, bool COHERENT = false>
__device__ __forceinline__ void lq_panel16_waves(double* __restrict__ Tc, int ld, int mrows, int nq, int k,
                                                double* __restrict__ V, int ldv, double* __restrict__ diagL,
                                                Lq16Panel* __restrict__ panel, double* __restrict__ dmaxbuf,
                                                double* __restrict__ lds) {
    __shared__ double s_sc[P16_RING][2];             // by slot: beta, the largest pivot so far
    __shared__ double s_lower[LQ16][LQ16];           // finished entries of L inside the panel
    __shared__ double s_S[LQ16][LQ16];               // v_a . v_b, a < b
    __shared__ double s_beta[LQ16], s_diag[LQ16];
    // (one block of 128 bytes: the dynamic LDS behind the static arrays must stay 16-byte aligned - a ds_read_b128
    // at an address that is not takes the slow path, 7x on the vector reads when two ints were declared here)
    __shared__ __attribute__((aligned(32))) int s_ctl[32];
    int& s_ready = s_ctl[0];                         // reflectors published so far
    int& s_broken = s_ctl[1];                        // a wait among the wavefronts of this workgroup gave up (it cannot,
                                                     // they are all resident: a defect) - the diagonal goes out as NaN
    int* const s_done = s_ctl + 16;                  // wavefronts that have read reflector b
    double* vrow = lds;                              // P16_RING x (256 E)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // (a scalar: what depends on it branches, not masks)
    const int nb = min(LQ16, mrows - k), L = nq - k;
    const int W = 256 * E;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memtime();
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define P16MARK(slot_) do { const long long now_ = __builtin_amdgcn_s_memtime(); t_sec[slot_] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define P16MARK(slot_) do { } while (0)
#endif
    double x[2][E][4];
    if (COHERENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = wv + 8 * h;
        const double* row = Tc + (long)(k + min(r, nb - 1)) * ld + k;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            // one 32-byte load (rows are 128-byte aligned from column k on; no branch: the loads go out together)
            const int j = 4 * (64 * e + lane);
            const double* src = row + min(j, 4 * ((L - 1) / 4));
            const dbl4 v = *(const dbl4*)src;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[h][e][i] = (r < nb && j + i < L) ? v[i] : 0.0;
        }
    }
    for (int e = tid; e < LQ16 * LQ16; e += P16_THREADS) {
        (&s_S[0][0])[e] = 0.0;
        (&s_lower[0][0])[e] = 0.0;
    }
    if (tid < LQ16) s_done[tid] = 0;
    if (tid == 0) {
        s_ready = 0;
        s_broken = 0;
    }
    double dmax = dmaxbuf[0];
    __syncthreads();
    P16MARK(0);   // load
    // Row b (held in xr by this wavefront, every earlier reflector applied) becomes reflector b: entries left of the
    // pivot are finished entries of L (kept aside, zero in the vector).  The row goes out before its norm is known;
    // the pivot entry, the scalars and the flag follow.  With sigma = |row|, s = sigma + |x0|: v0 = x0 - alpha =
    // sign(x0) s, v'v = 2 sigma s, beta = 2 / v'v = 1 / (|row|^2 + sigma |x0|).
    auto make_reflector = [&](const int b, double (&xr)[E][4], const double dmax_in, const int done_early) {
        double* vr = vrow + (b % P16_RING) * W;
        if (b >= P16_RING && done_early < P16_WAVES) {     // (never seen waiting: the slot's readers are three reflectors back)
            int spins = 0;
            while (__hip_atomic_load(&s_done[b - P16_RING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < P16_WAVES)
                if (++spins > P16_SPINS) { s_broken = 1; break; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 4 * lane + i;
            const bool left = j < b;                 // (only the first group of four columns can lie left: b < 16)
            if (left) s_lower[b][j] = xr[0][i];
            xr[0][i] = left ? 0.0 : xr[0][i];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            // (a lane's 32 bytes as two halves 1 KB apart: 16-byte accesses at a 16-byte stride have no bank conflicts,
            // at a 32-byte stride they take twice as long - and the LDS is what eight wavefronts share)
            *(dbl2*)(vr + 256 * e + 2 * lane) = dbl2{xr[e][0], xr[e][1]};
            *(dbl2*)(vr + 256 * e + 128 + 2 * lane) = dbl2{xr[e][2], xr[e][3]};
        }
        // eight partial sums: a dependent f64 multiply-add is 13 ns on this part, an independent one 3
        double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) p[4 * (e & 1) + i] = fma(xr[e][i], xr[e][i], p[4 * (e & 1) + i]);
        const double Db = wave_sum(((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        const int bi = b & 3;
        const double xsel = bi == 0 ? xr[0][0] : bi == 1 ? xr[0][1] : bi == 2 ? xr[0][2] : xr[0][3];
        const double x0 = lane_f64(xsel, b >> 2);
        const double sigma = fast_sqrt(Db);
        // what is left of a row that depends on the earlier ones is rounding noise: no reflector is built from
        // it (it would rotate the null-space basis by that noise); its pivot is recorded as exactly 0
        const bool live = sigma > REDUNDANT * dmax_in && sigma > 0.0;
        const double s = sigma + fabs(x0);
        const double alpha = !live ? 0.0 : (x0 >= 0.0 ? -sigma : sigma);
        const double v0 = !live ? x0 : (x0 >= 0.0 ? s : -s);
        const double bt = live ? fast_inverse(fma(sigma, fabs(x0), Db)) : 0.0;
        if (lane == (b >> 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[0][i] = i == bi ? v0 : xr[0][i];      // row b becomes its reflector vector
            vr[2 * (b >> 2) + (b & 1) + 128 * ((b >> 1) & 1)] = v0;      // column b = lane b / 4, entry b % 4 of group 0
        }
        if (lane == 0) {
            s_sc[b % P16_RING][0] = bt;
            s_sc[b % P16_RING][1] = fmax(dmax_in, sigma);
            s_beta[b] = bt;
            s_diag[b] = alpha;
        }
        // (fences over LDS only: a release over every address space would wait for this wavefront's stores of V rows)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __hip_atomic_store(&s_ready, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // one row of this wavefront in step b (v = reflector b on my columns): its product with v, and what follows
    // from where the row stands - a finished reflector a < b: v_a . v_b; a row below: the update, and for row b + 1
    // (NEXT: it can only be the wavefront's row in line) its turn as the next reflector
    auto row_step = [&](const int b, const int r, double (&xr)[E][4], const double (&v)[E][4], const double bt,
                        const double dmax_now, const bool next, const int done_early) {
        if (r == b || r >= nb) return;               // (uniform per wavefront)
        double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) p[4 * (e & 1) + i] = fma(xr[e][i], v[e][i], p[4 * (e & 1) + i]);
        const double D = wave_sum(((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        if (r < b) {
            if (lane == 0) s_S[r][b] = D;
            return;
        }
        const double f = bt * D;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[e][i] = fma(-f, v[e][i], xr[e][i]);
        if (next && r == b + 1) make_reflector(b + 1, xr, dmax_now, done_early);
    };
    // a row of V goes out when it is final - right after it became a reflector vector, from the wavefront that holds
    // it, behind that wavefront's other row (nobody waits for these stores before the kernel ends)
    auto store_row = [&](const int r, const double (&xr)[E][4], const bool zero) {
        double* vout = V + (long)r * ldv;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = 4 * (64 * e + lane);
            if (j + 3 < L) {
                *(dbl4*)(vout + j) = zero ? dbl4{0.0, 0.0, 0.0, 0.0} : dbl4{xr[e][0], xr[e][1], xr[e][2], xr[e][3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j + i < L) vout[j + i] = zero ? 0.0 : xr[e][i];
            }
        }
    };
    if (wv >= nb) store_row(wv, x[0], true);                 // (rows beyond nb: zero)
    if (wv + 8 >= nb) store_row(wv + 8, x[1], true);
    if (wv == 0) make_reflector(0, x[0], dmax, P16_WAVES);
    P16MARK(5);   // the first reflector
    // x[0] is the wavefront's row in line for the next reflector: row wv while b + 1 < 8, row wv + 8 from then on
    int r_line = wv, r_other = wv + 8;
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        if (b == 7) {
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double t = x[0][e][i];
                    x[0][e][i] = x[1][e][i];
                    x[1][e][i] = t;
                }
            r_line = wv + 8;
            r_other = wv;
        }
        {
            int spins = 0;
            while (__hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= b) {
                if (++spins > P16_SPINS) { s_broken = 1; break; }
                if (r_line != b + 1) __builtin_amdgcn_s_sleep(1);     // (the wavefront next in line polls at once)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        P16MARK(1);   // wait for reflector b
        const double* vr = vrow + (b % P16_RING) * W;
        double v[E][4];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const dbl2 q0 = *(const dbl2*)(vr + 256 * e + 2 * lane), q1 = *(const dbl2*)(vr + 256 * e + 128 + 2 * lane);
            v[e][0] = q0[0];
            v[e][1] = q0[1];
            v[e][2] = q1[0];
            v[e][3] = q1[1];
        }
        const double bt = s_sc[b % P16_RING][0];
        dmax = s_sc[b % P16_RING][1];
        // (the count the next reflector's slot must have reached, asked for with the vector: it is there when needed)
        const int done_early = b + 1 >= P16_RING ? __hip_atomic_load(&s_done[b + 1 - P16_RING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : P16_WAVES;
        P16MARK(2);   // vector and scalars
        row_step(b, r_line, x[0], v, bt, dmax, true, done_early);
        P16MARK(3);   // row in line
        // done with reading slot b (here, not behind the reads: counting waits for all of them, and the products
        // start on the first)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) __hip_atomic_fetch_add(&s_done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        row_step(b, r_other, x[1], v, bt, dmax, false, done_early);
        if (r_line == b + 1 && r_line < nb) store_row(r_line, x[0], false);
        if (b == 0 && wv == 0) store_row(0, x[0], false);      // (not in front of the loop: the wait for this wavefront's
                                                               // second row, still on its way then, would include the stores)
        P16MARK(4);   // other row
    }
    __syncthreads();
    // T in wavefront 0 (the rows of V went out as they became final), the finished entries of L and the diagonal from
    // the other wavefronts.
    // T by forward accumulation; row a of T depends only on itself: thread a does row a, the row in registers and
    // the loops unrolled (entries left of the diagonal are zero, so no bound on c is needed): the LDS reads of
    // V V' have static addresses and go out ahead of the dependent multiply-adds
    if (tid < LQ16) {
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
        for (int c = 0; c < LQ16; c += 4) *(dbl4*)&panel->T[a][c] = dbl4{Tr[c], Tr[c + 1], Tr[c + 2], Tr[c + 3]};
    } else if (tid >= 64 && tid < 64 + LQ16 * LQ16) {
        const int e = tid - 64, a = e / LQ16, b = e % LQ16;
        if (a < nb && b < a) Tc[(long)(k + a) * ld + k + b] = s_lower[a][b];
        if (a < nb && b == 0) diagL[k + a] = s_broken ? __builtin_nan("") : s_diag[a];
    }
    if (tid == 0) {
        panel->nb = nb;
        panel->pad = 0;
        dmaxbuf[0] = dmax;
    }
    P16MARK(6);   // store V, T
#ifdef OGSQP_TRACE
    if (tid == 0)
        for (int e = 0; e < 8; ++e) panel->tr[e] = t_sec[e];
#endif
#undef P16MARK
}

// The panel with the ROWS spread over the lanes (the form every panel had before the one above): lane l of every
// wavefront belongs to the rows 2 (l / 8) and 2 (l / 8) + 1, eight lanes per row pair and wavefront, 64 column slots
// per pair in all; two barriers per reflector, the products of all rows with the reflector vector through LDS.  Kept for
// rows beyond 1536 entries (E = 7, 8), where a wavefront of the form above would hold 2 x 32 doubles of rows and 32 of
// the vector per lane and spill (measured: 107 us per launch against 58).
// E: groups of 4 columns per lane (rows of up to 256 E entries from the panel's first column on).  Lane l of every
// wavefront holds the rows 2 (l / 8) and 2 (l / 8) + 1 on its columns: the reflector vector is read from LDS once
// per lane and step and serves both, in the products and in the update.
// COHERENT: the rows were written earlier in the SAME launch by other workgroups (k_lq_step16) with agent-scope
// (write-through) stores and the caller has seen their count: an agent-scope acquire fence drops whatever this
// compute unit and its XCD's L2 may still hold of them, then they are read like any rows (48 eight-byte
// agent-scope loads per lane instead took 4 us where the plain 32-byte loads take 1.3).
template <int E, bool COHERENT = false>
__device__ __forceinline__ void lq_panel16_lanes(double* __restrict__ Tc, int ld, int mrows, int nq, int k,
                                                double* __restrict__ V, int ldv, double* __restrict__ diagL,
                                                Lq16Panel* __restrict__ panel, double* __restrict__ dmaxbuf,
                                                double* __restrict__ lds) {
    __shared__ double s_part[2][P16_WAVES][LQ16];    // partial dot products (double-buffered by step parity)
    __shared__ double s_xpc[2][LQ16];                // every row's entry in the pivot column
    __shared__ double s_lower[LQ16][LQ16];           // finished entries of L inside the panel
    __shared__ double s_S[LQ16][LQ16];               // v_a . v_b, a < b
    __shared__ double s_T[LQ16][LQ16];
    __shared__ double s_beta[LQ16], s_diag[LQ16];
    double* vrow = lds;                              // 2 x (256 E): the current row (reflector vector before its pivot is set)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = lane >> 3, slot = wv * 8 + (lane & 7);
    const int nb = min(LQ16, mrows - k), L = nq - k;
    const int W = 256 * E;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memtime();
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define P16MARK(slot_) do { const long long now_ = __builtin_amdgcn_s_memtime(); t_sec[slot_] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define P16MARK(slot_) do { } while (0)
#endif
    double x[2][E][4];
    if (COHERENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 2 * rp + h;
        const double* row = Tc + (long)(k + min(r, nb - 1)) * ld + k;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            // one 32-byte load (rows are 128-byte aligned from column k on; no branch: the loads go out together)
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
    P16MARK(0);   // load
#pragma unroll
    for (int b = 0; b < LQ16; ++b) {
        if (b < nb) {                                // (uniform)
            double* vr = vrow + (b & 1) * W;
            // row b's lanes publish it: entries left of the pivot are finished entries of L (kept aside, zero in
            // the vector); every row publishes its entry in the pivot column b = slot b / 4, register b % 4
            if (rp == (b >> 1)) {
                // (only the first group of four columns can lie left of the pivot: b < 16)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 4 * slot + i;
                    const bool left = j < b;
                    if (left) s_lower[b][j] = x[b & 1][0][i];
                    x[b & 1][0][i] = left ? 0.0 : x[b & 1][0][i];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    dbl4 q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i] = x[b & 1][e][i];
                    *(dbl4*)(vr + 4 * (64 * e + slot)) = q;
                }
            }
            if (slot == (b >> 2)) {
                s_xpc[b & 1][2 * rp] = x[0][0][b & 3];
                s_xpc[b & 1][2 * rp + 1] = x[1][0][b & 3];
            }
            P16MARK(1);   // publish
            __syncthreads();
            P16MARK(2);   // barrier 1
            // the vector on my columns, once; products of my two rows with it
            double v[E][4];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const dbl4 q = *(const dbl4*)(vr + 4 * (64 * e + slot));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[e][i] = q[i];
            }
            // four partial sums per row: a dependent f64 multiply-add is 13 ns on this part, an independent one 3
            double pa[4] = {0.0, 0.0, 0.0, 0.0}, pb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pa[i] = fma(x[0][e][i], v[e][i], pa[i]);
                    pb[i] = fma(x[1][e][i], v[e][i], pb[i]);
                }
            double acc0 = oct_sum((pa[0] + pa[1]) + (pa[2] + pa[3]));
            double acc1 = oct_sum((pb[0] + pb[1]) + (pb[2] + pb[3]));
            if ((lane & 7) == 0) {
                s_part[b & 1][wv][2 * rp] = acc0;
                s_part[b & 1][wv][2 * rp + 1] = acc1;
            }
            P16MARK(3);   // read vector, products, 8-lane sums
            __syncthreads();
            P16MARK(4);   // barrier 2
            double Db = 0.0, D0 = 0.0, D1 = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < P16_WAVES; ++w8) {
                Db += s_part[b & 1][w8][b];
                D0 += s_part[b & 1][w8][2 * rp];
                D1 += s_part[b & 1][w8][2 * rp + 1];
            }
            const double x0 = s_xpc[b & 1][b];
            const double xr0 = s_xpc[b & 1][2 * rp], xr1 = s_xpc[b & 1][2 * rp + 1];
            const double sigma = sqrt(Db);
            // what is left of a row that depends on the earlier ones is rounding noise: no reflector is built from
            // it (it would rotate the null-space basis by that noise); its pivot is recorded as exactly 0
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
            // rows below b: row -= f v_b with v_b = the published row, its pivot entry lowered by alpha
            const double f0 = (2 * rp > b && 2 * rp < nb) ? bt * rv0 : 0.0;
            const double f1 = (2 * rp + 1 > b && 2 * rp + 1 < nb) ? bt * rv1 : 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[0][e][i] = fma(-f0, v[e][i], x[0][e][i]);
                    x[1][e][i] = fma(-f1, v[e][i], x[1][e][i]);
                }
            if (slot == (b >> 2)) {
                x[0][0][b & 3] = fma(f0, alpha, x[0][0][b & 3]);
                x[1][0][b & 3] = fma(f1, alpha, x[1][0][b & 3]);
                if (rp == (b >> 1)) x[b & 1][0][b & 3] = v0;     // row b becomes its reflector vector
            }
            P16MARK(5);   // scalars + update
        }
    }
    // T first, in wavefront 0 (which wrote V V' and the betas itself: no barrier), while the other wavefronts send
    // their part of V on its way
    // T by forward accumulation; row a of T depends only on itself: thread a does row a, the row in registers and
    // the loops unrolled (entries left of the diagonal are zero, so no bound on c is needed): the LDS reads of
    // V V' have static addresses and go out ahead of the dependent multiply-adds
    if (tid < LQ16) {
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
    // V (zero rows beyond nb), the finished entries of L, the diagonal
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 2 * rp + h;
        double* vout = V + (long)r * ldv;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = 4 * (64 * e + slot);
            if (j + 3 < L) {
                dbl4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = r < nb ? x[h][e][i] : 0.0;
                *(dbl4*)(vout + j) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j + i < L) vout[j + i] = r < nb ? x[h][e][i] : 0.0;
            }
        }
    }
    __syncthreads();
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
    P16MARK(6);   // store V, T
#ifdef OGSQP_TRACE
    if (tid == 0)
        for (int e = 0; e < 8; ++e) panel->tr[e] = t_sec[e];
#endif
#undef P16MARK
}

template <int E, bool COHERENT = false>
__device__ __forceinline__ void lq_panel16_body(double* __restrict__ Tc, int ld, int mrows, int nq, int k,
                                                double* __restrict__ V, int ldv, double* __restrict__ diagL,
                                                Lq16Panel* __restrict__ panel, double* __restrict__ dmaxbuf,
                                                double* __restrict__ lds) {
    if constexpr (E <= 6) lq_panel16_waves<E, COHERENT>(Tc, ld, mrows, nq, k, V, ldv, diagL, panel, dmaxbuf, lds);
    else lq_panel16_lanes<E, COHERENT>(Tc, ld, mrows, nq, k, V, ldv, diagL, panel, dmaxbuf, lds);
}

template <int E>
__global__ __launch_bounds__(P16_THREADS) void k_lq_panel16(double* __restrict__ Tc, int ld, int mrows, int nq, int k,
                                                           double* __restrict__ V, int ldv, double* __restrict__ diagL,
                                                           Lq16Panel* __restrict__ panel, double* __restrict__ dmaxbuf) {
    extern __shared__ __attribute__((aligned(32))) double lds[];
    lq_panel16_body<E>(Tc, ld, mrows, nq, k, V, ldv, diagL, panel, dmaxbuf, lds);
}

// U: blocks of 16 columns per wavefront (rows of up to 128 U entries from the panel's first column on); eight
// wavefronts per workgroup.  Rows are 128-byte aligned (leading dimension a multiple of 16, k a multiple of 16):
// lane (n, g) owns the four CONSECUTIVE columns 16 b + 4 g + i of row n in block b - one 32-byte load, and the
// sixteen lanes of a row group touch whole cache lines.  (An MFMA does not care which of its K slots - or M rows -
// stands for which column as long as both operands agree; the products below are indexed accordingly.)  Every
// global load of a phase is issued before the first result is used.
constexpr int A16_WAVES = 8;
#ifdef OGSQP_TRACE
__device__ long long g_apply16_trace[8];     // s_memrealtime ticks (10 ns) per section, one workgroup of the last launch
__device__ long long g_head16_trace[8];      // ... of head workgroup 0 of the last look-ahead launch
#define A16MARK(slot_) do { const long long now_ = __builtin_amdgcn_s_memrealtime(); t_sec[slot_] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define A16MARK(slot_) do { } while (0)
#endif
// HEAD (look-ahead, k_lq_step16): this workgroup is one of LQ_HEADS that share the 16 rows of the NEXT panel by
// COLUMNS (blocks uoff, uoff + ustride, ... per wavefront): their partial products W' meet in `wpart` (agent-scope
// stores and loads, a count in `cnt`), and the rows go out with agent-scope stores for the panel workgroup.
// LDS of the trailing update (declared once per kernel: the look-ahead kernel instantiates the body twice)
struct A16Shared {
    double w[A16_WAVES][64][4];                                     // partial W' of the wavefronts
    __attribute__((aligned(32))) double v[A16_WAVES][16][16];       // one 16 x 16 block of V per wavefront
};
struct Lq16Head {
    double* wpart;           // LQ_HEADS x 64 x 4
    unsigned* cnt;
    unsigned expect;
    int h;
    int* lost;               // set when the wait below gives up (the host then re-runs the subproblem with the
    int spin_limit;          // separate-launch forms, which wait for nothing)
};
// wait (one thread) until *word has reached `expect`; bounded - a workgroup that never comes would otherwise hang the
// device - and a wait that gives up says so
__device__ __forceinline__ void lq_wait_for(unsigned* word, unsigned expect, int* lost, int spin_limit) {
    int spins = 0;
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) {
        if (++spins > spin_limit) {    // (2^19 polls by default, about a second: a workgroup that comes this late is not coming)
            __hip_atomic_store(lost, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
#ifndef OGSQP_LQ_HEADS
#define OGSQP_LQ_HEADS 8
#endif
constexpr int LQ_HEADS = OGSQP_LQ_HEADS;
template <int U, bool HEAD = false>
__device__ __forceinline__ void lq_apply16_body(double* __restrict__ Tc, double* __restrict__ Jw, int ld, int mrows, int nq,
                                                int k, const double* __restrict__ V, int ldv,
                                                const Lq16Panel* __restrict__ panel, const int r0, const int rcount,
                                                A16Shared& sh, const int uoff = 0, const int ustride = 1,
                                                const Lq16Head head = Lq16Head()) {
    constexpr bool COHERENT = HEAD;
    double (&s_w)[A16_WAVES][64][4] = sh.w;
    double (&s_v)[A16_WAVES][16][16] = sh.v;
#define A16_BLOCK(u_) (wv + A16_WAVES * (uoff + ustride * (u_)))
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memrealtime();
    long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int nb = panel->nb, L = nq - k;
    A16MARK(0);   // panel header
    const int nblk = (L + 15) / 16;
    const int below = mrows - k - nb, nrows = below + nq;
    // rows r0 .. r0 + rcount - 1 (rcount <= 16: fewer rows per workgroup when 16 would leave compute units idle -
    // a workgroup streams at what ONE unit's miss queue sustains, ~26 GB/s); the other lanes repeat a row of the
    // group (same cache lines, nothing stored)
    const int r = r0 + n;
    const bool valid = n < rcount && r < nrows;
    const int rc = valid ? r : min(r0 + n % rcount, nrows - 1);
    double* row = (rc < below ? Tc + (long)(k + nb + rc) * ld : Jw + (long)(rc - below) * ld) + k;
    // my row's segment: blocks wv, wv + 8, ...
    dbl4 x[U], a[U];
    const double* vrow = V + (long)n * ldv;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = min(A16_BLOCK(u), nblk - 1);
        x[u] = *(const dbl4*)(row + 16 * b + 4 * g);
        a[u] = *(const dbl4*)(vrow + 16 * b + 4 * g);
    }
    const double t0 = panel->T[g][n], t1 = panel->T[4 + g][n], t2 = panel->T[8 + g][n], t3 = panel->T[12 + g][n];
    // W' = V X' (reflector 4 i + g in register i, matrix row n); K slot g of MFMA (u, i) is the column 16 b + 4 g + i
    // (four accumulators: a dependent MFMA is 27 ns, four chains run at the issue rate)
    d4 accs[4] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#ifdef OGSQP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A16MARK(1);   // loads of the row segments and of V
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j0 = 16 * A16_BLOCK(u) + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = j0 + i < L;
            x[u][i] = in ? x[u][i] : 0.0;
            accs[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? a[u][i] : 0.0, x[u][i], accs[i], 0, 0, 0);
        }
    }
    d4 acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (accs[0][i] + accs[1][i]) + (accs[2][i] + accs[3][i]);
    // the operands of the last product are V once more, reflector-major: M row m of block b is the column
    // 16 b + 4 (m & 3) + (m >> 2), so that the result lands in the layout the segment is held in.  They are NOT
    // read again (a second read of V was a third of this workgroup's traffic): the block this wavefront holds -
    // lane (n, g): V[n][16 b + 4 g + i] - goes through a 2 KB tile of LDS of its own and comes back as
    // V[4 t + g][16 b + pm]
    const int pm = 4 * (n & 3) + (n >> 2);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j0 = 16 * A16_BLOCK(u) + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = j0 + i < L ? a[u][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_w[wv][lane][i] = acc[i];
    A16MARK(2);   // first product
    __syncthreads();
    A16MARK(3);   // barrier
    double wt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double sum = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < A16_WAVES; ++w8) sum += s_w[w8][lane][i];
        wt[i] = sum;
    }
    if (HEAD) {
        // the other head workgroups' columns: partial sums through memory, in the order of the workgroups
        if (wv == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __hip_atomic_store(head.wpart + ((long)head.h * 64 + lane) * 4 + i, wt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(head.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lq_wait_for(head.cnt, head.expect, head.lost, head.spin_limit);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sum = 0.0;
#pragma unroll
            for (int hh = 0; hh < LQ_HEADS; ++hh)
                sum += __hip_atomic_load(head.wpart + ((long)hh * 64 + lane) * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wt[i] = sum;
        }
        A16MARK(6);   // partial products of the other head workgroups
    }
    // Z' = T' W': A (m' = n, k = 4 t + g) = T[4 t + g][n], B = register t of W'
    d4 z = d4{0.0, 0.0, 0.0, 0.0};
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, wt[0], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, wt[1], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(t2, wt[2], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(t3, wt[3], z, 0, 0, 0);
    // X' -= V' Z' block by block: A (M row n, k = 4 t + g) = V[4 t + g][16 b + pm], B = register t of Z'; result
    // register i of lane (n, g) = M row 4 i + g = column 16 b + 4 g + i of matrix row n
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // (LDS operations of one wavefront execute in order: the fences only keep the compiler from reordering)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        *(dbl4*)&s_v[wv][n][4 * g] = a[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double vt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) vt[t] = s_v[wv][4 * t + g][pm];
        d4 o = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) o = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[t], z[t], o, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[u][i] -= o[i];
        // block u goes out while the matrix cores work on block u + 1 (measured with -DOGSQP_TRACE at C3: the loads
        // take 10 us, the stores 6-7 us at the chip's HBM rate, this loop 4.8 us of MFMA issue - one after the other
        // they added up)
        const int j0 = 16 * A16_BLOCK(u) + 4 * g;
        if (valid) {
            if (COHERENT) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j0 + i < L) __hip_atomic_store(row + j0 + i, x[u][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (j0 + 3 < L) {
                *(dbl4*)(row + j0) = x[u];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j0 + i < L) row[j0 + i] = x[u][i];
            }
        }
    }
    A16MARK(4);   // T product, tiles, last product, stores issued
#ifdef OGSQP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A16MARK(5);   // stores
    if (!HEAD && r0 >= 5 * rcount && r0 < 6 * rcount && tid == 0)
        for (int e = 0; e < 8; ++e) g_apply16_trace[e] = t_sec[e];
    if (HEAD && head.h == 0 && tid == 0)
        for (int e = 0; e < 8; ++e) g_head16_trace[e] = t_sec[e];
#endif
#undef A16_BLOCK
}

template <int U>
__global__ __launch_bounds__(64 * A16_WAVES) void k_lq_apply16(double* __restrict__ Tc, double* __restrict__ Jw, int ld,
                                                              int mrows, int nq, int k, const double* __restrict__ V,
                                                              int ldv, const Lq16Panel* __restrict__ panel, int rpg) {
    __shared__ A16Shared sh;
    lq_apply16_body<U>(Tc, Jw, ld, mrows, nq, k, V, ldv, panel, (int)blockIdx.x * rpg, rpg, sh);
}

// Look-ahead inside one launch.  The panel factorisation is one workgroup's latency chain (26-33 us) and the
// trailing update the whole chip's bandwidth (21-29 us); run one after the other they idle each other's resource.
// Here the launch that applies panel k also factors panel k + 16 (into the other reflector buffer):
//   workgroups 1 .. LQ_HEADS  update the 16 rows of the next panel, an eighth of the COLUMNS each.  (A workgroup
//                             streams at what ONE compute unit's miss queue sustains, ~38 GB/s measured, and issues
//                             its 96 MFMAs per SIMD at one per 27 ns: all 16 rows in one workgroup take as long as the
//                             whole update (22-27 us), two rows per workgroup still 15 us because every one of them
//                             reads all of V and runs all of the MFMAs.)  Their partial products meet in memory
//                             (Lq16Head), the rows go out with agent-scope stores, and they count themselves done;
//   workgroup 0               waits for that count (the eight are resident before any other workgroup of the launch and
//                             wait only for each other), reads the rows past its L2 and factors the panel;
//   the others                the rest of the update, rpg rows each.
// The head rows' W' is summed in another order than k_lq_apply16 sums it (eight column slices instead of eight
// wavefronts): same reflectors to rounding, not to the bit.
template <int U, int E>
__global__ __launch_bounds__(64 * A16_WAVES) void k_lq_step16(double* __restrict__ Tc, double* __restrict__ Jw, int ld,
                                                             int mrows, int nq, int k, const double* __restrict__ V,
                                                             int ldv, const Lq16Panel* __restrict__ panel,
                                                             double* __restrict__ Vnext, Lq16Panel* __restrict__ pnext,
                                                             double* __restrict__ diagL, double* __restrict__ dmaxbuf, int rpg,
                                                             unsigned* __restrict__ sync, unsigned expect,
                                                             double* __restrict__ wpart, int* __restrict__ lost,
                                                             int spin_limit) {
    extern __shared__ __attribute__((aligned(32))) double lds[];
    __shared__ A16Shared sh;
    static_assert(64 * A16_WAVES == P16_THREADS, "one workgroup shape for both roles");
    constexpr int UH = (U + LQ_HEADS - 1) / LQ_HEADS;
    // (the heads come first in the grid: they are dispatched before everything else and start together)
    const int b = (int)blockIdx.x == LQ_HEADS ? 0 : (int)blockIdx.x < LQ_HEADS ? (int)blockIdx.x + 1 : (int)blockIdx.x;
    if (b == 0) {
        if (threadIdx.x == 0) lq_wait_for(sync + 1, expect, lost, spin_limit);
        __syncthreads();
        lq_panel16_body<E, true>(Tc, ld, mrows, nq, k + LQ16, Vnext, ldv, diagL, pnext, dmaxbuf, lds);
    } else if (b <= LQ_HEADS) {
        Lq16Head head;
        head.wpart = wpart;
        head.cnt = sync;
        head.expect = expect;
        head.h = b - 1;
        head.lost = lost;
        head.spin_limit = spin_limit;
        lq_apply16_body<UH, true>(Tc, Jw, ld, mrows, nq, k, V, ldv, panel, 0, LQ16, sh, b - 1, LQ_HEADS, head);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wavefront: its stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        lq_apply16_body<U>(Tc, Jw, ld, mrows, nq, k, V, ldv, panel, LQ16 + (b - 1 - LQ_HEADS) * rpg, rpg, sh);
    }
}
