// ogsqp_rows.h - the dual active-set method of the least-distance problem in ROTATED coordinates, one
// workgroup-parallel pass over the constraint rows per change (included by ogsqp.hip inside its anonymous
// namespace).
//
// The two older kernels (k_gi_iter, k_gi_coop) keep an orthonormal basis Q1 of the active normals next to the
// untouched constraint matrix W = [G; I] Y and pay for it with a chain of dependent global reductions per change
// (projections Q1'n, the component z, the pricing W y: ~12 memory round trips and 3-4 grid barriers, 40-66 us per
// change at C3) and with a serial Givens chain per removal.  Here neither exists.  Goldfarb & Idnani's method is
// run the way the restatement (oracle/slsqp_np.py: ldp_gi) states it - an orthogonal Q such that the active
// normals live in the first q coordinates of W Q - but Q is applied to the rows of W as it grows (W <- W Q, in
// place in GJ / Jw: J Q is as good a factor of B^-1 as J), so that
//
//   * the normal of the incoming row p in the current coordinates is row p itself: d = W[p], d1 = d[:q],
//     d2 = d[q:]; the step direction is [0; d2], |z|^2 = |d2|^2; no projection, no Gram-Schmidt;
//   * every row keeps its own constraint value W[i] y up to date from  g_i = W[i][q:] . d2  (dots[i] += t g_i),
//     which is the same dot product the Householder reflector of a full step needs
//     (W[i][q:] -= beta (g_i - alpha W[i][q]) v, v = d2 - alpha e1): ONE pass over the rows per change, row-local,
//     no communication between rows;
//   * a leaving row k is one reflector too, not a chain of q - k Givens rotations: with M the (square, here not
//     necessarily triangular) matrix of the active normals in the first q coordinates and its explicit inverse
//     M^-1 = RI, row k of RI is the direction of those coordinates that is orthogonal to every OTHER active
//     normal; the reflector that sends it to the last of them, e_{q-1}, applied to every row (and to the rows of
//     RI, which then lose row k and their last column), leaves the remaining normals in the first q - 1
//     coordinates.  No factor R is kept at all: r = RI d1 is all the method asks of it.
//
// What is serial per change is small: the dual direction r = RI d1 (rows of the inverse spread over the workgroups
// of the first kernel), the ratio test and the book-keeping (the workgroup that arrives last at the first kernel's
// ticket).  A change is two launches - k_rows_decide, k_rows_apply - and the boundary between them is the only
// synchronisation (1.5-1.9 us on this part, cheaper than a grid barrier); the host enqueues batches of pairs and
// looks at the state once per batch, kernels of a finished solve return at once.
//
// Warm start.  The rows that were active at the solution of the previous subproblem are appended to the LQ sweep
// of the equalities (og_qp_solve_dev), which leaves them triangular in the new coordinates: RI is the inverse of
// that triangle (k_rows_invert), the minimiser on them and its multipliers are two products with it; while a
// multiplier is negative the row with the most negative one is taken out (phase -1).  What is left is an S-pair,
// from which the dual method continues; late in an SQP run a subproblem then costs a handful of changes.

struct RowsDecision {
    int kind;        // 0 price only, 1 full step (p joins), 2 partial step (position k leaves), 3 warm start: position
                     // k leaves (reflector only), 4 warm start done: every row's value from y, 5 nothing at all (a pair
                     // enqueued for the warm start after it was over: k_rows_resident prices for itself)
    int q;           // active rows before this change = first tail coordinate
    int k;
    int dependent;
    double t, alpha, beta, beta_out;   // step; reflector of the incoming row; of the leaving one (vector in vvec)
};

struct RowsArgs {
    GiArgs g;                // g.y = y in the rotated coordinates; g.RI[0] = the inverse, one row per SLOT
    double* dots;            // mg + nq: W[i] . y of every row
    double* dvec;            // nr: the incoming normal (signed)
    double* rvec;            // qcap: dual direction r = RI d1
    double* vvec;            // qcap: reflector vector of a removal
    int* slot;               // qcap: storage row of RI per active position; positions >= q list the free rows
    GiPartial* price;        // G2: most violated row of every workgroup of k_rows_apply
    GiPartial* ratio;        // G1: ratio-test candidate of every workgroup of k_rows_decide
    RowsDecision* rec;
    int G1, G2;
    // warm start's removals spread over the grid of k_rows_decide (round 5): multipliers of the active rows before the
    // argmin, the flag a wait that gives up raises, the bound of that wait; 0: one workgroup does both products
    double* uval;
    int* lost;
    int spin_limit;
    int warm_spread;
    // round 6: the launches in front of k_rows_resident (ogsqp_resident.h) only serve the warm start's removals: with the
    // warm start over, k_rows_decide leaves a "price only" decision (the pass that follows changes nothing) and returns
    int only_warm;
};

constexpr int ROWS_SPREAD_TILE = 128;                 // rows of the inverse per tile of a 16-column block (stage A)
// extra LDS of the spread form behind the three vectors: the tile and the storage rows of the active positions
__host__ __device__ inline size_t rows_spread_lds_bytes(int qcap) {
    return (size_t)(ROWS_SPREAD_TILE * 16 + (qcap + 1) / 2 + 8) * sizeof(double);
}

constexpr int ROWS_THREADS = 256;
constexpr int ROWS_WAVES = ROWS_THREADS / 64;

// Stack row r < mg + nq (general rows, then one row per variable shared by its lower and upper bound).
__device__ __forceinline__ double* rows_ptr(const GiArgs& g, int r) {
    return (double*)(r < g.mg ? g.GJ + (long)r * g.ld : g.Jw + (long)(r - g.mg) * g.ld) + g.meq;
}

// Position k leaves (one workgroup): its storage row becomes the first free one, the lists close up, and the
// reflector that moves the freed direction to coordinate q-1 goes to vvec.  w = row k of RI (LDS, q entries) is
// turned into the reflector vector in place.  Returns beta of the reflector.
__device__ __forceinline__ double rows_leave(const RowsArgs& a, int q, int k, double* w, int* shifted, double* red) {
    const GiArgs& g = a.g;
    const int tid = threadIdx.x;
    const double* rowk = g.RI[0] + (long)a.slot[k] * g.qcap;
    double part = 0.0;
    for (int j = tid; j < q; j += ROWS_THREADS) {
        const double v = rowk[j];
        w[j] = v;
        part += v * v;
    }
    const double ww = block_sum(part, red);
    const double wl = w[q - 1];
    const double alpha = wl >= 0.0 ? -sqrt(ww) : sqrt(ww);
    const double v0 = wl - alpha;
    const double vv = ww - wl * wl + v0 * v0;
    __syncthreads();
    if (tid == 0) w[q - 1] = v0;
    __syncthreads();
    for (int j = tid; j < q; j += ROWS_THREADS) a.vvec[j] = w[j];
    // lists: positions k+1 .. q-1 move down, the freed storage row goes to position q-1
    const int leaving = g.act[k], freed = a.slot[k];
    for (int j = k + tid; j < q - 1; j += ROWS_THREADS) {
        shifted[2 * j] = g.act[j + 1];
        shifted[2 * j + 1] = a.slot[j + 1];
    }
    __syncthreads();
    for (int j = k + tid; j < q - 1; j += ROWS_THREADS) {
        g.act[j] = shifted[2 * j];
        a.slot[j] = shifted[2 * j + 1];
    }
    if (tid == 0) {
        a.slot[q - 1] = freed;
        g.u[leaving] = 0.0;
        g.isact[leaving] = 0;
    }
    __syncthreads();
    return vv > 0.0 ? 2.0 / vv : 0.0;
}

// ------------------------------------------------------------------------------------------
// First kernel of a change: who comes in, the dual direction, the step length, the book-keeping.
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_decide(RowsArgs a) {
    extern __shared__ double lds[];
    __shared__ double redv[ROWS_WAVES];
    __shared__ int redi[ROWS_WAVES];
    __shared__ double red[ROWS_WAVES];
    __shared__ double red2[2 * ROWS_WAVES];
    __shared__ int s_last;
    const GiArgs& g = a.g;
    GiState* st = g.st;
#ifdef OGSQP_TRACE
    long long t_mark = __builtin_amdgcn_s_memrealtime();
    long long t_sec[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RMARK(slot_) do { const long long now_ = __builtin_amdgcn_s_memrealtime(); t_sec[slot_] += now_ - t_mark; t_mark = now_; } while (0)
#else
#define RMARK(slot_) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap;
    // the partial results of the pricing are requested together with the state word (they are only looked at in
    // phase 0): one round trip instead of two.  At most 2048 of them - the grid of k_rows_apply
    constexpr int PRE = 8;
    double ppv[PRE];
    int ppi[PRE];
#pragma unroll
    for (int c = 0; c < PRE; ++c) {
        const int b = tid + ROWS_THREADS * c;
        ppv[c] = a.price[b < a.G2 ? b : 0].value;
        ppi[c] = a.price[b < a.G2 ? b : 0].index;
    }
    const int phase = st->phase;
    if (phase >= 2) return;
    if (a.only_warm && phase >= 0) {
        if (w == 0 && tid == 0) a.rec->kind = 5;
        return;
    }
    const int q = st->q;
    RMARK(0);   // state word
    double* d = lds;                  // nr: incoming normal
    double* rv = d + nr;              // qcap: dual direction / multipliers of the warm start
    double* aux = rv + qcap;          // qcap: leaving row of RI -> reflector vector / minimiser of the warm start
    int* shifted = (int*)d;           // (the normal is spent when the lists are shifted)
    const double* RI = g.RI[0];

    if (phase < 0) {
        // ---- warm start: minimiser on the warm rows, multipliers, the most negative one leaves --------------
        const bool spread = a.warm_spread != 0 && a.G1 > 1 && q > 0;
        if (!spread && w != 0) return;
        if (q == 0) {
            if (w == 0 && tid == 0) {
                a.rec->kind = 0;
                st->phase = 0;
            }
            return;
        }
        double* y1 = aux;
        double* bA = d;
        double worst = INFINITY;
        int kworst = 0x7fffffff;
        if (spread) {
            // Round 5.  A removal recomputes the point and its multipliers on the remaining rows - y1 = -RI' b_A, u = RI y1 -
            // and ONE workgroup streaming the q x q inverse twice at what one compute unit pulls (50 GB/s) took 52-90 us at
            // C3 and a millisecond at C5.  Both products are spread over the grid: (A) workgroup w owns blocks of 16
            // columns of y1 - tiles of 128 rows of the inverse go through LDS, two threads per column add up the even and
            // the odd rows in the order the one-workgroup form adds them -, a counter and a bounded wait make y1 whole,
            // (B) the rows of u are dealt to the wavefronts of the grid as the dual direction's are below, the last
            // workgroup at the ticket decides.  Same sums in the same order: same bits (OGSQP_WARM_SPREAD=0: the old form).
            double* tile = lds + (nr + 2 * qcap + 16);
            int* sl = (int*)(tile + ROWS_SPREAD_TILE * 16);
            for (int i = tid; i < q; i += ROWS_THREADS) {
                sl[i] = a.slot[i];
                bA[i] = g.bval[g.act[i]];
            }
            __syncthreads();
            const int nblk = (q + 15) >> 4;
            for (int cb = w; cb < nblk; cb += a.G1) {
                double acc = 0.0;                              // thread t < 32: column 16 cb + (t & 15), rows of parity t >> 4
                for (int r0 = 0; r0 < q; r0 += ROWS_SPREAD_TILE) {
                    double v[ROWS_SPREAD_TILE * 16 / ROWS_THREADS];
#pragma unroll
                    for (int k = 0; k < ROWS_SPREAD_TILE * 16 / ROWS_THREADS; ++k) {
                        const int e = tid + ROWS_THREADS * k, row = r0 + (e >> 4), col = 16 * cb + (e & 15);
                        v[k] = (row < q && col < q) ? RI[(long)sl[row < q ? row : 0] * qcap + col] : 0.0;
                    }
                    __syncthreads();                           // (the previous tile has been summed)
#pragma unroll
                    for (int k = 0; k < ROWS_SPREAD_TILE * 16 / ROWS_THREADS; ++k) tile[tid + ROWS_THREADS * k] = v[k];
                    __syncthreads();
                    if (tid < 32) {
                        const int col = tid & 15, rmax = min(ROWS_SPREAD_TILE, q - r0);
                        for (int rr = tid >> 4; rr < rmax; rr += 2) acc += tile[rr * 16 + col] * bA[r0 + rr];
                    }
                }
                if (tid < 64) {
                    const double odd = __shfl_down(acc, 16);
                    const int col = 16 * cb + tid;
                    if (tid < 16 && col < q) st_shared(a.rvec + col, -(acc + odd));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&st->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&st->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.G1) {
                    if (++spins > a.spin_limit) {
                        __hip_atomic_store(a.lost, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            for (int j = tid; j < q; j += ROWS_THREADS) y1[j] = ld_shared(a.rvec + j);
            __syncthreads();
            const int stride_w = a.G1 * ROWS_WAVES;
            for (int i0 = w * ROWS_WAVES + wave; i0 < q; i0 += 4 * stride_w) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                const double* row[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * stride_w;
                    row[u] = RI + (long)sl[i < q ? i : i0] * qcap;
                }
                for (int j = lane; j < q; j += 64) {
                    const double yj = y1[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] += row[u][j] * yj;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * stride_w;
                        if (i < q) {
                            st_shared(a.uval + i, acc[u]);
                            if (acc[u] < worst || (acc[u] == worst && i < kworst)) {
                                worst = acc[u];
                                kworst = i;
                            }
                        }
                    }
                }
            }
            if (lane != 0) {
                worst = INFINITY;
                kworst = 0x7fffffff;
            }
            block_argmin(worst, kworst, redv, redi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                st_shared(&a.ratio[w].value, worst);
                st_shared(&a.ratio[w].index, kworst);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned tk = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (tk == (unsigned)a.G1 - 1) ? 1 : 0;
            }
            __syncthreads();
            if (!s_last) return;
            if (tid == 0) {
                st->ticket = 0u;
                st->arrive = 0;                                // (every workgroup has passed the wait above)
            }
            worst = INFINITY;
            kworst = 0x7fffffff;
            for (int b = tid; b < a.G1; b += ROWS_THREADS) {
                const double cv = ld_shared(&a.ratio[b].value);
                const int ci = ld_shared(&a.ratio[b].index);
                if (cv < worst || (cv == worst && ci < kworst)) {
                    worst = cv;
                    kworst = ci;
                }
            }
            for (int i = tid; i < q; i += ROWS_THREADS) rv[i] = ld_shared(a.uval + i);
            block_argmin(worst, kworst, redv, redi);
        } else {
        for (int i = tid; i < q; i += ROWS_THREADS) bA[i] = g.bval[g.act[i]];
        __syncthreads();
        // (one workgroup reads the inverse twice per removal: what it costs is the number of round trips, so eight rows
        // of the first product and four of the second are in flight per thread, the sums in their old order - 82 us per removal at C3 before)
        for (int j = tid; j < q; j += ROWS_THREADS) {          // M' y1 = -b_A  ->  y1 = -RI' b_A
            // (two sums, even and odd rows, as ever - the order of the additions is part of the result - but eight
            // rows requested before the first of them is used)
            double acc0 = 0.0, acc1 = 0.0;
            int i = 0;
            for (; i + 7 < q; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = RI[(long)a.slot[i + u] * qcap + j];
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc0 += v[u] * bA[i + u];
                    acc1 += v[u + 1] * bA[i + u + 1];
                }
            }
            for (; i + 1 < q; i += 2) {
                acc0 += RI[(long)a.slot[i] * qcap + j] * bA[i];
                acc1 += RI[(long)a.slot[i + 1] * qcap + j] * bA[i + 1];
            }
            if (i < q) acc0 += RI[(long)a.slot[i] * qcap + j] * bA[i];
            y1[j] = -(acc0 + acc1);
        }
        __syncthreads();
        for (int i0 = 4 * wave; i0 < q; i0 += 4 * ROWS_WAVES) {    // y = N u  ->  u = RI y1, four rows per wavefront and trip
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            const double* row[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) row[u] = RI + (long)a.slot[min(i0 + u, q - 1)] * qcap;
            for (int j = lane; j < q; j += 64) {
                const double yj = y1[j];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += row[u][j] * yj;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    if (i < q) {
                        rv[i] = acc[u];
                        if (acc[u] < worst || (acc[u] == worst && i < kworst)) {
                            worst = acc[u];
                            kworst = i;
                        }
                    }
                }
            }
        }
        if (lane != 0) {
            worst = INFINITY;
            kworst = 0x7fffffff;
        }
        block_argmin(worst, kworst, redv, redi);
        }
        if (!(worst < 0.0)) {
            double part = 0.0;
            for (int i = tid; i < nr; i += ROWS_THREADS) {
                const double v = i < q ? y1[i] : 0.0;
                g.y[i] = v;
                part += v * v;
            }
            for (int i = tid; i < q; i += ROWS_THREADS) g.u[g.act[i]] = rv[i];
            const double ynorm = sqrt(block_sum(part, red));
            if (tid == 0) {
                RowsDecision rec;
                rec.kind = 4;
                rec.q = q;
                rec.k = 0;
                rec.dependent = 0;
                rec.t = rec.alpha = rec.beta = rec.beta_out = 0.0;
                *a.rec = rec;
                st->phase = 0;
                st->ynorm = ynorm;
            }
            return;
        }
        const int k = kworst;
        __syncthreads();
        const double beta_out = rows_leave(a, q, k, aux, shifted, red);
        if (tid == 0) {
            RowsDecision rec;
            rec.kind = 3;
            rec.q = q;
            rec.k = k;
            rec.dependent = 0;
            rec.t = rec.alpha = rec.beta = 0.0;
            rec.beta_out = beta_out;
            *a.rec = rec;
            st->q = q - 1;
            st->iters = st->iters + 1;
        }
        return;
    }

    // ---- the incoming row ------------------------------------------------------------------------------------
    // Everything whose address does not depend on what another workgroup is about to write is requested as early as
    // its address is known and in every workgroup - the lists and the multipliers of the active rows and y for the
    // decision (PRE entries per thread in registers; longer lists fall back to loading late), the first rows of the
    // inverse - so that the workgroup that draws the last ticket has one round trip left: the others' shares of r
    // and of the ratio test.  Measured (-DOGSQP_TRACE, tools/sqp_trace.sh; C3, first subproblems): 11.3 us per change
    // in this kernel (12.4 before the requests were ordered by who is waited for first) - state word 0.5, price
    // election 2.2 (875 partials), normal into LDS 2.0, norms 0.4, inverse rows x d1 1.1, stores + ticket 1.8, the
    // others' r 1.4, u and y 1.0, reflector and lists 1.0.  What is left is the chain itself: state -> price -> row p
    // -> (product) -> store acknowledged -> ticket -> the others' r, six dependent trips through memory of about a
    // microsecond each (agent-scope traffic goes past the L2), and seven workgroup reductions of 0.4 us.
    int pact[PRE], pslot[PRE];
    double pu[PRE], py[PRE];
    const int fresh_slot = a.slot[q < qcap ? q : 0];          // storage row of position q, should a row join
#pragma unroll
    for (int c = 0; c < PRE; ++c) {
        const int jj = tid + ROWS_THREADS * c;
        pact[c] = g.act[jj < q ? jj : 0];
        pslot[c] = a.slot[jj < qcap ? jj : 0];
        py[c] = g.y[jj < nr ? jj : 0];
    }
    const int stride = a.G1 * ROWS_WAVES;
    const int ifirst = w * ROWS_WAVES + wave;
    int fslot[4], fact[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ifirst + e * stride;
        fslot[e] = a.slot[i < q ? i : (ifirst < q ? ifirst : 0)];
        fact[e] = g.act[i < q ? i : (ifirst < q ? ifirst : 0)];
    }
    int p;
    if (phase == 0) {
        double v = INFINITY;
        int idx = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const bool in = tid + ROWS_THREADS * c < a.G2;
            const double pv = in ? ppv[c] : INFINITY;
            const int pi = in ? ppi[c] : 0x7fffffff;
            if (pv < v || (pv == v && pi < idx)) {
                v = pv;
                idx = pi;
            }
        }
        for (int b = tid + ROWS_THREADS * PRE; b < a.G2; b += ROWS_THREADS) {
            const double pv = a.price[b].value;
            const int pi = a.price[b].index;
            if (pv < v || (pv == v && pi < idx)) {
                v = pv;
                idx = pi;
            }
        }
        block_argmin(v, idx, redv, redi);
        if (!(v < 0.0)) {                         // every workgroup sees the same: solved
            if (w == 0 && tid == 0) st->phase = 2;
            return;
        }
        p = idx;
    } else {
        p = st->p;
    }
    RMARK(1);   // who comes in
    // the first RPRE * 64 entries of my first rows of the inverse (their addresses do not depend on who comes in;
    // the list entries they hang on have arrived during the election)
    constexpr int RPRE = 8;
    double rpre[4][RPRE];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double* rowe = RI + (long)fslot[e] * qcap;
#pragma unroll
        for (int c = 0; c < RPRE; ++c) {
            const int j = lane + 64 * c;
            rpre[e][c] = rowe[j < q ? j : 0];
        }
    }
    const int iters = st->iters + 1;
    const bool broken = iters > g.limit || q > nr || q > qcap || p < 0 || p >= mg + 2 * nq;
    const int psafe = broken ? 0 : p;
    double psign;
    const double* prow = stack_row(g, psafe, psign);
    for (int i = tid; i < nr; i += ROWS_THREADS) d[i] = psign * prow[i];
    const int prow_index = psafe < mg + nq ? psafe : psafe - nq;
    const double bval_p = g.bval[psafe], dots_p = a.dots[prow_index];
    const double up_before = phase == 0 ? 0.0 : st->up;
#pragma unroll
    for (int c = 0; c < PRE; ++c) pu[c] = g.u[tid + ROWS_THREADS * c < q ? pact[c] : 0];      // (act[] beyond q: anything)
    double fu[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fu[e] = g.u[ifirst < q ? fact[e] : 0];
    __syncthreads();
    RMARK(2);   // its normal in LDS
    // |d2|^2 and |d|^2 (every workgroup: the same sums in the same order)
    double part_zz = 0.0, part_nn = 0.0;
    for (int i = tid; i < nr; i += ROWS_THREADS) {
        const double v = d[i];
        part_nn += v * v;
        if (i >= q) part_zz += v * v;
    }
    // (one exchange for both: each the sum of its wavefronts' sums in their order, as block_sum has it)
    part_zz = wave_sum(part_zz);
    part_nn = wave_sum(part_nn);
    if (lane == 0) {
        red2[wave] = part_zz;
        red2[ROWS_WAVES + wave] = part_nn;
    }
    __syncthreads();
    double zz = 0.0, nn = 0.0;
#pragma unroll
    for (int wv = 0; wv < ROWS_WAVES; ++wv) {
        zz += red2[wv];
        nn += red2[ROWS_WAVES + wv];
    }
    RMARK(3);   // |d2|, |d|
    // ---- my rows of the inverse: dual direction and ratio test ------------------------------------------------
    {
        double t1 = INFINITY;
        int kdrop = 0x7fffffff;
        // four rows per trip: their loads are in flight together (a row is one memory round trip otherwise)
        for (int i0 = ifirst; i0 < q; i0 += 4 * stride) {
            const double* row[4];
            double uact[4], acc[4];
            const bool first = i0 == ifirst;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * stride;
                const int sl = first ? fslot[e] : a.slot[i < q ? i : i0];
                row[e] = RI + (long)sl * qcap;
                uact[e] = first ? fu[e] : g.u[g.act[i < q ? i : i0]];
                acc[e] = 0.0;
            }
            int jstart = lane;
            if (first) {                                       // (same sums in the same order as the loop below)
#pragma unroll
                for (int c = 0; c < RPRE; ++c) {
                    const int j = lane + 64 * c;
                    const double dj = j < q ? d[j < q ? j : 0] : 0.0;      // (beyond q: + 0, after the last real term)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += rpre[e][c] * dj;
                }
                jstart = lane + 64 * RPRE;
            }
            for (int j = jstart; j < q; j += 64) {
                const double dj = d[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += row[e][j] * dj;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * stride;
                const double ri = wave_sum(acc[e]);
                if (lane == 0 && i < q) {
                    st_shared(a.rvec + i, ri);
                    if (ri > 0.0) {
                        const double cand = uact[e] / ri;
                        if (cand < t1 || (cand == t1 && i < kdrop)) {
                            t1 = cand;
                            kdrop = i;
                        }
                    }
                }
            }
        }
        if (lane != 0) {
            t1 = INFINITY;
            kdrop = 0x7fffffff;
        }
        RMARK(4);   // my rows of the inverse times d1
        block_argmin(t1, kdrop, redv, redi);
        // (the barriers of the reduction order every wavefront's stores to r before thread 0's wait below only in
        // program order of ITS wavefront: each wavefront waits for its own)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            st_shared(&a.ratio[w].value, t1);
            st_shared(&a.ratio[w].index, kdrop);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == (unsigned)a.G1 - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
        RMARK(5);   // stores out, ticket
    }
    // ---- last workgroup: the decision ------------------------------------------------------------------------
    if (tid == 0) st->ticket = 0u;
    double t1 = INFINITY;
    int kdrop = 0x7fffffff;
    for (int b = tid; b < a.G1; b += ROWS_THREADS) {
        const double cv = ld_shared(&a.ratio[b].value);
        const int ci = ld_shared(&a.ratio[b].index);
        if (cv < t1 || (cv == t1 && ci < kdrop)) {
            t1 = cv;
            kdrop = ci;
        }
    }
    // the others' entries of r: requested together with their ratio-test candidates
    double pr[PRE];
#pragma unroll
    for (int c = 0; c < PRE; ++c) {
        const int jj = tid + ROWS_THREADS * c;
        pr[c] = ld_shared(a.rvec + (jj < q ? jj : 0));
    }
    block_argmin(t1, kdrop, redv, redi);
    RMARK(6);   // the others' r and candidates
    if (broken) {
        if (tid == 0) {
            st->phase = 3;
            st->iters = iters;
        }
        return;
    }
    const bool dependent = (q >= nr) || !(zz > (DEPENDENT * DEPENDENT) * nn);
    const double sp = bval_p + psign * dots_p;
    const double t2 = dependent ? INFINITY : -sp / zz;
    const double t = fmin(t1, t2);
    if (!(t < INFINITY)) {
        if (tid == 0) {
            st->phase = 4;
            st->iters = iters;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < PRE; ++c) {
        const int jj = tid + ROWS_THREADS * c;
        if (jj < q) {
            rv[jj] = pr[c];
            g.u[pact[c]] = pu[c] - t * pr[c];
        }
    }
    for (int jj = tid + ROWS_THREADS * PRE; jj < q; jj += ROWS_THREADS) {
        const double rj = ld_shared(a.rvec + jj);
        rv[jj] = rj;
        g.u[g.act[jj]] -= t * rj;
    }
    const double up = up_before + t;
    double part_yy = 0.0;
    {
        // (the partial sums of |y|^2 in the order of the plain loop: c ascending per thread)
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int i = tid + ROWS_THREADS * c;
            if (i < nr) {
                double yi = py[c];
                if (!dependent && i >= q) {
                    yi += t * d[i];
                    g.y[i] = yi;
                }
                part_yy += yi * yi;
            }
        }
        for (int i = tid + ROWS_THREADS * PRE; i < nr; i += ROWS_THREADS) {
            double yi = g.y[i];
            if (!dependent && i >= q) {
                yi += t * d[i];
                g.y[i] = yi;
            }
            part_yy += yi * yi;
        }
    }
    const double ynorm = sqrt(block_sum(part_yy, red));
    RMARK(7);   // u, y
    for (int i = tid; i < nr; i += ROWS_THREADS) a.dvec[i] = d[i];
    const bool full_step = (t2 < INFINITY) && (t2 <= t1);
    RowsDecision rec;
    rec.q = q;
    rec.k = 0;
    rec.dependent = dependent ? 1 : 0;
    rec.t = t;
    rec.alpha = rec.beta = rec.beta_out = 0.0;
    if (full_step) {
        // p joins: reflector H with H d2 = alpha e1 on the tail coordinates; the matrix of the active normals gets
        // the column [d1; alpha], its inverse the column -r / alpha and the row [0 .. 0, 1 / alpha]
        const double dq = d[q];
        const double alpha = dq >= 0.0 ? -sqrt(zz) : sqrt(zz);
        const double v0 = dq - alpha;
        const double vv = zz - dq * dq + v0 * v0;
        rec.kind = 1;
        rec.alpha = alpha;
        rec.beta = vv > 0.0 ? 2.0 / vv : 0.0;
        double* RIw = g.RI[0];
        const double inv = 1.0 / alpha;
        double* fresh = RIw + (long)fresh_slot * qcap;
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int i = tid + ROWS_THREADS * c;
            if (i < q) {
                RIw[(long)pslot[c] * qcap + q] = -pr[c] * inv;
                fresh[i] = 0.0;
            }
        }
        for (int i = tid + ROWS_THREADS * PRE; i < q; i += ROWS_THREADS) {
            RIw[(long)a.slot[i] * qcap + q] = -rv[i] * inv;
            fresh[i] = 0.0;
        }
        if (tid == 0) {
            fresh[q] = inv;
            g.act[q] = p;
            g.u[p] = up;
            g.isact[p] = 1;
            st->q = q + 1;
            st->phase = 0;
        }
    } else {
        // partial step: position k leaves; the reflector goes to k_rows_apply, which applies it to every row of the
        // constraint matrix and of the inverse
        const int k = kdrop;
        if (k < 0 || k >= q) {
            if (tid == 0) {
                st->phase = 3;
                st->dbg = -7;
                st->dbg2 = k;
            }
            return;
        }
        rec.kind = 2;
        rec.k = k;
        __syncthreads();
        rec.beta_out = rows_leave(a, q, k, aux, shifted, red);
        if (tid == 0) {
            st->q = q - 1;
            st->phase = 1;
            st->p = p;
            st->up = up;
        }
    }
    if (tid == 0) {
        *a.rec = rec;
        st->iters = iters;
        st->ynorm = ynorm;
    }
#ifdef OGSQP_TRACE
    RMARK(8);   // reflector, the inverse's new column / the leaving row
    if (tid == 0) {
        for (int e = 0; e < 9; ++e) st->tr[e] += t_sec[e];
        st->tr[10] += 1;
        if (!full_step) st->tr[11] += 1;
    }
#endif
#undef RMARK
}

// ------------------------------------------------------------------------------------------
// Second kernel of a change: the pass over the rows, one wavefront per row with the whole row in registers
// (TAIL coordinates per lane: rows of up to 64 * TAIL null-space entries).  The walk covers the constraint rows,
// then y (which lives in the same coordinates), then - when a row leaves - the rows of the inverse.
//
// Round 5 (80 us per change at C5, 143 MB at 1.8 TB/s).  The kernel was four residency rounds of
// wavefronts that each ran the chain  state word -> decision -> its vectors -> row -> sum -> store  once: its duration
// was round trips, not bytes.  Now (a) the state word, the decision and both vectors are requested together (none of
// their addresses depends on another's value; the masks that do are applied afterwards), (b) the two vectors share
// ONE array in LDS - the incoming normal lives on the coordinates >= q0, the leaving reflector on those < q0; in
// registers they cost 4 * TAIL VGPRs -, and (c) the 3 * TAIL lane masks of the row loop are no longer loop invariants
// (see the loop): 256 VGPRs + 232 AGPRs + spilled scalar pairs, one wavefront per SIMD, became a kernel that several
// wavefronts per SIMD fit, and the grid covers the rows in one or two residency rounds instead of eight.  The
// sums are the same sums in the same order (lane l adds its coordinates l, l + 64, ... in ascending order, then
// wave_sum): same bits as rounds 3-4.
template <int TAIL>
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_apply(RowsArgs a) {
    __shared__ double redv[ROWS_WAVES];
    __shared__ int redi[ROWS_WAVES];
    const GiArgs& g = a.g;
    GiState* st = g.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap;
    // ---- one round trip for everything whose address is known: state, decision, the two vectors.  The vectors are the
    // same for every row: they live in LDS (in registers they cost 4 * TAIL VGPRs and the kernel one wavefront per SIMD)
    __shared__ double s_c[64 * TAIL];
    // Short rows (up to 512 coordinates: C3) are requested WHOLE in the same trip, before the decision is known - a
    // constraint row's address depends on nothing the decision says; the part the change does not touch is masked out
    // of the registers afterwards (the launch is a chain of round trips, not of bytes: 7.1 us with the row requested
    // after the decision).  Rows of the inverse (their address hangs on a list entry) and longer rows load late.
    constexpr bool EARLY = TAIL <= 8;
    const int r_first = w * ROWS_WAVES + wave;
    const bool early = EARLY && r_first <= mg + nq;
    double xe[EARLY ? TAIL : 1];
    double dot_e = 0.0;
    if (early) {
        const double* row0 = r_first == mg + nq ? g.y : rows_ptr(g, r_first);
#pragma unroll
        for (int e = 0; e < (EARLY ? TAIL : 1); ++e) {
            const int j = lane + 64 * e;
            xe[e] = row0[j < nr ? j : 0];
        }
        dot_e = a.dots[r_first < mg + nq ? r_first : 0];
    }
    const int phase = st->phase;
    const double ynorm = st->ynorm;
    const RowsDecision rec = *a.rec;
    double draw[(TAIL + ROWS_WAVES - 1) / ROWS_WAVES], vraw[(TAIL + ROWS_WAVES - 1) / ROWS_WAVES];
#pragma unroll
    for (int c = 0; c < (TAIL + ROWS_WAVES - 1) / ROWS_WAVES; ++c) {
        const int j = tid + ROWS_THREADS * c;
        draw[c] = a.dvec[j < nr ? j : 0];
        vraw[c] = a.vvec[j < qcap ? j : 0];
    }
    if (phase >= 2 || rec.kind == 5) return;
    const int kind = rec.kind;
    const int nrows = mg + nq;
    const bool moves = (kind == 1 || kind == 2) && !rec.dependent;
    const bool leaves = kind == 2 || kind == 3;
    const double slack = FEASIBLE * ynorm;
    // s_c: the incoming normal's tail (d2, on the coordinates >= q0) and the leaving reflector's vector (< q0)
#pragma unroll
    for (int c = 0; c < (TAIL + ROWS_WAVES - 1) / ROWS_WAVES; ++c) {
        const int j = tid + ROWS_THREADS * c;
        if (j < 64 * TAIL)
            s_c[j] = j < rec.q ? ((leaves && j < qcap) ? vraw[c] : 0.0) : ((moves && j < nr) ? draw[c] : 0.0);
    }
    __syncthreads();
    // (volatile: the vector is loop-invariant, and the compiler would otherwise pull it back into 2 * TAIL registers)
    const volatile double* vc = s_c;
    // the first q0 coordinates of a row matter when a row leaves (its reflector lives there) and for the values
    // after a warm start; the tail when the incoming row moves the point - a full step at q0 = 300 of 468
    // coordinates streams a third of the matrix
    const bool head = leaves || kind == 4, tail = moves;
    const int extra = leaves ? rec.q - 1 : 0;              // rows of the inverse that stay (positions after the shift)
    const int last = nrows + extra, stride = a.G2 * ROWS_WAVES;
    double best = INFINITY;
    int besti = 0x7fffffff;

    for (int r = w * ROWS_WAVES + wave; r <= last; r += stride) {
        // (the grid gives every wavefront ONE row as a rule.  The masks below compare lane + 64 e with q0, nr, len: as
        // loop invariants the compiler kept all 3 * TAIL of them alive in scalar register pairs, spilled those into
        // vector registers and left the kernel one wavefront per SIMD at C5 - rounds 3-4.  An opaque copy per trip
        // makes them values of the trip: they are computed where they are used.)
        int q0 = rec.q, nrv = nr;
        asm volatile("" : "+s"(q0), "+s"(nrv));
        const bool is_y = r == nrows, is_inv = r > nrows;
        double* row = is_inv ? g.RI[0] + (long)a.slot[r - nrows - 1] * qcap : is_y ? g.y : rows_ptr(g, r);
        const int len = is_inv ? q0 : nrv;
        const bool have = early && r == r_first;              // (this wavefront's first row came with the first trip)
        double dot = (is_y || is_inv) ? 0.0 : (have ? dot_e : a.dots[r]);
        double xq = (kind == 1 && !is_inv && !have) ? row[q0] : 0.0;
        if (kind != 0) {
            double x[TAIL];
            // (the two sums of rounds 3-4, term for term: the normal's entries are zero below q0, the reflector's from q0 on)
            double acc_d = 0.0, acc_v = 0.0;
            if (have) {
                double pick = 0.0;
#pragma unroll
                for (int e = 0; e < (EARLY ? TAIL : 1); ++e) {
                    const int j = lane + 64 * e;
                    x[e] = (j < len && (j < q0 ? head : tail)) ? xe[e] : 0.0;
                    if (e == (q0 >> 6)) pick = xe[e];
                }
                if (kind == 1) xq = __shfl(pick, q0 & 63);    // entry q0 of the row: lane q0 % 64 of strip q0 / 64
            } else {
#pragma unroll
                for (int e = 0; e < TAIL; ++e) {
                    const int j = lane + 64 * e;
                    x[e] = (j < len && (j < q0 ? head : tail)) ? row[j] : 0.0;
                }
            }
#pragma unroll
            for (int e = 0; e < TAIL; ++e) {
                const int j = lane + 64 * e;
                const double cj = vc[j];
                acc_d += x[e] * (j < q0 ? 0.0 : cj);
                acc_v += x[e] * (j < q0 ? cj : 0.0);
            }
            if (kind == 4) {
                if (!is_y) {
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e < TAIL; ++e) {
                        const int j = lane + 64 * e;
                        acc += (j < q0) ? x[e] * g.y[j] : 0.0;
                    }
                    dot = wave_sum(acc);
                    if (lane == 0) a.dots[r] = dot;
                }
            } else {
                if (moves && !is_inv && (kind == 1 || !is_y)) {
                    const double gi = wave_sum(acc_d);
                    if (!is_y) {
                        dot += rec.t * gi;
                        if (lane == 0) a.dots[r] = dot;
                    }
                    if (kind == 1) {
                        // reflector of the incoming row on the tail: v = d2 - alpha e_q0
                        const double f = rec.beta * (gi - rec.alpha * xq);
#pragma unroll
                        for (int e = 0; e < TAIL; ++e) {
                            const int j = lane + 64 * e;
                            if (j >= q0 && j < nrv) row[j] = x[e] - f * (vc[j] - (j == q0 ? rec.alpha : 0.0));
                        }
                    }
                }
                if (leaves) {
                    // reflector of the leaving row on the first q0 coordinates
                    const double f = rec.beta_out * wave_sum(acc_v);
#pragma unroll
                    for (int e = 0; e < TAIL; ++e) {
                        const int j = lane + 64 * e;
                        if (j < q0) row[j] = x[e] - f * vc[j];
                    }
                }
            }
        }
        if (is_y || is_inv || leaves) continue;            // the same row p goes on after a removal: no pricing
        if (r < mg) {
            if (g.scale[r] > 0.0 && !g.isact[r]) {
                const double v = (g.bval[r] + dot) / g.scale[r] + g.own[r] + slack;
                if (v < best || (v == best && r < besti)) {
                    best = v;
                    besti = r;
                }
            }
        } else {
            const int lo = r, hi = r + nq;
            if (g.scale[lo] > 0.0 && !g.isact[lo]) {
                const double v = (g.bval[lo] + dot) / g.scale[lo] + g.own[lo] + slack;
                if (v < best || (v == best && lo < besti)) {
                    best = v;
                    besti = lo;
                }
            }
            if (g.scale[hi] > 0.0 && !g.isact[hi]) {
                const double v = (g.bval[hi] - dot) / g.scale[hi] + g.own[hi] + slack;
                if (v < best || (v == best && hi < besti)) {
                    best = v;
                    besti = hi;
                }
            }
        }
    }
    if (!leaves) {
        block_argmin(best, besti, redv, redi);
        if (tid == 0) {
            a.price[w].value = best;
            a.price[w].index = besti;
        }
    }
}

// Rounds 3-4's form of the pass (both vectors and the lane masks in registers, the row requested after the decision).
// Measured against the form above (profiles/r05_rows_ab.txt; same bits): rows of up to 512 coordinates (TAIL = 8, C3)
// 6.8-7.0 us against 8.0 - the barrier behind the vector's way through LDS costs more than the round trip it saves, and
// 104 VGPRs never were the problem at this size: THIS form stays the default there; 589 coordinates (TAIL = 16, C4) 10.9 us
// against 9.0: the form above.  OGSQP_ROWS=r4 / lds force one of them for rows of up to 1024 coordinates.
template <int TAIL>
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_apply_r4(RowsArgs a) {
    __shared__ double redv[ROWS_WAVES];
    __shared__ int redi[ROWS_WAVES];
    const GiArgs& g = a.g;
    GiState* st = g.st;
    if (st->phase >= 2) return;
    const RowsDecision rec = *a.rec;
    const int kind = rec.kind;
    if (kind == 5) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap;
    const int q0 = rec.q;
    const int nrows = mg + nq;
    const bool moves = (kind == 1 || kind == 2) && !rec.dependent;
    const bool leaves = kind == 2 || kind == 3;
    const double slack = FEASIBLE * st->ynorm;
    // per lane: the incoming normal's tail (d2, zero on the first q0 coordinates) and the leaving reflector's vector
    double dreg[TAIL], vreg[TAIL];
#pragma unroll
    for (int e = 0; e < TAIL; ++e) {
        const int j = lane + 64 * e;
        dreg[e] = (moves && j >= q0 && j < nr) ? a.dvec[j] : 0.0;
        vreg[e] = (leaves && j < q0) ? a.vvec[j] : 0.0;
    }
    double best = INFINITY;
    int besti = 0x7fffffff;
    const int extra = leaves ? q0 - 1 : 0;                 // rows of the inverse that stay (positions after the shift)
    for (int r = w * ROWS_WAVES + wave; r <= nrows + extra; r += a.G2 * ROWS_WAVES) {
        const bool is_y = r == nrows, is_inv = r > nrows;
        double* row = is_inv ? g.RI[0] + (long)a.slot[r - nrows - 1] * qcap : is_y ? g.y : rows_ptr(g, r);
        const int len = is_inv ? q0 : nr;
        double dot = (is_y || is_inv) ? 0.0 : a.dots[r];
        const double xq = (kind == 1 && !is_inv) ? row[q0] : 0.0;
        if (kind != 0) {
            double x[TAIL];
            double acc_d = 0.0, acc_v = 0.0;
            // the first q0 coordinates of a row matter when a row leaves (its reflector lives there) and for the
            // values after a warm start; the tail when the incoming row moves the point - a full step at q0 = 300
            // of 468 coordinates streams a third of the matrix
            const bool head = leaves || kind == 4, tail = moves;
#pragma unroll
            for (int e = 0; e < TAIL; ++e) {
                const int j = lane + 64 * e;
                x[e] = (j < len && (j < q0 ? head : tail)) ? row[j] : 0.0;
                acc_d += x[e] * dreg[e];
                acc_v += x[e] * vreg[e];
            }
            if (kind == 4) {
                if (!is_y) {
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e < TAIL; ++e) {
                        const int j = lane + 64 * e;
                        acc += (j < q0) ? x[e] * g.y[j] : 0.0;
                    }
                    dot = wave_sum(acc);
                    if (lane == 0) a.dots[r] = dot;
                }
            } else {
                if (moves && !is_inv && (kind == 1 || !is_y)) {
                    const double gi = wave_sum(acc_d);
                    if (!is_y) {
                        dot += rec.t * gi;
                        if (lane == 0) a.dots[r] = dot;
                    }
                    if (kind == 1) {
                        // reflector of the incoming row on the tail: v = d2 - alpha e_q0
                        const double f = rec.beta * (gi - rec.alpha * xq);
#pragma unroll
                        for (int e = 0; e < TAIL; ++e) {
                            const int j = lane + 64 * e;
                            if (j >= q0 && j < nr) row[j] = x[e] - f * (dreg[e] - (j == q0 ? rec.alpha : 0.0));
                        }
                    }
                }
                if (leaves) {
                    // reflector of the leaving row on the first q0 coordinates
                    const double f = rec.beta_out * wave_sum(acc_v);
#pragma unroll
                    for (int e = 0; e < TAIL; ++e) {
                        const int j = lane + 64 * e;
                        if (j < q0) row[j] = x[e] - f * vreg[e];
                    }
                }
            }
        }
        if (is_y || is_inv || leaves) continue;            // the same row p goes on after a removal: no pricing
        if (r < mg) {
            if (g.scale[r] > 0.0 && !g.isact[r]) {
                const double v = (g.bval[r] + dot) / g.scale[r] + g.own[r] + slack;
                if (v < best || (v == best && r < besti)) {
                    best = v;
                    besti = r;
                }
            }
        } else {
            const int lo = r, hi = r + nq;
            if (g.scale[lo] > 0.0 && !g.isact[lo]) {
                const double v = (g.bval[lo] + dot) / g.scale[lo] + g.own[lo] + slack;
                if (v < best || (v == best && lo < besti)) {
                    best = v;
                    besti = lo;
                }
            }
            if (g.scale[hi] > 0.0 && !g.isact[hi]) {
                const double v = (g.bval[hi] - dot) / g.scale[hi] + g.own[hi] + slack;
                if (v < best || (v == best && hi < besti)) {
                    best = v;
                    besti = hi;
                }
            }
        }
    }
    if (!leaves) {
        block_argmin(best, besti, redv, redi);
        if (tid == 0) {
            a.price[w].value = best;
            a.price[w].index = besti;
        }
    }
}

// Long rows (more than 16 x 64 null-space coordinates: C5's 2017).  Holding such a row in registers - 2 * TAIL of them,
// with TAIL lane masks per comparison - cost the register kernel its occupancy: 256 VGPRs + 232 AGPRs, ONE wavefront
// per SIMD, the 8 199 rows of C5 in eight residency rounds of round trips, 80 us per change for 143 MB (1.8 TB/s).
// Here the row is STREAMED in strips of 64 coordinates with run-time bounds - only the strips the change touches: the
// tail from q0 on when the point moves, the head below q0 when a row leaves -, the sums in the register kernel's order
// (lane l adds its coordinates l, l + 64, ... in ascending order; strips outside the range contributed exact zeros
// there), a few dozen registers, eight loads in flight per wavefront.  What the second pass needs of the row again comes
// out of LDS (STAGE: one row per wavefront next to the shared vector - 5 x 8 nr bytes per workgroup, two workgroups per
// compute unit at C5) or, for rows too long for that, from the caches.  Same bits as k_rows_apply<TAIL>
// (OGSQP_ROWS=reg selects the register kernels for every length; tests/test_slsqp_core.py compares the two).
template <bool STAGE>
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_apply_stream(RowsArgs a) {
    extern __shared__ double lds[];
    __shared__ double redv[ROWS_WAVES];
    __shared__ int redi[ROWS_WAVES];
    const GiArgs& g = a.g;
    GiState* st = g.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap;
    const int nrp = (nr + 63) & ~63;
    double* s_c = lds;
    double* s_x = lds + nrp + (STAGE ? wave * nrp : 0);
    const int phase = st->phase;
    const double ynorm = st->ynorm;
    const RowsDecision rec = *a.rec;
    if (phase >= 2 || rec.kind == 5) return;
    const int kind = rec.kind;
    const int q0 = rec.q;
    const int nrows = mg + nq;
    const bool moves = (kind == 1 || kind == 2) && !rec.dependent;
    const bool leaves = kind == 2 || kind == 3;
    const double slack = FEASIBLE * ynorm;
    // the incoming normal's tail (coordinates >= q0) and the leaving reflector's vector (< q0) in one array
    for (int j = tid; j < nrp; j += ROWS_THREADS) {
        const double dv = a.dvec[j < nr ? j : 0], vv = a.vvec[j < qcap ? j : 0];
        s_c[j] = j < q0 ? ((leaves && j < qcap) ? vv : 0.0) : ((moves && j < nr) ? dv : 0.0);
    }
    __syncthreads();
    const bool head = leaves || kind == 4, tail = moves;
    const int extra = leaves ? q0 - 1 : 0;
    const int last = nrows + extra, stride = a.G2 * ROWS_WAVES;
    double best = INFINITY;
    int besti = 0x7fffffff;
    for (int r = w * ROWS_WAVES + wave; r <= last; r += stride) {
        const bool is_y = r == nrows, is_inv = r > nrows;
        double* row = is_inv ? g.RI[0] + (long)a.slot[r - nrows - 1] * qcap : is_y ? g.y : rows_ptr(g, r);
        const int len = is_inv ? q0 : nr;
        double dot = (is_y || is_inv) ? 0.0 : a.dots[r];
        const double xq = (kind == 1 && !is_inv) ? row[q0] : 0.0;
        if (kind != 0) {
            // strips [e_lo, e_hi) of 64 coordinates hold everything the change reads of this row
            const int e_lo = head ? 0 : q0 >> 6;
            const int e_hi = tail ? (len + 63) >> 6 : (min(q0, len) + 63) >> 6;
            double acc_d = 0.0, acc_v = 0.0, acc_y = 0.0;
#pragma unroll 8
            for (int e = e_lo; e < e_hi; ++e) {
                const int j = lane + 64 * e;
                const double xv = (j < len && (j < q0 ? head : tail)) ? row[j] : 0.0;
                if (STAGE) s_x[j] = xv;
                const double cj = s_c[j];
                acc_d += xv * (j < q0 ? 0.0 : cj);
                acc_v += xv * (j < q0 ? cj : 0.0);
                if (kind == 4) acc_y += (j < q0) ? xv * g.y[j] : 0.0;
            }
            if (kind == 4) {
                if (!is_y) {
                    dot = wave_sum(acc_y);
                    if (lane == 0) a.dots[r] = dot;
                }
            } else {
                if (moves && !is_inv && (kind == 1 || !is_y)) {
                    const double gi = wave_sum(acc_d);
                    if (!is_y) {
                        dot += rec.t * gi;
                        if (lane == 0) a.dots[r] = dot;
                    }
                    if (kind == 1) {
                        // reflector of the incoming row on the tail: v = d2 - alpha e_q0
                        const double f = rec.beta * (gi - rec.alpha * xq);
                        const int t_hi = (nr + 63) >> 6;
#pragma unroll 8
                        for (int e = q0 >> 6; e < t_hi; ++e) {
                            const int j = lane + 64 * e;
                            if (j >= q0 && j < nr) {
                                const double xv = STAGE ? s_x[j] : row[j];
                                row[j] = xv - f * (s_c[j] - (j == q0 ? rec.alpha : 0.0));
                            }
                        }
                    }
                }
                if (leaves) {
                    // reflector of the leaving row on the first q0 coordinates
                    const double f = rec.beta_out * wave_sum(acc_v);
                    const int h_hi = (q0 + 63) >> 6;
#pragma unroll 8
                    for (int e = 0; e < h_hi; ++e) {
                        const int j = lane + 64 * e;
                        if (j < q0) {
                            const double xv = STAGE ? s_x[j] : row[j];
                            row[j] = xv - f * s_c[j];
                        }
                    }
                }
            }
        }
        if (is_y || is_inv || leaves) continue;            // the same row p goes on after a removal: no pricing
        if (r < mg) {
            if (g.scale[r] > 0.0 && !g.isact[r]) {
                const double v = (g.bval[r] + dot) / g.scale[r] + g.own[r] + slack;
                if (v < best || (v == best && r < besti)) {
                    best = v;
                    besti = r;
                }
            }
        } else {
            const int lo = r, hi = r + nq;
            if (g.scale[lo] > 0.0 && !g.isact[lo]) {
                const double v = (g.bval[lo] + dot) / g.scale[lo] + g.own[lo] + slack;
                if (v < best || (v == best && lo < besti)) {
                    best = v;
                    besti = lo;
                }
            }
            if (g.scale[hi] > 0.0 && !g.isact[hi]) {
                const double v = (g.bval[hi] - dot) / g.scale[hi] + g.own[hi] + slack;
                if (v < best || (v == best && hi < besti)) {
                    best = v;
                    besti = hi;
                }
            }
        }
    }
    if (!leaves) {
        block_argmin(best, besti, redv, redi);
        if (tid == 0) {
            a.price[w].value = best;
            a.price[w].index = besti;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Start of the active-set loop.  nwarm = 0: empty active set.  Otherwise the rows warm[0 .. nwarm) were appended
// to the LQ sweep: row j of the extension holds L(j, 0..j) at Tc[(meq + j) * ld + meq + i], its diagonal at
// diagL[meq + j]; the matrix of the active normals is L'.  A vanishing pivot (a warm row that depends on the
// equalities or on the warm rows before it) falls back to the empty set - the coordinates are as good as any.
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_init(RowsArgs a, const double* __restrict__ diagL,
                                                           const int* __restrict__ warm, int nwarm,
                                                           const double* __restrict__ dthresh, const int* flag) {
    __shared__ int s_dead;
    const GiArgs& g = a.g;
    const int tid = threadIdx.x;
    const long i = (long)blockIdx.x * blockDim.x + tid;
    const int mt = g.mg + 2 * g.nq;
    if (i < mt) {
        g.u[i] = 0.0;
        g.isact[i] = 0;
    }
    if (i < g.nr) g.y[i] = 0.0;
    if (i < g.mg + g.nq) a.dots[i] = 0.0;
    if (i < g.qcap) a.slot[i] = (int)i;
    if (blockIdx.x != gridDim.x - 1) return;                   // the last workgroup owns the state
    if (tid == 0) s_dead = 0;
    __syncthreads();
    const double tiny = dthresh[0];
    for (int j = tid; j < nwarm; j += ROWS_THREADS) {
        const int c = warm[j];
        if (!(fabs(diagL[g.meq + j]) > tiny) || !(g.scale[c] > 0.0)) s_dead = 1;
    }
    __syncthreads();
    const int q = (s_dead || flag[1]) ? 0 : nwarm;
    for (int j = tid; j < q; j += ROWS_THREADS) g.act[j] = warm[j];
    if (tid == 0) {
        GiState s;
        s.phase = flag[1] ? 4 : (q > 0 ? -1 : 0);
        s.p = -1;
        s.q = q;
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
        RowsDecision rec;
        rec.kind = 0;
        rec.q = 0;
        rec.k = 0;
        rec.dependent = 0;
        rec.t = rec.alpha = rec.beta = rec.beta_out = 0.0;
        *a.rec = rec;
    }
}

// isact of the warm rows (after k_rows_init decided whether they are used)
__global__ void k_rows_mark(RowsArgs a, const int* __restrict__ warm, int nwarm) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nwarm && j < a.g.st->q && a.g.st->phase < 0) a.g.isact[warm[j]] = 1;
}

// The inverse of the warm start's triangle: the matrix of the active normals is M = L' (upper triangular,
// M(i, l) = L(l, i): column l of M is row l of the sweep's extension, contiguous).  One workgroup per column c of
// the inverse: x = e_c, then for l = c .. 0: x_l /= M(l, l), x[0..l) -= x_l M(0..l, l).  Row i of RI (slot i) gets x_i.
__global__ __launch_bounds__(ROWS_THREADS) void k_rows_invert(RowsArgs a, const double* __restrict__ Tc,
                                                             const double* __restrict__ diagL) {
    extern __shared__ double lds[];
    const GiArgs& g = a.g;
    const GiState* st = g.st;
    if (st->phase >= 0) return;
    const int q = st->q, c = blockIdx.x, qcap = g.qcap, tid = threadIdx.x;
    if (c >= q) return;
    double* x = lds;
    for (int i = tid; i <= c; i += ROWS_THREADS) x[i] = i == c ? 1.0 : 0.0;
    __syncthreads();
    for (int l = c; l >= 0; --l) {
        const double* col = Tc + (long)(g.meq + l) * g.ld + g.meq;
        const double xl = x[l] / diagL[g.meq + l];
        __syncthreads();
        if (tid == 0) x[l] = xl;
        for (int i = tid; i < l; i += ROWS_THREADS) x[i] -= xl * col[i];
        __syncthreads();
    }
    double* RI = g.RI[0];
    for (int i = tid; i < q; i += ROWS_THREADS) RI[(long)i * qcap + c] = i <= c ? x[i] : 0.0;
}

// Rows of the warm start that are bounds: +- rows of the work factor (before the sweep).
__global__ void k_rows_gather_bounds(const double* __restrict__ Jw, int ld, int nq, int mg, const int* __restrict__ warm,
                                     int first, int nwarm, double* __restrict__ Text) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, j = first + blockIdx.y;
    if (k >= nq || j >= nwarm) return;
    const int c = warm[j] - mg;                                // lower bound of variable c, or upper bound of c - nq
    const double sign = c < nq ? 1.0 : -1.0;
    const int i = c < nq ? c : c - nq;
    Text[(long)j * ld + k] = sign * Jw[(long)i * ld + k];
}
