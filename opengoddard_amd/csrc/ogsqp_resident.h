// ogsqp_resident.h - the active-set loop of ogsqp_rows.h as ONE launch whose rows never leave the compute units
// (included by ogsqp.hip inside its anonymous namespace, after ogsqp_rows.h).
//
// k_rows_decide + k_rows_apply pay per change for two kernel boundaries and for a chain of dependent trips through
// memory: state word -> price partials -> row p -> product -> store acknowledged -> ticket -> the others' r ->
// decision -> (boundary) -> decision + vectors + row -> stores -> (boundary): 19-23 us at C3, of which the bytes are
// nothing.  What the chain carries is small - one row of W, the dual direction, a few scalars - and what it walks
// over fits the chip: the stack W = [G; +-I] Y plus the inverse RI are (mg + nq + qcap) rows of a few hundred
// null-space coordinates, one WAVEFRONT PER ROW with the row in its compute unit's LDS (15 rows of 4 KB at C3: 60 of
// the 160 KB) is 2 551 wavefronts at C3 and 15 of them per compute unit is 171 workgroups.  (The rows were in
// registers first: 16 VGPRs of a budget of 128 - and a kernel of 15 000 instructions with every loop unrolled over the
// row's strips, five dozen live lane masks and 59-117 spilled vector registers.  Out of LDS the loops are loops and
// the kernel has no template parameter.)  So:
//
//   * the grid is one workgroup of 16 wavefronts per 15 rows, all resident at once (<= one per compute unit); every
//     wavefront loads its row once, keeps it in LDS - and its constraint value W[i] y in registers - over ALL changes
//     of the subproblem and stores it once at the end; the sixteenth wavefront of a workgroup carries y;
//   * what Goldfarb & Idnani's step needs of the shared state - y, the multipliers, the lists, the incoming normal, the
//     dual direction, the leaving reflector - is REPLICATED in every workgroup's LDS and every workgroup takes the
//     same decision from the same numbers in the same order (the sums in k_rows_decide's order: 256 threads striding
//     the vector, wave_sum, four partial sums added in turn): nobody waits for a deciding workgroup;
//   * what the workgroups tell each other per change is three small messages - (1) each workgroup's most violated row:
//     its price and, with it, the ROW ITSELF (so that whoever reads the prices finds the winner's row already there:
//     one trip through memory instead of election -> owner publishes -> everybody reads) - and the price goes out as
//     soon as the step of the change before is known, BEFORE the pass over the rows (a row's price depends on its
//     value W[i] y, which moves by t g_i, not on the row): the pass and its barrier hide behind the exchange -, (3) the entries of
//     r = RI d1 from the owners of the inverse's rows, (4) on a partial step the leaving row of the inverse from its
//     owner; after a partial step the same incoming row goes on and its owner alone publishes it again (2) - through
//     a mailbox in HBM whose records validate themselves: a double travels as two 64-bit words, each carrying 32 bits
//     of payload and the 32-bit number of its exchange (one 16-byte store, one 16-byte load, agent scope).  A reader
//     spins on the record it needs and has the value the moment it lands: ONE trip through memory per message, no
//     counter, no fence, no re-arming, nothing to reset between launches (the exchange numbers go on counting in a
//     device word).  Every kind of message has two areas used alternately, and between two uses of an area there is
//     always an exchange in which EVERY workgroup publishes after it has read the earlier one: (1) is all-to-all, and
//     with (4) every workgroup leaves a record saying it has come this far, which everybody waits for - a chain of
//     partial steps has no (1).  So nobody overwrites what a slow reader is still reading.
//
// Same arithmetic in the same order as the two-launch form - pricing, election (lowest index wins a tie), norms,
// r = RI d1 lane by lane, ratio test, step, u, y, |y|, the reflectors and their application row by row - so the
// iterates are its BITS (tests/test_slsqp_core.py::test_gpu_resident_active_set_gives_the_bits_of_the_two_launch_form)
// and the change counts still equal the restatement's.  Per change at C3: see DESIGN.md section 9 / profiles/r06_*.
//
// The warm start's removals (phase -1: two products with the whole inverse per removal) stay with k_rows_decide /
// k_rows_apply, which the host runs - in `only_warm` form: they return at once when there is nothing of that kind to
// do - in front of this kernel.  A workgroup that is not resident never answers: waits are bounded, a wait that
// gives up raises the sweep's `lost` flag, every workgroup leaves WITHOUT writing anything back (the state in memory
// is the state the kernel started from) and the host goes on with the two-launch form.

constexpr int RES_THREADS = 1024;
constexpr int RES_WAVES = RES_THREADS / 64;
constexpr int RES_ROWS = RES_WAVES - 1;             // rows per workgroup; the last wavefront carries y
constexpr int RES_MAX_WG = 256;                     // one workgroup per compute unit at most
constexpr int RES_MAX_LEN = 1024;                   // records of a vector message
constexpr int RES_VEC_CAP = RES_MAX_LEN + RES_MAX_WG + 64;   // a vector, and one record per workgroup behind it
typedef unsigned long long res_u64;
typedef res_u64 res_rec __attribute__((ext_vector_type(2)));

// records per workgroup of area (1): price, index, the row's value, its b, the row
__host__ __device__ inline size_t res_e1_stride(int nr) { return (size_t)nr + 4; }
__host__ __device__ inline size_t res_mail_records(int workgroups, int nr) {
    return 2 * (size_t)workgroups * res_e1_stride(nr) + 3 * 2 * (size_t)RES_VEC_CAP;
}

struct ResArgs {
    RowsArgs r;
    res_u64* mail;                   // res_mail_records records of two 64-bit words
    unsigned* seq;                   // 4 exchange counters that go on counting from launch to launch
    int NW;                          // workgroups = ceil((mg + nq + qcap) / 15)
};

// LDS of a workgroup: 15 rows (of the stack: nr coordinates; of the inverse: qcap), two vectors of the null space, three
// of the active set, four lists
__host__ __device__ inline size_t res_lds_bytes(int nr, int qcap) {
    const size_t nrp = (size_t)(nr + 63) & ~(size_t)63, qcp = (size_t)(qcap + 63) & ~(size_t)63;
    const size_t rowp = nrp > qcp ? nrp : qcp;
    return (RES_ROWS * rowp + 2 * nrp + 3 * qcp) * sizeof(double) + 4 * qcp * sizeof(int) + 64;
}

__device__ __forceinline__ void mb_put(res_u64* rec, double v, unsigned tag) {
    const res_u64 b = (res_u64)__double_as_longlong(v), t = (res_u64)tag << 32;
    res_rec r;
    r.x = t | (b & 0xffffffffull);
    r.y = t | (b >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(rec), "v"(r) : "memory");
}

// -> the value of exchange `tag`; polls until both words carry the tag.  ok = false: gave up (flag raised).
__device__ __forceinline__ double mb_get(const res_u64* rec, unsigned tag, int spin_limit, int* lost, bool& ok) {
    int spins = 0;
    while (true) {
        res_rec r;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(rec) : "memory");
        if ((unsigned)(r.x >> 32) == tag && (unsigned)(r.y >> 32) == tag)
            return __longlong_as_double((long long)((r.x & 0xffffffffull) | (r.y << 32)));
        ++spins;
        if (spins > spin_limit || ((spins & 127) == 0 && __hip_atomic_load(lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(lost, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
            return 0.0;
        }
    }
}

// Sum over the first 256 threads of the workgroup in block_sum's order for a 256-thread workgroup (k_rows_decide's):
// wave_sum, then the four wavefronts' sums added in turn.  Every thread of the workgroup gets the total.
// (no barrier in front: between two uses of red4 in the loop below there always is one)
__device__ __forceinline__ double res_sum256(double v, double* red4) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red4[threadIdx.x >> 6] = v;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) total += red4[w];
    return total;
}

// (value, index) minimum, ties to the lower index, over the wavefront - the result of block_argmin's tree (the order
// is total, so any tree gives it), but through DPP moves instead of 18 dependent ds_bpermute round trips per wavefront
// (block_argmin's __shfl_xor: 1.2 us of the change, twice per change).
template <int CTRL>
__device__ __forceinline__ void argmin_step(double& v, int& idx) {
    const double ov = dpp_f64<CTRL>(v);
    const int oi = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xf, 0xf, true);
    if (ov < v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
    }
}
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
    argmin_step<0xB1>(v, idx);     // quad_perm [1,0,3,2]
    argmin_step<0x4E>(v, idx);     // quad_perm [2,3,0,1]
    argmin_step<0x124>(v, idx);    // row_ror 4
    argmin_step<0x128>(v, idx);    // row_ror 8: every lane holds the minimum of its row of 16
    double bv = lane_f64(v, 0);
    int bi = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ov = lane_f64(v, r);
        const int oi = __builtin_amdgcn_readlane(idx, r);
        if (ov < bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    v = bv;
    idx = bi;
}
// ... over what the wavefronts left in redv / redi (one entry each, a barrier behind them): every wavefront for itself
__device__ __forceinline__ void res_argmin_of_waves(double& v, int& idx, const double* redv, const int* redi) {
    const int l = threadIdx.x & 63;
    v = l < RES_WAVES ? redv[l] : INFINITY;
    idx = l < RES_WAVES ? redi[l] : 0x7fffffff;
    wave_argmin(v, idx);
}

// A value every lane of the wavefront holds alike, moved to scalar registers (the loop carries two dozen of them - the
// state of the method, a row's b / scale / own: in vector registers they crowd the register file).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__global__ __launch_bounds__(RES_THREADS) void k_rows_resident(ResArgs A) {
    extern __shared__ double lds[];
    __shared__ double redv[RES_WAVES];
    __shared__ int redi[RES_WAVES];
    __shared__ double red4[4], red8[8];
    __shared__ int redi4[4];
    __shared__ double s_msg[2];                      // what travels with the incoming row: its constraint value, its b
    const RowsArgs& a = A.r;
    const GiArgs& g = a.g;
    GiState* st = g.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int nr = g.nr, mg = g.mg, nq = g.nq, qcap = g.qcap, nrows = mg + nq, NW = A.NW;
    const int nrp = (nr + 63) & ~63, qcp = (qcap + 63) & ~63;
    const int phase0 = st->phase;
    if (phase0 < 0 || phase0 >= 2) return;           // (the warm start is not over / nothing left to do: every workgroup sees the same)
    int q = uni(st->q), iters = uni(st->iters), p = uni(st->p), phase = uni(phase0);
    const int warm_removals = iters;                 // (what the two-launch pairs in front did: the host's hint for the next subproblem)
    double up = uni(st->up), ynorm = uni(st->ynorm);
    unsigned c1 = uni(A.seq[0]), c2 = uni(A.seq[1]), c3 = uni(A.seq[2]), c4 = uni(A.seq[3]);
    const int spin_limit = a.spin_limit;
    int* lost = a.lost;

    double* d = lds;                                 // nrp: incoming normal (signed), zero behind nr
    double* y = d + nrp;                             // nrp
    double* rv = y + nrp;                            // qcp: dual direction
    double* aux = rv + qcp;                          // qcp: leaving row of the inverse -> its reflector's vector
    double* upos = aux + qcp;                        // qcp: multipliers by active position
    int* act = (int*)(upos + qcp);                   // qcp
    int* slot = act + qcp;                           // qcp: storage row of the inverse per position (a permutation)
    int* posof = slot + qcp;                         // qcp: its inverse
    int* scratch = posof + qcp;                      // qcp
    const int rowp = nrp > qcp ? nrp : qcp;
    double* xr = (double*)(scratch + qcp) + 8 + (size_t)(wave < RES_ROWS ? wave : 0) * rowp;      // this wavefront's row

    const size_t S1 = res_e1_stride(nr);
    res_u64* const mailE1 = A.mail;
    res_u64* const mailE2 = mailE1 + 2 * (2 * (size_t)NW * S1);
    res_u64* const mailE3 = mailE2 + 2 * (size_t)(2 * RES_VEC_CAP);
    res_u64* const mailE4 = mailE3 + 2 * (size_t)(2 * RES_VEC_CAP);

    // ---- what this wavefront owns -----------------------------------------------------------------------------
    const bool carrier = wave < RES_ROWS;
    const int gw = carrier ? w * RES_ROWS + wave : -1;
    const bool is_row = carrier && gw < nrows;       // a row of the stack: general row gw, or the bounds of variable gw - mg
    const bool is_inv = carrier && gw >= nrows && gw < nrows + qcap;
    const int myslot = gw - nrows;                   // storage row of the inverse
    double dot = 0.0;
    double b0 = 0.0, s0 = 0.0, o0 = 0.0, b1 = 0.0, s1 = 0.0, o1 = 0.0;
    int ia0 = 1, ia1 = 1;
    if (carrier) {
        const double* src = is_row ? rows_ptr(g, gw) : g.RI[0] + (long)(is_inv ? myslot : 0) * qcap;
        const int len = is_row ? nr : (is_inv ? qcap : 0);
        for (int j = lane; j < rowp; j += 64) xr[j] = j < len ? src[j] : 0.0;
        if (is_row) {
            dot = uni(a.dots[gw]);
            b0 = uni(g.bval[gw]); s0 = uni(g.scale[gw]); o0 = uni(g.own[gw]); ia0 = uni(g.isact[gw]);
            if (gw >= mg) {
                const int hi = gw + nq;
                b1 = uni(g.bval[hi]); s1 = uni(g.scale[hi]); o1 = uni(g.own[hi]); ia1 = uni(g.isact[hi]);
            }
        }
    }
    for (int i = tid; i < nrp; i += RES_THREADS) {
        y[i] = i < nr ? g.y[i] : 0.0;
        d[i] = 0.0;
    }
    for (int i = tid; i < qcp; i += RES_THREADS) {
        const int ai = i < q ? g.act[i] : 0;
        act[i] = ai;
        upos[i] = i < q ? g.u[ai] : 0.0;
        slot[i] = i < qcap ? a.slot[i] : 0;
        aux[i] = 0.0;
        rv[i] = 0.0;
    }
    __syncthreads();
    for (int i = tid; i < qcap; i += RES_THREADS) posof[slot[i]] = i;
    __syncthreads();

#ifdef OGSQP_TRACE
    // (the clocks of two lanes of workgroup 0 - the first wavefront's and the last one's - kept in LDS: sixteen 64-bit
    // counters per thread in registers cost the traced build its place in the register file)
    __shared__ long long s_tr[32];
    if (tid < 32) s_tr[tid] = 0;
    long long t_mark = __builtin_amdgcn_s_memrealtime();
    long long n_changes = 0, n_partial = 0;
#define SMARK(slot_) do { const long long now_ = __builtin_amdgcn_s_memrealtime(); \
        if (lane == 0 && (wave == 0 || wave == RES_WAVES - 1)) s_tr[(slot_) + (wave == 0 ? 0 : 16)] += now_ - t_mark; \
        t_mark = now_; } while (0)
#else
#define SMARK(slot_) do { } while (0)
#endif
    // what this wavefront's row is worth (k_rows_apply's pricing): into redv / redi, for the election of the next change
    auto price_row = [&]() {
        double best = INFINITY;
        int besti = 0x7fffffff;
        if (is_row) {
            const double slack = FEASIBLE * ynorm;
            if (gw < mg) {
                if (s0 > 0.0 && !ia0) {
                    best = (b0 + dot) / s0 + o0 + slack;
                    besti = gw;
                }
            } else {
                const int lo = gw, hi = gw + nq;
                if (s0 > 0.0 && !ia0) {
                    const double v = (b0 + dot) / s0 + o0 + slack;
                    if (v < best || (v == best && lo < besti)) {
                        best = v;
                        besti = lo;
                    }
                }
                if (s1 > 0.0 && !ia1) {
                    const double v = (b1 - dot) / s1 + o1 + slack;
                    if (v < best || (v == best && hi < besti)) {
                        best = v;
                        besti = hi;
                    }
                }
            }
        }
        if (!(best < INFINITY)) {                    // (a NaN price is no candidate, as `v < best` has it in k_rows_apply)
            best = INFINITY;
            besti = 0x7fffffff;
        }
        if (lane == 0) {
            redv[wave] = best;
            redi[wave] = besti;
        }
    };
    // the workgroup's best of what its wavefronts left in redv / redi goes out as exchange (1) - price and index at once,
    // the row itself (publish_best_row) once it holds what the change in progress makes of it
    double bv_pub = INFINITY;
    int bi_pub = 0x7fffffff;
    auto publish_best = [&]() {
        res_argmin_of_waves(bv_pub, bi_pub, redv, redi);
        bv_pub = uni(bv_pub);
        bi_pub = uni(bi_pub);
        ++c1;
        res_u64* const mine = mailE1 + 2 * (((c1 & 1u) * (size_t)NW + (size_t)w) * S1);
        if (tid == RES_THREADS - 64) {
            mb_put(mine, bv_pub, c1);
            mb_put(mine + 2, (double)bi_pub, c1);
        }
    };
    auto publish_best_row = [&]() {
        if (!(bv_pub < 0.0)) return;
        const int brow = bi_pub < nrows ? bi_pub : bi_pub - nq;
        if (gw != brow) return;
        res_u64* const mine = mailE1 + 2 * (((c1 & 1u) * (size_t)NW + (size_t)w) * S1);
        const double sg = bi_pub < nrows ? 1.0 : -1.0;
#pragma unroll 4
        for (int j = lane; j < nr; j += 64) mb_put(mine + 2 * (size_t)(4 + j), sg * xr[j], c1);
        if (lane == 0) {
            mb_put(mine + 4, dot, c1);
            mb_put(mine + 6, bi_pub < nrows ? b0 : b1, c1);
        }
    };
    if (phase == 0) {
        price_row();
        __syncthreads();
        publish_best();
        publish_best_row();
    }
    __syncthreads();
    int exit_dbg = 0, exit_dbg2 = 0;
    while (true) {
        int q0;
        double psign;
        int prow_index;
        if (phase == 0) {
            // ---- (1) who comes in: every workgroup's best is under way since the step of the change before was known
            // (its price and index BEFORE the pass over the rows, the row itself behind it): the election reads them
            res_u64* const area = mailE1 + 2 * (((c1 & 1u) * (size_t)NW) * S1);
            SMARK(0);
            bool ok = true;
            double v = INFINITY;
            int idx = 0x7fffffff;
            if (tid < NW) {
                const res_u64* rec = area + 2 * ((size_t)tid * S1);
                v = mb_get(rec, c1, spin_limit, lost, ok);
                idx = (int)mb_get(rec + 2, c1, spin_limit, lost, ok);
            }
            if (tid < RES_MAX_WG) {                  // (the wavefronts that may hold candidates)
                wave_argmin(v, idx);
                if (lane == 0) {
                    red4[wave] = v;
                    redi4[wave] = idx;
                }
            }
            if (__syncthreads_or(ok ? 0 : 1)) return;
            v = red4[0];
            idx = redi4[0];
#pragma unroll
            for (int k = 1; k < RES_MAX_WG / 64; ++k)
                if (red4[k] < v || (red4[k] == v && redi4[k] < idx)) {
                    v = red4[k];
                    idx = redi4[k];
                }
            v = uni(v);
            idx = uni(idx);
            SMARK(1);
            if (!(v < 0.0)) {                        // solved
                phase = 2;
                break;
            }
            p = idx;
            iters += 1;
            if (iters > g.limit || q > nr || q > qcap || p < 0 || p >= mg + 2 * nq) {
                phase = 3;
                break;
            }
            q0 = q;
            prow_index = p < nrows ? p : p - nq;
            psign = p < nrows ? 1.0 : -1.0;
            // the winner's row lies in its workgroup's record
            const res_u64* from = area + 2 * ((size_t)(prow_index / RES_ROWS) * S1);
            for (int j = tid; j < nr + 2; j += RES_THREADS) {
                if (j < nr) d[j] = mb_get(from + 2 * (size_t)(4 + j), c1, spin_limit, lost, ok);
                else s_msg[j - nr] = mb_get(from + 2 * (size_t)(2 + (j - nr)), c1, spin_limit, lost, ok);
            }
            if (__syncthreads_or(ok ? 0 : 1)) return;
            SMARK(2);
        } else {
            // ---- (2) after a partial step the same row comes in again: from its owner ----------------------------
            iters += 1;
            if (iters > g.limit || q > nr || q > qcap || p < 0 || p >= mg + 2 * nq) {
                phase = 3;
                break;
            }
            q0 = q;
            prow_index = p < nrows ? p : p - nq;
            psign = p < nrows ? 1.0 : -1.0;
            ++c2;
            res_u64* const area = mailE2 + 2 * (size_t)((c2 & 1u) * RES_VEC_CAP);
            if (gw == prow_index) {
#pragma unroll 4
                for (int j = lane; j < nr; j += 64) mb_put(area + 2 * (size_t)j, psign * xr[j], c2);
                if (lane == 0) {
                    mb_put(area + 2 * (size_t)nr, dot, c2);
                    mb_put(area + 2 * (size_t)(nr + 1), p < nrows ? b0 : b1, c2);
                }
            }
            bool ok = true;
            for (int j = tid; j < nr + 2; j += RES_THREADS) {
                const double v = mb_get(area + 2 * (size_t)j, c2, spin_limit, lost, ok);
                if (j < nr) d[j] = v;
                else s_msg[j - nr] = v;
            }
            if (__syncthreads_or(ok ? 0 : 1)) return;
            SMARK(3);
        }
        const double dots_p = uni(s_msg[0]), bval_p = uni(s_msg[1]);
        const int e0 = q0 >> 6, ne = nrp >> 6, eh = (q0 + 63) >> 6;      // strips of 64 coordinates: tail from e0, head below eh
        // ---- (3) r = RI d1: every owner of a row of the inverse its entry ------------------------------------
        double my_r = 0.0, gi_early = 0.0;
        int mypos = is_inv ? uni(posof[myslot]) : 0x7fffffff;
        {
            ++c3;
            res_u64* const area = mailE3 + 2 * (size_t)((c3 & 1u) * RES_VEC_CAP);
            if (is_inv && mypos < q0) {
                double acc = 0.0;
#pragma unroll 4
                for (int e = 0; e < eh; ++e) {
                    const int j = lane + 64 * e;
                    const double xv = xr[j], dv = d[j];
                    acc += j < q0 ? xv * dv : 0.0;
                }
                my_r = uni(wave_sum(acc));
                if (lane == 0) mb_put(area + 2 * (size_t)mypos, my_r, c3);
            }
            SMARK(5);
            // this wavefront's row times the incoming normal's tail: wanted by the pass below whenever the point moves, and
            // independent of r - formed while r is under way (k_rows_apply_r4's sum, lane by lane in ascending order)
            // (not y's: the step moves y before the pass)
            if (is_row) {
                double acc_d = 0.0;
#pragma unroll 4
                for (int e = e0; e < ne; ++e) {
                    const int j = lane + 64 * e;
                    const double xv = xr[j], dv = d[j];
                    acc_d += j >= q0 ? xv * dv : 0.0;
                }
                gi_early = uni(wave_sum(acc_d));
            }
            // |d2|^2 and |d|^2 in k_rows_decide's order (256 threads striding the vector), by the last four wavefronts
            // while the first ones wait for r
            if (tid >= RES_THREADS - 256) {
                const int t256 = tid - (RES_THREADS - 256);
                double part_zz = 0.0, part_nn = 0.0;
                for (int i = t256; i < nr; i += 256) {
                    const double v = d[i];
                    part_nn += v * v;
                    if (i >= q0) part_zz += v * v;
                }
                part_zz = wave_sum(part_zz);
                part_nn = wave_sum(part_nn);
                if (lane == 0) {
                    red8[wave - (RES_WAVES - 4)] = part_zz;
                    red8[4 + wave - (RES_WAVES - 4)] = part_nn;
                }
            }
            bool ok = true;
            for (int j = tid; j < q0; j += RES_THREADS) rv[j] = mb_get(area + 2 * (size_t)j, c3, spin_limit, lost, ok);
            if (__syncthreads_or(ok ? 0 : 1)) return;
        }
        double zz = 0.0, nn = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
            zz += red8[wv];
            nn += red8[4 + wv];
        }
        zz = uni(zz);
        nn = uni(nn);
        SMARK(6);
        // ratio test over the active positions (lowest position wins a tie)
        double t1 = INFINITY;
        int kdrop = 0x7fffffff;
        for (int j = tid; j < q0; j += RES_THREADS) {
            const double ri = rv[j];
            if (ri > 0.0) {
                const double cand = upos[j] / ri;
                if (cand < t1 || (cand == t1 && j < kdrop)) {
                    t1 = cand;
                    kdrop = j;
                }
            }
        }
        wave_argmin(t1, kdrop);
        if (lane == 0) {
            redv[wave] = t1;
            redi[wave] = kdrop;
        }
        __syncthreads();
        res_argmin_of_waves(t1, kdrop, redv, redi);
        t1 = uni(t1);
        kdrop = uni(kdrop);
        SMARK(7);
        // ---- the decision, by everybody ----------------------------------------------------------------------
        const bool dependent = (q0 >= nr) || !(zz > (DEPENDENT * DEPENDENT) * nn);
        const double sp = bval_p + psign * dots_p;
        const double t2 = dependent ? INFINITY : -sp / zz;
        const double t = fmin(t1, t2);
        if (!(t < INFINITY)) {
            phase = 4;
            break;
        }
        for (int j = tid; j < q0; j += RES_THREADS) upos[j] = upos[j] - t * rv[j];
        const double up_before = phase == 0 ? 0.0 : up;
        up = up_before + t;
        {
            double part_yy = 0.0;
            if (tid < 256)
                for (int i = tid; i < nr; i += 256) {
                    double yi = y[i];
                    if (!dependent && i >= q0) {
                        yi += t * d[i];
                        y[i] = yi;
                    }
                    part_yy += yi * yi;
                }
            ynorm = uni(sqrt(res_sum256(part_yy, red4)));
        }
        SMARK(8);
        const bool full_step = (t2 < INFINITY) && (t2 <= t1);
        double alpha = 0.0, beta = 0.0, beta_out = 0.0;
        const bool leaves = !full_step;
        const bool moves = !dependent;
        if (full_step) {
            // p joins: reflector H with H d2 = alpha e1 on the tail coordinates; the inverse gets the column -r / alpha
            // and the row [0 .. 0, 1 / alpha]
            const double dq = uni(d[q0]);
            alpha = dq >= 0.0 ? -sqrt(zz) : sqrt(zz);
            const double v0 = dq - alpha;
            const double vv = zz - dq * dq + v0 * v0;
            beta = vv > 0.0 ? 2.0 / vv : 0.0;
            const double inv = 1.0 / alpha;
            if (is_inv) {
                if (mypos < q0) {
                    if (lane == 0) xr[q0] = -my_r * inv;
                } else if (mypos == q0) {
                    for (int j = lane; j <= q0; j += 64) xr[j] = j < q0 ? 0.0 : inv;
                }
            }
            if (gw == prow_index) {
                if (p < nrows) ia0 = 1;
                else ia1 = 1;
            }
            if (tid == 0) {
                act[q0] = p;
                upos[q0] = up;
            }
            q = q0 + 1;
            phase = 0;
            SMARK(9);
        } else {
            // ---- (4) partial step: position k leaves; its row of the inverse from its owner -------------------
            const int k = kdrop;
            if (k < 0 || k >= q0) {
                phase = 3;
                exit_dbg = -7;
                exit_dbg2 = k;
                break;
            }
            ++c4;
            res_u64* const area = mailE4 + 2 * (size_t)((c4 & 1u) * RES_VEC_CAP);
            if (tid == RES_THREADS - 64) mb_put(area + 2 * (size_t)(RES_MAX_LEN + w), 1.0, c4);     // this workgroup has come this far
            if (is_inv && mypos == k) {
#pragma unroll 4
                for (int j = lane; j < q0; j += 64) mb_put(area + 2 * (size_t)j, xr[j], c4);
            }
            bool ok = true;
            for (int j = tid; j < q0 + NW; j += RES_THREADS) {
                const double v = mb_get(area + 2 * (size_t)(j < q0 ? j : RES_MAX_LEN + (j - q0)), c4, spin_limit, lost, ok);
                if (j < q0) aux[j] = v;
            }
            if (__syncthreads_or(ok ? 0 : 1)) return;
            // the reflector that moves the freed direction to coordinate q0 - 1 (rows_leave, its sums in its order)
            double part = 0.0;
            if (tid < 256)
                for (int j = tid; j < q0; j += 256) {
                    const double v = aux[j];
                    part += v * v;
                }
            const double ww = uni(res_sum256(part, red4));
            const double wl = uni(aux[q0 - 1]);
            const double al = wl >= 0.0 ? -sqrt(ww) : sqrt(ww);
            const double v0 = wl - al;
            const double vv = ww - wl * wl + v0 * v0;
            beta_out = vv > 0.0 ? 2.0 / vv : 0.0;
            const int leaving = uni(act[k]), freed = uni(slot[k]);
            __syncthreads();
            if (tid == 0) aux[q0 - 1] = v0;
            // lists: positions k + 1 .. q0 - 1 move down, the freed storage row goes to position q0 - 1
            for (int j = k + tid; j < q0 - 1; j += RES_THREADS) {
                scratch[j] = act[j + 1];
                rv[j] = upos[j + 1];
            }
            __syncthreads();
            for (int j = k + tid; j < q0 - 1; j += RES_THREADS) {
                act[j] = scratch[j];
                upos[j] = rv[j];
                scratch[j] = slot[j + 1];
            }
            __syncthreads();
            for (int j = k + tid; j < q0 - 1; j += RES_THREADS) {
                const int sl = scratch[j];
                slot[j] = sl;
                posof[sl] = j;
            }
            if (tid == 0) {
                slot[q0 - 1] = freed;
                posof[freed] = q0 - 1;
            }
            {
                const int lrow = leaving < nrows ? leaving : leaving - nq;
                if (gw == lrow) {
                    if (leaving < nrows) ia0 = 0;
                    else ia1 = 0;
                }
            }
            __syncthreads();
            mypos = is_inv ? uni(posof[myslot]) : 0x7fffffff;
            q = q0 - 1;
            phase = 1;
            SMARK(10);
#ifdef OGSQP_TRACE
            ++n_partial;
#endif
        }
        // every row's constraint value moves with the point (t g_i, the product formed while r was under way) - and with
        // it the row's price: after a full step the next election's prices go out NOW, before the rows are touched (the
        // pass and its closing barrier hide behind the exchange's way through memory)
        if (is_row && moves) dot += t * gi_early;
        if (phase == 0) {
            price_row();
            __syncthreads();
            publish_best();
        }
        // ---- the pass over the rows, each in its wavefront's LDS (k_rows_apply_r4's arithmetic) ----------------
        // the first q0 coordinates of a row matter when a row leaves (its reflector lives there), the tail when the
        // incoming row moves the point.  Lane l adds its coordinates l, l + 64, ... in ascending order, then wave_sum: the
        // sums of the register kernels (what a change does not read of a row entered those as exact zeros; d is zero
        // behind nr, the rows behind their length)
        {
            double* const row = carrier ? xr : y;    // the last wavefront: y, which lives in the same coordinates and has no
                                                     // value of its own (the row "y" of k_rows_apply)
            const bool tail_turn = moves && (is_row || (!carrier && full_step));
            const bool head_turn = leaves && (is_row || !carrier || (is_inv && mypos < q0 - 1));   // ... and the inverse's rows that stay
            if (tail_turn) {
                double gi = gi_early;                // (a row's: formed before r arrived - row and d have not changed since)
                if (!carrier) {
                    double acc_d = 0.0;
#pragma unroll 4
                    for (int e = e0; e < ne; ++e) {
                        const int j = lane + 64 * e;
                        const double xv = row[j], dv = d[j];
                        acc_d += j >= q0 ? xv * dv : 0.0;
                    }
                    gi = uni(wave_sum(acc_d));
                }
                if (full_step) {
                    const double f = beta * (gi - alpha * row[q0]);
#pragma unroll 4
                    for (int e = e0; e < ne; ++e) {
                        const int j = lane + 64 * e;
                        const double xv = row[j], dv = d[j];
                        if (j >= q0 && j < nr) row[j] = xv - f * (dv - (j == q0 ? alpha : 0.0));
                    }
                }
            }
            if (head_turn) {
                double acc_v = 0.0;
#pragma unroll 4
                for (int e = 0; e < eh; ++e) {
                    const int j = lane + 64 * e;
                    const double xv = row[j], vj = aux[j];
                    acc_v += j < q0 ? xv * vj : 0.0;
                }
                const double f = beta_out * uni(wave_sum(acc_v));
#pragma unroll 4
                for (int e = 0; e < eh; ++e) {
                    const int j = lane + 64 * e;
                    const double xv = row[j], vj = aux[j];
                    if (j < q0) row[j] = xv - f * vj;
                }
            }
        }
        if (phase == 0) publish_best_row();          // (the row as this change leaves it)
        SMARK(11);
        __syncthreads();
        SMARK(12);
#ifdef OGSQP_TRACE
        ++n_changes;
#endif
    }

    // ---- everything back to memory: rows, values, the shared state (workgroup 0) ------------------------------
    if (is_row) {
        double* dst = (double*)rows_ptr(g, gw);
        for (int j = lane; j < nr; j += 64) dst[j] = xr[j];
        if (lane == 0) {
            // (nothing was written while the loop ran - a launch that gives up leaves memory as it found it -: which of
            // this row's constraints are active now, and zero multipliers for those that are not)
            a.dots[gw] = dot;
            g.isact[gw] = ia0;
            if (!ia0) g.u[gw] = 0.0;
            if (gw >= mg) {
                g.isact[gw + nq] = ia1;
                if (!ia1) g.u[gw + nq] = 0.0;
            }
        }
    } else if (is_inv) {
        double* dst = g.RI[0] + (long)myslot * qcap;
        for (int j = lane; j < qcap; j += 64) dst[j] = xr[j];
    }
    if (w == 0) {
        __syncthreads();
        for (int i = tid; i < nr; i += RES_THREADS) g.y[i] = y[i];
        for (int i = tid; i < qcap; i += RES_THREADS) {
            a.slot[i] = slot[i];
            if (i < q) {
                g.act[i] = act[i];
                g.u[act[i]] = upos[i];
            }
        }
        if (tid == 0) {
            st->phase = phase;
            st->q = q;
            st->iters = iters;
            st->p = p;
            st->up = up;
            st->ynorm = ynorm;
            st->warm_removals = warm_removals;
            if (exit_dbg) {
                st->dbg = exit_dbg;
                st->dbg2 = exit_dbg2;
            }
            A.seq[0] = c1;
            A.seq[1] = c2;
            A.seq[2] = c3;
            A.seq[3] = c4;
        }
#ifdef OGSQP_TRACE
        __syncthreads();
        if (tid == 0) {
            for (int e = 0; e < 13; ++e) st->tr[16 + e] += s_tr[e];
            st->tr[16 + 13] += s_tr[16 + 11];
            st->tr[38] += n_changes;
            st->tr[39] += n_partial;
        }
#endif
    }
}
#undef SMARK
