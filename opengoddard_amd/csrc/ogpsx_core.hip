// ogpsx_core.hip -- libogpsx.so: the generic runtime behind include/ogpsx.h.
//
// LGL construction (host function + HIP kernel from one shared source, og_lgl.h), SciPy's
// forward-difference step rule, and problem handles that bind a compiled callback module
// (libogk_<hash>.so, built from ogk_kernels.hip + a generated header) to a device.
// Reference lines replaced by each entry point are listed in include/ogpsx.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/ogpsx.h"
#include "og_lgl.h"
#include "ogk.h"

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

// ------------------------------------------------------------------------------ LGL kernels
// One thread per node: Newton iteration for tau_k, then P_{N-1}(tau_k) and the weight.
__global__ void lgl_nodes_kernel(int N, double* tau, double* w, double* pval) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= N) return;
    const double t = oglgl::node(N, k);
    const double p = oglgl::legendre(N - 1, t);
    tau[k] = t;
    pval[k] = p;
    w[k] = oglgl::weight(N, p);
}

// One thread per matrix entry, consecutive lanes along a row => coalesced stores.
__global__ void lgl_dmat_kernel(int N, const double* tau, const double* pval, double* D) {
    const int l = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int k = (int)blockIdx.y;
    if (l >= N) return;
    D[(long)k * N + l] = oglgl::dmat(N, k, l, tau[k], tau[l], pval[k], pval[l]);
}

typedef int (*ogk_get_info_fn)(ogk_info*);
typedef int (*ogk_launch_fn)(const ogk_args*, int, void*);

// A callback module is one shared object, or several PARTS of it that were compiled side by side (build.py:
// <module>.so, <module>.p1.so, ...; each holds some of the kernels and answers OGK_OTHER_PART for the modes of the
// others).  Calls go to the part that holds the mode; which one that is is remembered per mode.
struct ogk_module {
    static constexpr int MAX_PARTS = 4, MAX_MODES = 16;
    ogk_launch_fn part[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    void* handle[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    int n_parts = 0;
    signed char home[MAX_MODES];
    ogk_module() {
        for (int i = 0; i < MAX_MODES; ++i) home[i] = -1;
    }
    explicit operator bool() const { return n_parts > 0; }
    int operator()(const ogk_args* a, int mode, void* stream) {
        if (mode >= 0 && mode < MAX_MODES && home[mode] >= 0) return part[(int)home[mode]](a, mode, stream);
        for (int i = 0; i < n_parts; ++i) {
            const int rc = part[i](a, mode, stream);
            if (rc != OGK_OTHER_PART) {
                if (mode >= 0 && mode < MAX_MODES) home[mode] = (signed char)i;
                return rc;
            }
        }
        return (int)hipErrorInvalidDeviceFunction;         // no part of the module holds this mode
    }
};

}  // namespace

struct og_problem_s {
    int device = 0;
    int n = 0, m = 0, m_eq = 0, m_ineq = 0;
    void* module = nullptr;
    ogk_module launch;
    double* d_dfrag = nullptr;
    double* d_cvec = nullptr;
    int64_t dfrag_off[OGK_MAX_PHASE] = {0};
    // staging buffers for the host-pointer entry points
    double* d_x = nullptr;
    double* d_h = nullptr;
    double* d_f0 = nullptr;
    double* d_jt = nullptr;
    size_t jt_capacity = 0;
    // sweep scratch written by every evaluation (ogk.h): base products / dynamics terms / F0-F0
    double* d_y0 = nullptr;
    double* d_xop = nullptr;
    double* d_t0 = nullptr;
    double* d_z = nullptr;
    int* d_flags = nullptr;             // two non-finite-row counters used alternately, then the ticket of the
    int flag_slot = 0;                  // fused launch (evaluation workgroups that have finished, ever)
    int n_eval_blocks = 0;
    double* d_trace = nullptr;          // phase stamps of -DOGK_TRACE kernel builds (tools/trace_fused.py)
    // how og_fd_sweep_dev runs: 5 evaluation + structured sweep in one launch (default), 1 the same as two
    // launches (OGPSX_SWEEP=split), 2 evaluation + dense sweep (OGPSX_SWEEP=dense).  og_fd_columns_dev (the
    // sweep alone, F(x0) supplied) uses 1 or 2.
    int sweep_mode = 5;
    int exact_mode = 4;                 // 4 structured (default), 3 dense (OGPSX_SWEEP=dense)
    hipStream_t stream = nullptr;
    // persistent-zero output buffers (og_jt_register_dev): the sweep writes only what can be non-zero.  Each
    // registration owns one word of d_state (ogk.h: jt_state); word 0 is the stand-in for unregistered buffers.
    struct jt_reg {
        double* ptr;
        int lo, hi;
        int slot;                       // d_state[2*slot] = state word, d_state[2*slot + 1] = launch counter
    };
    std::vector<jt_reg> regs;
    uint32_t* d_state = nullptr;
    bool fused_ok = false;              // the module can run evaluation + sweep as one launch
    int* nf_read = nullptr;             // where the most recent evaluation left its count of non-finite rows
    // static pattern of J_T (og_pattern): entries of column j are indptr[j]..indptr[j+1] of the packed order
    bool have_pattern = false;
    std::vector<int64_t> indptr;
    std::vector<int32_t> rows;
    // the same pattern as runs of consecutive rows per column (the defect rows of a variable's own state are one run of
    // N): the host's scatter of a downloaded block is one memcpy per run instead of one store per entry
    std::vector<int64_t> run_ptr;       // n + 1: runs of column j = [run_ptr[j], run_ptr[j + 1])
    std::vector<int32_t> run_row, run_len;
    std::vector<int64_t> run_off;       // per run: where it goes in a full n x m matrix (j * m + row): the scatter is one
                                        // flat loop over runs that asks for the destination lines a few runs ahead
    hipEvent_t down_ev[4] = {nullptr, nullptr, nullptr, nullptr};    // the packed block comes down in up to four pieces
    int64_t* d_indptr = nullptr;
    int32_t* d_rows = nullptr;          // the pattern's row indices, flat (pack / unpack read them coalesced)
    // host-pointer entry points: pinned staging ([x | h] up, [packed non-zeros | F | non-finite count] down)
    double* h_up = nullptr;
    double* h_xh = nullptr;             // mapped host matrix: [x | h] the launch reads in place
    double* h_down = nullptr;
    double* d_down = nullptr;
    size_t down_capacity = 0;
    // host matrices registered as persistent-zero (og_jt_register_host)
    struct host_reg {
        double* ptr;
        int lo, hi;
        bool dirty;                     // holds a NaN fill: zero it before the next scatter
        double* mapped;                 // round 6: the matrix is page-locked and mapped into the device's address space
                                        // (hipHostRegister): the one-launch sweep writes its non-zeros - and F(x0) into
                                        // the pinned staging buffer - straight over PCIe: no packed copy, no host scatter;
                                        // nullptr: the packed download + scatter (registration refused, OGPSX_HOST=staged)
        // Which of the two is faster is the HOST's doing (page size and IOMMU decide what 83 541 scattered 8-byte writes into
        // host pages cost; on every box of round 6 the mapped form won, 0.043-0.052 against 0.10-0.12 ms at C3), so it is
        // measured, not assumed: the first ten sweeps into a mapped matrix time both - two warm-ups and three timed calls
        // each, the fastest call of each counts - and the faster form serves from then on (OGPSX_HOST=mapped / staged decide
        // without the trial).
        int choice = 0;                 // 0 undecided, 1 mapped, 2 staged
        int calls = 0;
        double t_mapped = 0.0, t_staged = 0.0;
        bool last_staged = false;       // the last call scattered on the host (a NaN fill of its is the host's to clean)
        bool registered_here = false;   // hipHostRegister was this library's doing (memory from og_pinned_alloc needs none)
        double best_mapped = 1e30, best_staged = 1e30;
    };
    std::vector<host_reg> host_regs;
    // column sharding (og_shard_plan)
    int shard_world = 0;
    int64_t shard_block_vals = 0;
    int64_t* d_shard_off = nullptr;
    void* shard_comm = nullptr;         // ncclComm_t of this rank (og_shard_comm_init), one process per GPU
    int shard_rank = -1;
};
static const int OG_MAX_JT_REGS = 63;
static const size_t OG_TRACE_DOUBLES = (size_t)1 << 20;     // 16384 workgroups x 8 wavefronts x 8 stamps

namespace {

void fill_args(og_problem_s* p, ogk_args* a, const double* x, const double* h, double* f0,
               double* jt, int lo, int hi, bool new_launch = true) {
    memset(a, 0, sizeof *a);
    a->x0 = x;
    a->h = h;
    a->dfrag = p->d_dfrag;
    a->cvec = p->d_cvec;
    a->f0 = f0;
    a->y0 = p->d_y0;
    a->xop = p->d_xop;
    a->t0 = p->d_t0;
    a->z = p->d_z;
    a->nonfinite = p->d_flags + p->flag_slot;
    a->nonfinite_next = p->d_flags + (p->flag_slot ^ 1);
    a->ready = reinterpret_cast<unsigned*>(p->d_flags + 2);
    a->trace = p->d_trace;
    a->jt = jt;
    a->col_lo = lo;
    a->col_hi = hi;
    // a registered buffer (exactly this block of columns at this address) is written sparsely; its two words
    // (state, launch counter) live on the device - nothing about the history of the buffer is a launch argument
    a->jt_sparse = 0;
    a->jt_bump = 0;
    a->jt_state = p->d_state;
    a->jt_launches = p->d_state + 1;
    a->nonfinite_result = p->d_flags + 4;
    if (!new_launch && p->nf_read) a->nonfinite = p->nf_read;     // pack / unpack: the count the last evaluation left
    if (jt)
        for (size_t i = 0; i < p->regs.size(); ++i) {
            auto& r = p->regs[i];
            if (r.ptr == jt && r.lo == lo && r.hi == hi) {
                a->jt_sparse = 1;
                a->jt_state = p->d_state + 2 * r.slot;
                a->jt_launches = a->jt_state + 1;
                break;
            }
        }
    memcpy(a->dfrag_off, p->dfrag_off, sizeof(a->dfrag_off));
}

int launch_failed(og_problem_s*, int rc, const char* where) {
    return fail(100 + rc, std::string(where) + ": " + hipGetErrorString((hipError_t)rc));
}

// the handle's own Jacobian buffer (host-pointer entry points): sized for this block of columns and
// registered as a persistent-zero buffer, so that every sweep into it writes the non-zeros only
int own_jt(og_problem_s* p, int lo, int hi) {
    const size_t need = (size_t)(hi - lo) * (size_t)p->m;
    if (need > p->jt_capacity) {
        if (p->d_jt) {
            og_jt_unregister_dev(p, p->d_jt);
            OG_HIP(hipFree(p->d_jt));
        }
        p->d_jt = nullptr;
        p->jt_capacity = 0;
        OG_HIP(hipMalloc(&p->d_jt, sizeof(double) * need));
        p->jt_capacity = need;
    }
    for (auto& r : p->regs)
        if (r.ptr == p->d_jt && r.lo == lo && r.hi == hi) return 0;
    return og_jt_register_dev(p, p->d_jt, lo, hi, p->stream);
}

// the static pattern, once per handle: entries per column from the module (mode 6), prefix sums here, row
// indices from the module again (mode 7)
int ensure_pattern(og_problem_s* p) {
    if (p->have_pattern) return 0;
    OG_HIP(hipSetDevice(p->device));
    const int n = p->n;
    int32_t* d_cnt = nullptr;
    OG_HIP(hipMalloc(&d_cnt, sizeof(int32_t) * (size_t)n));
    ogk_args a;
    fill_args(p, &a, nullptr, nullptr, nullptr, nullptr, 0, 0, false);
    a.pint = d_cnt;
    int rc = p->launch(&a, 6, p->stream);
    std::vector<int32_t> cnt((size_t)n);
    hipError_t e = rc ? (hipError_t)rc : hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int32_t) * (size_t)n,
                                                        hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    hipFree(d_cnt);
    if (e != hipSuccess) return fail(100 + (int)e, std::string("og_pattern: ") + hipGetErrorString(e));
    p->indptr.assign((size_t)n + 1, 0);
    for (int j = 0; j < n; ++j) p->indptr[(size_t)j + 1] = p->indptr[(size_t)j] + cnt[(size_t)j];
    const size_t nnz = (size_t)p->indptr[(size_t)n];
    OG_HIP(hipMalloc(&p->d_indptr, sizeof(int64_t) * ((size_t)n + 1)));
    OG_HIP(hipMemcpy(p->d_indptr, p->indptr.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
    int32_t* d_rows = nullptr;
    OG_HIP(hipMalloc(&d_rows, sizeof(int32_t) * (nnz ? nnz : 1)));
    a.pint = d_rows;
    a.poff = p->d_indptr;
    rc = p->launch(&a, 7, p->stream);
    p->rows.assign(nnz, 0);
    e = rc ? (hipError_t)rc : hipMemcpyAsync(p->rows.data(), d_rows, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost,
                                             p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) {
        hipFree(d_rows);
        return fail(100 + (int)e, std::string("og_pattern: ") + hipGetErrorString(e));
    }
    p->d_rows = d_rows;
    p->run_ptr.assign((size_t)n + 1, 0);
    p->run_row.clear();
    p->run_len.clear();
    p->run_off.clear();
    for (int j = 0; j < n; ++j) {
        const int64_t lo = p->indptr[(size_t)j], hi = p->indptr[(size_t)j + 1];
        for (int64_t i = lo; i < hi;) {
            int64_t e = i + 1;
            while (e < hi && p->rows[(size_t)e] == p->rows[(size_t)e - 1] + 1) ++e;
            p->run_row.push_back(p->rows[(size_t)i]);
            p->run_len.push_back((int32_t)(e - i));
            p->run_off.push_back((int64_t)j * (int64_t)p->m + (int64_t)p->rows[(size_t)i]);
            i = e;
        }
        p->run_ptr[(size_t)j + 1] = (int64_t)p->run_row.size();
    }
    p->have_pattern = true;
    return 0;
}

// pinned staging of the host-pointer entry points
int ensure_staging(og_problem_s* p, size_t down_doubles) {
    if (!p->h_up) OG_HIP(hipHostMalloc(&p->h_up, sizeof(double) * 2 * (size_t)p->n, hipHostMallocDefault));
    if (down_doubles > p->down_capacity) {
        if (p->h_down) OG_HIP(hipHostFree(p->h_down));
        if (p->d_down) OG_HIP(hipFree(p->d_down));
        p->h_down = p->d_down = nullptr;
        p->down_capacity = 0;
        OG_HIP(hipHostMalloc(&p->h_down, sizeof(double) * down_doubles, hipHostMallocDefault));
        OG_HIP(hipMalloc(&p->d_down, sizeof(double) * down_doubles));
        p->down_capacity = down_doubles;
    }
    return 0;
}

// x (and h) to the device through the pinned buffer: one copy
int upload_point(og_problem_s* p, const double* x, const double* hstep) {
    int rc = ensure_staging(p, p->down_capacity ? p->down_capacity : (size_t)p->m + 1);
    if (rc) return rc;
    memcpy(p->h_up, x, sizeof(double) * (size_t)p->n);
    if (hstep) memcpy(p->h_up + p->n, hstep, sizeof(double) * (size_t)p->n);
    // d_x and d_h are one allocation: [x | h]
    OG_HIP(hipMemcpyAsync(p->d_x, p->h_up, sizeof(double) * (size_t)p->n * (hstep ? 2 : 1), hipMemcpyHostToDevice,
                          p->stream));
    return 0;
}

og_problem_s::host_reg* find_host_reg(og_problem_s* p, const double* JT, int lo, int hi) {
    for (auto& r : p->host_regs)
        if (r.ptr == JT && r.lo == lo && r.hi == hi) return &r;
    return nullptr;
}

// After a sweep / exact Jacobian into the handle's own registered buffer: bring the result to the host.  A
// registered host matrix receives the packed non-zeros (one pinned copy together with F and the count of
// non-finite rows) and a scatter; anything else, and any sweep with non-finite rows, the dense block.
// OGPSX_TIMING=1: where a host-pointer sweep spends its time (upload enqueue, launch, copy enqueue, wait, scatter),
// averaged over 64 calls, on stderr
struct host_clock {
    bool on = getenv("OGPSX_TIMING") != nullptr;
    double t[6] = {0, 0, 0, 0, 0, 0};
    double last = 0.0;
    int calls = 0;
    static double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void start() { if (on) last = now(); }
    void mark(int slot) {
        if (!on) return;
        const double n = now();
        t[slot] += n - last;
        last = n;
    }
    void done() {
        if (!on || ++calls < 64) return;
        fprintf(stderr, "[ogpsx timing] host sweep, us per call: upload %.1f, launch %.1f, copy enqueue %.1f, wait %.1f, "
                        "F + flags %.1f, scatter %.1f\n", 1e6 * t[0] / calls, 1e6 * t[1] / calls, 1e6 * t[2] / calls,
                1e6 * t[3] / calls, 1e6 * t[4] / calls, 1e6 * t[5] / calls);
        for (double& v : t) v = 0.0;
        calls = 0;
    }
} g_host_clock;

int download_block(og_problem_s* p, int lo, int hi, double* JT, double* F0, const double* d_src = nullptr,
                   bool already_packed = false) {
    const size_t need = (size_t)(hi - lo) * (size_t)p->m;
    if (!d_src) d_src = p->d_jt;
    og_problem_s::host_reg* reg = find_host_reg(p, JT, lo, hi);
    if (!reg) {
        OG_HIP(hipMemcpyAsync(JT, d_src, sizeof(double) * need, hipMemcpyDeviceToHost, p->stream));
        if (F0) OG_HIP(hipMemcpyAsync(F0, p->d_f0, sizeof(double) * p->m, hipMemcpyDeviceToHost, p->stream));
        OG_HIP(hipStreamSynchronize(p->stream));
        return 0;
    }
    reg->last_staged = true;
    int rc = ensure_pattern(p);
    if (rc) return rc;
    const int64_t first = p->indptr[(size_t)lo], nnz = p->indptr[(size_t)hi] - first;
    const size_t down = (size_t)nnz + (size_t)p->m + 1;
    rc = ensure_staging(p, down);
    if (rc) return rc;
    if (!already_packed) {              // (the one-launch sweep of og_fd_sweep wrote d_down itself)
        ogk_args a;
        fill_args(p, &a, p->d_x, p->d_h, p->d_f0, const_cast<double*>(d_src), lo, hi, false);
        a.poff = p->d_indptr;
        a.pind = p->d_indptr;
        a.prow = p->d_rows;
        a.pvals = p->d_down - first;
        a.ptail = p->d_down + nnz;
        rc = p->launch(&a, 8, p->stream);
        if (rc) return fail(100 + rc, std::string("og_fd_sweep: pack: ") + hipGetErrorString((hipError_t)rc));
    }
    // The packed block comes down in up to four pieces, each with an event behind it, and the host scatters piece c
    // while piece c + 1 crosses PCIe (round 6; measured before, C3: 39 us waiting for upload + launch + ONE copy, then
    // 43 us of scatter - the destination lines of an 18.6 MB matrix are cache misses: the loop asks for them a few runs
    // ahead).  F(x0) and the count of non-finite rows travel behind the last piece.
    static const int pieces_wanted = [] { const char* e = getenv("OGPSX_DOWN_PIECES"); return e ? atoi(e) : 1; }();
    const int64_t run_lo = p->run_ptr[(size_t)lo], run_hi = p->run_ptr[(size_t)hi];
    const int pieces = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(4, pieces_wanted), nnz / 16384));
    int64_t piece_run[5], piece_val[5];
    piece_run[0] = run_lo;
    piece_val[0] = 0;
    {
        // cut at column boundaries nearest to equal shares of the values
        int col = lo;
        for (int c = 1; c < pieces; ++c) {
            const int64_t want = first + nnz * c / pieces;
            col = (int)(std::lower_bound(p->indptr.begin() + col, p->indptr.begin() + hi, want) - p->indptr.begin());
            piece_run[c] = p->run_ptr[(size_t)col];
            piece_val[c] = p->indptr[(size_t)col] - first;
        }
        piece_run[pieces] = run_hi;
        piece_val[pieces] = nnz;
    }
    for (int c = 0; c < pieces; ++c) {
        if (!p->down_ev[c]) OG_HIP(hipEventCreateWithFlags(&p->down_ev[c], hipEventDisableTiming));
        const size_t from = (size_t)piece_val[c], to = c + 1 == pieces ? down : (size_t)piece_val[c + 1];
        if (to > from)
            OG_HIP(hipMemcpyAsync(p->h_down + from, p->d_down + from, sizeof(double) * (to - from), hipMemcpyDeviceToHost,
                                  p->stream));
        OG_HIP(hipEventRecord(p->down_ev[c], p->stream));
    }
    g_host_clock.mark(2);
    if (reg->dirty) {                    // (a NaN fill from the sweep before: the matrix is zeroed while the copies run)
        memset(JT, 0, sizeof(double) * need);
        reg->dirty = false;
    }
    const int64_t base = (int64_t)lo * (int64_t)p->m;
    const int64_t* off = p->run_off.data();
    const int32_t* len = p->run_len.data();
    const double* v = p->h_down;
    static const int AHEAD = [] { const char* e = getenv("OGPSX_AHEAD"); return e ? atoi(e) : 24; }();
    for (int c = 0; c < pieces; ++c) {
        while (true) {                   // (polling: a blocking wait costs a wake-up per piece)
            const hipError_t e = hipEventQuery(p->down_ev[c]);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) return fail(100 + (int)e, std::string("og_fd_sweep: download: ") + hipGetErrorString(e));
        }
        if (c == 0) g_host_clock.mark(3);
        const int64_t k1 = piece_run[c + 1];
        for (int64_t k = piece_run[c]; k < k1; ++k) {
            if (k + AHEAD < run_hi) __builtin_prefetch(JT + (off[k + AHEAD] - base), 1, 3);
            const int32_t n_ = len[k];
            double* dst = JT + (off[k] - base);
            if (n_ <= 4) {
                for (int32_t i = 0; i < n_; ++i) dst[i] = v[i];
            } else {
                memcpy(dst, v, sizeof(double) * (size_t)n_);
            }
            v += n_;
        }
    }
    g_host_clock.mark(5);
    const bool bad = p->h_down[(size_t)nnz + (size_t)p->m] != 0.0;
    if (F0) memcpy(F0, p->h_down + nnz, sizeof(double) * (size_t)p->m);
    g_host_clock.mark(4);
    if (bad) {                           // rows of NaN in every column: the dense block (over what was scattered)
        OG_HIP(hipMemcpyAsync(JT, d_src, sizeof(double) * need, hipMemcpyDeviceToHost, p->stream));
        OG_HIP(hipStreamSynchronize(p->stream));
        reg->dirty = true;
    }
    g_host_clock.done();
    return 0;
}

}  // namespace

// launch-floor probe (og_probe_launch): a kernel that does nothing, in the geometry asked for
__global__ void og_probe_kernel(int unused) { (void)unused; }

extern "C" {

const char* og_last_error(void) { return g_error.c_str(); }

int og_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int og_probe_launch(int32_t blocks, int32_t threads, int32_t count, void* hip_stream) {
    if (blocks < 1 || threads < 1 || threads > 1024 || count < 0) return fail(1, "og_probe_launch: bad geometry");
    for (int i = 0; i < count; ++i)
        hipLaunchKernelGGL(og_probe_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)hip_stream, 0);
    OG_HIP(hipGetLastError());
    return 0;
}

int og_lgl(int32_t N, double* tau, double* w, double* D) {
    if (N < 3) return fail(1, "og_lgl: N must be >= 3");
    if (!tau || !w || !D) return fail(1, "og_lgl: null output");
    std::vector<double> p(N);
    for (int k = 0; k < N; ++k) tau[k] = oglgl::node(N, k);
    for (int k = 0; k < N; ++k) {
        p[k] = oglgl::legendre(N - 1, tau[k]);
        w[k] = oglgl::weight(N, p[k]);
    }
    for (int k = 0; k < N; ++k)
        for (int l = 0; l < N; ++l)
            D[(long)k * N + l] = oglgl::dmat(N, k, l, tau[k], tau[l], p[k], p[l]);
    return 0;
}

int og_lgl_dev(int32_t N, double* d_tau, double* d_w, double* d_D, void* hip_stream) {
    if (N < 3) return fail(1, "og_lgl_dev: N must be >= 3");
    hipStream_t s = (hipStream_t)hip_stream;
    double* d_p = nullptr;
    OG_HIP(hipMalloc(&d_p, sizeof(double) * N));
    hipLaunchKernelGGL(lgl_nodes_kernel, dim3((N + 63) / 64), dim3(64), 0, s, N, d_tau, d_w, d_p);
    hipLaunchKernelGGL(lgl_dmat_kernel, dim3((N + 63) / 64, N), dim3(64), 0, s, N, d_tau, d_p, d_D);
    OG_HIP(hipGetLastError());
    OG_HIP(hipStreamSynchronize(s));
    OG_HIP(hipFree(d_p));
    return 0;
}

int og_fd_step(int32_t n, const double* x, const double* lb, const double* ub, double* h) {
    // scipy/optimize/_numdiff.py:500-515 (absolute step with zero-step fallback) followed by
    // _adjust_scheme_to_bounds(x0, h, 1, '1-sided', lb, ub), scipy/optimize/_numdiff.py:44-70.
    const double abs_step = 1.4901161193847656e-08;          // scipy/optimize/_slsqp_py.py:33
    const double root_eps = 1.4901161193847656e-08;          // EPS**0.5 for the 2-point scheme
    const double inf = std::numeric_limits<double>::infinity();
    bool unbounded = true;
    for (int i = 0; i < n; ++i)
        if (!(lb[i] == -inf && ub[i] == inf)) unbounded = false;
    for (int i = 0; i < n; ++i) {
        double hi = abs_step;
        const double dx = (x[i] + hi) - x[i];
        if (dx == 0.0) {
            const double sign = (x[i] >= 0.0) ? 1.0 : -1.0;
            hi = root_eps * sign * std::fmax(1.0, std::fabs(x[i]));
        }
        if (!unbounded) {
            const double lower = x[i] - lb[i], upper = ub[i] - x[i];
            const double xn = x[i] + hi;
            const bool violated = (xn < lb[i]) || (xn > ub[i]);
            const bool fitting = std::fabs(hi) <= std::fmax(lower, upper);
            double adj = hi;
            if (violated && fitting) adj = -adj;
            if (!fitting) adj = (upper >= lower) ? upper : -lower;
            hi = adj;
        }
        h[i] = hi;
    }
    return 0;
}

int og_problem_create(const og_desc* desc, og_handle* out) {
    if (!desc || !out) return fail(1, "og_problem_create: null argument");
    *out = nullptr;
    if (desc->abi_version != OG_ABI_VERSION) return fail(2, "og_problem_create: ABI version mismatch");
    if (desc->n_phase < 1 || desc->n_phase > OGK_MAX_PHASE)
        return fail(2, "og_problem_create: unsupported phase count");
    if (!desc->module_path) return fail(2, "og_problem_create: module_path is null");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(3, "og_problem_create: no HIP device available");
    if (desc->device < 0 || desc->device >= ndev)
        return fail(3, "og_problem_create: device ordinal out of range");
    OG_HIP(hipSetDevice(desc->device));

    void* mod = dlopen(desc->module_path, RTLD_NOW | RTLD_LOCAL);
    if (!mod) return fail(4, std::string("og_problem_create: dlopen failed: ") + dlerror());
    ogk_get_info_fn get_info = (ogk_get_info_fn)dlsym(mod, "ogk_get_info");
    ogk_launch_fn launch = (ogk_launch_fn)dlsym(mod, "ogk_launch");
    if (!get_info || !launch) {
        dlclose(mod);
        return fail(4, "og_problem_create: module lacks ogk_get_info/ogk_launch");
    }
    ogk_info info;
    memset(&info, 0, sizeof info);
    get_info(&info);
    bool ok = info.abi == OGK_ABI && info.n == desc->n && info.m_eq == desc->m_eq &&
              info.m_ineq == desc->m_ineq && info.n_phase == desc->n_phase &&
              info.n_cvec == desc->n_cvec;
    for (int i = 0; ok && i < desc->n_phase; ++i) ok = info.phase_nodes[i] == desc->nodes[i];
    if (!ok) {
        dlclose(mod);
        return fail(5, "og_problem_create: descriptor does not match the compiled module");
    }

    og_problem_s* p = new og_problem_s();
    p->device = desc->device;
    p->n = info.n;
    p->m = info.m;
    p->m_eq = info.m_eq;
    p->m_ineq = info.m_ineq;
    p->module = mod;
    p->launch.part[0] = launch;
    p->launch.handle[0] = mod;
    p->launch.n_parts = 1;
    {
        // further parts of the module, if it was built in parts: <path minus ".so">.p<k>.so
        std::string stem(desc->module_path);
        if (stem.size() > 3 && stem.compare(stem.size() - 3, 3, ".so") == 0) stem.resize(stem.size() - 3);
        for (int k = 1; k < ogk_module::MAX_PARTS; ++k) {
            const std::string path = stem + ".p" + std::to_string(k) + ".so";
            void* more = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!more) break;
            ogk_launch_fn fn = (ogk_launch_fn)dlsym(more, "ogk_launch");
            if (!fn) {
                dlclose(more);
                break;
            }
            p->launch.part[p->launch.n_parts] = fn;
            p->launch.handle[p->launch.n_parts] = more;
            ++p->launch.n_parts;
        }
    }

    // pack every phase's D into MFMA operand order and upload
    std::vector<double> frag;
    long total = 0;
    for (int i = 0; i < desc->n_phase; ++i) {
        p->dfrag_off[i] = total;
        total += ogk_frag_size(desc->nodes[i]);
    }
    frag.assign((size_t)total, 0.0);
    for (int i = 0; i < desc->n_phase; ++i) {
        const int N = desc->nodes[i];
        std::vector<double> tmp;
        const double* D = (desc->D && desc->D[i]) ? desc->D[i] : nullptr;
        if (!D) {
            // no matrix supplied: build it on the device with the LGL kernels (bit-identical to
            // og_lgl, tests/test_gpu_parity.py) and bring it back for operand packing
            double *d_tau = nullptr, *d_w = nullptr, *d_D = nullptr;
            tmp.resize((size_t)N * N);
            hipError_t le = hipMalloc(&d_tau, sizeof(double) * N);
            if (le == hipSuccess) le = hipMalloc(&d_w, sizeof(double) * N);
            if (le == hipSuccess) le = hipMalloc(&d_D, sizeof(double) * (size_t)N * N);
            int rc = (le == hipSuccess) ? og_lgl_dev(N, d_tau, d_w, d_D, nullptr) : 100 + (int)le;
            if (!rc && hipMemcpy(tmp.data(), d_D, sizeof(double) * (size_t)N * N,
                                 hipMemcpyDeviceToHost) != hipSuccess)
                rc = 101;
            hipFree(d_tau);
            hipFree(d_w);
            hipFree(d_D);
            if (rc) {
                og_problem_destroy(p);
                return fail(rc, "og_problem_create: device LGL construction failed");
            }
            D = tmp.data();
        }
        ogk_frag_pack(N, D, frag.data() + p->dfrag_off[i]);
    }
    hipError_t e = hipMalloc(&p->d_dfrag, sizeof(double) * (size_t)total);
    if (e == hipSuccess)
        e = hipMemcpy(p->d_dfrag, frag.data(), sizeof(double) * (size_t)total, hipMemcpyHostToDevice);
    if (e == hipSuccess && desc->n_cvec > 0) {
        e = hipMalloc(&p->d_cvec, sizeof(double) * (size_t)desc->n_cvec);
        if (e == hipSuccess)
            e = hipMemcpy(p->d_cvec, desc->cvec, sizeof(double) * (size_t)desc->n_cvec,
                          hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&p->d_x, sizeof(double) * 2 * (size_t)p->n);      // [x | h]
    if (e == hipSuccess) p->d_h = p->d_x + p->n;
    if (e == hipSuccess) e = hipMalloc(&p->d_f0, sizeof(double) * (size_t)p->m);
    if (e == hipSuccess) e = hipMalloc(&p->d_y0, sizeof(double) * (size_t)(info.n_y0 > 0 ? info.n_y0 : 1));
    if (e == hipSuccess) e = hipMalloc(&p->d_xop, sizeof(double) * (size_t)(info.n_y0 > 0 ? info.n_y0 : 1));
    if (e == hipSuccess) e = hipMalloc(&p->d_t0, sizeof(double) * (size_t)p->m);
    if (e == hipSuccess) e = hipMalloc(&p->d_z, sizeof(double) * (size_t)p->m);
    // {non-finite counters of the two-launch form (alternating), ticket, counter of the one-launch form, its result}
    if (e == hipSuccess) e = hipMalloc(&p->d_flags, 8 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(p->d_flags, 0, 8 * sizeof(int));
    if (e == hipSuccess && getenv("OGPSX_TRACE")) {
        e = hipMalloc(&p->d_trace, sizeof(double) * OG_TRACE_DOUBLES);
        if (e == hipSuccess) e = hipMemset(p->d_trace, 0, sizeof(double) * OG_TRACE_DOUBLES);
    }
    // one pair per registration, pair 0 for unregistered buffers, the last pair for og_shard_unpack_dev of a rank
    // that owns no columns (state: no NaN fill; launches so far: 0)
    if (e == hipSuccess) e = hipMalloc(&p->d_state, 2 * (OG_MAX_JT_REGS + 2) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(p->d_state, 0xff, 2 * (OG_MAX_JT_REGS + 2) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(p->d_state + 2 * (OG_MAX_JT_REGS + 1) + 1, 0, sizeof(uint32_t));
    p->n_eval_blocks = info.n_eval_blocks;
    p->fused_ok = info.fused_ok != 0;
    const char* mode_env = getenv("OGPSX_SWEEP");
    // One launch (evaluation + structured sweep) whenever the output is a registered persistent-zero buffer;
    // measured against the two-launch form with such a buffer (bench step, us, one launch / two): C2 4.8 / 8.1,
    // C3 7.6 / 12.4, C4 13.9 / 22.6, C5 23.9 / 33.0.  An unregistered buffer always takes two launches (the
    // module decides: the one-launch form never fills).
    p->sweep_mode = 5;
    if (mode_env && std::string(mode_env) == "dense") p->sweep_mode = 2, p->exact_mode = 3;
    if (mode_env && std::string(mode_env) == "split") p->sweep_mode = 1;
    if (mode_env && std::string(mode_env) == "fused") p->sweep_mode = 5;
    if (p->n_eval_blocks <= 0 && p->sweep_mode == 5) p->sweep_mode = 1;
    if (e == hipSuccess) e = hipStreamCreate(&p->stream);
    if (e != hipSuccess) {
        og_problem_destroy(p);
        return fail(100 + (int)e, std::string("og_problem_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return 0;
}

void og_problem_destroy(og_handle p) {
    if (!p) return;
    hipSetDevice(p->device);
    og_shard_comm_destroy(p);
    if (p->stream) hipStreamDestroy(p->stream);
    hipFree(p->d_dfrag);
    hipFree(p->d_cvec);
    hipFree(p->d_x);
    hipFree(p->d_f0);
    hipFree(p->d_jt);
    hipFree(p->d_y0);
    hipFree(p->d_xop);
    hipFree(p->d_t0);
    hipFree(p->d_z);
    hipFree(p->d_flags);
    hipFree(p->d_state);
    hipFree(p->d_trace);
    hipFree(p->d_indptr);
    hipFree(p->d_rows);
    hipFree(p->d_down);
    hipFree(p->d_shard_off);
    for (auto& r : p->host_regs)
        if (r.mapped && r.registered_here) (void)hipHostUnregister(r.ptr);
    (void)hipGetLastError();            // (a matrix its owner has already unmapped or freed: not this handle's error)
    if (p->h_up) hipHostFree(p->h_up);
    if (p->h_xh) hipHostFree(p->h_xh);
    if (p->h_down) hipHostFree(p->h_down);
    for (hipEvent_t ev : p->down_ev)
        if (ev) hipEventDestroy(ev);
    for (int k = 1; k < p->launch.n_parts; ++k)
        if (p->launch.handle[k]) dlclose(p->launch.handle[k]);
    if (p->module) dlclose(p->module);
    delete p;
}

int og_sweep_mode(og_handle p) { return p ? p->sweep_mode : 0; }

int og_device_read(int32_t device, const void* d_src, void* dst, int64_t bytes) {
    if (!d_src || !dst || bytes < 0) return fail(1, "og_device_read: bad argument");
    OG_HIP(hipSetDevice(device));
    OG_HIP(hipDeviceSynchronize());
    OG_HIP(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

int og_trace_read(og_handle p, double* out, int64_t count) {
    if (!p || !out) return fail(1, "og_trace_read: null argument");
    if (!p->d_trace) return fail(1, "og_trace_read: the handle was created without OGPSX_TRACE=1");
    if (count < 0 || (size_t)count > OG_TRACE_DOUBLES) return fail(1, "og_trace_read: bad count");
    OG_HIP(hipSetDevice(p->device));
    OG_HIP(hipDeviceSynchronize());
    OG_HIP(hipMemcpy(out, p->d_trace, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    OG_HIP(hipMemset(p->d_trace, 0, sizeof(double) * OG_TRACE_DOUBLES));
    return 0;
}

int og_jt_register_dev(og_handle p, double* d_JT, int32_t lo, int32_t hi, void* hip_stream) {
    if (!p || !d_JT) return fail(1, "og_jt_register_dev: null argument");
    if (lo < 0 || hi > p->n || lo >= hi) return fail(1, "og_jt_register_dev: bad column range");
    OG_HIP(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)hip_stream;
    og_problem_s::jt_reg* reg = nullptr;
    for (auto& r : p->regs)
        if (r.ptr == d_JT) reg = &r;
    if (!reg) {
        if ((int)p->regs.size() >= OG_MAX_JT_REGS)
            return fail(6, "og_jt_register_dev: too many registered buffers on this handle");
        // a free state word
        std::vector<char> used(OG_MAX_JT_REGS + 1, 0);
        for (auto& r : p->regs) used[r.slot] = 1;
        int slot = 1;
        while (used[slot]) ++slot;
        p->regs.push_back({d_JT, lo, hi, slot});
        reg = &p->regs.back();
    }
    reg->lo = lo;
    reg->hi = hi;
    OG_HIP(hipMemsetAsync(d_JT, 0, sizeof(double) * (size_t)(hi - lo) * (size_t)p->m, s));
    OG_HIP(hipMemsetAsync(p->d_state + 2 * reg->slot, 0xff, sizeof(uint32_t), s));        // state: no NaN fill
    OG_HIP(hipMemsetAsync(p->d_state + 2 * reg->slot + 1, 0, sizeof(uint32_t), s));       // launches so far
    return 0;
}

int og_jt_unregister_dev(og_handle p, double* d_JT) {
    if (!p) return fail(1, "og_jt_unregister_dev: null handle");
    for (size_t i = 0; i < p->regs.size(); ++i)
        if (p->regs[i].ptr == d_JT) {
            p->regs.erase(p->regs.begin() + (long)i);
            return 0;
        }
    return fail(1, "og_jt_unregister_dev: buffer is not registered");
}

int og_jt_register_host(og_handle p, double* JT, int32_t lo, int32_t hi) {
    if (!p || !JT) return fail(1, "og_jt_register_host: null argument");
    if (lo < 0 || hi > p->n || lo >= hi) return fail(1, "og_jt_register_host: bad column range");
    int rc = ensure_pattern(p);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * (size_t)(hi - lo) * (size_t)p->m;
    memset(JT, 0, bytes);
    for (auto& r : p->host_regs)
        if (r.ptr == JT) {
            r.lo = lo, r.hi = hi, r.dirty = false;
            if (r.mapped && og_jt_register_dev(p, r.mapped, lo, hi, p->stream) == 0) {
                OG_HIP(hipStreamSynchronize(p->stream));
                memset(JT, 0, bytes);
            }
            return 0;
        }
    // Round 6 (0.093 ms per host-pointer sweep at C3, of which 39 us are upload + launch + the packed copy
    // and 43 us the host's scatter of a cache-cold staging buffer): the matrix is page-locked and mapped, the sweep writes
    // its structural non-zeros into it over PCIe itself - the same persistent-zero protocol as a device buffer
    // (og_jt_register_dev on the mapped address: state words on the device, a NaN fill cleans itself with the next
    // sweep).  A host that refuses the registration keeps the packed path.
    double* mapped = nullptr;
    bool registered_here = false;
    static const bool staged = [] { const char* e = getenv("OGPSX_HOST"); return e && std::string(e) == "staged"; }();
    if (!staged && p->sweep_mode == 5 && p->fused_ok) {
        OG_HIP(hipSetDevice(p->device));
        void* dev = nullptr;
        // memory that came from og_pinned_alloc / hipHostMalloc is mapped already (and in the driver's own large fragments:
        // a handful of translations per sweep where a page-locked malloc'ed matrix of 4 KB pages needs 4 500)
        hipPointerAttribute_t attr;
        const bool pinned_already = hipPointerGetAttributes(&attr, JT) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (pinned_already || hipHostRegister(JT, bytes, hipHostRegisterMapped) == hipSuccess) {
            if (hipHostGetDevicePointer(&dev, JT, 0) == hipSuccess && dev &&
                og_jt_register_dev(p, (double*)dev, lo, hi, p->stream) == 0 &&
                hipStreamSynchronize(p->stream) == hipSuccess) {
                mapped = (double*)dev;
                memset(JT, 0, bytes);
                registered_here = !pinned_already;
            } else if (!pinned_already) {
                (void)hipHostUnregister(JT);
            }
        }
        (void)hipGetLastError();
    }
    og_problem_s::host_reg fresh;
    fresh.ptr = JT;
    fresh.lo = lo;
    fresh.hi = hi;
    fresh.dirty = false;
    fresh.mapped = mapped;
    fresh.registered_here = registered_here;
    p->host_regs.push_back(fresh);
    return 0;
}

int og_pinned_alloc(int64_t bytes, void** out) {
    if (bytes <= 0 || !out) return fail(1, "og_pinned_alloc: bad arguments");
    void* ptr = nullptr;
    const hipError_t e = hipHostMalloc(&ptr, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(100 + (int)e, std::string("og_pinned_alloc: ") + hipGetErrorString(e));
    }
    *out = ptr;
    return 0;
}

int og_pinned_free(void* ptr) {
    if (!ptr) return 0;
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(100 + (int)e, std::string("og_pinned_free: ") + hipGetErrorString(e));
    }
    return 0;
}

int og_jt_host_path(og_handle p, const double* JT, int32_t* path) {
    if (!p || !path) return fail(1, "og_jt_host_path: null argument");
    *path = -1;
    for (auto& r : p->host_regs)
        if (r.ptr == JT) *path = !r.mapped ? 2 : r.choice;
    return 0;
}

int og_jt_unregister_host(og_handle p, double* JT) {
    if (!p) return fail(1, "og_jt_unregister_host: null handle");
    for (size_t i = 0; i < p->host_regs.size(); ++i)
        if (p->host_regs[i].ptr == JT) {
            if (p->host_regs[i].mapped) {
                (void)hipStreamSynchronize(p->stream);
                (void)og_jt_unregister_dev(p, p->host_regs[i].mapped);
                if (p->host_regs[i].registered_here) (void)hipHostUnregister(JT);
                (void)hipGetLastError();
            }
            p->host_regs.erase(p->host_regs.begin() + (long)i);
            return 0;
        }
    return fail(1, "og_jt_unregister_host: matrix is not registered");
}

int og_pattern(og_handle p, int32_t lo, int32_t hi, int64_t* nnz, int64_t* indptr, int32_t* rows) {
    if (!p) return fail(1, "og_pattern: null handle");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_pattern: bad column range");
    int rc = ensure_pattern(p);
    if (rc) return rc;
    const int64_t first = p->indptr[(size_t)lo], count = p->indptr[(size_t)hi] - first;
    if (nnz) *nnz = count;
    if (indptr)
        for (int j = lo; j <= hi; ++j) indptr[j - lo] = p->indptr[(size_t)j] - first;
    if (rows && count) memcpy(rows, p->rows.data() + first, sizeof(int32_t) * (size_t)count);
    return 0;
}

int og_pack_dev(og_handle p, const double* d_JT, int32_t lo, int32_t hi, double* d_vals, void* hip_stream) {
    if (!p || !d_JT || !d_vals) return fail(1, "og_pack_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_pack_dev: bad column range");
    int rc = ensure_pattern(p);
    if (rc) return rc;
    ogk_args a;
    fill_args(p, &a, nullptr, nullptr, p->d_f0, const_cast<double*>(d_JT), lo, hi, false);
    a.poff = p->d_indptr;
    a.pind = p->d_indptr;
    a.prow = p->d_rows;
    a.pvals = d_vals - p->indptr[(size_t)lo];
    rc = p->launch(&a, 8, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_pack_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_unpack_dev(og_handle p, const double* d_vals, int32_t lo, int32_t hi, double* d_JT, void* hip_stream) {
    if (!p || !d_JT || !d_vals) return fail(1, "og_unpack_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_unpack_dev: bad column range");
    int rc = ensure_pattern(p);
    if (rc) return rc;
    ogk_args a;
    fill_args(p, &a, nullptr, nullptr, p->d_f0, nullptr, lo, lo, false);
    a.jt = d_JT - (size_t)lo * (size_t)p->m;       // the kernel indexes rows by absolute column
    a.jt_sparse = 0;                               // plain scatter: no fill
    a.poff = p->d_indptr;
    a.pind = p->d_indptr;
    a.prow = p->d_rows;
    a.pvals = const_cast<double*>(d_vals) - p->indptr[(size_t)lo];
    a.ulo = lo;
    a.uhi = hi;
    rc = p->launch(&a, 9, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_unpack_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_shard_plan(og_handle p, int32_t world, int32_t* block_cols, int64_t* block_vals) {
    if (!p) return fail(1, "og_shard_plan: null handle");
    if (world < 1) return fail(1, "og_shard_plan: world must be >= 1");
    int rc = ensure_pattern(p);
    if (rc) return rc;
    OG_HIP(hipSetDevice(p->device));
    const int n = p->n, B = (n + world - 1) / world;
    int64_t worst = 0;
    for (int r = 0; r < world; ++r) {
        const int lo = std::min(n, r * B), hi = std::min(n, lo + B);
        worst = std::max(worst, p->indptr[(size_t)hi] - p->indptr[(size_t)lo]);
    }
    std::vector<int64_t> off((size_t)n);
    for (int j = 0; j < n; ++j) {
        const int r = j / B;
        off[(size_t)j] = (int64_t)r * worst + p->indptr[(size_t)j] - p->indptr[(size_t)(r * B)];
    }
    if (p->d_shard_off) OG_HIP(hipFree(p->d_shard_off));
    p->d_shard_off = nullptr;
    OG_HIP(hipMalloc(&p->d_shard_off, sizeof(int64_t) * (size_t)(n ? n : 1)));
    OG_HIP(hipMemcpy(p->d_shard_off, off.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
    p->shard_world = world;
    p->shard_block_vals = worst;
    if (block_cols) *block_cols = B;
    if (block_vals) *block_vals = worst;
    return 0;
}

int og_shard_pack_dev(og_handle p, int32_t rank, const double* d_JT_block, double* d_send, void* hip_stream) {
    if (!p || !d_JT_block || !d_send) return fail(1, "og_shard_pack_dev: null argument");
    if (!p->shard_world || rank < 0 || rank >= p->shard_world)
        return fail(1, "og_shard_pack_dev: no shard plan, or rank out of range");
    const int n = p->n, B = (n + p->shard_world - 1) / p->shard_world;
    const int lo = std::min(n, rank * B), hi = std::min(n, lo + B);
    ogk_args a;
    fill_args(p, &a, nullptr, nullptr, p->d_f0, const_cast<double*>(d_JT_block), lo, hi, false);
    a.poff = p->d_shard_off;
    a.pind = p->d_indptr;
    a.prow = p->d_rows;
    a.pvals = d_send - (int64_t)rank * p->shard_block_vals;
    int rc = p->launch(&a, 8, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_shard_pack_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_shard_sweep_dev(og_handle p, int32_t rank, const double* d_x, const double* d_h, double* d_JT_block,
                       double* d_F0, double* d_send, void* hip_stream) {
    if (!p || !d_x || !d_h || !d_JT_block || !d_F0 || !d_send) return fail(1, "og_shard_sweep_dev: null argument");
    if (!p->shard_world || rank < 0 || rank >= p->shard_world)
        return fail(1, "og_shard_sweep_dev: no shard plan, or rank out of range");
    const int n = p->n, B = (n + p->shard_world - 1) / p->shard_world;
    const int lo = std::min(n, rank * B), hi = std::min(n, lo + B);
    if (hi <= lo) return og_eval_dev(p, d_x, d_F0, hip_stream);        // more ranks than columns: F(x0) only
    ogk_args a;
    fill_args(p, &a, d_x, d_h, d_F0, d_JT_block, lo, hi);
    if (p->sweep_mode == 5 && p->fused_ok && a.jt_sparse) {
        // ONE launch: F(x0), the block's non-zeros into J_T and, the same values, into this rank's message
        a.nonfinite = p->d_flags + 3;
        p->nf_read = a.nonfinite_result;
        a.poff = p->d_shard_off;
        a.pvals = d_send - (int64_t)rank * p->shard_block_vals;
        const int rc = p->launch(&a, 5, hip_stream);
        if (rc) return launch_failed(p, rc, "og_shard_sweep_dev");
        return 0;
    }
    const int rc = og_fd_sweep_dev(p, d_x, d_h, lo, hi, d_JT_block, d_F0, hip_stream);
    return rc ? rc : og_shard_pack_dev(p, rank, d_JT_block, d_send, hip_stream);
}

int og_shard_unpack_dev(og_handle p, int32_t rank, const double* d_recv, double* d_JT_full, void* hip_stream) {
    if (!p || !d_recv || !d_JT_full) return fail(1, "og_shard_unpack_dev: null argument");
    if (!p->shard_world || rank < 0 || rank >= p->shard_world)
        return fail(1, "og_shard_unpack_dev: no shard plan, or rank out of range");
    const int n = p->n, B = (n + p->shard_world - 1) / p->shard_world;
    const int lo = std::min(n, rank * B), hi = std::min(n, lo + B);
    ogk_args a;
    // the registration of this rank's own block inside the replica tells whether the previous step left a NaN
    // fill behind; same generation as the sweep that was just enqueued (no new launch into the block)
    fill_args(p, &a, nullptr, nullptr, p->d_f0, d_JT_full + (size_t)lo * (size_t)p->m, lo, hi, false);
    if (!a.jt_sparse && hi > lo)
        return fail(1, "og_shard_unpack_dev: this rank's block of the replica is not a registered buffer");
    a.jt = d_JT_full;
    a.jt_sparse = 1;
    a.poff = p->d_shard_off;
    a.pind = p->d_indptr;
    a.prow = p->d_rows;
    a.pvals = const_cast<double*>(d_recv);
    a.ulo = 0;
    a.uhi = n;
    int rc = 0;
    if (hi <= lo) {
        // more ranks than columns: no sweep of this rank keeps the replica's NaN history, so the unpack does - its
        // own pair of words, one count per step, and the kernel marks a step whose F(x0) had non-finite rows
        a.jt_state = p->d_state + 2 * (OG_MAX_JT_REGS + 1);
        a.jt_launches = a.jt_state + 1;
        a.jt_bump = 1;
        rc = p->launch(&a, 10, hip_stream);
    }
    if (!rc) rc = p->launch(&a, 9, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_shard_unpack_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_problem_dims(og_handle p, int32_t* n, int32_t* m, int32_t* m_eq, int32_t* m_ineq) {
    if (!p) return fail(1, "og_problem_dims: null handle");
    if (n) *n = p->n;
    if (m) *m = p->m;
    if (m_eq) *m_eq = p->m_eq;
    if (m_ineq) *m_ineq = p->m_ineq;
    return 0;
}

int og_eval_dev(og_handle p, const double* d_x, double* d_F, void* hip_stream) {
    if (!p || !d_x || !d_F) return fail(1, "og_eval_dev: null argument");
    ogk_args a;
    p->flag_slot ^= 1;                      // this evaluation counts non-finite rows into a fresh slot
    fill_args(p, &a, d_x, nullptr, d_F, nullptr, 0, 0);
    p->nf_read = a.nonfinite;
    int rc = p->launch(&a, 0, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_eval_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_fd_sweep_dev(og_handle p, const double* d_x, const double* d_h, int32_t lo, int32_t hi,
                    double* d_JT, double* d_F0, void* hip_stream) {
    if (!p || !d_x || !d_h || !d_JT || !d_F0) return fail(1, "og_fd_sweep_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_fd_sweep_dev: bad column range");
    ogk_args a;
    fill_args(p, &a, d_x, d_h, d_F0, d_JT, lo, hi);
    int rc;
    if (p->sweep_mode == 5 && p->fused_ok && a.jt_sparse && hi > lo) {
        // ONE launch.  Its arguments depend on nothing but the pointers: the count of non-finite rows, the ticket
        // and the buffer's launch number are kept by the kernel itself (a captured graph can be replayed)
        a.nonfinite = p->d_flags + 3;
        p->nf_read = a.nonfinite_result;
        rc = p->launch(&a, 5, hip_stream);
    } else {
        p->flag_slot ^= 1;
        a.nonfinite = p->d_flags + p->flag_slot;
        a.nonfinite_next = p->d_flags + (p->flag_slot ^ 1);
        p->nf_read = a.nonfinite;
        a.jt_bump = a.jt_sparse;                    // F(x0) first: the sweep subtracts it; it also counts the launch
        rc = p->launch(&a, 0, hip_stream);
        if (!rc) rc = p->launch(&a, p->sweep_mode == 5 ? 1 : p->sweep_mode, hip_stream);
    }
    if (rc) return launch_failed(p, rc, "og_fd_sweep_dev");
    return 0;
}

int og_fd_columns_dev(og_handle p, const double* d_x, const double* d_h, int32_t lo, int32_t hi,
                      double* d_JT, const double* d_F0, void* hip_stream) {
    if (!p || !d_x || !d_h || !d_JT || !d_F0) return fail(1, "og_fd_columns_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_fd_columns_dev: bad column range");
    ogk_args a;
    fill_args(p, &a, d_x, d_h, const_cast<double*>(d_F0), d_JT, lo, hi);
    int rc = a.jt_sparse ? p->launch(&a, 10, hip_stream) : 0;       // count this launch into the registered buffer
    if (!rc) rc = p->launch(&a, p->sweep_mode == 5 ? 1 : p->sweep_mode, hip_stream);
    if (rc) return launch_failed(p, rc, "og_fd_columns_dev");
    return 0;
}

int og_jacobian_exact_dev(og_handle p, const double* d_x, int32_t lo, int32_t hi, double* d_JT, double* d_F0,
                          void* hip_stream) {
    if (!p || !d_x || !d_JT || !d_F0) return fail(1, "og_jacobian_exact_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_jacobian_exact_dev: bad column range");
    ogk_args a;
    p->flag_slot ^= 1;
    fill_args(p, &a, d_x, nullptr, d_F0, d_JT, lo, hi);
    p->nf_read = a.nonfinite;
    a.jt_bump = a.jt_sparse;
    int rc = p->launch(&a, 0, hip_stream);          // F(x0) and the base collocation products
    if (!rc) rc = p->launch(&a, p->exact_mode, hip_stream);   // forward-mode derivatives, column by column
    if (rc) return launch_failed(p, rc, "og_jacobian_exact_dev");
    return 0;
}

int og_eval(og_handle p, const double* x, double* F) {
    if (!p || !x || !F) return fail(1, "og_eval: null argument");
    OG_HIP(hipSetDevice(p->device));
    int rc = upload_point(p, x, nullptr);
    if (rc) return rc;
    rc = og_eval_dev(p, p->d_x, p->d_f0, p->stream);
    if (rc) return rc;
    OG_HIP(hipMemcpyAsync(p->h_down, p->d_f0, sizeof(double) * p->m, hipMemcpyDeviceToHost, p->stream));
    OG_HIP(hipStreamSynchronize(p->stream));
    memcpy(F, p->h_down, sizeof(double) * (size_t)p->m);
    return 0;
}

int og_fd_sweep(og_handle p, const double* x, const double* hstep, int32_t lo, int32_t hi,
                double* JT, double* F0) {
    if (!p || !x || !hstep || !JT) return fail(1, "og_fd_sweep: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_fd_sweep: bad column range");
    OG_HIP(hipSetDevice(p->device));
    const size_t need = (size_t)(hi - lo) * (size_t)p->m;
    if (need == 0) {                     // empty column range: only F(x) is produced
        if (!F0) return 0;
        return og_eval(p, x, F0);
    }
    {
        const int rcj = own_jt(p, lo, hi);
        if (rcj) return rcj;
    }
    g_host_clock.start();
    // (mapped matrix: the launch also reads x | h in place, out of a pinned buffer of their own - non-coherent host memory,
    // cached in L2 for the launch and visible at its boundary - instead of waiting for their copy: 5 us of a call.
    // OGPSX_HOST=copyx keeps the copy.)
    static const bool readx_mode = [] { const char* e = getenv("OGPSX_HOST"); return !(e && std::string(e) == "copyx"); }();
    og_problem_s::host_reg* mreg = find_host_reg(p, JT, lo, hi);
    static const int forced = [] {
        const char* e = getenv("OGPSX_HOST");
        return (e && (std::string(e) == "mapped" || std::string(e) == "copyx")) ? 1 : 0;
    }();
    bool use_mapped = mreg && mreg->mapped && p->sweep_mode == 5 && p->fused_ok;
    if (use_mapped && forced) mreg->choice = 1;
    const bool trial = use_mapped && mreg->choice == 0 && !forced;
    if (use_mapped && !forced) use_mapped = mreg->choice == 1 || (mreg->choice == 0 && mreg->calls < 5);
    const double t_call = trial ? host_clock::now() : 0.0;
    auto trial_done = [&] {
        // calls 0-1 mapped (warm-up), 2-4 mapped timed, 5-6 packed (warm-up: pattern, staging, its device buffer), 7-9 packed
        // timed; the FASTEST call of each decides (a mean would carry one-time costs of the first calls)
        if (!trial) return;
        const double dt = host_clock::now() - t_call;
        const int c = mreg->calls++;
        if (c >= 2 && c <= 4) mreg->best_mapped = std::min(mreg->best_mapped, dt);
        if (c >= 7 && c <= 9) mreg->best_staged = std::min(mreg->best_staged, dt);
        mreg->t_mapped = mreg->best_mapped;
        mreg->t_staged = mreg->best_staged;
        if (mreg->calls >= 10) mreg->choice = mreg->best_mapped <= mreg->best_staged ? 1 : 2;
    };
    const bool in_place = readx_mode && use_mapped;
    int rc = in_place ? 0 : upload_point(p, x, hstep);
    if (rc) return rc;
    g_host_clock.mark(0);
    if (og_problem_s::host_reg* reg = mreg; use_mapped) {
        // mapped host matrix: ONE launch writes the non-zeros into the caller's matrix and F(x0) + the count of non-finite
        // rows into the pinned staging buffer, both over PCIe; the host waits for the launch and is done
        rc = ensure_staging(p, (size_t)p->m + 1);
        if (rc) return rc;
        if (reg->dirty && reg->last_staged) {        // (a NaN block the packed path downloaded: not the device's to clean)
            memset(JT, 0, sizeof(double) * (size_t)(hi - lo) * (size_t)p->m);
            reg->dirty = false;
        }
        reg->last_staged = false;
        void* tail = nullptr;
        OG_HIP(hipHostGetDevicePointer(&tail, p->h_down, 0));
        const double *kx = p->d_x, *kh = p->d_h;
        if (in_place) {
            if (!p->h_xh) OG_HIP(hipHostMalloc(&p->h_xh, sizeof(double) * 2 * (size_t)p->n, hipHostMallocNonCoherent | hipHostMallocMapped));
            memcpy(p->h_xh, x, sizeof(double) * (size_t)p->n);
            memcpy(p->h_xh + p->n, hstep, sizeof(double) * (size_t)p->n);
            void* dxh = nullptr;
            OG_HIP(hipHostGetDevicePointer(&dxh, p->h_xh, 0));
            kx = (const double*)dxh;
            kh = kx + p->n;
        }
        ogk_args a;
        fill_args(p, &a, kx, kh, (double*)tail, reg->mapped, lo, hi);
        if (a.jt_sparse) {
            a.nonfinite = p->d_flags + 3;
            p->nf_read = a.nonfinite_result;
            a.ptail = (double*)tail;
            rc = p->launch(&a, 5, p->stream);
            if (rc) return launch_failed(p, rc, "og_fd_sweep");
            g_host_clock.mark(1);
            OG_HIP(hipStreamSynchronize(p->stream));
            g_host_clock.mark(3);
            if (F0) memcpy(F0, p->h_down, sizeof(double) * (size_t)p->m);
            // (rows of NaN in every column were written by the launch itself and the next sweep cleans them, as in a
            // device buffer; a packed download into this matrix in between - the exact mode - zeroes it first)
            reg->dirty = p->h_down[(size_t)p->m] != 0.0;
            g_host_clock.mark(4);
            g_host_clock.done();
            trial_done();
            return 0;
        }
    }
    if (find_host_reg(p, JT, lo, hi) && p->sweep_mode == 5 && p->fused_ok) {
        // registered host matrix: ONE launch leaves the packed non-zeros, F(x0) and the count of non-finite rows
        // contiguous in the staging buffer, one pinned copy brings them down
        rc = ensure_pattern(p);
        if (rc) return rc;
        const int64_t first = p->indptr[(size_t)lo], nnz = p->indptr[(size_t)hi] - first;
        rc = ensure_staging(p, (size_t)nnz + (size_t)p->m + 1);
        if (rc) return rc;
        ogk_args a;
        fill_args(p, &a, p->d_x, p->d_h, p->d_down + nnz, p->d_jt, lo, hi);
        if (a.jt_sparse) {
            a.nonfinite = p->d_flags + 3;
            p->nf_read = a.nonfinite_result;
            a.poff = p->d_indptr;
            a.pvals = p->d_down - first;
            a.ptail = p->d_down + nnz;
            rc = p->launch(&a, 5, p->stream);
            if (rc) return launch_failed(p, rc, "og_fd_sweep");
            g_host_clock.mark(1);
            rc = download_block(p, lo, hi, JT, F0, nullptr, true);
            trial_done();
            return rc;
        }
    }
    rc = og_fd_sweep_dev(p, p->d_x, p->d_h, lo, hi, p->d_jt, p->d_f0, p->stream);
    if (rc) return rc;
    return download_block(p, lo, hi, JT, F0);
}

int og_jacobian_exact(og_handle p, const double* x, int32_t lo, int32_t hi, double* JT, double* F0) {
    if (!p || !x || !JT) return fail(1, "og_jacobian_exact: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_jacobian_exact: bad column range");
    OG_HIP(hipSetDevice(p->device));
    const size_t need = (size_t)(hi - lo) * (size_t)p->m;
    if (need == 0) {
        if (!F0) return 0;
        return og_eval(p, x, F0);
    }
    {
        const int rcj = own_jt(p, lo, hi);
        if (rcj) return rcj;
    }
    int rc = upload_point(p, x, nullptr);
    if (rc) return rc;
    rc = og_jacobian_exact_dev(p, p->d_x, lo, hi, p->d_jt, p->d_f0, p->stream);
    if (rc) return rc;
    return download_block(p, lo, hi, JT, F0);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Column sharding over several devices of one node from ONE process:
// og_comm_init names the devices (and brings up one RCCL communicator per device, ncclCommInitAll, when
// librccl is loadable); an og_multi handle holds one sub-handle, one stream and one full replica of J_T
// per device.  A sweep: every device gets x and h, sweeps ITS block of columns into its replica, packs the
// non-zeros; one grouped ncclAllGather over xGMI (or, OGPSX_GATHER=peer, G-1 hipMemcpyPeerAsync per device)
// exchanges the packed blocks; every device scatters the others' non-zeros into its replica.  The host
// result comes from device 0.
// ------------------------------------------------------------------------------------------------
namespace {

typedef void* ogn_comm;
struct ogn_unique_id { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
struct rccl_api {
    void* lib = nullptr;
    int (*CommInitAll)(ogn_comm*, int, const int*) = nullptr;
    int (*GetUniqueId)(ogn_unique_id*) = nullptr;
    int (*CommInitRank)(ogn_comm*, int, ogn_unique_id, int) = nullptr;
    int (*CommDestroy)(ogn_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ogn_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
const int OGN_FLOAT64 = 8;              // ncclFloat64 (rccl.h)

struct comm_state {
    std::vector<int> devs;
    std::vector<ogn_comm> comms;        // empty: peer copies
    rccl_api api;
} g_comm;

bool load_rccl(rccl_api* api) {
    // a librccl that is already in the process (torch loads its own) wins over the system one
    const char* names[] = {nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        void* lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) continue;
        void* sym = dlsym(lib, "ncclCommInitAll");
        if (!sym) continue;
        api->lib = lib;
        api->CommInitAll = (int (*)(ogn_comm*, int, const int*))sym;
        api->GetUniqueId = (int (*)(ogn_unique_id*))dlsym(lib, "ncclGetUniqueId");
        api->CommInitRank = (int (*)(ogn_comm*, int, ogn_unique_id, int))dlsym(lib, "ncclCommInitRank");
        api->CommDestroy = (int (*)(ogn_comm))dlsym(lib, "ncclCommDestroy");
        api->AllGather = (int (*)(const void*, void*, size_t, int, ogn_comm, hipStream_t))dlsym(lib, "ncclAllGather");
        api->GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
        api->GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        api->GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        return api->CommDestroy && api->AllGather && api->GroupStart && api->GroupEnd;
    }
    return false;
}

}  // namespace

struct og_multi_s {
    int G = 0;
    int n = 0, m = 0, B = 0;
    int64_t block_vals = 0;
    bool rccl = false;
    std::vector<int> devs;              // the handle's own copy of the device list ...
    std::vector<ogn_comm> comms;        // ... and its OWN communicators (ncclCommInitAll at og_multi_create): a later
                                        // og_comm_init, or a second handle on other devices, cannot pull them away
    std::vector<og_problem_s*> sub;
    std::vector<double*> d_full;        // n x m replica per device (rows of the device's block are registered)
    std::vector<double*> d_send, d_recv;
    std::vector<hipEvent_t> packed;     // peer mode: block g's packed values are ready
    std::vector<hipEvent_t> fetched;    // peer mode: device g has copied the other blocks of the previous step
    std::vector<hipEvent_t> uploaded;   // the pinned upload buffer of device g has been read
    bool stepped = false;
};

extern "C" {

// ---- one process per GPU: this rank's own RCCL communicator (the launcher distributes the unique id) ----
int og_shard_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(1, "og_shard_comm_unique_id: null argument");
    if (!g_comm.api.lib && !load_rccl(&g_comm.api)) return fail(7, "og_shard_comm_unique_id: librccl could not be loaded");
    if (!g_comm.api.GetUniqueId) return fail(7, "og_shard_comm_unique_id: ncclGetUniqueId not found");
    ogn_unique_id id;
    const int rc = g_comm.api.GetUniqueId(&id);
    if (rc != 0) return fail(7, std::string("og_shard_comm_unique_id: ") + (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rc) : "?"));
    memcpy(id128, id.internal, sizeof id.internal);
    return 0;
}

int og_shard_comm_init(og_handle p, const uint8_t* id128, int32_t rank, int32_t world) {
    if (!p || !id128) return fail(1, "og_shard_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(1, "og_shard_comm_init: bad rank / world");
    if (!g_comm.api.lib && !load_rccl(&g_comm.api)) return fail(7, "og_shard_comm_init: librccl could not be loaded");
    if (!g_comm.api.CommInitRank) return fail(7, "og_shard_comm_init: ncclCommInitRank not found");
    if (p->shard_world != world) {
        const int rcp = og_shard_plan(p, world, nullptr, nullptr);
        if (rcp) return rcp;
    }
    og_shard_comm_destroy(p);
    OG_HIP(hipSetDevice(p->device));
    ogn_unique_id id;
    memcpy(id.internal, id128, sizeof id.internal);
    ogn_comm comm = nullptr;
    const int rc = g_comm.api.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(7, std::string("og_shard_comm_init: ncclCommInitRank: ") +
                                    (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rc) : "?"));
    p->shard_comm = comm;
    p->shard_rank = rank;
    return 0;
}

int og_shard_comm_info(og_handle p, int32_t* nranks, int32_t* rank, int32_t* device) {
    if (!p) return fail(1, "og_shard_comm_info: null handle");
    if (!p->shard_comm) return fail(1, "og_shard_comm_info: call og_shard_comm_init first");
    // what RCCL itself says about the communicator (not what the launcher asked for)
    typedef int (*query_fn)(ogn_comm, int*);
    const query_fn count = (query_fn)dlsym(g_comm.api.lib, "ncclCommCount");
    const query_fn user = (query_fn)dlsym(g_comm.api.lib, "ncclCommUserRank");
    const query_fn dev = (query_fn)dlsym(g_comm.api.lib, "ncclCommCuDevice");
    if (!count || !user || !dev) return fail(7, "og_shard_comm_info: ncclCommCount / UserRank / CuDevice not found");
    int v[3] = {-1, -1, -1};
    const int rc = count((ogn_comm)p->shard_comm, &v[0]) | user((ogn_comm)p->shard_comm, &v[1]) |
                   dev((ogn_comm)p->shard_comm, &v[2]);
    if (rc != 0) return fail(7, "og_shard_comm_info: an RCCL query failed");
    if (nranks) *nranks = v[0];
    if (rank) *rank = v[1];
    if (device) *device = v[2];
    return 0;
}

void og_shard_comm_destroy(og_handle p) {
    if (p && p->shard_comm && g_comm.api.CommDestroy) g_comm.api.CommDestroy((ogn_comm)p->shard_comm);
    if (p) p->shard_comm = nullptr;
}

int og_shard_all_gather_dev(og_handle p, const double* d_send, double* d_recv, void* hip_stream) {
    if (!p || !d_send || !d_recv) return fail(1, "og_shard_all_gather_dev: null argument");
    if (!p->shard_comm) return fail(1, "og_shard_all_gather_dev: call og_shard_comm_init first");
    const int rc = g_comm.api.AllGather(d_send, d_recv, (size_t)p->shard_block_vals, OGN_FLOAT64, (ogn_comm)p->shard_comm,
                                        (hipStream_t)hip_stream);
    if (rc != 0) return fail(7, std::string("og_shard_all_gather_dev: ncclAllGather: ") +
                                    (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rc) : "?"));
    return 0;
}

int og_comm_init(int32_t G, const int32_t* devs) {
    if (G < 1 || !devs) return fail(1, "og_comm_init: need at least one device");
    og_comm_finalize();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(3, "og_comm_init: no HIP device available");
    bool distinct = true;
    for (int i = 0; i < G; ++i) {
        if (devs[i] < 0 || devs[i] >= ndev) return fail(3, "og_comm_init: device ordinal out of range");
        for (int k = 0; k < i; ++k) distinct = distinct && devs[k] != devs[i];
    }
    g_comm.devs.assign(devs, devs + G);
    const char* mode = getenv("OGPSX_GATHER");
    const bool want_peer = mode && std::string(mode) == "peer";
    if (!distinct && !want_peer)
        return fail(1, "og_comm_init: a device is listed twice (only allowed with OGPSX_GATHER=peer, for tests)");
    if (!want_peer) {
        if (!load_rccl(&g_comm.api)) {
            if (mode && std::string(mode) == "rccl") return fail(7, "og_comm_init: librccl could not be loaded");
        } else {
            g_comm.comms.assign((size_t)G, nullptr);
            const int rc = g_comm.api.CommInitAll(g_comm.comms.data(), G, g_comm.devs.data());
            if (rc != 0) {
                g_comm.comms.clear();
                return fail(7, std::string("og_comm_init: ncclCommInitAll: ") +
                                   (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rc) : "?"));
            }
        }
    }
    if (g_comm.comms.empty())            // peer copies: let every pair of distinct devices reach each other
        for (int i = 0; i < G; ++i)
            for (int k = 0; k < G; ++k)
                if (devs[i] != devs[k]) {
                    hipSetDevice(devs[i]);
                    hipDeviceEnablePeerAccess(devs[k], 0);      // "already enabled" is fine
                    (void)hipGetLastError();
                }
    return 0;
}

void og_comm_finalize(void) {
    for (ogn_comm c : g_comm.comms)
        if (c && g_comm.api.CommDestroy) g_comm.api.CommDestroy(c);
    g_comm.comms.clear();
    g_comm.devs.clear();
}

int og_comm_size(void) { return (int)g_comm.devs.size(); }
int og_comm_uses_rccl(void) { return g_comm.comms.empty() ? 0 : 1; }

void og_multi_destroy(og_multi mh) {
    if (!mh) return;
    for (int g = 0; g < (int)mh->sub.size(); ++g) {
        if (!mh->sub[(size_t)g]) continue;
        hipSetDevice(mh->sub[(size_t)g]->device);
        if ((size_t)g < mh->d_full.size()) hipFree(mh->d_full[(size_t)g]);
        if ((size_t)g < mh->d_send.size()) hipFree(mh->d_send[(size_t)g]);
        if ((size_t)g < mh->d_recv.size()) hipFree(mh->d_recv[(size_t)g]);
        if ((size_t)g < mh->packed.size() && mh->packed[(size_t)g]) hipEventDestroy(mh->packed[(size_t)g]);
        if ((size_t)g < mh->fetched.size() && mh->fetched[(size_t)g]) hipEventDestroy(mh->fetched[(size_t)g]);
        if ((size_t)g < mh->uploaded.size() && mh->uploaded[(size_t)g]) hipEventDestroy(mh->uploaded[(size_t)g]);
        og_problem_destroy(mh->sub[(size_t)g]);
    }
    for (ogn_comm c : mh->comms)
        if (c && g_comm.api.CommDestroy) g_comm.api.CommDestroy(c);
    delete mh;
}

int og_multi_create(const og_desc* desc, og_multi* out) {
    if (!desc || !out) return fail(1, "og_multi_create: null argument");
    *out = nullptr;
    const int G = (int)g_comm.devs.size();
    if (G < 1) return fail(1, "og_multi_create: call og_comm_init first");
    og_multi_s* mh = new og_multi_s();
    mh->G = G;
    mh->devs = g_comm.devs;
    mh->rccl = !g_comm.comms.empty();
    if (mh->rccl) {
        // communicators of this handle's own (the ones og_comm_init made stay with the process-wide state)
        mh->comms.assign((size_t)G, nullptr);
        const int rcn = g_comm.api.CommInitAll(mh->comms.data(), G, mh->devs.data());
        if (rcn != 0) {
            mh->comms.clear();
            delete mh;
            return fail(7, std::string("og_multi_create: ncclCommInitAll: ") +
                               (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rcn) : "?"));
        }
    }
    mh->sub.assign((size_t)G, nullptr);
    mh->fetched.assign((size_t)G, nullptr);
    mh->uploaded.assign((size_t)G, nullptr);
    mh->d_full.assign((size_t)G, nullptr);
    mh->d_send.assign((size_t)G, nullptr);
    mh->d_recv.assign((size_t)G, nullptr);
    mh->packed.assign((size_t)G, nullptr);
    for (int g = 0; g < G; ++g) {
        og_desc d = *desc;
        d.device = mh->devs[(size_t)g];
        int rc = og_problem_create(&d, &mh->sub[(size_t)g]);
        if (rc) {
            og_multi_destroy(mh);
            return rc;
        }
    }
    mh->n = mh->sub[0]->n;
    mh->m = mh->sub[0]->m;
    for (int g = 0; g < G; ++g) {
        og_problem_s* p = mh->sub[(size_t)g];
        int32_t B = 0;
        int64_t bv = 0;
        int rc = og_shard_plan(p, G, &B, &bv);
        if (rc) {
            og_multi_destroy(mh);
            return rc;
        }
        mh->B = B;
        mh->block_vals = bv;
        const size_t full = sizeof(double) * (size_t)mh->n * (size_t)mh->m;
        const size_t blk = sizeof(double) * (size_t)(bv ? bv : 1);
        hipError_t e = hipSetDevice(p->device);
        if (e == hipSuccess) e = hipMalloc(&mh->d_full[(size_t)g], full);
        if (e == hipSuccess) e = hipMemsetAsync(mh->d_full[(size_t)g], 0, full, p->stream);
        if (e == hipSuccess) e = hipMalloc(&mh->d_send[(size_t)g], blk);
        if (e == hipSuccess) e = hipMemsetAsync(mh->d_send[(size_t)g], 0, blk, p->stream);
        if (e == hipSuccess) e = hipMalloc(&mh->d_recv[(size_t)g], blk * (size_t)G);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mh->packed[(size_t)g], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mh->fetched[(size_t)g], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mh->uploaded[(size_t)g], hipEventDisableTiming);
        if (e != hipSuccess) {
            og_multi_destroy(mh);
            return fail(100 + (int)e, std::string("og_multi_create: ") + hipGetErrorString(e));
        }
        const int lo = std::min(mh->n, g * B), hi = std::min(mh->n, lo + B);
        if (hi > lo) {
            rc = og_jt_register_dev(p, mh->d_full[(size_t)g] + (size_t)lo * (size_t)mh->m, lo, hi, p->stream);
            if (rc) {
                og_multi_destroy(mh);
                return rc;
            }
        }
    }
    *out = mh;
    return 0;
}

int og_multi_devices(og_multi mh) { return mh ? mh->G : 0; }

int og_multi_replica_dev(og_multi mh, int32_t g, double** d_JT_full, double** d_F0, void** hip_stream) {
    if (!mh || g < 0 || g >= mh->G) return fail(1, "og_multi_replica_dev: bad argument");
    if (d_JT_full) *d_JT_full = mh->d_full[(size_t)g];
    if (d_F0) *d_F0 = mh->sub[(size_t)g]->d_f0;
    if (hip_stream) *hip_stream = (void*)mh->sub[(size_t)g]->stream;
    return 0;
}

// enqueue one sharded sweep on every device (x and hstep are host vectors); no synchronisation of the devices.
// Safe to call again before the previous step has finished: the pinned upload buffer is reused only after its
// copy has been read (a host wait on an event that is normally long past), and in peer mode a device overwrites
// its message only after every other device has fetched the previous one.
static int multi_enqueue(og_multi_s* mh, const double* x, const double* hstep) {
    const int G = mh->G, n = mh->n, B = mh->B;
    for (int g = 0; g < G; ++g) {
        og_problem_s* p = mh->sub[(size_t)g];
        OG_HIP(hipSetDevice(p->device));
        if (mh->stepped) OG_HIP(hipEventSynchronize(mh->uploaded[(size_t)g]));
        int rc = upload_point(p, x, hstep);
        if (rc) return rc;
        OG_HIP(hipEventRecord(mh->uploaded[(size_t)g], p->stream));
        if (mh->stepped && !mh->rccl && G > 1)
            for (int r = 0; r < G; ++r)
                if (r != g) OG_HIP(hipStreamWaitEvent(p->stream, mh->fetched[(size_t)r], 0));
        const int lo = std::min(n, g * B), hi = std::min(n, lo + B);
        double* block = mh->d_full[(size_t)g] + (size_t)lo * (size_t)mh->m;
        if (G > 1 || mh->rccl) {         // (one device with RCCL: the collective still runs - a plumbing check)
            // sweep of the device's block, its non-zeros straight into its message (one launch)
            rc = og_shard_sweep_dev(p, g, p->d_x, p->d_h, block, p->d_f0, mh->d_send[(size_t)g], p->stream);
            if (rc) return rc;
            if (!mh->rccl) OG_HIP(hipEventRecord(mh->packed[(size_t)g], p->stream));
        } else {
            if (hi > lo) rc = og_fd_sweep_dev(p, p->d_x, p->d_h, lo, hi, block, p->d_f0, p->stream);
            else rc = og_eval_dev(p, p->d_x, p->d_f0, p->stream);
            if (rc) return rc;
        }
    }
    if (G == 1 && !mh->rccl) {
        mh->stepped = true;
        return 0;
    }
    const size_t count = (size_t)mh->block_vals;
    if (mh->rccl) {
        int rc = g_comm.api.GroupStart();
        for (int g = 0; g < G && rc == 0; ++g)
            rc = g_comm.api.AllGather(mh->d_send[(size_t)g], mh->d_recv[(size_t)g], count, OGN_FLOAT64,
                                      mh->comms[(size_t)g], mh->sub[(size_t)g]->stream);
        const int rc2 = g_comm.api.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0)
            return fail(7, std::string("og_multi_fd_sweep: ncclAllGather: ") +
                               (g_comm.api.GetErrorString ? g_comm.api.GetErrorString(rc) : "?"));
    } else {
        for (int g = 0; g < G; ++g) {
            og_problem_s* p = mh->sub[(size_t)g];
            OG_HIP(hipSetDevice(p->device));
            for (int r = 0; r < G; ++r) {
                if (r == g) continue;
                OG_HIP(hipStreamWaitEvent(p->stream, mh->packed[(size_t)r], 0));
                OG_HIP(hipMemcpyPeerAsync(mh->d_recv[(size_t)g] + (size_t)r * count, p->device, mh->d_send[(size_t)r],
                                          mh->sub[(size_t)r]->device, sizeof(double) * count, p->stream));
            }
            OG_HIP(hipEventRecord(mh->fetched[(size_t)g], p->stream));
        }
    }
    mh->stepped = true;
    for (int g = 0; g < G; ++g) {
        og_problem_s* p = mh->sub[(size_t)g];
        OG_HIP(hipSetDevice(p->device));
        int rc = og_shard_unpack_dev(p, g, mh->d_recv[(size_t)g], mh->d_full[(size_t)g], p->stream);
        if (rc) return rc;
    }
    return 0;
}

int og_multi_fd_sweep_enqueue(og_multi mh, const double* x, const double* hstep) {
    if (!mh || !x || !hstep) return fail(1, "og_multi_fd_sweep_enqueue: null argument");
    return multi_enqueue(mh, x, hstep);
}

int og_multi_synchronize(og_multi mh) {
    if (!mh) return fail(1, "og_multi_synchronize: null handle");
    for (int g = 0; g < mh->G; ++g) {
        OG_HIP(hipSetDevice(mh->sub[(size_t)g]->device));
        OG_HIP(hipStreamSynchronize(mh->sub[(size_t)g]->stream));
    }
    return 0;
}

int og_multi_fd_sweep(og_multi mh, const double* x, const double* hstep, double* JT, double* F0) {
    if (!mh || !x || !hstep || !JT) return fail(1, "og_multi_fd_sweep: null argument");
    int rc = multi_enqueue(mh, x, hstep);
    if (rc) return rc;
    // devices 1.. finish their replicas too before the call returns (one call in flight per handle)
    for (int g = 1; g < mh->G; ++g) {
        OG_HIP(hipSetDevice(mh->sub[(size_t)g]->device));
        OG_HIP(hipStreamSynchronize(mh->sub[(size_t)g]->stream));
    }
    og_problem_s* p0 = mh->sub[0];
    OG_HIP(hipSetDevice(p0->device));
    return download_block(p0, 0, mh->n, JT, F0, mh->d_full[0]);
}

int og_multi_jt_register_host(og_multi mh, double* JT) {
    if (!mh) return fail(1, "og_multi_jt_register_host: null handle");
    return og_jt_register_host(mh->sub[0], JT, 0, mh->n);
}

int og_multi_eval(og_multi mh, const double* x, double* F) {
    if (!mh) return fail(1, "og_multi_eval: null handle");
    return og_eval(mh->sub[0], x, F);
}

}  // extern "C"
