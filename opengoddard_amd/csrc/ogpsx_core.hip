// ogpsx_core.hip -- libogpsx.so: the generic runtime behind include/ogpsx.h.
//
// LGL construction (host function + HIP kernel from one shared source, og_lgl.h), SciPy's
// forward-difference step rule, and problem handles that bind a compiled callback module
// (libogk_<hash>.so, built from ogk_kernels.hip + a generated header) to a device.
// Reference lines replaced by each entry point are listed in include/ogpsx.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

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

}  // namespace

struct og_problem_s {
    int device = 0;
    int n = 0, m = 0, m_eq = 0, m_ineq = 0;
    void* module = nullptr;
    ogk_launch_fn launch = nullptr;
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
        int slot;
        unsigned launches;
    };
    std::vector<jt_reg> regs;
    int last_reg = -1;                  // registration the most recent fill_args matched (-1: none)
    uint32_t* d_state = nullptr;
};
static const int OG_MAX_JT_REGS = 63;
static const size_t OG_TRACE_DOUBLES = (size_t)1 << 20;     // 16384 workgroups x 8 wavefronts x 8 stamps

namespace {

void fill_args(og_problem_s* p, ogk_args* a, const double* x, const double* h, double* f0,
               double* jt, int lo, int hi) {
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
    // a registered buffer (exactly this block of columns at this address) is written sparsely
    a->jt_sparse = 0;
    a->jt_gen = 1;
    a->jt_state = p->d_state;
    p->last_reg = -1;
    if (jt)
        for (size_t i = 0; i < p->regs.size(); ++i) {
            auto& r = p->regs[i];
            if (r.ptr == jt && r.lo == lo && r.hi == hi) {
                a->jt_sparse = 1;
                a->jt_gen = ++r.launches;
                a->jt_state = p->d_state + r.slot;
                p->last_reg = (int)i;
                break;
            }
        }
    memcpy(a->dfrag_off, p->dfrag_off, sizeof(a->dfrag_off));
}

// a launch that was not accepted has not happened as far as the buffer's generation count goes
int launch_failed(og_problem_s* p, int rc, const char* where) {
    if (p->last_reg >= 0) p->regs[(size_t)p->last_reg].launches -= 1;
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

}  // namespace

extern "C" {

const char* og_last_error(void) { return g_error.c_str(); }

int og_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
    p->launch = launch;

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
    if (e == hipSuccess) e = hipMalloc(&p->d_x, sizeof(double) * (size_t)p->n);
    if (e == hipSuccess) e = hipMalloc(&p->d_h, sizeof(double) * (size_t)p->n);
    if (e == hipSuccess) e = hipMalloc(&p->d_f0, sizeof(double) * (size_t)p->m);
    if (e == hipSuccess) e = hipMalloc(&p->d_y0, sizeof(double) * (size_t)(info.n_y0 > 0 ? info.n_y0 : 1));
    if (e == hipSuccess) e = hipMalloc(&p->d_xop, sizeof(double) * (size_t)(info.n_y0 > 0 ? info.n_y0 : 1));
    if (e == hipSuccess) e = hipMalloc(&p->d_t0, sizeof(double) * (size_t)p->m);
    if (e == hipSuccess) e = hipMalloc(&p->d_z, sizeof(double) * (size_t)p->m);
    if (e == hipSuccess) e = hipMalloc(&p->d_flags, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(p->d_flags, 0, 4 * sizeof(int));
    if (e == hipSuccess && getenv("OGPSX_TRACE")) {
        e = hipMalloc(&p->d_trace, sizeof(double) * OG_TRACE_DOUBLES);
        if (e == hipSuccess) e = hipMemset(p->d_trace, 0, sizeof(double) * OG_TRACE_DOUBLES);
    }
    if (e == hipSuccess) e = hipMalloc(&p->d_state, (OG_MAX_JT_REGS + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(p->d_state, 0xff, (OG_MAX_JT_REGS + 1) * sizeof(uint32_t));
    p->n_eval_blocks = info.n_eval_blocks;
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
    if (p->stream) hipStreamDestroy(p->stream);
    hipFree(p->d_dfrag);
    hipFree(p->d_cvec);
    hipFree(p->d_x);
    hipFree(p->d_h);
    hipFree(p->d_f0);
    hipFree(p->d_jt);
    hipFree(p->d_y0);
    hipFree(p->d_xop);
    hipFree(p->d_t0);
    hipFree(p->d_z);
    hipFree(p->d_flags);
    hipFree(p->d_state);
    hipFree(p->d_trace);
    if (p->module) dlclose(p->module);
    delete p;
}

int og_sweep_mode(og_handle p) { return p ? p->sweep_mode : 0; }

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
        p->regs.push_back({d_JT, lo, hi, slot, 0u});
        reg = &p->regs.back();
    }
    reg->lo = lo;
    reg->hi = hi;
    reg->launches = 0;
    OG_HIP(hipMemsetAsync(d_JT, 0, sizeof(double) * (size_t)(hi - lo) * (size_t)p->m, s));
    OG_HIP(hipMemsetAsync(p->d_state + reg->slot, 0xff, sizeof(uint32_t), s));
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
    int rc = p->launch(&a, 0, hip_stream);
    if (rc) return fail(100 + rc, std::string("og_eval_dev: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int og_fd_sweep_dev(og_handle p, const double* d_x, const double* d_h, int32_t lo, int32_t hi,
                    double* d_JT, double* d_F0, void* hip_stream) {
    if (!p || !d_x || !d_h || !d_JT || !d_F0) return fail(1, "og_fd_sweep_dev: null argument");
    if (lo < 0 || hi > p->n || lo > hi) return fail(1, "og_fd_sweep_dev: bad column range");
    ogk_args a;
    p->flag_slot ^= 1;
    fill_args(p, &a, d_x, d_h, d_F0, d_JT, lo, hi);
    int rc;
    if (p->sweep_mode == 5 && hi > lo) {
        // one launch (the module falls back to two when d_JT is not a registered buffer)
        rc = p->launch(&a, 5, hip_stream);
    } else {
        rc = p->launch(&a, 0, hip_stream);          // F(x0) first: the sweep subtracts it
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
    int rc = p->launch(&a, p->sweep_mode == 5 ? 1 : p->sweep_mode, hip_stream);
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
    int rc = p->launch(&a, 0, hip_stream);          // F(x0) and the base collocation products
    if (!rc) rc = p->launch(&a, p->exact_mode, hip_stream);   // forward-mode derivatives, column by column
    if (rc) return launch_failed(p, rc, "og_jacobian_exact_dev");
    return 0;
}

int og_eval(og_handle p, const double* x, double* F) {
    if (!p || !x || !F) return fail(1, "og_eval: null argument");
    OG_HIP(hipSetDevice(p->device));
    OG_HIP(hipMemcpyAsync(p->d_x, x, sizeof(double) * p->n, hipMemcpyHostToDevice, p->stream));
    int rc = og_eval_dev(p, p->d_x, p->d_f0, p->stream);
    if (rc) return rc;
    OG_HIP(hipMemcpyAsync(F, p->d_f0, sizeof(double) * p->m, hipMemcpyDeviceToHost, p->stream));
    OG_HIP(hipStreamSynchronize(p->stream));
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
    OG_HIP(hipMemcpyAsync(p->d_x, x, sizeof(double) * p->n, hipMemcpyHostToDevice, p->stream));
    OG_HIP(hipMemcpyAsync(p->d_h, hstep, sizeof(double) * p->n, hipMemcpyHostToDevice, p->stream));
    int rc = og_fd_sweep_dev(p, p->d_x, p->d_h, lo, hi, p->d_jt, p->d_f0, p->stream);
    if (rc) return rc;
    if (need)
        OG_HIP(hipMemcpyAsync(JT, p->d_jt, sizeof(double) * need, hipMemcpyDeviceToHost, p->stream));
    if (F0)
        OG_HIP(hipMemcpyAsync(F0, p->d_f0, sizeof(double) * p->m, hipMemcpyDeviceToHost, p->stream));
    OG_HIP(hipStreamSynchronize(p->stream));
    return 0;
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
    OG_HIP(hipMemcpyAsync(p->d_x, x, sizeof(double) * p->n, hipMemcpyHostToDevice, p->stream));
    int rc = og_jacobian_exact_dev(p, p->d_x, lo, hi, p->d_jt, p->d_f0, p->stream);
    if (rc) return rc;
    OG_HIP(hipMemcpyAsync(JT, p->d_jt, sizeof(double) * need, hipMemcpyDeviceToHost, p->stream));
    if (F0) OG_HIP(hipMemcpyAsync(F0, p->d_f0, sizeof(double) * p->m, hipMemcpyDeviceToHost, p->stream));
    OG_HIP(hipStreamSynchronize(p->stream));
    return 0;
}

}  // extern "C"
