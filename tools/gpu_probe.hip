// gpu_probe.hip -- hardware facts the design relies on, checked on the MI355X itself.
//   1. v_mfma_f64_16x16x4_f64 accumulates as a k-ordered fma chain (bit-identical to a scalar
//      fma loop) and uses the C/D layout  col = lane & 15, row = (lane >> 4) + 4*reg.
//   2. f64 division / sqrt and the og_math.h functions give the same bits on gfx950 as on the
//      host CPU (the basis of the bit-exact CPU twin, oracle/twin.cpp).
//   3. achievable HBM bandwidth for a fill and a copy (context for roofline numbers).
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I opengoddard_amd/csrc tools/gpu_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "og_math.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define CK(e)                                                                     \
    do {                                                                          \
        hipError_t r_ = (e);                                                      \
        if (r_ != hipSuccess) {                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__); \
            exit(2);                                                              \
        }                                                                         \
    } while (0)

// A: 16 x K (row-major), B: K x 16 (row-major), out: 16 x 16
__global__ void mfma_probe(const double* A, const double* B, double* out, int K) {
    const int lane = threadIdx.x;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < K / 4; ++ks) {
        const int k = ks * 4 + (lane >> 4);
        const double a = A[(lane & 15) * K + k];
        const double b = B[k * 16 + (lane & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (lane >> 4) + 4 * reg, col = lane & 15;
        out[row * 16 + col] = acc[reg];
    }
}

__global__ void math_probe(const double* x, const double* y, double* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[0 * n + i] = x[i] / y[i];
    o[1 * n + i] = ogm::sqrt_(ogm::fabs_(x[i]));
    o[2 * n + i] = ogm::exp_(x[i]);
    o[3 * n + i] = ogm::sin_(x[i]);
    o[4 * n + i] = ogm::cos_(x[i]);
    o[5 * n + i] = ogm::log_(ogm::fabs_(y[i]));
    o[6 * n + i] = x[i] * y[i] + y[i];            // must NOT be contracted into an fma
    o[7 * n + i] = ogm::tan_(x[i]);
    o[8 * n + i] = ogm::atan2_(x[i], y[i]);
    o[9 * n + i] = ogm::asin_(x[i] / 30.0);
    o[10 * n + i] = ogm::acos_(x[i] / 30.0);
}

__global__ void fill_kernel(double4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_double4(1.0, 2.0, 3.0, 4.0);
}

__global__ void copy_kernel(const double4* s, double4* d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}

// MFMA / VALU f64 rate probes: NACC independent accumulator chains per wave, ITER steps.
template <int NACC>
__global__ void mfma_rate(double* out, int iters, double a0, double b0) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void fma_rate(double* out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, b, acc[i]);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void div_rate(double* out, int iters, double a0) {
    double acc = a0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) acc = 1.0 + 1.0 / acc;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double urand() { return (double)rand() / RAND_MAX; }

template <class F>
static float time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("devices=%d name=%s arch=%s CUs=%d clock=%d MHz mem=%.1f GB\n", ndev, prop.name,
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000,
           prop.totalGlobalMem / 1e9);
    int bad = 0;

    // ---- 1. MFMA chain order
    for (int K : {4, 20, 80, 128, 200}) {
        std::vector<double> A(16 * K), B(K * 16), ref(256), got(256);
        srand(1234 + K);
        for (auto& v : A) v = (urand() - 0.5) * exp(8 * (urand() - 0.5));
        for (auto& v : B) v = (urand() - 0.5) * exp(8 * (urand() - 0.5));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double acc = 0.0;
                for (int k = 0; k < K; ++k) acc = __builtin_fma(A[i * K + k], B[k * 16 + j], acc);
                ref[i * 16 + j] = acc;
            }
        double *dA, *dB, *dO;
        CK(hipMalloc(&dA, A.size() * 8));
        CK(hipMalloc(&dB, B.size() * 8));
        CK(hipMalloc(&dO, 256 * 8));
        CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
        CK(hipMemcpy(got.data(), dO, 256 * 8, hipMemcpyDeviceToHost));
        int diff = 0;
        double maxrel = 0;
        for (int i = 0; i < 256; ++i)
            if (got[i] != ref[i]) {
                ++diff;
                maxrel = fmax(maxrel, fabs(got[i] - ref[i]) / fabs(ref[i]));
            }
        printf("mfma_f64_16x16x4 K=%3d: %d/256 differ from the k-ordered fma chain (max rel %.3g)\n",
               K, diff, maxrel);
        bad += diff;
        hipFree(dA), hipFree(dB), hipFree(dO);
    }

    // ---- 2. scalar math parity host <-> device
    {
        const int n = 1 << 20;
        std::vector<double> x(n), y(n), o(11 * (size_t)n);
        srand(99);
        for (int i = 0; i < n; ++i) {
            x[i] = (urand() - 0.5) * 60.0;
            y[i] = (urand() - 0.5) * exp(20 * (urand() - 0.5));
            if (y[i] == 0.0) y[i] = 1.0;
        }
        double *dx, *dy, *dout;
        CK(hipMalloc(&dx, n * 8));
        CK(hipMalloc(&dy, n * 8));
        CK(hipMalloc(&dout, 11 * (size_t)n * 8));
        CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(math_probe, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
        CK(hipMemcpy(o.data(), dout, 11 * (size_t)n * 8, hipMemcpyDeviceToHost));
        const char* names[11] = {"div", "sqrt", "exp", "sin", "cos", "log", "mul+add", "tan",
                                 "atan2", "asin", "acos"};
        for (int f = 0; f < 11; ++f) {
            int diff = 0;
            for (int i = 0; i < n; ++i) {
                double h;
                switch (f) {
                    case 0: h = x[i] / y[i]; break;
                    case 1: h = ogm::sqrt_(ogm::fabs_(x[i])); break;
                    case 2: h = ogm::exp_(x[i]); break;
                    case 3: h = ogm::sin_(x[i]); break;
                    case 4: h = ogm::cos_(x[i]); break;
                    case 5: h = ogm::log_(ogm::fabs_(y[i])); break;
                    case 6: { volatile double t = x[i] * y[i]; h = t + y[i]; } break;
                    case 7: h = ogm::tan_(x[i]); break;
                    case 8: h = ogm::atan2_(x[i], y[i]); break;
                    case 9: h = ogm::asin_(x[i] / 30.0); break;
                    default: h = ogm::acos_(x[i] / 30.0); break;
                }
                const double d = o[(size_t)f * n + i];
                if (!(d == h) && !(d != d && h != h)) ++diff;
            }
            printf("host/device %-7s: %d of %d differ\n", names[f], diff, n);
            bad += diff;
        }
        hipFree(dx), hipFree(dy), hipFree(dout);
    }

    // ---- 2b. f64 MFMA / FMA / division rates (one wave, then the whole chip)
    {
        double* dout;
        CK(hipMalloc(&dout, sizeof(double) * 256 * 1024 * 16));
        const int iters = 20000;
        float t1 = time_ms([&] { hipLaunchKernelGGL(mfma_rate<1>, dim3(1), dim3(64), 0, 0, dout, iters, 1.0, 1e-9); });
        float t4 = time_ms([&] { hipLaunchKernelGGL(mfma_rate<4>, dim3(1), dim3(64), 0, 0, dout, iters, 1.0, 1e-9); });
        float t8 = time_ms([&] { hipLaunchKernelGGL(mfma_rate<8>, dim3(1), dim3(64), 0, 0, dout, iters, 1.0, 1e-9); });
        printf("mfma_f64_16x16x4, one wave: dependent chain %.1f ns/MFMA; 4 chains %.1f ns/MFMA; 8 chains %.1f ns/MFMA\n",
               t1 * 1e6 / iters, t4 * 1e6 / (4.0 * iters), t8 * 1e6 / (8.0 * iters));
        float tc = time_ms([&] { hipLaunchKernelGGL(mfma_rate<4>, dim3(256 * 4), dim3(256), 0, 0, dout, iters, 1.0, 1e-9); });
        printf("mfma_f64_16x16x4, 1024 blocks x 4 waves: %.2f TFLOP/s\n",
               1024.0 * 4 * 4 * iters * 2048.0 / (tc * 1e-3) / 1e12);
        float f1 = time_ms([&] { hipLaunchKernelGGL(fma_rate<1>, dim3(1), dim3(64), 0, 0, dout, iters * 8, 1.0, 1e-9); });
        float f8 = time_ms([&] { hipLaunchKernelGGL(fma_rate<8>, dim3(1), dim3(64), 0, 0, dout, iters * 8, 1.0, 1e-9); });
        printf("v_fma_f64, one wave: dependent chain %.2f ns/FMA; 8 chains %.2f ns/FMA\n",
               f1 * 1e6 / (iters * 8.0), f8 * 1e6 / (iters * 64.0));
        float fc = time_ms([&] { hipLaunchKernelGGL(fma_rate<8>, dim3(256 * 8), dim3(256), 0, 0, dout, iters, 1.0, 1e-9); });
        printf("v_fma_f64, 2048 blocks x 4 waves x 8 chains: %.2f TFLOP/s\n",
               2048.0 * 256 * 8 * iters * 2.0 / (fc * 1e-3) / 1e12);
        float d1 = time_ms([&] { hipLaunchKernelGGL(div_rate, dim3(1), dim3(64), 0, 0, dout, iters, 3.0); });
        printf("f64 add+div dependent chain, one wave: %.1f ns per (add, div)\n", d1 * 1e6 / iters);
        float tl = time_ms([&] { hipLaunchKernelGGL(fma_rate<1>, dim3(1), dim3(64), 0, 0, dout, 1, 1.0, 1e-9); }, 20);
        printf("empty-ish kernel, event to event: %.2f us\n", tl * 1e3);
        hipFree(dout);
    }

    // ---- 3. HBM fill / copy bandwidth
    {
        const size_t bytes = (size_t)2 << 30;
        double4 *a, *b;
        CK(hipMalloc(&a, bytes));
        CK(hipMalloc(&b, bytes));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const size_t n = bytes / sizeof(double4);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it)
                hipLaunchKernelGGL(fill_kernel, dim3(256 * 8), dim3(256), 0, 0, a, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("fill  2 GiB: %.2f TB/s\n", 5.0 * bytes / (ms * 1e-3) / 1e12);
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it)
                hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("copy  2 GiB: %.2f TB/s (read+write)\n", 5.0 * 2 * bytes / (ms * 1e-3) / 1e12);
        }
        hipFree(a), hipFree(b);
    }
    printf(bad ? "PROBE: MISMATCHES PRESENT\n" : "PROBE: ALL BIT-EXACT\n");
    return bad ? 1 : 0;
}
