// xcd_probe.hip - (1) which XCD does workgroup b of a small grid land on?  (2) round trip of a value between two
// workgroups through memory: agent-scope atomics (past the L2) against L2-coherent accesses (store + load that only skips
// the L1) when both workgroups sit on the same XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); return 2; } } while (0)

__global__ void where(int* xcc, int* hwid) {
    if (threadIdx.x == 0) {
        int x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x;
        hwid[blockIdx.x] = h;
    }
}

__device__ __forceinline__ double load_l2(const double* p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_l2(double* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// mode 0: agent-scope atomics; mode 1: sc0 loads / stores (coherent in the XCD's L2 only)
// (every wait is bounded by 200 000 polls, and the first wait that gives up ends the kernel on BOTH sides through a flag
// polled with agent scope: accesses that are not coherent between the two workgroups are a result, not a hang)
__global__ void pingpong(double* mail, int partner, int trips, int mode, long long* ticks, int* xcc) {
    const int b = blockIdx.x;
    if (b != 0 && b != partner) return;
    if (threadIdx.x != 0) return;
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc[b == 0 ? 0 : 1] = x;
    double* mine = mail + (b == 0 ? 0 : 64);
    double* theirs = mail + (b == 0 ? 64 : 0);
    int* gave_up = (int*)(mail + 128);
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    int done = 0;
    for (int k = 1; k <= trips; ++k) {
        if (b == 0) {
            if (mode == 0) __hip_atomic_store(mine, (double)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else store_l2(mine, (double)k);
        }
        long spins = 0;
        bool ok = false;
        while (true) {
            const double v = mode == 0 ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : load_l2(theirs);
            if (v == (double)k) { ok = true; break; }
            if (++spins > 200000 || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        if (!ok) {
            __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (b != 0) {
            if (mode == 0) __hip_atomic_store(mine, (double)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else store_l2(mine, (double)k);
        }
        done = k;
    }
    if (b == 0) {
        ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
        ticks[1] = done;
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int *xcc, *hw;
    CK(hipMalloc(&xcc, 4096 * 4));
    CK(hipMalloc(&hw, 4096 * 4));
    for (int G : {4, 8, 16, 40}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(where, dim3(G), dim3(512), 0, 0, xcc, hw);
            CK(hipDeviceSynchronize());
            std::vector<int> h(G);
            CK(hipMemcpy(h.data(), xcc, G * 4, hipMemcpyDeviceToHost));
            printf("grid %2d (launch %d): XCC of workgroup b =", G, rep);
            for (int b = 0; b < G; ++b) printf(" %d", h[b]);
            printf("\n");
        }
    }
    double* mail;
    long long* ticks;
    CK(hipMalloc(&mail, 4096));
    CK(hipMalloc(&ticks, 64));
    for (int partner : {1, 8, 16}) {
        for (int mode : {0, 1}) {
            CK(hipMemset(mail, 0, 4096));
            const int trips = 2000;
            hipLaunchKernelGGL(pingpong, dim3(partner + 1), dim3(64), 0, 0, mail, partner, trips, mode, ticks, xcc);
            CK(hipDeviceSynchronize());
            long long t[2];
            int x[2];
            CK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
            if (t[1] == trips)
                printf("ping-pong workgroups 0 <-> %2d (XCC %d / %d), %s: %.0f ns per round trip (two hand-offs)\n", partner, x[0],
                       x[1], mode == 0 ? "agent-scope atomics" : "sc0 store / sc0 load   ", t[0] * 10.0 / trips);
            else
                printf("ping-pong workgroups 0 <-> %2d (XCC %d / %d), %s: NOT COHERENT - gave up in round trip %lld\n", partner,
                       x[0], x[1], mode == 0 ? "agent-scope atomics" : "sc0 store / sc0 load   ", t[1] + 1);
        }
    }
    return 0;
}
