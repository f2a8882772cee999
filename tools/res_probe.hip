// tools/res_probe.hip - what an exchange between resident workgroups costs on this part.
// Measurements behind csrc/ogsqp_resident.h: self-validating 16-byte records (32 bits of payload + the 32-bit number of
// the exchange per 64-bit word), stored and polled with agent-scope (sc1) accesses.
//   hipcc --offload-arch=gfx950 -O3 tools/res_probe.hip -o tools/_build/res_probe && tools/_build/res_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef u64 rec2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void put(u64* rec, double v, unsigned tag) {
    const u64 b = (u64)__double_as_longlong(v), t = (u64)tag << 32;
    rec2 r;
    r.x = t | (b & 0xffffffffull);
    r.y = t | (b >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(rec), "v"(r) : "memory");
}
__device__ __forceinline__ bool tryget(const u64* rec, unsigned tag, double& v) {
    rec2 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(rec) : "memory");
    if ((unsigned)(r.x >> 32) != tag || (unsigned)(r.y >> 32) != tag) return false;
    v = __longlong_as_double((long long)((r.x & 0xffffffffull) | (r.y << 32)));
    return true;
}
__device__ __forceinline__ double get(const u64* rec, unsigned tag) {
    double v = 0.0;
    int spins = 0;
    while (!tryget(rec, tag, v))
        if (++spins > (1 << 22)) break;
    return v;
}
// two loads in flight per poll
__device__ __forceinline__ double get2(const u64* rec, unsigned tag) {
    int spins = 0;
    while (true) {
        rec2 a, b;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tglobal_load_dwordx4 %1, %2, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(rec) : "memory");
        if ((unsigned)(a.x >> 32) == tag && (unsigned)(a.y >> 32) == tag)
            return __longlong_as_double((long long)((a.x & 0xffffffffull) | (a.y << 32)));
        if ((unsigned)(b.x >> 32) == tag && (unsigned)(b.y >> 32) == tag)
            return __longlong_as_double((long long)((b.x & 0xffffffffull) | (b.y << 32)));
        if (++spins > (1 << 22)) return 0.0;
    }
}

// mode 0: all-to-all, one record per workgroup at `stride` records, threads tid < NW poll one record each
// mode 1: the same with two loads in flight
// mode 2: one workgroup (round-robin) broadcasts `len` records, everybody polls them (thread per record)
// mode 3: all-to-all through a counter: store, atomic add, lane 0 polls the counter, then everybody loads
// mode 4: all-to-all, but ONE wavefront polls all records (NW <= 64 * 4 in four trips) and tells the others through LDS
__global__ __launch_bounds__(1024) void k_probe(u64* mail, unsigned* counter, int NW, int rounds, int mode, int stride,
                                               int len, long long* ticks, double* sink) {
    __shared__ double s_v[1024];
    const int tid = threadIdx.x, w = blockIdx.x;
    unsigned tag = 0;
    double acc = 0.0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        ++tag;
        u64* area = mail + 2 * (size_t)((tag & 1u) * (size_t)(mode == 2 ? len : NW * stride));
        if (mode == 0 || mode == 1 || mode == 4) {
            if (tid == 960) put(area + 2 * (size_t)w * stride, (double)(w + r), tag);
            double v = 0.0;
            if (mode == 4) {
                if (tid < 64)
                    for (int j = tid; j < NW; j += 64) v += get(area + 2 * (size_t)j * stride, tag);
            } else if (tid < NW) {
                v = mode == 0 ? get(area + 2 * (size_t)tid * stride, tag) : get2(area + 2 * (size_t)tid * stride, tag);
            }
            s_v[tid] = v;
            __syncthreads();
            acc += s_v[(tid * 7) & 63];
            __syncthreads();
        } else if (mode == 2) {
            const int src = r % NW;
            if (w == src && tid < 64)
                for (int j = tid; j < len; j += 64) put(area + 2 * (size_t)j, (double)(j + r), tag);
            double v = 0.0;
            for (int j = tid; j < len; j += 1024) v += get(area + 2 * (size_t)j, tag);
            s_v[tid] = v;
            __syncthreads();
            acc += s_v[(tid * 7) & 1023];
            __syncthreads();
            // (everybody must have read before the next broadcast into this area: an all-to-all record, as (1) gives it)
            ++tag;
            u64* bar = mail + 2 * (size_t)(2 * len) + 2 * (size_t)((tag & 1u) * NW);
            if (tid == 960) put(bar + 2 * (size_t)w, 1.0, tag);
            if (tid < NW) v = get(bar + 2 * (size_t)tid, tag);
            __syncthreads();
        } else {
            u64* area3 = mail + 2 * (size_t)((tag & 1u) * NW);
            if (tid == 0) {
                __hip_atomic_store(area3 + 2 * (size_t)w, (u64)(w + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NW * tag) { }
            }
            __syncthreads();
            double v = 0.0;
            if (tid < NW) v = (double)__hip_atomic_load(area3 + 2 * (size_t)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_v[tid] = v;
            __syncthreads();
            acc += s_v[(tid * 7) & 63];
            __syncthreads();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[w] = t1 - t0;
    sink[(size_t)w * 1024 + tid] = acc;
}

// ping-pong between two workgroups, one lane each: `rounds` there-and-back hand-offs
__global__ void k_pingpong(u64* mail, int rounds, long long* ticks) {
    const int w = blockIdx.x;
    if (threadIdx.x != 0) return;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        if (w == 0) {
            put(mail, (double)r, (unsigned)r);
            (void)get(mail + 64, (unsigned)r);
        } else {
            (void)get(mail, (unsigned)r);
            put(mail + 64, (double)r, (unsigned)r);
        }
    }
    ticks[w] = __builtin_amdgcn_s_memrealtime() - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    u64* mail;
    unsigned* counter;
    long long* ticks;
    double* sink;
    const size_t mail_words = 2 * 2 * (size_t)256 * 1100 + 4096;
    CK(hipMalloc(&mail, mail_words * sizeof(u64)));
    CK(hipMalloc(&counter, 64));
    CK(hipMalloc(&ticks, 256 * sizeof(long long)));
    CK(hipMalloc(&sink, 256 * 1024 * sizeof(double)));
    std::vector<long long> h(256);
    const int rounds = 2000;
    auto run = [&](int NW, int mode, int stride, int len, const char* what) {
        CK(hipMemset(mail, 0, mail_words * sizeof(u64)));
        CK(hipMemset(counter, 0, 64));
        hipLaunchKernelGGL(k_probe, dim3(NW), dim3(1024), 0, 0, mail, counter, NW, rounds, mode, stride, len, ticks, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), ticks, NW * sizeof(long long), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (int i = 0; i < NW; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-58s NW %3d stride %4d len %4d: %6.2f us per round\n", what, NW, stride, len, 0.01 * (double)mx / rounds);
    };
    {
        CK(hipMemset(mail, 0, mail_words * sizeof(u64)));
        hipLaunchKernelGGL(k_pingpong, dim3(2), dim3(64), 0, 0, mail, rounds, ticks);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), ticks, 2 * sizeof(long long), hipMemcpyDeviceToHost));
        printf("ping-pong, two workgroups, one lane each: %.2f us there and back = %.2f us per hand-off\n",
               0.01 * (double)h[0] / rounds, 0.005 * (double)h[0] / rounds);
    }
    const int sizes[] = {2, 8, 32, 64, 128, 171, 256};
    for (int NW : sizes) run(NW, 0, 1, 0, "all-to-all, records packed, a thread per record");
    for (int NW : {64, 171, 256}) run(NW, 0, 472, 0, "all-to-all, records 7.5 KB apart");
    for (int NW : {64, 171, 256}) run(NW, 1, 1, 0, "all-to-all, packed, two loads in flight per poll");
    for (int NW : {64, 171, 256}) run(NW, 4, 1, 0, "all-to-all, packed, ONE wavefront polls");
    for (int NW : {64, 171, 256}) run(NW, 3, 1, 0, "all-to-all through store + counter + poll + load");
    for (int NW : {64, 171, 256}) run(NW, 2, 1, 470, "broadcast of 470 records (+ the all-to-all that closes it)");
    for (int NW : {171}) run(NW, 2, 1, 64, "broadcast of 64 records (+ the all-to-all that closes it)");
    return 0;
}
