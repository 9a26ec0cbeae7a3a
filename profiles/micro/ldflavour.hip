// Latency of a dependent 12-byte load by cache-policy bits on gfx950 (plain / sc0 / sc1 / sc0 sc1), one lane and 25 lanes of a
// wavefront, 4 MB footprint (L2-resident), and what a write-through (sc1) store + s_waitcnt vmcnt(0) costs.  For the persistent
// form of the in-place A-trous iteration: rows another workgroup wrote are read with sc1, own rows with sc0.
//   hipcc --offload-arch=gfx950 -O3 ldflavour.hip -o ldflavour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE> __global__ __launch_bounds__(64) void k(const uint32_t *buf, uint32_t mask, int steps, int active, uint32_t *out, unsigned long long *ticks)
{
    const int lane = threadIdx.x;
    uint32_t i = (blockIdx.x * 64u + lane) * 2654435761u & mask;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane < active)
        for (int s = 0; s < steps; s++) {
            uint32_t v;
            const uint32_t *p = buf + (size_t)i * 4;
            if (MODE == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            if (MODE == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            if (MODE == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            i = v & mask;
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = i;
}
template <int MODE> __global__ __launch_bounds__(64) void kst(uint32_t *buf, int steps, unsigned long long *ticks)
{
    uint32_t *p = buf + (size_t)(blockIdx.x * 64 + threadIdx.x) * 32;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; s++) {
        uint32_t v = s;
        if (MODE == 0) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
        if (MODE == 1) asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
        if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main()
{
    const size_t n = (size_t)1 << 18;       // 4 MB of 16-byte records
    std::vector<uint32_t> h(n * 4);
    uint64_t z = 88172645463325252ull;
    for (auto &x : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; x = (uint32_t)z; }
    uint32_t *d, *out; unsigned long long *ticks;
    hipMalloc(&d, n * 16); hipMalloc(&out, 64 * 64 * 4); hipMalloc(&ticks, 64 * 8);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    const int steps = 2000;
    const char *names[5] = {"plain", "sc0", "sc1", "sc0 sc1", "nt"};
    for (int mode = 0; mode < 5; mode++)
        for (int active : {1, 25, 64}) {
            unsigned long long t;
            auto run = [&](int st) {
                if (mode == 0) k<0><<<1, 64>>>(d, n - 1, st, active, out, ticks);
                if (mode == 1) k<1><<<1, 64>>>(d, n - 1, st, active, out, ticks);
                if (mode == 2) k<2><<<1, 64>>>(d, n - 1, st, active, out, ticks);
                if (mode == 3) k<3><<<1, 64>>>(d, n - 1, st, active, out, ticks);
                if (mode == 4) k<4><<<1, 64>>>(d, n - 1, st, active, out, ticks);
            };
            run(4000); run(steps);
            hipDeviceSynchronize();
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("load  %-8s lanes %2d : %7.1f ns per dependent load\n", names[mode], active, t * 10.0 / steps);
        }
    const char *snames[3] = {"plain", "sc1", "sc0 sc1"};
    for (int mode = 0; mode < 3; mode++) {
        unsigned long long t;
        if (mode == 0) kst<0><<<1, 64>>>(d, steps, ticks);
        if (mode == 1) kst<1><<<1, 64>>>(d, steps, ticks);
        if (mode == 2) kst<2><<<1, 64>>>(d, steps, ticks);
        hipDeviceSynchronize();
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("store %-8s + wait   : %7.1f ns per store acknowledged\n", snames[mode], t * 10.0 / steps);
    }
    return 0;
}
