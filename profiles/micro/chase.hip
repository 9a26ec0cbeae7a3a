// Dependent-load latency of one traversal step on gfx950: every lane chases its own chain of 64-byte records
// (4 x global_load_dwordx4 + one wait, as mesh_walk does).  Reports ns per step for different footprints,
// active lanes per wavefront and wavefronts in flight.   hipcc --offload-arch=gfx950 -O3 chase.hip -o chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_record64(const void *p, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &e)
{
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e) : "v"(p) : "memory");
}
__global__ __launch_bounds__(64) void k_chase(const uint32_t *recs, uint32_t mask, int steps, int active, int valu, uint32_t *out, unsigned long long *ticks)
{
    const int lane = threadIdx.x;
    uint32_t i = (blockIdx.x * 64u + lane) * 2654435761u & mask;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane < active) {
        for (int s = 0; s < steps; s++) {
            f32x4 a, b, c, e;
            load_record64(recs + (size_t)i * 16, a, b, c, e);
            i = __float_as_uint(a.x) & mask;
            float x = b.x;
            for (int v = 0; v < valu; v++) x = x * 1.0001f + c.y;    // dependent VALU chain standing in for the slab tests
            acc += x + e.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = i + (uint32_t)acc;
}
int main()
{
    const size_t max_recs = (size_t)1 << 24;    // 1 GiB of 64-byte records
    std::vector<uint32_t> h(max_recs * 16);
    uint32_t *d; hipMalloc(&d, max_recs * 64);
    uint32_t *out; hipMalloc(&out, 8192 * 64 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8192 * 8);
    std::vector<unsigned long long> ht(8192);
    const int steps = 2000;
    for (size_t recs : {(size_t)1 << 14, (size_t)1 << 18, (size_t)1 << 20, (size_t)1 << 24}) {
        uint64_t z = 88172645463325252ull;
        for (size_t r = 0; r < recs; r++) for (int k = 0; k < 16; k++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[r * 16 + k] = (uint32_t)z; }
        hipMemcpy(d, h.data(), recs * 64, hipMemcpyHostToDevice);
        for (int waves : {1, 1024, 3072, 6144})
            for (int active : {1, 16, 64})
                for (int valu : {0, 100}) {
                    k_chase<<<waves, 64>>>(d, (uint32_t)(recs - 1), 200, active, valu, out, ticks);   // warm caches
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    k_chase<<<waves, 64>>>(d, (uint32_t)(recs - 1), steps, active, valu, out, ticks);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    hipMemcpy(ht.data(), ticks, waves * 8, hipMemcpyDeviceToHost);
                    double mean = 0; unsigned long long mx = 0; for (int w = 0; w < waves; w++) { mean += ht[w]; if (ht[w] > mx) mx = ht[w]; }
                    mean /= waves;
                    printf("footprint %7.1f MB waves %5d lanes %2d valu %3d : %7.1f ns/step mean-wave, %7.1f max-wave, kernel %8.3f ms\n",
                           recs * 64 / 1048576.0, waves, active, valu, mean * 10.0 / steps, mx * 10.0 / steps, ms);
                }
    }
    return 0;
}
