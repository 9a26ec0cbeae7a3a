// What does one traversal step cost the memory pipeline when EVERY lane of EVERY resident wavefront fetches its own 64..72-byte
// record (the bulk of k_trace: 4 wavefronts per SIMD, 64 divergent lanes)?  Dependent chase over 64-byte records, variants of HOW the
// record is fetched:
//   0  per lane: 4 x global_load_dwordx4 + 1 x dwordx2 (what mesh_walk does today: 5 instructions x 64 lanes = 320 L1 tag look-ups)
//   1  per lane: 4 x global_load_dwordx4
//   2  per QUAD: instruction r fetches the record of the quad's lane r, lane j of the quad taking 16-byte chunk j - the four lanes
//      of a quad read one contiguous 64 bytes, which the texture-address unit coalesces into ONE look-up (4 instructions x 16
//      look-ups); the chunks are then handed to their owners through LDS (ds_write_b128 x 4, ds_read_b128 x 4)
//   3  as 2 without the LDS hand-over (memory side alone)
//   4  per lane: 1 x global_load_dwordx4 (a 16-byte record: lower bound of the per-lane form)
//   5  per QUAD as 2, but the four fetches are LDS-DMA (global_load_lds_dwordx4: the chunk of lane L lands at plane r + 16 L, no VGPR,
//      no ds_write); lane L then reads its own record back with 4 x ds_read_b128 from plane (L & 3), columns 4 (L / 4) .. + 3
// Reports ns per step per wavefront and steps per microsecond per CU, for 4 wavefronts per SIMD resident on every CU.
//   hipcc --offload-arch=gfx950 -O3 fetchrate.hip -o fetchrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rate(const uint8_t *recs, uint32_t mask, int steps, int valu, uint32_t *out)
{
    __shared__ f32x4 xch[4 * 64];
    __shared__ __attribute__((aligned(16))) uint8_t dma[4 * 1040];
    const int lane = threadIdx.x;
    uint32_t i = (blockIdx.x * 64u + lane) * 2654435761u & mask;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        f32x4 a, b, c, e;
        f32x2 f = {0.0f, 0.0f};
        if (MODE == 0 || MODE == 1) {
            const uint8_t *p = recs + (size_t)i * 64;
            if (MODE == 0)
                asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %5, off offset:16\n\tglobal_load_dwordx4 %2, %5, off offset:32\n\t"
                             "global_load_dwordx4 %3, %5, off offset:48\n\tglobal_load_dwordx2 %4, %5, off offset:64\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e), "=&v"(f) : "v"(p) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\t"
                             "global_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e) : "v"(p) : "memory");
        } else if (MODE == 2 || MODE == 3) {
            // the record indices of the quad's four lanes
            const int q0 = lane & ~3, j = lane & 3;
            const uint32_t i0 = (uint32_t)__shfl((int)i, q0 + 0, 64), i1 = (uint32_t)__shfl((int)i, q0 + 1, 64);
            const uint32_t i2 = (uint32_t)__shfl((int)i, q0 + 2, 64), i3 = (uint32_t)__shfl((int)i, q0 + 3, 64);
            const uint8_t *p0 = recs + (size_t)i0 * 64 + j * 16, *p1 = recs + (size_t)i1 * 64 + j * 16;
            const uint8_t *p2 = recs + (size_t)i2 * 64 + j * 16, *p3 = recs + (size_t)i3 * 64 + j * 16;
            f32x4 r0, r1, r2, r3;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\t"
                         "global_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
            if (MODE == 2) {
                // chunk j of record (q0 + r) goes to slot [j][q0 + r]; lane L then reads slots [0..3][L]
                xch[j * 64 + q0 + 0] = r0; xch[j * 64 + q0 + 1] = r1; xch[j * 64 + q0 + 2] = r2; xch[j * 64 + q0 + 3] = r3;
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): one wavefront, no barrier needed
                a = xch[0 * 64 + lane]; b = xch[1 * 64 + lane]; c = xch[2 * 64 + lane]; e = xch[3 * 64 + lane];
            } else {
                a = r0; b = r1; c = r2; e = r3;
                // the owner of the next pointer is the quad's lane 0 in register set j
                const float nx = j == 0 ? r0.x : j == 1 ? r1.x : j == 2 ? r2.x : r3.x;
                a.x = __shfl(nx, q0, 64);
            }
        } else if (MODE == 5) {
            const int q0 = lane & ~3, j = lane & 3;
            const uint32_t i0 = (uint32_t)__shfl((int)i, q0 + 0, 64), i1 = (uint32_t)__shfl((int)i, q0 + 1, 64);
            const uint32_t i2 = (uint32_t)__shfl((int)i, q0 + 2, 64), i3 = (uint32_t)__shfl((int)i, q0 + 3, 64);
            const uint8_t *p0 = recs + (size_t)i0 * 64 + j * 16, *p1 = recs + (size_t)i1 * 64 + j * 16;
            const uint8_t *p2 = recs + (size_t)i2 * 64 + j * 16, *p3 = recs + (size_t)i3 * 64 + j * 16;
            const uint32_t lds0 = (uint32_t)(uintptr_t)dma;      // plane r at lds0 + r * 1040 (16 bytes of padding: the quad's four planes fall on different banks)
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_add_u32 m0, m0, 1040\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_add_u32 m0, m0, 1040\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                         "s_add_u32 m0, m0, 1040\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                         "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(lds0) : "memory", "scc");
            const uint32_t rd = lds0 + (uint32_t)j * 1040u + (uint32_t)q0 * 16u;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e) : "v"(rd) : "memory");
        } else {
            const uint8_t *p = recs + (size_t)i * 64;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
            b = a; c = a; e = a;
        }
        i = __float_as_uint(a.x) & mask;
        float x = b.x;
        for (int v = 0; v < valu; v++) x = x * 1.0001f + c.y;    // dependent VALU chain standing in for the slab tests
        acc += x + e.w + f.x;
    }
    out[blockIdx.x * 64 + lane] = i + (uint32_t)acc;
}

template <int MODE> static float run(const uint8_t *d, uint32_t mask, int steps, int valu, int waves, uint32_t *out)
{
    k_rate<MODE><<<waves, 64>>>(d, mask, 50, valu, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_rate<MODE><<<waves, 64>>>(d, mask, steps, valu, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const size_t max_recs = (size_t)1 << 22;    // 256 MiB of 64-byte records
    std::vector<uint32_t> h(max_recs * 16);
    uint64_t z = 88172645463325252ull;
    for (size_t r = 0; r < max_recs; r++) for (int k = 0; k < 16; k++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[r * 16 + k] = (uint32_t)z; }
    uint8_t *d; hipMalloc(&d, max_recs * 64 + 256);
    hipMemcpy(d, h.data(), max_recs * 64, hipMemcpyHostToDevice);
    uint32_t *out; hipMalloc(&out, 8192 * 64 * 4);
    const int steps = 400, cus = 256;
    {   // the cooperative forms must walk the same chains as the per-lane form
        std::vector<uint32_t> o1(1024 * 64), o2(1024 * 64), o5(1024 * 64);
        run<1>(d, (1u << 20) - 1, 100, 0, 1024, out); hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost);
        run<2>(d, (1u << 20) - 1, 100, 0, 1024, out); hipMemcpy(o2.data(), out, o2.size() * 4, hipMemcpyDeviceToHost);
        run<5>(d, (1u << 20) - 1, 100, 0, 1024, out); hipMemcpy(o5.data(), out, o5.size() * 4, hipMemcpyDeviceToHost);
        size_t bad2 = 0, bad5 = 0;
        for (size_t k = 0; k < o1.size(); k++) { bad2 += o1[k] != o2[k]; bad5 += o1[k] != o5[k]; }
        printf("self-check: mode 2 vs 1 mismatches %zu, mode 5 (LDS-DMA) vs 1 mismatches %zu of %zu\n", bad2, bad5, o1.size());
    }
    for (size_t recs : {(size_t)1 << 16, (size_t)1 << 20, (size_t)1 << 22}) {      // 4 MB (L2), 64 MB (the BVH of config 4: Infinity Cache), 256 MB
        for (int waves : {1024, 4096}) {
            for (int valu : {0, 120}) {
                float ms[6];
                ms[0] = run<0>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                ms[1] = run<1>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                ms[2] = run<2>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                ms[3] = run<3>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                ms[4] = run<4>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                ms[5] = run<5>(d, (uint32_t)(recs - 1), steps, valu, waves, out);
                printf("footprint %6.1f MB waves %5d valu %3d : ns/step/wave", recs * 64 / 1048576.0, waves, valu);
                for (int m = 0; m < 6; m++) printf("  m%d %7.1f", m, ms[m] * 1e6 / steps);
                printf("   | steps/us/CU");
                for (int m = 0; m < 6; m++) printf(" %6.2f", (double)waves * steps / (ms[m] * 1e3) / cus);
                printf("\n");
            }
        }
    }
    return 0;
}
