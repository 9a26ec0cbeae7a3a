// experiments/ycge_atrous_persist_groups.hip.h - EXPERIMENT (measured, rejected; compiled only with -DYCGE_EXPERIMENTS=1, i.e. into
// lib/var_experiments.so, which the parity test of the in-place A-trous forms loads for YCGE_POST_MODE=4).
//
// The in-place A-trous iteration as one persistent launch whose bands hand over per GROUP of levels (a band starts a group when the band above
// has finished the same group): it trails its neighbour by a whole group, like the launch form - no gain over it.  The production form
// hands over per LEVEL (k_atrous_stream, ycge_post.hip).
#pragma once

namespace ycge {

template <int G>
__global__ __launch_bounds__(32 * G) void k_atrous_persist(const AtrousParams A, float *__restrict__ buf, const float *__restrict__ statw,
                                                           const uint8_t *__restrict__ sky, const uint32_t *__restrict__ pixels,
                                                           const uint32_t *__restrict__ off, int levels, int K, int n_bands, int rows_per_band,
                                                           uint32_t wx, uint32_t *__restrict__ progress, uint32_t epoch, int xcd_local)
{
    __shared__ __attribute__((aligned(16))) PostSharedT<G> sh;
    int b = (int)blockIdx.x;
    if (xcd_local) { const int per_xcd = (n_bands + 7) / 8; b = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8; }
    if (b >= n_bands) return;
    const int groups = (levels + K - 1) / K;
    const uint32_t *o = off + (size_t)b * (levels + 1);
    const int grp = threadIdx.x >> 5, t = threadIdx.x & 31;
    const BandWindow W = {b * rows_per_band, rows_per_band, wx, 1u, 0};
    const uint32_t n_ent = (uint32_t)rows_per_band * wx;
    for (int g = 0; g < groups; g++) {
        const int t0 = g * K, t1 = t0 + K < levels ? t0 + K : levels;
        const uint32_t pass_lo = o[t0], pass_hi = o[t1];
        if (pass_lo < pass_hi) {
            if (b > 0)      // every lane polls the one word: no one-lane loop in front of the barriers below
                while ((int32_t)(__hip_atomic_load(&progress[(size_t)(b - 1) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < g + 1)
                    __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (uint32_t e = threadIdx.x; e < n_ent; e += 32 * G) sh.ent[e].x = YCGE_POST_NONE;
            uint32_t p1 = pixels[(size_t)pass_lo * G + grp];
            uint32_t p2 = pass_lo + 1 < pass_hi ? pixels[(size_t)(pass_lo + 1) * G + grp] : YCGE_POST_NONE;
            PassData D1 = pass_fetch<true, true>(A, W, buf, statw, sky, p1, t);
            lds_barrier();              // table cleared
            for (uint32_t i = pass_lo; i < pass_hi; i++) {
                const uint32_t p3 = i + 2 < pass_hi ? pixels[(size_t)(i + 2) * G + grp] : YCGE_POST_NONE;
                const PassData D2 = pass_fetch<true, true>(A, W, buf, statw, sky, p2, t);
                pass_compute<true>(A, p1, D1, sh);
                p1 = p2; D1 = D2; p2 = p3;
            }
            for (uint32_t e = threadIdx.x; e < n_ent; e += 32 * G) {
                const uint4 en = sh.ent[e];
                if (en.x != YCGE_POST_NONE) st3_dev(buf, en.x, f3(__uint_as_float(en.y), __uint_as_float(en.z), __uint_as_float(en.w)));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's colours are in memory
            __syncthreads();                                      // ... and everybody's; the table is free again
        }
        if (threadIdx.x == 0) __hip_atomic_store(&progress[(size_t)b * 32], epoch + (uint32_t)g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace ycge
