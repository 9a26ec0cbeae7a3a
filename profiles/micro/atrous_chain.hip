// The dependent chain of ONE pass of the in-place A-trous iteration (csrc/ycge_post.hip: pass_compute), run by one wavefront that has a
// compute unit to itself and everything but the chain removed (no prefetch for the next pass, no other wavefronts on the SIMD, no
// neighbouring band to wait for): the LDS look-up of the tap rewritten one pass ago -> luminance difference -> IEEE division by cPhi ->
// the binary64 exponential -> four products -> LDS -> the 25 ordered adds of a component -> DPP -> IEEE reciprocal -> product -> LDS
// insert -> barrier.  Each pass's output is the next pass's tap (the (-1, 0) tap of the reference's scan order, RaytraceRenderer.cs:718),
// so nothing overlaps between passes.  What it prints - shader clocks and microseconds per pass - times the levels of a frame
// (W/2 + 3H/2: the recurrence T(x, y) = 1 + max(T(x - 2, y), T(x + 4, y - 2)) of the in-place iteration at step 2) is a FLOOR for
// that iteration however it is scheduled: compare profiles/r05 (post stage) and DESIGN section 8.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../yetanotherconsolegameengine_amd/csrc atrous_chain.hip -o atrous_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ycge_math.h"

__global__ __launch_bounds__(64) void k_chain(int passes, float c_phi, float *out, unsigned long long *clocks)
{
    __shared__ float val[2][4][28];
    __shared__ uint4 ent[64];
    const int t = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (threadIdx.x < 64) ent[threadIdx.x] = make_uint4(threadIdx.x, __float_as_uint(0.4f + 0.001f * t), __float_as_uint(0.5f), __float_as_uint(0.3f));
    __syncthreads();
    const float c0x = 0.41f, c0y = 0.52f, c0z = 0.33f;
    const float wn = 0.9f, wz = 0.8f, wa = 0.95f, w_base = 0.0625f;
    uint32_t slot = 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < passes; p++) {
        const uint4 en = ent[(slot + (uint32_t)t) & 63u];                    // the tap's new colour, written one pass ago
        const float cjx = __uint_as_float(en.y), cjy = __uint_as_float(en.z), cjz = __uint_as_float(en.w);
        const float dl = fabsf((0.2126f * cjx + 0.7152f * cjy + 0.0722f * cjz) - (0.2126f * c0x + 0.7152f * c0y + 0.0722f * c0z));
        const float wc = ycge::m_exp(-dl / c_phi);
        const float w = w_base * wc * wn * wz * wa;
        if (t < 25) { val[g][0][t] = cjx * w; val[g][1][t] = cjy * w; val[g][2][t] = cjz * w; val[g][3][t] = w; }
        float acc = 0.0f;
        if (t < 4) {
            float v[28];
            const float4 *row = (const float4 *)&val[g][t][0];
#pragma unroll
            for (int k = 0; k < 6; k++) { const float4 q = row[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
            v[24] = val[g][t][24];
#pragma unroll
            for (int k = 0; k < 25; k++) acc = acc + v[k];
        }
        const float wsum = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xFF, 0xF, 0xF, true));
        slot = (slot + 1u) & 63u;
        if (wsum > 1e-8f && t < 4) {
            const float inv = 1.0f / wsum;
            (&ent[(slot + (uint32_t)t) & 63u].x)[(t + 1) & 3] = t < 3 ? __float_as_uint(acc * inv) : ((slot + (uint32_t)t) & 63u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; out[0] = __uint_as_float(ent[3].y); }
}

int main()
{
    float *out; unsigned long long *clk;
    hipMalloc(&out, 64); hipMalloc(&clk, 64);
    const int passes = 20000;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, passes, 3.0f, out, clk);
        hipDeviceSynchronize();
        unsigned long long h[2]; float o;
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
        const double us = (double)h[1] / 100.0 / passes;        // s_memrealtime: 100 MHz
        printf("one wavefront alone: %.0f shader clocks, %.3f us per pass (%d passes; result %g)\n", (double)h[0] / passes, us, passes, o);
        if (rep == 2)
            for (int wh : {1080, 2160}) {
                const int w = wh == 1080 ? 1920 : 3840, levels = w / 2 + 3 * wh / 2;
                printf("  %dx%d: %d levels x %.3f us = %.2f ms - a floor for the in-place iteration however it is scheduled\n", w, wh, levels, us, levels * us * 1e-3);
            }
    }
    return 0;
}
