// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of
// a wide streaming read; WRITE_SIZE is uncalibrated): a copy of a KNOWN byte count, far larger than the 256 MiB Infinity Cache, with
// the access widths the trace kernels use - 16 B per lane streaming (k_copy16) and 4 B per lane (k_copy4, the per-pixel result stores).
//   hipcc --offload-arch=gfx950 -O3 copycal.hip -o copycal ; rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./copycal   (then WRITE_SIZE)
// Prints the bytes each launch reads and writes; profiles/summarize.py divides them by the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy4(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main()
{
    const size_t bytes = (size_t)1 << 30;          // 1 GiB read + 1 GiB written per launch
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_copy16, dim3(256 * 16), dim3(256), 0, 0, (const f32x4 *)a, (f32x4 *)b, bytes / 16);
        hipLaunchKernelGGL(k_copy4, dim3(256 * 16), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("copycal bytes_read_per_launch=%zu bytes_written_per_launch=%zu\n", bytes, bytes);
    return 0;
}
