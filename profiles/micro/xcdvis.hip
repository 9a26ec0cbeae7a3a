// Cross-XCD visibility of global stores by cache-policy bits on gfx950, measured: workgroup B (one XCD) first READS a line (so its
// L1 / L2 hold it), workgroup A (another XCD) then writes it and raises a flag, B reads the line again.  How often B still sees the
// old value, for every store flavour x load flavour - what a persistent multi-workgroup kernel may rely on without cache-wide
// write-back / invalidate fences (persistent form of the in-place A-trous iteration).
//   hipcc --offload-arch=gfx950 -O3 xcdvis.hip -o xcdvis
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t ld_mode(const uint32_t *p, int mode)
{
    uint32_t v;
    if (mode == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if (mode == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if (mode == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_mode(uint32_t *p, uint32_t v, int mode)
{
    if (mode == 0) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (mode == 1) asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// flags: [0] = B ready (round + 1), [32] = A wrote (round + 1); both polled with device-scope atomic RMWs (always coherent)
__global__ __launch_bounds__(64) void k_vis(uint32_t *data, uint32_t *flags, int rounds, int st, int ldm, int b_block, uint32_t *result)
{
    if (threadIdx.x != 0) return;
    const bool is_a = blockIdx.x == 0, is_b = (int)blockIdx.x == b_block;
    if (!is_a && !is_b) return;
    if (is_a) result[2] = xcc_id(); else result[3] = xcc_id();
    uint32_t stale = 0, hang = 0;
    for (int r = 0; r < rounds && !hang; r++) {
        uint32_t *line = data + (size_t)r * 64;              // a fresh 256-byte-apart line per round, initially 0
        if (is_b) {
            const uint32_t before = ld_mode(line, 0);        // plain read: the line is now in B's L1 and L2
            if (before != 0) stale |= 0x80000000u;
            atomicExch(&flags[0], (uint32_t)r + 1u);
            uint32_t spins = 0;
            while (atomicAdd(&flags[32], 0u) < (uint32_t)r + 1u) if (++spins > 20000000u) { hang = 1; break; }
            const uint32_t after = ld_mode(line, ldm);
            if (after != (uint32_t)r + 1u) stale++;
        } else {
            uint32_t spins = 0;
            while (atomicAdd(&flags[0], 0u) < (uint32_t)r + 1u) if (++spins > 20000000u) { hang = 1; break; }
            st_mode(line, (uint32_t)r + 1u, st);
            atomicExch(&flags[32], (uint32_t)r + 1u);
        }
    }
    if (is_b) { result[0] = stale; result[1] = hang; }
}
int main()
{
    const int rounds = 2000;
    uint32_t *data, *flags, *result;
    hipMalloc(&data, (size_t)rounds * 256); hipMalloc(&flags, 512); hipMalloc(&result, 64);
    const char *sn[3] = {"plain", "sc1", "sc0 sc1"}, *ln[4] = {"plain", "sc0", "sc1", "sc0 sc1"};
    for (int b_block : {1, 8}) {            // block 1: the next XCD under round-robin dispatch; block 8: the same XCD as block 0
        for (int st = 0; st < 3; st++)
            for (int ldm = 0; ldm < 4; ldm++) {
                hipMemset(data, 0, (size_t)rounds * 256); hipMemset(flags, 0, 512); hipMemset(result, 0, 64);
                hipDeviceSynchronize();
                k_vis<<<16, 64>>>(data, flags, rounds, st, ldm, b_block, result);
                hipDeviceSynchronize();
                uint32_t h[4];
                hipMemcpy(h, result, 16, hipMemcpyDeviceToHost);
                printf("writer XCC %u, reader XCC %u (block %d): store %-8s load %-8s : stale %4u of %d%s%s\n", h[2], h[3], b_block, sn[st], ln[ldm],
                       h[0] & 0x7fffffffu, rounds, (h[0] >> 31) ? " (first read not 0!)" : "", h[1] ? " HANG" : "");
            }
    }
    return 0;
}
