// Micro-test (round 6): does LDS-DMA (global_load_lds_dwordx4, LDS base in M0) reach LDS addresses above 64 KiB on gfx950 (160 KiB of LDS)?
// One 256-thread block copies 128 pieces of 1 KiB to LDS offsets 0 .. 128 KiB by DMA, reads them back with ds_read and compares.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsdma_hi.hip -o /tmp/ldsdma_hi && /tmp/ldsdma_hi
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void glds16(const void *base_uniform, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(256) void k(const unsigned *src, unsigned *bad)
{
    __shared__ __attribute__((aligned(16))) unsigned s[128 * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const char *)s);
    for (int k = 0; k < 32; ++k) {
        const int piece = wave * 32 + k;
        glds16((const char *)src + (size_t)piece * 1024, (unsigned)lane * 16u, base + (unsigned)piece * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int piece = 0; piece < 128; ++piece) {
        const unsigned v = s[piece * 256 + threadIdx.x];
        if (v != src[piece * 256 + threadIdx.x]) atomicAdd(&bad[piece], 1u);
    }
}
int main()
{
    unsigned *src, *bad, h[128 * 256], hb[128];
    hipMalloc(&src, sizeof(h)); hipMalloc(&bad, sizeof(hb));
    for (int e = 0; e < 128 * 256; ++e) h[e] = 0x10000u + e;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(bad, 0, sizeof(hb));
    k<<<1, 256>>>(src, bad);
    hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    int first_bad = -1, nbad = 0;
    for (int p = 0; p < 128; ++p) if (hb[p]) { if (first_bad < 0) first_bad = p; ++nbad; }
    printf("LDS-DMA to 128 KiB of LDS: %d of 128 pieces wrong, first wrong piece %d (LDS offset %d KiB)\n", nbad, first_bad, first_bad);
    return 0;
}
