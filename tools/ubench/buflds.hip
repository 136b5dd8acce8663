// Does `buffer_load_dwordx4 ... lds` (MUBUF LDS-DMA, 16 bytes per lane) zero-fill out-of-range lanes on gfx950?
// Build: hipcc --offload-arch=gfx950 -O2 -o buflds buflds.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *src, unsigned *dst, int nbytes)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
    for (int t = threadIdx.x; t < 512; t += 64) lds[t] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // lanes 0..47 in range, 48..55 beyond the end, 56..63 "negative" (wrapped) offsets
    unsigned off = lane * 16u;
    if (lane >= 48 && lane < 56) off = (unsigned)nbytes + (lane - 48) * 16u;
    if (lane >= 56) off = 0xFFFFFF00u + (lane - 56) * 16u;
    const unsigned ldsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds + 1024;   // second half
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(rsrc), "s"(ldsaddr) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = threadIdx.x; t < 512; t += 64) dst[t] = lds[t];
}
int main()
{
    const int n = 48 * 4;
    std::vector<unsigned> h(n + 64);
    for (int i = 0; i < n + 64; ++i) h[i] = 0x1000u + i;
    unsigned *s, *d;
    hipMalloc(&s, (n + 64) * 4); hipMalloc(&d, 512 * 4);
    hipMemcpy(s, h.data(), (n + 64) * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(s, d, n * 4);
    std::vector<unsigned> o(512);
    hipMemcpy(o.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) if (o[t] != 0xDEADBEEFu) ++bad;                 // first half untouched
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const unsigned want = l < 48 ? 0x1000u + l * 4 + e : 0u;
            if (o[256 + l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %08x want %08x\n", l, e, o[256 + l * 4 + e], want); ++bad; }
        }
    printf("buffer_load_dwordx4 lds: %s (%d mismatches)\n", bad ? "MISMATCH" : "in-range lanes copied, out-of-range lanes zero-filled", bad);
    return bad != 0;
}
