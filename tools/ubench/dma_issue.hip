// Micro-benchmark (round 5): what does it cost a wave to ISSUE n LDS-DMA loads (global_load_lds_dwordx4, 64 lanes x 16 B each) back to back,
// and how long until they have all landed -- against the same n loads into registers?  256 CUs x 2 blocks x 4 waves, every wave reads its own
// 1 KB pieces of a 256 MB buffer (HBM) or of a 8 MB buffer (L2-resident after the warm-up launch).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_issue.hip -o /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N, int DMA> __global__ __launch_bounds__(256, 2) void k(const char *src, size_t span, float *out, unsigned long long *cyc, int reps)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * 256];          // 64 KB: 16 KB per wave
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds + wid * 16384;
    unsigned long long t_issue = 0, t_all = 0;
    f32x4 acc = {0, 0, 0, 0};
    size_t pos = ((size_t)blockIdx.x * 4 + wid) * 1024 * N;
    for (int r = 0; r < reps; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        f32x4 v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const char *p = src + (pos + (size_t)i * 1024) % span + lane * 16;
            if constexpr (DMA) glds16(p, lds0 + i * 1024);
            else v[i] = *(const f32x4 *)p;
        }
        if constexpr (!DMA) { asm volatile("" ::: "memory"); }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
#pragma unroll
            for (int i = 0; i < N; ++i) acc += v[i];
        }
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0; t_all += t2 - t0;
        pos += (size_t)gridDim.x * 4 * 1024 * N;
    }
    if (DMA) acc[0] += lds[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1];
    if (lane == 0) { cyc[(blockIdx.x * 4 + wid) * 2] = t_issue; cyc[(blockIdx.x * 4 + wid) * 2 + 1] = t_all; }
}
template <int N, int DMA> void run(const char *src, size_t span, float *out, unsigned long long *cyc, const char *what)
{
    const int reps = 64, grid = 512;
    k<N, DMA><<<grid, 256>>>(src, span, out, cyc, reps);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    k<N, DMA><<<grid, 256>>>(src, span, out, cyc, reps);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[512 * 4 * 2];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < grid * 4; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
    a /= grid * 4.0 * reps; b /= grid * 4.0 * reps;
    printf("%-4s %-9s n = %2d: issue %7.0f cycles (%5.0f per load), landed after %7.0f; %6.2f TB/s\n", what, DMA ? "LDS-DMA" : "registers", N, a, a / N, b,
           (double)grid * 4 * reps * N * 1024 / (ms * 1e-3) / 1e12);
}
int main()
{
    char *src; float *out; unsigned long long *cyc;
    const size_t big = 256ull << 20, small = 8ull << 20;
    (void)hipMalloc(&src, big); (void)hipMemset(src, 1, big); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&cyc, 512 * 4 * 2 * 8);
#define BOTH(N) run<N, 1>(src, span, out, cyc, what); run<N, 0>(src, span, out, cyc, what)
    for (int pass = 0; pass < 2; ++pass) {
        const size_t span = pass ? small : big; const char *what = pass ? "L2" : "HBM";
        BOTH(1); BOTH(2); BOTH(4); BOTH(8); BOTH(16);
    }
    return 0;
}
