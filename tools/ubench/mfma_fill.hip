// Micro-benchmark (round 5): are v_mfma_f32_16x16x4_f32 and fp32 VALU FMAs independent pipes on gfx950?  One asm block holds 8 MFMAs
// (two accumulator chains) with N independent v_fmac_f32 after each; cycles per MFMA for N = 0..8 at 1, 2, 4 waves per SIMD, and the
// same with the bf16 MFMA 16x16x32 for comparison.   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fill.hip -o /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define F1 "v_fmac_f32 %2, %10, %11\n"
#define F2 F1 "v_fmac_f32 %3, %10, %11\n"
#define F3 F2 "v_fmac_f32 %4, %10, %11\n"
#define F4 F3 "v_fmac_f32 %5, %10, %11\n"
#define F5 F4 "v_fmac_f32 %6, %10, %11\n"
#define F6 F5 "v_fmac_f32 %7, %10, %11\n"
#define F7 F6 "v_fmac_f32 %8, %10, %11\n"
#define F8 F7 "v_fmac_f32 %9, %10, %11\n"
#define F0 ""
#define M32A "v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n"
#define M32B "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n"
#define M16A "v_mfma_f32_16x16x32_bf16 %0, %14, %15, %0\n"
#define M16B "v_mfma_f32_16x16x32_bf16 %1, %14, %15, %1\n"
#define BODY(MA, MB, F) MA F MB F MA F MB F MA F MB F MA F MB F
#define OPS : "+v"(a0), "+v"(a1), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(x), "v"(y), "v"(pa), "v"(pb), "v"(ha), "v"(hb)
template <int N, int BF, int NT> __global__ __launch_bounds__(NT) void k(float *out, const float *in, unsigned long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float x = in[lane], y = in[64 + lane], pa = in[128 + lane], pb = in[192 + lane];
    s16x8 ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (short)(0x3c00 + lane + e); hb[e] = (short)(0x3c00 + 2 * lane + e); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (BF == 0) {
            if constexpr (N == 0) asm volatile(BODY(M32A, M32B, F0) OPS);
            if constexpr (N == 1) asm volatile(BODY(M32A, M32B, F1) OPS);
            if constexpr (N == 2) asm volatile(BODY(M32A, M32B, F2) OPS);
            if constexpr (N == 4) asm volatile(BODY(M32A, M32B, F4) OPS);
            if constexpr (N == 6) asm volatile(BODY(M32A, M32B, F6) OPS);
            if constexpr (N == 8) asm volatile(BODY(M32A, M32B, F8) OPS);
        } else if constexpr (BF == 1) {
            if constexpr (N == 0) asm volatile(BODY(M16A, M16B, F0) OPS);
            if constexpr (N == 2) asm volatile(BODY(M16A, M16B, F2) OPS);
            if constexpr (N == 4) asm volatile(BODY(M16A, M16B, F4) OPS);
            if constexpr (N == 8) asm volatile(BODY(M16A, M16B, F8) OPS);
        } else {                                            // fillers only
            if constexpr (N == 4) asm volatile(BODY("", "", F4) OPS);
            if constexpr (N == 8) asm volatile(BODY("", "", F8) OPS);
        }
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + wid] = t1 - t0;
    float s = a0[0] + a1[1];
    for (int e = 0; e < 8; ++e) s += f[e];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int N, int BF, int NT> void run(float *out, float *in, unsigned long long *cyc)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<N, BF, NT><<<256, NT>>>(out, in, cyc, iters);        // warm
    hipEventRecord(e0, 0);
    k<N, BF, NT><<<256, NT>>>(out, in, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    const int nw = NT / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) mx += (double)h[b * 16 + w];
    mx /= 256.0 * nw;
    printf("%s + %d fillers each, %d waves/SIMD: %6.1f cycles per slot and wave, %6.1f per slot and SIMD; wall clock %6.2f ns per slot and SIMD (s_memtime ticks at %.2f GHz)\n", BF == 0 ? "f32 MFMA 16x16x4  " : BF == 1 ? "bf16 MFMA 16x16x32" : "no MFMA           ",
           N, NT / 256, mx / iters / 8, mx / iters / 8 / (NT / 256), ms * 1e6 / iters / 8 / (NT / 256), mx / (ms * 1e6));
}
#define RUN3(N, BF) run<N, BF, 256>(out, in, cyc); run<N, BF, 512>(out, in, cyc); run<N, BF, 1024>(out, in, cyc)
int main()
{
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    float h[4096];
    for (int e = 0; e < 4096; ++e) h[e] = 0.001f * (e % 97) - 0.04f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    RUN3(0, 0); RUN3(1, 0); RUN3(2, 0); RUN3(4, 0); RUN3(6, 0); RUN3(8, 0);
    RUN3(0, 1); RUN3(2, 1); RUN3(4, 1); RUN3(8, 1);
    RUN3(4, 2); RUN3(8, 2);
    return 0;
}
