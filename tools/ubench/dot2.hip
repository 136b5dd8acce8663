// Micro-benchmark: issue rate of v_dot2c_f32_bf16 vs v_fma_f32 vs (unpack + v_fma) on gfx950, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, const unsigned *in, int iters)
{
    unsigned a = in[threadIdx.x], b = in[threadIdx.x + 64];
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) acc[u] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), acc[u], false);
            else if (MODE == 1) acc[u] = fmaf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), acc[u]);
            else {   // unpack lo/hi of a, multiply with two fp32 weights
                const float lo = __builtin_bit_cast(float, a << 16), hi = __builtin_bit_cast(float, a & 0xFFFF0000u);
                acc[u] = fmaf(lo, __builtin_bit_cast(float, b), acc[u]);
                acc[(u + 1) & 7] = fmaf(hi, __builtin_bit_cast(float, b), acc[(u + 1) & 7]);
            }
            a += 0x10001u;   // keep the compiler from hoisting
        }
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int waves_per_simd, int iters, float *out, unsigned *in)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;   // 4 SIMDs x waves
    k<MODE><<<256, threads>>>(out, in, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE><<<256, threads>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}
int main()
{
    float *out; unsigned *in;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 4096); hipMemset(in, 1, 4096);
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {
        const float t0 = run<0>(w, iters, out, in), t1 = run<1>(w, iters, out, in), t2 = run<2>(w, iters, out, in);
        // per wave: iters*8 instructions (mode 2: 8 * (2 unpack + 2 fma + 1 add))
        const double cyc = 2.4e6;   // cycles per ms at 2.4 GHz
        printf("%d waves/SIMD: dot2c %.2f ms (%.1f cyc/instr/SIMD)  fma %.2f ms (%.1f)  unpack+2fma(+add) %.2f ms (%.1f cyc per 5 instr)\n", w,
               t0, t0 * cyc / (iters * 8.0 * w), t1, t1 * cyc / (iters * 8.0 * w), t2, t2 * cyc / (iters * 8.0 * w));
    }
    return 0;
}
