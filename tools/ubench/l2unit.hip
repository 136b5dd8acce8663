// Micro-benchmark behind stem123w.h (round 5): what bounds one "L2 unit" of the stem (16 pixels x 24 channels: 9 x (b128 + b64) LDS
// tap reads, 54 depthwise FMAs, 12 fp32 MFMAs 16x16x4, 8 clamps, 2 LDS stores) on one SIMD at 1, 2 and 4 waves per SIMD?
// Modes: 0 plain FMAs only, 1 DPP FMAs only, 2 MFMAs only, 3 FMAs + MFMAs (operands in registers), 4 LDS reads only,
// 5 the whole unit with DPP FMAs, 6 the whole unit with plain FMAs (54 weight registers), 7 LDS reads + stores only.
// Prints shader cycles (s_memtime) per unit and SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2unit.hip -o /tmp/l2unit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void fmac_qp(float &acc, float w, float x, int j)
{
    switch (j) {
    case 0: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    case 1: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    case 2: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    default: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    }
}

constexpr int PW = 35, PIX = 24;                       // a1 patch: 11 x 35 pixels of 24 floats
constexpr int A1_FLOATS = 11 * PW * PIX, L2_FLOATS = 9 * 33 * 28;

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float *out, const float *in, unsigned long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float smem[A1_FLOATS + L2_FLOATS];
    float *a1p = smem, *l2 = smem + A1_FLOATS;
    for (int e = threadIdx.x; e < A1_FLOATS + L2_FLOATS; e += NT) smem[e] = in[e & 4095];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    float w54[9][6], w18[6][3], pwf[6][2];
    for (int s = 0; s < 6; ++s) {
        for (int t = 0; t < 9; ++t) w54[t][s] = in[(t * 6 + s) * 64 + lane];
        for (int g = 0; g < 3; ++g) w18[s][g] = in[1024 + (g * 6 + s) * 64 + lane];
        pwf[s][0] = in[2048 + s * 64 + lane]; pwf[s][1] = in[2048 + 512 + s * 64 + lane];
    }
    const f32x4 bA = *(const f32x4 *)(in + 4 * q), bB = *(const f32x4 *)(in + 16 + 4 * q);
    const int rb = (wid >> 1) % 8, half = wid & 1, pos = 1 + 16 * half + i;
    int ro4[3], ro2[3];
    for (int kx = 0; kx < 3; ++kx) {
        const int pcol = pos + kx;
        const int sl4 = q ^ ((pcol >> 2) & 1), sl2 = 4 + ((q >> 1) ^ ((pcol >> 3) & 1));
        ro4[kx] = (rb * PW + pcol) * PIX + 4 * sl4;
        ro2[kx] = (rb * PW + pcol) * PIX + 4 * sl2 + 2 * (q & 1);
    }
    const int l2w = (rb * 33 + pos) * 28 + 4 * q;
    __syncthreads();
    f32x4 v4[3][3];
    f32x2 v2[3][3];
    for (int r = 0; r < 3; ++r)
        for (int kx = 0; kx < 3; ++kx) { v4[r][kx] = *(const f32x4 *)(a1p + ro4[kx] + r * PW * PIX); v2[r][kx] = *(const f32x2 *)(a1p + ro2[kx] + r * PW * PIX); }
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { v4[r][kx] = *(const f32x4 *)(a1p + ro4[kx] + r * PW * PIX); v2[r][kx] = *(const f32x2 *)(a1p + ro2[kx] + r * PW * PIX); }
        }
        float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == 0 || MODE == 3 || MODE == 6) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int t = r * 3 + kx;
                    dv[0] = fmaf(v4[r][kx][0], w54[t][0], dv[0]); dv[1] = fmaf(v4[r][kx][1], w54[t][1], dv[1]);
                    dv[2] = fmaf(v4[r][kx][2], w54[t][2], dv[2]); dv[3] = fmaf(v4[r][kx][3], w54[t][3], dv[3]);
                    dv[4] = fmaf(v2[r][kx][0], w54[t][4], dv[4]); dv[5] = fmaf(v2[r][kx][1], w54[t][5], dv[5]);
                }
        }
        if constexpr (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int t = r * 3 + kx, g = t >> 2, j = t & 3;
                    fmac_qp(dv[0], w18[0][g], v4[r][kx][0], j); fmac_qp(dv[1], w18[1][g], v4[r][kx][1], j);
                    fmac_qp(dv[2], w18[2][g], v4[r][kx][2], j); fmac_qp(dv[3], w18[3][g], v4[r][kx][3], j);
                    fmac_qp(dv[4], w18[4][g], v2[r][kx][0], j); fmac_qp(dv[5], w18[5][g], v2[r][kx][1], j);
                }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 6; ++s) dv[s] = v4[0][0][s & 3];
        }
        if constexpr (MODE == 4 || MODE == 7) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) asm volatile("" :: "v"(v4[r][kx]), "v"(v2[r][kx]));
        }
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int s = 0; s < 6; ++s) sink += dv[s];
            asm volatile("" : "+v"(v4[0][0]), "+v"(v4[1][1]), "+v"(v2[2][2]));
        }
        if constexpr (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
            f32x4 acc0 = bA, acc1 = bB;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][0], dv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][1], dv[s], acc1, 0, 0, 0);
            }
            const float cap = __builtin_inff();
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] = __builtin_amdgcn_fmed3f(acc0[r], 0.f, cap); acc1[r] = __builtin_amdgcn_fmed3f(acc1[r], 0.f, cap); }
            if constexpr (MODE == 5 || MODE == 6) {
                *(f32x4 *)(l2 + l2w) = acc0;
                if (q < 2) *(f32x4 *)(l2 + l2w + 16) = acc1;
            } else {
                sink += acc0[0] + acc1[1];
                asm volatile("" : "+v"(v4[0][0]), "+v"(v4[1][1]), "+v"(v2[2][2]));
            }
        }
        if constexpr (MODE == 7) {
            *(f32x4 *)(l2 + l2w) = v4[0][0];
            if (q < 2) *(f32x4 *)(l2 + l2w + 16) = v4[1][1];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + wid] = t1 - t0;
    out[blockIdx.x * NT + threadIdx.x] = sink + l2[threadIdx.x];
}

template <int MODE, int NT> void run(float *out, float *in, unsigned long long *cyc, int iters, const char *name)
{
    k<MODE, NT><<<256, NT>>>(out, in, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, NT><<<256, NT>>>(out, in, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    const int nw = NT / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) mx += (double)h[b * 16 + w];
    mx /= 256.0 * nw;
    const int wps = NT / 256;
    // every wave does `iters` units; a SIMD hosts wps waves: cycles per unit and SIMD = wave cycles / (iters * wps)
    printf("mode %d %-34s %d waves/SIMD: %8.1f cycles per unit and wave, %7.1f per unit and SIMD  (%.3f ms)\n", MODE, name, wps, mx / iters, mx / iters / wps, ms);
}

int main()
{
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    float h[4096];
    srand(1);
    for (int e = 0; e < 4096; ++e) h[e] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 4000;
#define RUNALL(M, NAME) run<M, 256>(out, in, cyc, iters, NAME); run<M, 512>(out, in, cyc, iters, NAME); run<M, 1024>(out, in, cyc, iters, NAME)
    RUNALL(0, "54 plain FMAs");
    RUNALL(1, "54 DPP FMAs");
    RUNALL(2, "12 MFMAs + 8 clamps");
    run<3, 256>(out, in, cyc, iters, "54 FMAs + 12 MFMAs (regs)"); run<3, 512>(out, in, cyc, iters, "54 FMAs + 12 MFMAs (regs)");
    RUNALL(4, "9 x (b128 + b64) LDS reads");
    RUNALL(7, "LDS reads + 2 LDS stores");
    RUNALL(5, "whole unit, DPP FMAs");
    run<6, 256>(out, in, cyc, iters, "whole unit, plain FMAs"); run<6, 512>(out, in, cyc, iters, "whole unit, plain FMAs");
    return 0;
}
