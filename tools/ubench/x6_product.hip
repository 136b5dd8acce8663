// Unit test (round 6) of the three-way bf16 split product on v_mfma_f32_16x16x32_bf16, outside the convolution kernel:
// one wave computes M[co][tile] = sum_ci U[ci][co] V[tile][ci] (24 x 24 x 16) with the lane layout of wino6.hip and compares with fp64.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/x6_product.hip -o /tmp/x6_product && /tmp/x6_product
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float ssub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float resid(float v) { return ssub(v, __uint_as_float(__float_as_uint(v) & 0xffff0000u)); }
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ void split6(const float *v, u32x4 &P1, u32x4 &P2, u32x4 &P3)
{
    float r[6], s[6];
    for (int e = 0; e < 6; ++e) { r[e] = resid(v[e]); s[e] = resid(r[e]); }
    P1 = (u32x4){pack_hi(v[0], v[1]), pack_hi(v[2], v[3]), pack_hi(v[4], v[5]), 0u};
    P2 = (u32x4){pack_hi(r[0], r[1]), pack_hi(r[2], r[3]), pack_hi(r[4], r[5]), 0u};
    P3 = (u32x4){pack_hi(s[0], s[1]), pack_hi(s[2], s[3]), pack_hi(s[4], s[5]), 0u};
}
// U: [ci 24][co 24], V: [tile 16][ci 24], M: [tile 16][co 24]; mode 0: six products, 1: v1u1 only, 2: pieces of lane dumped
__global__ void k(const float *U, const float *V, float *M, unsigned *dump, int mode)
{
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    float v[6];
    for (int e = 0; e < 6; ++e) v[e] = V[i * 24 + (e < 4 ? 4 * q + e : 16 + 2 * q + (e - 4))];
    u32x4 P1, P2, P3;
    split6(v, P1, P2, P3);
    if (dump) { for (int d = 0; d < 4; ++d) { dump[(0 * 64 + lane) * 4 + d] = P1[d]; dump[(1 * 64 + lane) * 4 + d] = P2[d]; dump[(2 * 64 + lane) * 4 + d] = P3[d]; } }
    for (int nt = 0; nt < 2; ++nt) {
        const int co = i + 16 * nt;
        float u[6] = {0, 0, 0, 0, 0, 0};
        if (co < 24) for (int e = 0; e < 6; ++e) u[e] = U[(e < 4 ? 4 * q + e : 16 + 2 * q + (e - 4)) * 24 + co];
        u32x4 U1, U2, U3;
        split6(u, U1, U2, U3);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};
        if (mode == 0) {
            m = mfma16(U3, P1, m); m = mfma16(U1, P3, m); m = mfma16(U2, P2, m);
            m = mfma16(U2, P1, m); m = mfma16(U1, P2, m); m = mfma16(U1, P1, m);
        } else {
            m = mfma16(U1, P1, m);
        }
        // lane (tile i, q): rows 4q + r of this N tile
        for (int r = 0; r < 4; ++r) { const int c = 16 * nt + 4 * q + r; if (c < 24) M[i * 24 + c] = m[r]; }
    }
}
int main()
{
    float hU[576], hV[384], hM[384];
    srand(1);
    for (int e = 0; e < 576; ++e) hU[e] = (float)rand() / RAND_MAX - 0.5f;
    for (int e = 0; e < 384; ++e) hV[e] = ((float)rand() / RAND_MAX - 0.3f) * 3.f;
    float *U, *V, *M; unsigned *dump;
    (void)hipMalloc(&U, sizeof(hU)); (void)hipMalloc(&V, sizeof(hV)); (void)hipMalloc(&M, sizeof(hM)); (void)hipMalloc(&dump, 3 * 64 * 4 * 4);
    (void)hipMemcpy(U, hU, sizeof(hU), hipMemcpyHostToDevice); (void)hipMemcpy(V, hV, sizeof(hV), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(U, V, M, dump, mode);
        (void)hipMemcpy(hM, M, sizeof(hM), hipMemcpyDeviceToHost);
        double worst = 0, big = 0;
        for (int t = 0; t < 16; ++t) for (int co = 0; co < 24; ++co) {
            double ref = 0, mag = 0;
            for (int ci = 0; ci < 24; ++ci) { ref += (double)hU[ci * 24 + co] * hV[t * 24 + ci]; mag += fabs((double)hU[ci * 24 + co] * hV[t * 24 + ci]); }
            worst = fmax(worst, fabs(hM[t * 24 + co] - ref) / mag); big = fmax(big, fabs(ref));
        }
        printf("mode %d (%s): worst |err| / sum|terms| = %.3e  (fp32 eps 6e-8, bf16 eps 3.9e-3)\n", mode, mode ? "v1u1 only" : "six products", worst);
    }
    unsigned hd[3 * 64 * 4];
    (void)hipMemcpy(hd, dump, sizeof(hd), hipMemcpyDeviceToHost);
    printf("lane 0: v = %a %a ; P1.d0 = %08x P2.d0 = %08x P3.d0 = %08x P1.d3 = %08x\n", hV[0], hV[1], hd[0], hd[256], hd[512], hd[3]);
    return 0;
}
