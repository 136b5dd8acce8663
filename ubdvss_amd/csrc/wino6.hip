// Dense dilated 3x3 convolution 24 -> 24, fp32 in / fp32 out, as Winograd F(2x2, 3x3) on the dilation sub-grids with the 16
// transform-domain GEMMs computed as EXACT THREE-WAY bf16 SPLIT PRODUCTS on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//
// Reference semantics: conv_bn(dilation_rate=d) (semantic_segmentation/net.py:298-304): 'same' zero padding d per side,
// cross-correlation, + bias + ReLU (and, for the last layer of an inference pass with one output channel, the 1x1 head of
// net.py:308-311 in the epilogue).  Same tiling, addressing and transforms as wino.hip (the fp32-MFMA form, kept for the data
// gradient and as the reference side of the bit-level tests); what changes is the arithmetic of the products.
//
// Why: the fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate -- 32 cycles of its SIMD per instruction, 192 per
// group of 16 tiles -- and the fp32 vector work of the transforms does not overlap with it (times add: DESIGN.md 5.1a), which
// left the layer at 36 us for 32 x 128 x 128 maps.  The bf16 MFMA of the same shape class takes 16 cycles for EIGHT times the
// K, holds the vector issue port for 8 of them only, and fp32 vector instructions run beside it.
// How, without giving up fp32 results: every fp32 value v is the EXACT sum of three bf16 values obtained by truncation,
//     v = v1 + v2 + v3,   v1 = hi16(v),  v2 = hi16(v - v1),  v3 = hi16(v - v1 - v2)       (8 + 8 + 8 = 24 significand bits),
// so a product of two fp32 values is the sum of nine bf16 x bf16 products (each exact in fp32).  The six with i + j <= 4 are
// kept (v1u1, v1u2, v2u1, v1u3, v3u1, v2u2); the three dropped ones are <= 2^-24 of the product -- the size of the rounding
// of an fp32 multiply.  K = 24 input channels fit ONE K = 32 step, so a (transform point, N tile) costs 6 bf16 MFMAs where the
// fp32 form cost 6 fp32 MFMAs: 16 x 2 x 6 x 16 = 3072 matrix-pipe cycles per group instead of 6144.  The six products are
// chained into one accumulator from the smallest term to the largest so that the big term is rounded once.
// The weights U = G g G^T are split once, at pack time; the transformed samples V = B^T D B are split on the vector ALU
// (v_and / v_sub / v_perm: 5.5 instructions per value) beside the MFMAs.  Scaling the input by a power of two scales every
// piece by it: f(2x) = 2 f(x) stays bit-exact.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// U fragments of one layer: [xi = a*4+b (16)][nt (2)][piece (3)][lane (64)] x 4 dwords (8 bf16 k-slots of the lane's k-group)
//   lane = (m = lane & 15 : output channel co = m + 16 nt, zero rows for co >= 24;  q = lane >> 4 : k-group)
//   k-slot e of group q holds input channel ci = e < 4 ? 4q + e : 16 + 2q + (e - 4) for e < 6, zero for e = 6, 7
//   (the same channel-to-lane map as the sample registers of the kernel: the input transform stays lane-local)
struct wino6_pack_args { size_t off_dil_k[UBD_NUM_DIL]; };

__global__ void pack_wino6_kernel(const float *__restrict__ params, unsigned *__restrict__ out, wino6_pack_args a)
{
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int total = UBD_NUM_DIL * 16 * 2 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, nt = (idx >> 6) & 1, xi = (idx >> 7) & 15, L = idx >> 11;
        const int q = lane >> 4, co = (lane & 15) + 16 * nt;
        const int ta = xi >> 2, tb = xi & 3;
        unsigned p[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (co < UBD_C) {
            const float *wk = params + a.off_dil_k[L];
            for (int e = 0; e < 6; ++e) {
                const int ci = e < 4 ? 4 * q + e : 16 + 2 * q + (e - 4);
                float v = 0.f;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        v += G[ta][ky] * G[tb][kx] * wk[((ky * 3 + kx) * UBD_C + ci) * UBD_C + co];
                // exact three-way truncation split
                const unsigned b1 = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(b1);
                const unsigned b2 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(b2);
                const unsigned b3 = __float_as_uint(r2) & 0xffff0000u;
                const int sh = (e & 1) ? 0 : 16;              // even slot: low half of the dword
                p[0][e >> 1] |= b1 >> sh;
                p[1][e >> 1] |= b2 >> sh;
                p[2][e >> 1] |= b3 >> sh;
            }
        }
        unsigned *o = out + ((size_t)(idx >> 6) * 3 * 64 + lane) * 4;      // idx >> 6 = (L * 16 + xi) * 2 + nt
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            *(u32x4 *)(o + pc * 64 * 4) = (u32x4){p[pc][0], p[pc][1], p[pc][2], p[pc][3]};
    }
}

void ubd_launch_pack_wino6(const ubd_handle *h, const float *params, unsigned *out, hipStream_t st)
{
    wino6_pack_args a;
    for (int k = 0; k < UBD_NUM_DIL; ++k) a.off_dil_k[k] = h->off_dil_k[k];
    hipLaunchKernelGGL(pack_wino6_kernel, dim3(48), dim3(256), 0, st, params, out, a);
}

// the instructions themselves, so that no pass re-packs or re-associates them (the file is also built with -fno-slp-vectorize)
__device__ __forceinline__ float sadd(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ssub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 vadd(f32x4 a, f32x4 b) { return (f32x4){sadd(a[0], b[0]), sadd(a[1], b[1]), sadd(a[2], b[2]), sadd(a[3], b[3])}; }
__device__ __forceinline__ f32x4 vsub(f32x4 a, f32x4 b) { return (f32x4){ssub(a[0], b[0]), ssub(a[1], b[1]), ssub(a[2], b[2]), ssub(a[3], b[3])}; }
__device__ __forceinline__ f32x2 vadd(f32x2 a, f32x2 b) { return (f32x2){sadd(a[0], b[0]), sadd(a[1], b[1])}; }
__device__ __forceinline__ f32x2 vsub(f32x2 a, f32x2 b) { return (f32x2){ssub(a[0], b[0]), ssub(a[1], b[1])}; }
// Sums that CONSUME matrix-pipe results stay plain C++ (component-wise, so that nothing packs them): hipcc's hazard recogniser does
// not look inside inline asm, so an asm v_add_f32 that reads an MFMA destination gets none of the wait states the read needs
// (observed: register 0 of the result correct, registers 1..3 stale).
__device__ __forceinline__ f32x4 madd(f32x4 a, f32x4 b) { return (f32x4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ f32x4 msub(f32x4 a, f32x4 b) { return (f32x4){a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}; }
// residual of the truncation to bf16: v - hi16(v), exact
__device__ __forceinline__ float resid(float v) { return ssub(v, __uint_as_float(__float_as_uint(v) & 0xffff0000u)); }
// {hi16(lo), hi16(hi)} as one dword of two bf16 k-slots
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef UBD_STAMPS   // diagnostic build only (tools/build_diag.sh)
static unsigned long long *g_wino6_stamps = nullptr;
extern "C" void ubd_debug_set_stamps_wino6(void *p) { g_wino6_stamps = (unsigned long long *)p; }
#define WSTAMP(k) do { if (stamps && lane == 0) stamps[((size_t)blockIdx.x * W6_WAVES + wave_in_block) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif

#define W6_WAVES 8
struct w6samples {
    f32x4 v4[4][4];
    f32x2 v2[4][4];
};

// EPI 0: y = relu(conv + bias);  EPI 2: logits = head(relu(conv + bias)) (one output channel; y is never written)
template <int EPI>
__global__ __launch_bounds__(64 * W6_WAVES) void dilconv_wino6_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                     const unsigned *__restrict__ ufrag,
                                                                     const float *__restrict__ bias, int n, int h, int w, int d,
                                                                     int log2d, unsigned in_bytes, const float *__restrict__ head
#ifdef UBD_STAMPS
                                                                     , unsigned long long *__restrict__ stamps
#endif
                                                                     )
{
    __shared__ __attribute__((aligned(16))) unsigned s_u[UBD_WINO6_FRAG_U32];       // 96 KiB
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    // products are issued as D = U^T . V^T (A operand = weights): D col = lane & 15 = tile, row = 4q + reg = channel, so
    // a lane ends up with four consecutive channels (4q.. of N-tile 0, 16+4q.. of N-tile 1 for q < 2) of its own tile
    f32x4 bA = *(const f32x4 *)(bias + 4 * q), bB = {0.f, 0.f, 0.f, 0.f};
    if (q < 2) bB = *(const f32x4 *)(bias + 16 + 4 * q);
    f32x4 hA = {0.f, 0.f, 0.f, 0.f}, hB = {0.f, 0.f, 0.f, 0.f};
    float hbias = 0.f;
    if constexpr (EPI == 2) {
        hA = *(const f32x4 *)(head + 4 * q);
        if (q < 2) hB = *(const f32x4 *)(head + 16 + 4 * q);
        hbias = head[UBD_C];
    }
    const int dm1 = d - 1;
    const int half_rows = ((h + 2 * d - 1) / (2 * d)) * d;       // rows y that pair with y + d
    const int half_cols = ((w + 2 * d - 1) / (2 * d)) * d;
    const int groups_x = (half_cols + 15) >> 4;
    const int total = n * half_rows * groups_x;
    // XCD-aware split (see dilconv_f32_kernel)
    const int xcd = blockIdx.x & 7;
    const int nblk_x = (gridDim.x + 7 - xcd) >> 3;
    const int chunk = (total + 7) >> 3;
    const int g_begin = xcd * chunk;
    const int g_end = (g_begin + chunk < total) ? g_begin + chunk : total;
    const int stride = nblk_x * W6_WAVES;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, (int)(EPI == 2 ? in_bytes / UBD_C : in_bytes), 0x00020000);
    const unsigned oob = in_bytes;

    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int g = g_begin + (int)(blockIdx.x >> 3) * W6_WAVES + wave_in_block;
    const int g_last = g_end - 1;

    // separable offsets as in wino.hip: wave-uniform row term + per-lane column term; an out-of-image term is 2^30, so the
    // sum of any invalid pair is out of range for the descriptor and the load returns the zero padding
    const unsigned BIG = 0x40000000u;
    unsigned cq4[4];
    const unsigned dq2 = 64u - 8u * q;
    auto tile_col = [&](int gg) {
        const int gx = (int)((unsigned)gg % (unsigned)groups_x);
        const int tcol = gx * 16 + i;
        return ((tcol >> log2d) << (log2d + 1)) + (tcol & dm1);
    };
    auto set_cols = [&](int gg) {
        const int xj = tile_col(gg);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ix = xj + (b - 1) * d;
            const bool ok = (unsigned)ix < (unsigned)w;
            const unsigned cb = (unsigned)ix * (unsigned)(UBD_C * 4);
            cq4[b] = ok ? cb + 16u * q : BIG;
        }
    };
    auto row_term = [&](int gg, int dy) {
        const int rs = (int)((unsigned)gg / (unsigned)groups_x);
        const int s = (int)((unsigned)rs % (unsigned)half_rows);
        const int img = (int)((unsigned)rs / (unsigned)half_rows);
        const int iy = ((s >> log2d) << (log2d + 1)) + (s & dm1) + dy;
        return (iy >= 0 && iy < h) ? (unsigned)((img * h + iy) * w) * (unsigned)(UBD_C * 4) : BIG;
    };
    auto load_row = [&](w6samples &D, int gg, int a) {
        const unsigned rb = row_term(gg, (a - 1) * d);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(rb + cq4[b]), 0, 0);
            u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(rb + cq4[b] + dq2), 0, 0);
            D.v4[a][b] = __builtin_bit_cast(f32x4, r4);
            D.v2[a][b] = __builtin_bit_cast(f32x2, r2);
        }
    };

    w6samples D;
    WSTAMP(0);
    set_cols(g < g_last ? g : g_last);
    load_row(D, g < g_last ? g : g_last, 0);
    load_row(D, g < g_last ? g : g_last, 2);
    {
        // U (96 KiB) to LDS by LDS-DMA: 96 pieces of 1 KiB, 12 per wave
        constexpr int PER_WAVE = UBD_WINO6_FRAG_U32 * 4 / 1024 / W6_WAVES;
        const unsigned lds_u = ubd_lds_addr(s_u);
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int piece = wave_in_block * PER_WAVE + k;
            ubd_glds16_sbase((const char *)ufrag + (size_t)piece * 1024, (unsigned)lane * 16u, lds_u + (unsigned)piece * 1024u);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WSTAMP(1);
    if (g >= g_end) return;
    int gcount = 0;
    const u32x4 *su4 = (const u32x4 *)s_u + lane;
    for (;;) {
        f32x4 Y[2][2][2];      // [output row rr][output col c][nt]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            // T = (B^T D)[a], V[a][b] = (T B)[b]
            f32x4 V4[4];
            f32x2 V2[4];
            {
                f32x4 T4[4];
                f32x2 T2[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (a == 0) { T4[b] = vsub(D.v4[0][b], D.v4[2][b]); T2[b] = vsub(D.v2[0][b], D.v2[2][b]); }
                    else if (a == 1) { T4[b] = vadd(D.v4[1][b], D.v4[2][b]); T2[b] = vadd(D.v2[1][b], D.v2[2][b]); }
                    else if (a == 2) { T4[b] = vsub(D.v4[2][b], D.v4[1][b]); T2[b] = vsub(D.v2[2][b], D.v2[1][b]); }
                    else { T4[b] = vsub(D.v4[1][b], D.v4[3][b]); T2[b] = vsub(D.v2[1][b], D.v2[3][b]); }
                }
                V4[0] = vsub(T4[0], T4[2]); V4[1] = vadd(T4[1], T4[2]); V4[2] = vsub(T4[2], T4[1]); V4[3] = vsub(T4[1], T4[3]);
                V2[0] = vsub(T2[0], T2[2]); V2[1] = vadd(T2[1], T2[2]); V2[2] = vsub(T2[2], T2[1]); V2[3] = vsub(T2[1], T2[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (a == 0) load_row(D, g, 1);                   // sample row 0 is dead
            if (a == 2) load_row(D, g, 3);                   // sample row 2 is dead
            if (a == 3) {
                const int gn = g + stride;
                set_cols(gn < g_last ? gn : g_last);
                load_row(D, gn < g_last ? gn : g_last, 0);
                load_row(D, gn < g_last ? gn : g_last, 2);
            }
            f32x4 Z0[2], Z1[2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = a * 4 + b;
                // this point's weights: 2 N tiles x 3 pieces, requested before the split so that their latency hides under it
                u32x4 U[2][3];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) U[nt][pc] = su4[((xi * 2 + nt) * 3 + pc) * 64];
                // three-way split of the lane's six values of V[a][b]
                const float v0 = V4[b][0], v1 = V4[b][1], v2 = V4[b][2], v3 = V4[b][3], v4 = V2[b][0], v5 = V2[b][1];
                const float r0 = resid(v0), r1 = resid(v1), r2 = resid(v2), r3 = resid(v3), r4 = resid(v4), r5 = resid(v5);
                const float s0 = resid(r0), s1 = resid(r1), s2 = resid(r2), s3 = resid(r3), s4 = resid(r4), s5 = resid(r5);
                const u32x4 P1 = {pack_hi(v0, v1), pack_hi(v2, v3), pack_hi(v4, v5), 0u};
                const u32x4 P2 = {pack_hi(r0, r1), pack_hi(r2, r3), pack_hi(r4, r5), 0u};
                const u32x4 P3 = {pack_hi(s0, s1), pack_hi(s2, s3), pack_hi(s4, s5), 0u};
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 m = {0.f, 0.f, 0.f, 0.f};
                    m = mfma16(U[nt][2], P1, m);             // smallest terms first: the large one is rounded once
                    m = mfma16(U[nt][0], P3, m);
                    m = mfma16(U[nt][1], P2, m);
                    m = mfma16(U[nt][1], P1, m);
                    m = mfma16(U[nt][0], P2, m);
                    m = mfma16(U[nt][0], P1, m);
                    // output transform along b
                    if (b == 0) Z0[nt] = m;
                    else if (b == 1) { Z0[nt] = madd(Z0[nt], m); Z1[nt] = m; }
                    else if (b == 2) { Z0[nt] = madd(Z0[nt], m); Z1[nt] = msub(Z1[nt], m); }
                    else Z1[nt] = msub(Z1[nt], m);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // accumulate along a
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (a == 0) { Y[0][0][nt] = Z0[nt]; Y[0][1][nt] = Z1[nt]; }
                else if (a == 1) { Y[0][0][nt] = madd(Y[0][0][nt], Z0[nt]); Y[0][1][nt] = madd(Y[0][1][nt], Z1[nt]); Y[1][0][nt] = Z0[nt]; Y[1][1][nt] = Z1[nt]; }
                else if (a == 2) { Y[0][0][nt] = madd(Y[0][0][nt], Z0[nt]); Y[0][1][nt] = madd(Y[0][1][nt], Z1[nt]); Y[1][0][nt] = msub(Y[1][0][nt], Z0[nt]); Y[1][1][nt] = msub(Y[1][1][nt], Z1[nt]); }
                else { Y[1][0][nt] = msub(Y[1][0][nt], Z0[nt]); Y[1][1][nt] = msub(Y[1][1][nt], Z1[nt]); }
            }
        }

        // ---- epilogue: lane = (tile i of the group, channel quarter q); registers = 4 consecutive channels
        {
            const int xo0 = tile_col(g);
            const int rs_e = (int)((unsigned)g / (unsigned)groups_x);
            const int s_e = (int)((unsigned)rs_e % (unsigned)half_rows);
            const int img = (int)((unsigned)rs_e / (unsigned)half_rows);
            const int y0 = ((s_e >> log2d) << (log2d + 1)) + (s_e & dm1);
            unsigned st0[2], st1[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int xo = xo0 + c * d;
                const unsigned cb = (unsigned)xo * (unsigned)(UBD_C * 4) + 16u * q;
                st0[c] = xo < w ? cb : BIG;
                st1[c] = (xo < w && q < 2) ? cb + 64u : BIG;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int yo = y0 + rr * d;
                const unsigned rb = row_term(g, rr * d);     // BIG below the image
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int xo = xo0 + c * d;
                    const bool ok = yo < h && xo < w;
                    f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] = fmaxf(Y[rr][c][0][r] + bA[r], 0.f); v1[r] = fmaxf(Y[rr][c][1][r] + bB[r], 0.f); }
                    if constexpr (EPI == 2) {
                        float part = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v0[r], hA[r], part);
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v1[r], hB[r], part);      // hB = 0 for q >= 2
                        part += __shfl_xor(part, 16, 64);
                        part += __shfl_xor(part, 32, 64);
                        const unsigned pix = (unsigned)(img * h + yo) * (unsigned)w + (unsigned)xo;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part + hbias), yrsrc, (int)((ok && q == 0) ? pix * 4u : oob), 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), yrsrc, (int)(rb + st0[c]), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), yrsrc, (int)(rb + st1[c]), 0, 0);
                    }
                }
            }
        }
        ++gcount;
        if (gcount <= 5) WSTAMP(1 + gcount);
        g += stride;
        if (g >= g_end) break;
    }
    WSTAMP(7);
}

#ifdef UBD_STAMPS
#define WSTAMP_ARG , g_wino6_stamps
#else
#define WSTAMP_ARG
#endif
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// frag: this layer's UBD_WINO6_FRAG_U32 packed dwords; epi 0 / 2 as in ubd_launch_dilconv_wino
void ubd_launch_dilconv_wino6(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, int dilation,
                              const float *in, float *out, int n, int H4, int W4, hipStream_t st, const float *head)
{
    const unsigned in_bytes = (unsigned)((size_t)n * H4 * W4 * UBD_C * 4);
    const int d = dilation;
    const long half_rows = ((H4 + 2 * d - 1) / (2 * d)) * d, half_cols = ((W4 + 2 * d - 1) / (2 * d)) * d;
    const long groups = (long)n * half_rows * ((half_cols + 15) / 16);
    int grid = ubd_grid_for(groups, h->num_cus, W6_WAVES, 1);     // 96 KiB of LDS: one 8-wave block per CU
    grid = (grid + 7) / 8 * 8;
    if (epi == 2)
        hipLaunchKernelGGL((dilconv_wino6_kernel<2>), dim3(grid), dim3(64 * W6_WAVES), 0, st, in, out, frag, bias, n, H4, W4, d, ilog2(d), in_bytes, head WSTAMP_ARG);
    else
        hipLaunchKernelGGL((dilconv_wino6_kernel<0>), dim3(grid), dim3(64 * W6_WAVES), 0, st, in, out, frag, bias, n, H4, W4, d, ilog2(d), in_bytes, nullptr WSTAMP_ARG);
}
