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

// All arithmetic below is plain C++ on vector types (no inline asm):
//   * hipcc's scheduler and hazard recogniser see every instruction -- an asm v_add_f32 that reads an MFMA destination gets none of
//     the wait states the read needs (observed with the first version of this kernel: register 0 of the result correct, 1..3 stale);
//   * the two-component adds compile to v_pk_add_f32.  Measured on gfx950 (tools/ubench/valu_ops.hip): a wave issues a simple vector
//     instruction every ~2.9 ns when it runs alone and the SIMD sustains one per ~1.1 ns from 2+ waves, while v_pk_add_f32 takes
//     ~2.25 ns per instruction at ANY occupancy.  At the two waves per SIMD this kernel's 250 registers allow, the waves are bound by
//     their own issue interval, so halving the number of add instructions halves their time (the fp32-MFMA kernel next door avoided the
//     packed form for the opposite reason: beside fp32 MFMAs it sits in the MFMA's shadow at one wave per SIMD).
//   Other per-instruction prices that shaped the code (same file): v_perm_b32 / v_cvt_pk_bf16_f32 / v_max_f32 / v_add3_u32 / any
//   instruction with an SGPR source ~1.95 ns, v_and_b32 with a literal 1.05 ns, s_nop and scalar ALU ~1.9 ns of the wave's time.
// residual of the truncation to bf16: v - hi16(v), exact
__device__ __forceinline__ f32x4 resid(f32x4 v) { return v - __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, v) & 0xffff0000u); }
__device__ __forceinline__ f32x2 resid(f32x2 v) { return v - __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, v) & 0xffff0000u); }
// {hi16(lo), hi16(hi)} as one dword of two bf16 k-slots
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ u32x4 pack6(f32x4 a, f32x2 b) { return (u32x4){pack_hi(a[0], a[1]), pack_hi(a[2], a[3]), pack_hi(b[0], b[1]), 0u}; }
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

#ifndef W6_WAVES
#define W6_WAVES 8
#endif
struct w6samples {
    f32x4 v4[4][4];
    f32x2 v2[4][4];
};
// launch geometry (host): everything the group decode needs, so that the loop holds no integer division
struct w6geom {
    int n, h, w, d, log2d;
    unsigned in_bytes;
    unsigned groups_x, half_rows, total;
    unsigned magic_gx, magic_hr;      // floor(v / groups_x) = umulhi(v, magic_gx) for v < total (ubd_tile_decoder's rule)
    int wpb;                          // waves per block that take groups (1..8): small launches spread over more CUs (all 8 waves copy the weights)
};
// addresses of one group: wave-uniform row terms (scalar registers; they ride in the buffer instructions' soffset, which IS part of
// the hardware range check: tools/ubench/buf_soffset.hip) and per-lane column terms.  An invalid row or column is 2^30 (host: tensor
// bytes <= 2^30), so any sum with an invalid term is out of range: loads return the zero padding, stores are dropped.
struct w6addr {
    unsigned row[4];      // byte offset of pixel (sample row a, x = 0)
    unsigned lrow[2];     // EPI 2: byte offset of logit (output row rr, x = 0)
    unsigned c4[4];       // column b: this lane's 16-byte chunk (channels 4q ..)
    unsigned c2[4];       // column b: this lane's 8-byte chunk (channels 16 + 2q, 17 + 2q)
};

// EPI 0: y = relu(conv + bias);  EPI 2: logits = head(relu(conv + bias)) (one output channel; y is never written)
template <int EPI>
__global__ __launch_bounds__(64 * W6_WAVES) void dilconv_wino6_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                     const unsigned *__restrict__ ufrag,
                                                                     const float *__restrict__ bias, w6geom G, const float *__restrict__ head
#ifdef UBD_STAMPS
                                                                     , unsigned long long *__restrict__ stamps
#endif
                                                                     )
{
    __shared__ __attribute__((aligned(16))) unsigned s_u[UBD_WINO6_FRAG_U32];       // 96 KiB
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int h = G.h, w = G.w, d = G.d, log2d = G.log2d, dm1 = G.d - 1;
    // Bias and head weights stay in registers for the whole kernel (8 + 8): loaded in the epilogue they cost a vmcnt(0) wait behind the
    // NEXT group's sample rows, which are in flight by then.  (An LDS table read in the epilogue was tried and looked corrupted in a few
    // per cent of the groups; the cause was the store hazard described at the stores below -- a LATER value in a store's data register --
    // not the table.  Registers are the cheaper home anyway: no reads in the epilogue.)
    const f32x4 bA = *(const f32x4 *)(bias + 4 * q);
    const f32x4 bB = q < 2 ? *(const f32x4 *)(bias + 16 + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 hA = {0.f, 0.f, 0.f, 0.f}, hB = {0.f, 0.f, 0.f, 0.f};
    float hbias = 0.f;
    if constexpr (EPI == 2) {
        hA = *(const f32x4 *)(head + 4 * q);
        if (q < 2) hB = *(const f32x4 *)(head + 16 + 4 * q);
        hbias = head[UBD_C];            // wave-uniform: one scalar load
    }
    // XCD-aware split (see dilconv_f32_kernel)
    const int xcd = blockIdx.x & 7;
    const int nblk_x = (gridDim.x + 7 - xcd) >> 3;
    const int chunk = (int)((G.total + 7) >> 3);
    const int g_begin = xcd * chunk;
    const int g_end = (g_begin + chunk < (int)G.total) ? g_begin + chunk : (int)G.total;
    const int stride = nblk_x * G.wpb;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)G.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, (int)(EPI == 2 ? G.in_bytes / UBD_C : G.in_bytes), 0x00020000);

    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int g = g_begin + (int)(blockIdx.x >> 3) * G.wpb + wave_in_block;
    const int g_last = g_end - 1;
    const unsigned BIG = 0x40000000u;
    const unsigned lane4 = 16u * q, lane2 = 64u + 8u * q;

    auto set_addr = [&](w6addr &A, int gg) {
        const unsigned rs = G.groups_x == 1u ? (unsigned)gg : __umulhi((unsigned)gg, G.magic_gx);           // gg / groups_x
        const unsigned gx = (unsigned)gg - rs * G.groups_x;
        const unsigned img = G.half_rows == 1u ? rs : __umulhi(rs, G.magic_hr);                             // rs / half_rows
        const unsigned s = rs - img * G.half_rows;
        const int y0 = (int)(((s >> log2d) << (log2d + 1)) + (s & (unsigned)dm1));
        const unsigned img_row = img * (unsigned)h;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int iy = y0 + (a - 1) * d;
            const bool ok = iy >= 0 && iy < h;
            A.row[a] = ok ? (img_row + (unsigned)iy) * (unsigned)w * (unsigned)(UBD_C * 4) : BIG;
            if (EPI == 2 && (a == 1 || a == 2)) A.lrow[a - 1] = ok ? (img_row + (unsigned)iy) * (unsigned)w * 4u : BIG;
        }
        const int tcol = (int)gx * 16 + i;
        const int xj = ((tcol >> log2d) << (log2d + 1)) + (tcol & dm1);   // this lane's tile column (pixel x of output (., 0))
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ix = xj + (b - 1) * d;
            const bool ok = (unsigned)ix < (unsigned)w;
            const unsigned cb = (unsigned)ix * (unsigned)(UBD_C * 4);
            A.c4[b] = ok ? cb + lane4 : BIG;
            A.c2[b] = ok ? cb + lane2 : BIG;
        }
    };
    auto load_row = [&](w6samples &D, const w6addr &A, int a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            D.v4[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)A.c4[b], (int)A.row[a], 0));
            D.v2[a][b] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)A.c2[b], (int)A.row[a], 0));
        }
    };

    // Sample rows are fetched just in time so that at most three of the four are live.  The first samples are requested before the
    // block copies U into LDS: their latency hides behind the copy.
    w6samples D;
    w6addr A, An;
    WSTAMP(0);
    set_addr(A, g < g_last ? g : g_last);
    load_row(D, A, 0);
    load_row(D, A, 2);
    load_row(D, A, 1);
    // U (96 KiB) to LDS by LDS-DMA: 96 pieces of 1 KiB, 12 per wave, handed out in ROUNDS (round r, wave w -> piece 8 r + w), so that the
    // pieces of the first transform points are the first to land: pieces 0..23 (points 0..3, the first transform row) are rounds 0..2.
    // The block starts computing behind THOSE (barrier A); the other nine rounds land under the first row of the first group and are
    // awaited once, before point 4's weights are read (barrier B; waves without a group wait there too, then leave).  The copy used to
    // stand in front of every launch whole: ~3.5 us of a 33 us launch, x 6 launches per pass.
    // (asm form: outside hipcc's bookkeeping -- its own waits for the sample rows can only over-wait, the DMAs being younger.)
    constexpr int PER_WAVE = UBD_WINO6_FRAG_U32 * 4 / 1024 / W6_WAVES;          // 12
    constexpr int EARLY_ROUNDS = 4 * 6 / W6_WAVES;                                // 3: the 24 pieces of points 0..3
    static_assert(PER_WAVE - EARLY_ROUNDS == 9, "the counted wait below is written for nine late rounds");
    {
        const unsigned lds_u = ubd_lds_addr(s_u);
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int piece = k * W6_WAVES + wave_in_block;
            ubd_glds16_sbase((const char *)ufrag + (size_t)piece * 1024, (unsigned)lane * 16u, lds_u + (unsigned)piece * 1024u);
        }
    }
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");          // the sample rows (older) and this wave's rounds 0..2 have landed
    __builtin_amdgcn_s_barrier();                             // A: points 0..3 of the weights are in LDS
    WSTAMP(1);
    if (g >= g_end || wave_in_block >= G.wpb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // B
        return;
    }
    bool weights_pending = true;                              // wave-uniform: barrier B is still owed (first group only)
    [[maybe_unused]] int gcount = 0;
    const u32x4 *su4 = (const u32x4 *)s_u + lane;
    // Schedule of one transform point xi = 4a + b (16 per group, fully unrolled):
    //   1. request the point's weights (6 ds_read_b128: 2 N tiles x 3 pieces);
    //   2. split the lane's six values of V[a][b] into three bf16 pieces (12 v_and + 6 v_pk_add + 9 v_perm; hides the LDS latency);
    //   3. 12 MFMAs, the two N tiles' chains alternating (no back-to-back dependent pair), with the vector work that does not
    //      feed them in the gaps: the output transform of point xi - 1 (whose accumulators are complete by now) and, at b = 3,
    //      the input transform of row a + 1.
    for (;;) {
        f32x4 Y[2][2][2];      // [output row rr][output col c][nt]
        f32x4 Z0[2], Z1[2];    // output transform along b of the current row a
        f32x4 V4[4];           // V[a][b], channels 4q .. 4q+3
        f32x2 V2[4];           //          channels 16+2q, 17+2q
        // V[a] = (B^T D)[a] B from the sample rows that row a needs
        auto make_v = [&](int a) {
            f32x4 T4[4];
            f32x2 T2[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (a == 0) { T4[b] = D.v4[0][b] - D.v4[2][b]; T2[b] = D.v2[0][b] - D.v2[2][b]; }
                else if (a == 1) { T4[b] = D.v4[1][b] + D.v4[2][b]; T2[b] = D.v2[1][b] + D.v2[2][b]; }
                else if (a == 2) { T4[b] = D.v4[2][b] - D.v4[1][b]; T2[b] = D.v2[2][b] - D.v2[1][b]; }
                else { T4[b] = D.v4[1][b] - D.v4[3][b]; T2[b] = D.v2[1][b] - D.v2[3][b]; }
            }
            V4[0] = T4[0] - T4[2]; V4[1] = T4[1] + T4[2]; V4[2] = T4[2] - T4[1]; V4[3] = T4[1] - T4[3];
            V2[0] = T2[0] - T2[2]; V2[1] = T2[1] + T2[2]; V2[2] = T2[2] - T2[1]; V2[3] = T2[1] - T2[3];
        };
        // output transform of point (a, b) with accumulators m: along b into Z, at b == 3 along a into Y
        auto consume = [&](int a, int b, const f32x4 (&m)[2]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (b == 0) Z0[nt] = m[nt];
                else if (b == 1) { Z0[nt] += m[nt]; Z1[nt] = m[nt]; }
                else if (b == 2) { Z0[nt] += m[nt]; Z1[nt] -= m[nt]; }
                else {
                    Z1[nt] -= m[nt];
                    if (a == 0) { Y[0][0][nt] = Z0[nt]; Y[0][1][nt] = Z1[nt]; }
                    else if (a == 1) { Y[0][0][nt] += Z0[nt]; Y[0][1][nt] += Z1[nt]; Y[1][0][nt] = Z0[nt]; Y[1][1][nt] = Z1[nt]; }
                    else if (a == 2) { Y[0][0][nt] += Z0[nt]; Y[0][1][nt] += Z1[nt]; Y[1][0][nt] -= Z0[nt]; Y[1][1][nt] -= Z1[nt]; }
                    else { Y[1][0][nt] -= Z0[nt]; Y[1][1][nt] -= Z1[nt]; }
                }
            }
        };
        make_v(0);                                           // sample rows 0 and 2 (requested under the previous group, like row 1)
        // Software pipeline over the 16 points: the MFMA phase of point xi carries, two vector instructions behind each MFMA (the part
        // of an MFMA's 16 cycles in which the vector issue port is free), the truncation residuals of point xi + 1; the tail of the
        // phase requests the weights of xi + 1, packs its three operand quads, and transforms the accumulators of xi.
        u32x4 U[2][3];
        u32x4 P1, P2, P3;
        f32x4 r4, t4;
        f32x2 r2, t2;
        auto load_u = [&](int xi) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) U[nt][pc] = su4[((xi * 2 + nt) * 3 + pc) * 64];
        };
        load_u(0);
        r4 = resid(V4[0]); t4 = resid(r4); r2 = resid(V2[0]); t2 = resid(r2);
        P1 = pack6(V4[0], V2[0]); P2 = pack6(r4, r2); P3 = pack6(t4, t2);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int a = xi >> 2, b = xi & 3;
            __builtin_amdgcn_sched_barrier(0);
            f32x4 m[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            // smallest terms first: the large one is rounded once
            m[0] = mfma16(U[0][2], P1, m[0]); m[1] = mfma16(U[1][2], P1, m[1]);
            m[0] = mfma16(U[0][0], P3, m[0]); m[1] = mfma16(U[1][0], P3, m[1]);
            m[0] = mfma16(U[0][1], P2, m[0]); m[1] = mfma16(U[1][1], P2, m[1]);
            m[0] = mfma16(U[0][1], P1, m[0]); m[1] = mfma16(U[1][1], P1, m[1]);
            m[0] = mfma16(U[0][0], P2, m[0]); m[1] = mfma16(U[1][0], P2, m[1]);
            m[0] = mfma16(U[0][0], P1, m[0]); m[1] = mfma16(U[1][0], P1, m[1]);
            // in the MFMAs' shadows: residuals of the next point of this row (b < 3: V[b + 1] is there; the next row's V does not exist yet)
            if (b < 3) { r4 = resid(V4[b + 1]); t4 = resid(r4); r2 = resid(V2[b + 1]); t2 = resid(r2); }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two vector instructions in its shadow
            }
            __builtin_amdgcn_sched_barrier(0);
            // tail: weights of the next point (the MFMAs above have read theirs), its operands, this point's output transform
            if (xi == 3 && weights_pending) {                 // first group of the launch: the rest of the weight copy (rounds 3..11)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // B
                weights_pending = false;
            }
            if (xi < 15) load_u(xi + 1);
            if (b == 3 && a < 3) {
                make_v(a + 1);
                if (a == 1) load_row(D, A, 3);               // sample row 2 is dead (rows 1 and 3 make row 3)
                if (a == 2) {
                    // all sample rows are dead: rows 0 and 2 of the next group.  Unconditional (clamped) so that hipcc counts
                    // the outstanding loads exactly.
                    const int gn = g + stride;
                    set_addr(An, gn < g_last ? gn : g_last);
                    load_row(D, An, 0);
                    load_row(D, An, 2);
                }
                r4 = resid(V4[0]); t4 = resid(r4); r2 = resid(V2[0]); t2 = resid(r2);
            }
            if (xi < 15) { const int bn = (b + 1) & 3; P1 = pack6(V4[bn], V2[bn]); P2 = pack6(r4, r2); P3 = pack6(t4, t2); }
            consume(a, b, m);
#ifdef UBD_STAMPS
            if (gcount == 1 && (xi & 3) == 3) WSTAMP(2 + (xi >> 2));
#endif
        }
        // Row 1 of the NEXT group is requested here, BEFORE this group's stores: the vector-memory counter retires in issue order, so a
        // load issued after the stores is only known to have landed once the stores have been acknowledged -- the wait for row 1 (three
        // points into the next group) then ended with the stores' round trip to L2, ~1 us later than the load itself (the kernel ran
        // in 27.6 us without its stores and 21 us without its loads against 33 us, round 6).  Rows 0 and 2 went out at point 11.
        load_row(D, An, 1);

        // ---- epilogue: lane = (tile i of the group, channel quarter q); registers = 4 consecutive channels.
        // Output (rr, c) is the pixel of sample (row 1 + rr, column 1 + c): its address terms are already there.
        {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 v0 = Y[rr][c][0] + bA, v1 = Y[rr][c][1] + bB;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] = fmaxf(v0[r], 0.f); v1[r] = fmaxf(v1[r], 0.f); }
                    if constexpr (EPI == 2) {
                        float part = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v0[r], hA[r], part);
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v1[r], hB[r], part);      // hB = 0 for q >= 2
                        part += __shfl_xor(part, 16, 64);                                   // sum over the four channel quarters
                        part += __shfl_xor(part, 32, 64);
                        // logit offset = pixel * 4: the column term is (c4 - 16q) / 24 for lane quarter 0
                        const unsigned lcol = (q == 0 && A.c4[1 + c] != BIG) ? A.c4[1 + c] / (unsigned)UBD_C : BIG;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part + hbias), yrsrc, (int)(lcol + A.lrow[rr]), 0, 0);
                    } else {
                        const unsigned st1 = (q < 2 && A.c4[1 + c] != BIG) ? A.c4[1 + c] + 64u : BIG;
                        // Stores take the row term in the VECTOR offset: with it in soffset, hipcc's hazard recogniser assumes that a wide
                        // store followed at once by a vector write to its data registers is safe (true of older chips) and inserts no
                        // wait state; on gfx950 ~0.1 % of the pixels of a 32 x 128 x 128 launch then carried a LATER value (an address
                        // integer) in one dword, different ones run to run (tools/wino6_chk_bias.py, round 6).
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), yrsrc, (int)(A.c4[1 + c] + A.row[1 + rr]), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), yrsrc, (int)(st1 + A.row[1 + rr]), 0, 0);
                    }
                }
        }
#ifdef UBD_STAMPS
        if (gcount == 1) WSTAMP(6);
        if (gcount == 0) WSTAMP(1);                          // re-stamped: start of the second group
#endif
        ++gcount;
        g += stride;
        if (g >= g_end) break;
        A = An;
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
    const int d = dilation;
    const long half_rows = ((H4 + 2 * d - 1) / (2 * d)) * d, half_cols = ((W4 + 2 * d - 1) / (2 * d)) * d;
    const long groups_x = (half_cols + 15) / 16;
    const long groups = (long)n * half_rows * groups_x;
    w6geom G;
    G.n = n; G.h = H4; G.w = W4; G.d = d; G.log2d = ilog2(d);
    G.in_bytes = (unsigned)((size_t)n * H4 * W4 * UBD_C * 4);
    G.groups_x = (unsigned)groups_x; G.half_rows = (unsigned)half_rows; G.total = (unsigned)groups;
    // floor(v / D) = (v * m) >> 32 with m = floor((2^32 - 1) / D) + 1, exact while v * D < 2^32: v < groups <= 2^30 / 96 / 4 pixels
    // ... / 16 and D <= 2^12 for every supported shape (tensor bytes <= 2^30); D = 1 needs the identity
    G.magic_gx = groups_x == 1 ? 0u : (unsigned)(0xFFFFFFFFul / (unsigned long)groups_x + 1ul);
    G.magic_hr = half_rows == 1 ? 0u : (unsigned)(0xFFFFFFFFul / (unsigned long)half_rows + 1ul);
    // 96 KiB of LDS: one 8-wave block per CU.  A launch with fewer than eight groups per CU (one 512 x 512 image: 256 groups) lets only
    // `wpb` waves of a block take groups, so that the groups spread over all CUs instead of filling 32 of them (batch-1 latency,
    // predict.py:73-78: 10.2 -> 7 us per layer); the other waves still copy their share of the weights and leave.
    int wpb = (int)((groups + h->num_cus - 1) / h->num_cus);
    if (wpb < 1) wpb = 1;
    if (wpb > W6_WAVES) wpb = W6_WAVES;
    G.wpb = wpb;
    int grid = ubd_grid_for(groups, h->num_cus, wpb, 1);
    grid = (grid + 7) / 8 * 8;
    if (epi == 2)
        hipLaunchKernelGGL((dilconv_wino6_kernel<2>), dim3(grid), dim3(64 * W6_WAVES), 0, st, in, out, frag, bias, G, head WSTAMP_ARG);
    else
        hipLaunchKernelGGL((dilconv_wino6_kernel<0>), dim3(grid), dim3(64 * W6_WAVES), 0, st, in, out, frag, bias, G, nullptr WSTAMP_ARG);
}
