// Dense dilated 3x3 convolution 24 -> 24 as Winograd F(2x2, 3x3) on the dilation sub-grids, fp32 MFMA.
//
// Reference semantics: conv_bn(dilation_rate=d) (semantic_segmentation/net.py:298-304): 'same' zero
// padding d per side, cross-correlation, + bias + ReLU.  A convolution with dilation d is an ordinary
// 3x3 convolution on each of the d x d sub-grids {(y, x) : y = ry mod d, x = rx mod d}, so the minimal
// filtering algorithm F(2x2,3x3) (Lavin & Gray) applies with every "neighbour" d pixels away:
//     Y = A^T [ (G g G^T) .* (B^T D B) ] A,   D = 4x4 input samples spaced d, Y = 2x2 outputs spaced d,
// 16 multiplies per output pair-of-pairs instead of 36  =>  2.25x fewer MACs than the direct form.
// Mapping on gfx950 (64-wide waves, v_mfma_f32_16x16x4_f32):
//   * a wave owns a GROUP of 16 tiles = 16 consecutive "left-half" columns x one "top-half" row
//     (columns x and x+d, rows y and y+d form one tile), i.e. 64 output pixels;
//   * input transform B^T D B is lane-local: lane (i = lane&15 : tile, q = lane>>4 : channel quarter) holds
//     the same 4+2 channel registers per sample as the direct kernel, loaded with buffer_load_dwordx4/x2
//     (hardware zero fill = padding);
//   * the 16 transform-domain products are 16 small GEMMs [16 tiles x 24 ci] x [24 ci x 24 co] on the
//     MFMA pipe (6 k-steps x 2 N-tiles each; N = 24 padded to 32), B operands (the pre-transformed
//     weights U = G g G^T, packed per lane) streamed from LDS with ds_read_b64;
//   * the products are issued with the weights as the A operand, so the MFMA result layout is lane = tile,
//     registers = 4 consecutive output channels: the output transform A^T M A is lane-local too and every output
//     pixel leaves as 16-byte stores; rows of the transform domain are processed one at a time so that only
//     4 of the 16 products are live (<= 256 VGPRs, two waves per SIMD);
//   * epilogue: bias + ReLU (forward) or ReLU mask of the layer below (data gradient, run with the
//     spatially flipped / channel-transposed kernel).
// fp32 Winograd F(2,3) is not bit-identical to a fused-multiply-add chain (observed |err| ~1e-7 relative);
// parity tests bound it well inside the 1e-3 logit tolerance.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// U fragments of one layer: [xi = a*4+b (16)][j (6)][lane (64)][nt (2)]
//   ci = j < 4 ? 4*q + j : 16 + 2*q + (j-4), co = (lane&15) + 16*nt (zero for co >= 24)
//   U_xi[ci][co] = sum_{ky,kx} G[a][ky] G[b][kx] g[ky][kx][ci][co]
// transpose == 0: g = W (forward);  transpose == 1: g[ky][kx][ci][co] = W[2-ky][2-kx][co][ci] (dgrad)
struct wino_pack_args {
    size_t off_dil_k[UBD_NUM_DIL];
    int transpose;
    int round_dtype;      // UBD_F32: exact; UBD_BF16 / UBD_F16: kernel values rounded to that type first (the copy the
                          // 16-bit forward pass multiplies with), so that the data gradient matches the forward pass
};

__global__ void pack_wino_kernel(const float *__restrict__ params, float *__restrict__ out, wino_pack_args a)
{
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int total = UBD_NUM_DIL * UBD_WINO_FRAG_FLOATS;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int L = idx / UBD_WINO_FRAG_FLOATS;
        int r = idx % UBD_WINO_FRAG_FLOATS;
        const int nt = r & 1, lane = (r >> 1) & 63, xj = r >> 7, j = xj % 6, xi = xj / 6;
        const int q = lane >> 4, co = (lane & 15) + 16 * nt;
        const int ci = j < 4 ? 4 * q + j : 16 + 2 * q + (j - 4);
        const int ta = xi >> 2, tb = xi & 3;
        float v = 0.f;
        if (co < UBD_C) {
            const float *wk = params + a.off_dil_k[L];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    float g = a.transpose ? wk[(((2 - ky) * 3 + (2 - kx)) * UBD_C + co) * UBD_C + ci]
                                          : wk[((ky * 3 + kx) * UBD_C + ci) * UBD_C + co];
                    if (a.round_dtype == UBD_BF16) g = (float)(__bf16)g;
                    else if (a.round_dtype == UBD_F16) g = (float)(_Float16)g;
                    v += G[ta][ky] * G[tb][kx] * g;
                }
        }
        out[idx] = v;
    }
}

void ubd_launch_pack_wino(const ubd_handle *h, const float *params, float *out, int transpose, hipStream_t st)
{
    wino_pack_args a;
    for (int k = 0; k < UBD_NUM_DIL; ++k) a.off_dil_k[k] = h->off_dil_k[k];
    a.transpose = transpose;
    a.round_dtype = transpose ? h->cfg.dtype : UBD_F32;
    hipLaunchKernelGGL(pack_wino_kernel, dim3(96), dim3(256), 0, st, params, out, a);
}

// Component-wise fp32 add that stays scalar: hipcc lowers vector fadd to v_pk_add_f32, which issues slower beside MFMAs
// than the two v_add_f32 it replaces (MI355X_MICROARCH: packed f32 VALU is an anti-lever next to the matrix pipe).
// Written as the instruction itself so that no pass can re-pack the components (the file is also built with
// -fno-slp-vectorize).
__device__ __forceinline__ float sadd(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 vadd(f32x4 a, f32x4 b) { return (f32x4){sadd(a[0], b[0]), sadd(a[1], b[1]), sadd(a[2], b[2]), sadd(a[3], b[3])}; }
__device__ __forceinline__ f32x2 vadd(f32x2 a, f32x2 b) { return (f32x2){sadd(a[0], b[0]), sadd(a[1], b[1])}; }
__device__ __forceinline__ float ssub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 vsub(f32x2 a, f32x2 b) { return (f32x2){ssub(a[0], b[0]), ssub(a[1], b[1])}; }

#ifdef UBD_STAMPS   // diagnostic build only (tools/build_diag.sh)
static unsigned long long *g_wino_stamps = nullptr;
extern "C" void ubd_debug_set_stamps_wino(void *p) { g_wino_stamps = (unsigned long long *)p; }
#define WSTAMP(k) do { if (stamps && lane == 0) stamps[((size_t)blockIdx.x * 4 + wave_in_block) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif

struct wsamples {
    f32x4 v4[4][4];
    f32x2 v2[4][4];
};

// TAUX: element type of the ReLU-mask source (EPI 1): fp32 or a 16-bit activation type
template <int EPI, typename TAUX>
__global__ __launch_bounds__(256, 3) void dilconv_wino_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              const float *__restrict__ ufrag,
                                                              const void *__restrict__ aux_, int n, int h, int w, int d,
                                                              int log2d, unsigned in_bytes, const float *__restrict__ head
#ifdef UBD_STAMPS
                                                              , unsigned long long *__restrict__ stamps
#endif
                                                              )
{
    const TAUX *__restrict__ aux = (const TAUX *)aux_;
    __shared__ __attribute__((aligned(16))) float s_u[UBD_WINO_FRAG_FLOATS];       // 48 KiB
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    // products are issued as D = U^T . V^T (A operand = weights): D col = lane & 15 = tile, row = 4q + reg = channel, so
    // a lane ends up with four consecutive channels (4q.. of N-tile 0, 16+4q.. of N-tile 1 for q < 2) of its own tile
    f32x4 bA = {0.f, 0.f, 0.f, 0.f}, bB = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != 1) {
        bA = *(const f32x4 *)((const float *)aux_ + 4 * q);
        if (q < 2) bB = *(const f32x4 *)((const float *)aux_ + 16 + 4 * q);
    }
    // EPI 2 (last hidden layer of an inference pass with one output channel): the 1x1 head (net.py:308-311) is applied
    // to the lane's channels in the epilogue and the activation itself is never written; head = 24 weights + bias
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
    const int stride = nblk_x * 4;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, (int)(EPI == 2 ? in_bytes / UBD_C : in_bytes), 0x00020000);   // output: same shape (EPI 2: one logit per pixel)
    const unsigned oob = in_bytes;

    // wave-uniform group index kept in SGPRs
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int g = g_begin + (int)(blockIdx.x >> 3) * 4 + wave_in_block;
    const int g_last = g_end - 1;

    // Offsets are separable: a wave-uniform row term (SGPR) plus a per-lane column term that already holds the lane's
    // channel offset.  A term outside the image is 2^30 (host: tensor bytes <= 2^30), so the sum of any invalid pair is
    // out of range for the buffer descriptor and the load returns the zero padding: one v_add per load, no compares or
    // selects in the sample loop.
    const unsigned BIG = 0x40000000u;
    unsigned cq4[4];                                          // column terms of the group whose rows are being fetched
    const unsigned dq2 = 64u - 8u * q;                        // b64 load (channels 16 + 2q, 17 + 2q) relative to the b128 load (4q ..)
    auto tile_col = [&](int gg) {
        const int gx = (int)((unsigned)gg % (unsigned)groups_x);
        const int tcol = gx * 16 + i;
        return ((tcol >> log2d) << (log2d + 1)) + (tcol & dm1);           // this lane's tile column (pixel x of output (., 0))
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
    auto row_term = [&](int gg, int dy) {                     // byte offset of pixel (row of the group + dy, x = 0); wave-uniform
        const int rs = (int)((unsigned)gg / (unsigned)groups_x);
        const int s = (int)((unsigned)rs % (unsigned)half_rows);
        const int img = (int)((unsigned)rs / (unsigned)half_rows);
        const int iy = ((s >> log2d) << (log2d + 1)) + (s & dm1) + dy;
        return (iy >= 0 && iy < h) ? (unsigned)((img * h + iy) * w) * (unsigned)(UBD_C * 4) : BIG;
    };
    // one row (a) of the 4 x 4 sample array of group gg: 4 samples x (4 + 2) channel registers
    auto load_row = [&](wsamples &D, int gg, int a) {
        const unsigned rb = row_term(gg, (a - 1) * d);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(rb + cq4[b]), 0, 0);
            u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(rb + cq4[b] + dq2), 0, 0);     // v_add3_u32
            D.v4[a][b] = __builtin_bit_cast(f32x4, r4);
            D.v2[a][b] = __builtin_bit_cast(f32x2, r2);
        }
    };

    // Sample rows are fetched just in time so that at most three of the four are live (72 instead of 96 VGPRs ->
    // three waves per SIMD): rows 0 and 2 arrive before the group starts (issued under the previous group's last 48
    // MFMAs), row 1 is fetched under the MFMAs of transform row 0, row 3 under those of transform row 2.
    // The first samples are requested before the block copies U into LDS: their latency hides behind the copy.
    wsamples D;
    WSTAMP(0);
    // U (48 KiB) goes to LDS by LDS-DMA, 12 pieces of 1 KiB per wave, issued as asm (common.h): no VGPR round trip and no
    // ds_write pass, and the fetch overlaps the first sample requests.  vmcnt(0) retires this wave's pieces whatever
    // hipcc does with the sample loads around it (they are needed next anyway); the raw barrier publishes everyone's.
    // First version: 12 x (global_load_dwordx4 + ds_write_b128) + __syncthreads() = 10-12 k cycles of a 69 k-cycle launch
    // (in-kernel stamps).
    set_cols(g < g_last ? g : g_last);
    load_row(D, g < g_last ? g : g_last, 0);
    load_row(D, g < g_last ? g : g_last, 2);
    {
        constexpr int PER_WAVE = UBD_WINO_FRAG_FLOATS * 4 / 1024 / 4;
        const unsigned lds_u = ubd_lds_addr(s_u);
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int piece = wave_in_block * PER_WAVE + k;       // scalar base + 32-bit lane offset: half the issue cost of a 64-bit address
            ubd_glds16_sbase((const char *)ufrag + (size_t)piece * 1024, (unsigned)lane * 16u, lds_u + (unsigned)piece * 1024u);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WSTAMP(1);
    if (g >= g_end) return;
    int gcount = 0;
    for (;;) {
        // ---- transform-domain rows a = 0..3
        f32x4 Y[2][2][2];      // [output row rr][output col c][nt]; first written at a == 0 (row 0) / a == 1 (row 1)
        f32x2 ucur[6], unext[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) ucur[j] = *(const f32x2 *)(s_u + ((0 * 6 + j) * 64 + lane) * 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            // T = (B^T D)[a]  (per column b, 4+2 channel registers), V[a][b] = (T B)[b]
            f32x4 V4[4];
            f32x2 V2[4];
            {
                f32x4 T4[4];
                f32x2 T2[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (a == 0) { T4[b] = D.v4[0][b] - D.v4[2][b]; T2[b] = vsub(D.v2[0][b], D.v2[2][b]); }
                    else if (a == 1) { T4[b] = vadd(D.v4[1][b], D.v4[2][b]); T2[b] = vadd(D.v2[1][b], D.v2[2][b]); }
                    else if (a == 2) { T4[b] = D.v4[2][b] - D.v4[1][b]; T2[b] = vsub(D.v2[2][b], D.v2[1][b]); }
                    else { T4[b] = D.v4[1][b] - D.v4[3][b]; T2[b] = vsub(D.v2[1][b], D.v2[3][b]); }
                }
                V4[0] = T4[0] - T4[2]; V4[1] = vadd(T4[1], T4[2]); V4[2] = T4[2] - T4[1]; V4[3] = T4[1] - T4[3];
                V2[0] = vsub(T2[0], T2[2]); V2[1] = vadd(T2[1], T2[2]); V2[2] = vsub(T2[2], T2[1]); V2[3] = vsub(T2[1], T2[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (a == 0) load_row(D, g, 1);                   // sample row 0 is dead
            if (a == 2) load_row(D, g, 3);                   // sample row 2 is dead
            if (a == 3) {
                // all sample rows are dead: rows 0 and 2 of the next group under the last 48 MFMAs + epilogue.
                // Unconditional (clamped) so that hipcc counts the outstanding loads exactly.
                const int gn = g + stride;
                set_cols(gn < g_last ? gn : g_last);
                load_row(D, gn < g_last ? gn : g_last, 0);
                load_row(D, gn < g_last ? gn : g_last, 2);
            }
            f32x4 M[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = a * 4 + b;
                if (xi < 15) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) unext[j] = *(const f32x2 *)(s_u + (((xi + 1) * 6 + j) * 64 + lane) * 2);
                }
                f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float av = j < 4 ? V4[b][j] : V2[b][j - 4];
                    m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ucur[j][0], av, m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ucur[j][1], av, m1, 0, 0, 0);
                }
                M[b][0] = m0; M[b][1] = m1;
#pragma unroll
                for (int j = 0; j < 6; ++j) ucur[j] = unext[j];
                __builtin_amdgcn_sched_barrier(0);
            }
            // output transform along b, then accumulate along a
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 z0 = vadd(vadd(M[0][nt], M[1][nt]), M[2][nt]);
                const f32x4 z1 = M[1][nt] - M[2][nt] - M[3][nt];
                if (a == 0) { Y[0][0][nt] = z0; Y[0][1][nt] = z1; }
                else if (a == 1) { Y[0][0][nt] = vadd(Y[0][0][nt], z0); Y[0][1][nt] = vadd(Y[0][1][nt], z1); Y[1][0][nt] = z0; Y[1][1][nt] = z1; }
                else if (a == 2) { Y[0][0][nt] = vadd(Y[0][0][nt], z0); Y[0][1][nt] = vadd(Y[0][1][nt], z1); Y[1][0][nt] -= z0; Y[1][1][nt] -= z1; }
                else { Y[1][0][nt] -= z0; Y[1][1][nt] -= z1; }
            }
        }

        // ---- epilogue: lane = (tile i of the group, channel quarter q); registers = 4 consecutive channels
        {
            const int xo0 = tile_col(g);
            const int rs_e = (int)((unsigned)g / (unsigned)groups_x);
            const int s_e = (int)((unsigned)rs_e % (unsigned)half_rows);
            const int img = (int)((unsigned)rs_e / (unsigned)half_rows);
            const int y0 = ((s_e >> log2d) << (log2d + 1)) + (s_e & dm1);
            unsigned st0[2], st1[2];                         // store column terms: channels 4q.. and 16 + 4q.. (q < 2)
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
                    const unsigned e = ((unsigned)(img * h + yo) * (unsigned)w + (unsigned)xo) * (unsigned)UBD_C + 4u * (unsigned)q;   // element index (EPI 1 mask loads)
                    const unsigned o0 = rb + st0[c];
                    const unsigned o1 = rb + st1[c];
                    f32x4 v0, v1;
                    if constexpr (EPI != 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = fmaxf(Y[rr][c][0][r] + bA[r], 0.f); v1[r] = fmaxf(Y[rr][c][1][r] + bB[r], 0.f); }
                    } else {
                        float mk0[4] = {0.f, 0.f, 0.f, 0.f}, mk1[4] = {0.f, 0.f, 0.f, 0.f};
                        if (ok) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) mk0[r] = (float)aux[e + r];
                            if (q < 2) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) mk1[r] = (float)aux[e + 16 + r];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = mk0[r] > 0.f ? Y[rr][c][0][r] : 0.f; v1[r] = mk1[r] > 0.f ? Y[rr][c][1][r] : 0.f; }
                    }
                    if constexpr (EPI == 2) {
                        float part = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v0[r], hA[r], part);
#pragma unroll
                        for (int r = 0; r < 4; ++r) part = fmaf(v1[r], hB[r], part);      // hB = 0 for q >= 2
                        part += __shfl_xor(part, 16, 64);                                   // sum over the four channel quarters
                        part += __shfl_xor(part, 32, 64);
                        const unsigned pix = (unsigned)(img * h + yo) * (unsigned)w + (unsigned)xo;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part + hbias), yrsrc, (int)((ok && q == 0) ? pix * 4u : oob), 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), yrsrc, (int)o0, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), yrsrc, (int)o1, 0, 0);
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
#define WSTAMP_ARG , g_wino_stamps
#else
#define WSTAMP_ARG
#endif
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// frag: this layer's UBD_WINO_FRAG_FLOATS packed floats; aux: bias (epi 0) or mask source (epi 1)
// aux_dtype: element type of `aux` for epi 1 (UBD_F32 / UBD_BF16 / UBD_F16); epi 0 ignores it (bias is fp32)
void ubd_launch_dilconv_wino(const ubd_handle *h, int epi, const float *frag, const void *aux, int aux_dtype, int dilation,
                             const float *in, float *out, int n, int H4, int W4, hipStream_t st, const float *head)
{
    const unsigned in_bytes = (unsigned)((size_t)n * H4 * W4 * UBD_C * 4);
    const int d = dilation;
    const long half_rows = ((H4 + 2 * d - 1) / (2 * d)) * d, half_cols = ((W4 + 2 * d - 1) / (2 * d)) * d;
    const long groups = (long)n * half_rows * ((half_cols + 15) / 16);
    int grid = ubd_grid_for(groups, h->num_cus, 4, 3);     // <= 168 VGPRs in every form: three waves per SIMD (the data-gradient form ran two until round 5)
    grid = (grid + 7) / 8 * 8;
    if (epi == 0)
        hipLaunchKernelGGL((dilconv_wino_kernel<0, float>), dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, d, ilog2(d), in_bytes, nullptr WSTAMP_ARG);
    else if (epi == 2)       // out = logits (n, H4, W4, 1); head = 24 weights followed by the bias
        hipLaunchKernelGGL((dilconv_wino_kernel<2, float>), dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, d, ilog2(d), in_bytes, head WSTAMP_ARG);
    else if (aux_dtype == UBD_F32)
        hipLaunchKernelGGL((dilconv_wino_kernel<1, float>), dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, d, ilog2(d), in_bytes, nullptr WSTAMP_ARG);
    else if (aux_dtype == UBD_BF16)
        hipLaunchKernelGGL((dilconv_wino_kernel<1, __bf16>), dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, d, ilog2(d), in_bytes, nullptr WSTAMP_ARG);
    else
        hipLaunchKernelGGL((dilconv_wino_kernel<1, _Float16>), dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, d, ilog2(d), in_bytes, nullptr WSTAMP_ARG);
}
