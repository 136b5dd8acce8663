// bf16-gradient kernels of the UBD_BF16 train step (included by backward.hip).
//
// BASELINE.json configs[2]/[3] ("bf16 train step"): activations AND the gradient tensors that flow between layers
// (G = dL/dZ) are stored in bf16, every product is accumulated in fp32, weight gradients / master weights / Adam
// state are fp32.  With both GEMM operands in bf16 the weight gradient of a dilated layer runs on
// v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate) and becomes an HBM/LDS-bound kernel.
// (UBD_F16 keeps fp32 gradient tensors: fp16 gradients of a mean loss over ~1e6 pixels underflow without loss scaling.)
//
//   head_dx16_kernel      G9 = (dlogits . hk^T) * (A9 > 0), stored as T
//   dil_wgrad16_kernel    dW[t][ci][co] = sum_p X[p+off_t][ci] G[p][co], db = sum_p G: M = 216 (+ ones row), N = 24,
//                         K = pixels; X and G tiles staged by LDS-DMA as plain [pixel][24 ch] rows, both operands read
//                         with ds_read_b64_tr_b16 (the K index = pixel is the row index of both LDS images)
//   (data gradient)       dilconv16_kernel<T, 1> in fwd16.hip
//   (separable layers)    sepb16_kernel in sepbwd16.h
#pragma once

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T>
__global__ __launch_bounds__(256) void head_dx16_kernel(const float *__restrict__ dlogits, const unsigned short *__restrict__ a9,
                                                        const float *__restrict__ hk, unsigned short *__restrict__ g, long npix, int k_out)
{
    __shared__ __attribute__((aligned(16))) float s_kT[(UBD_MAX_CLASSES + 1) * UBD_C];     // head kernel transposed: [k][c]
    for (int t = threadIdx.x; t < UBD_C * k_out; t += blockDim.x) { const int c = t / k_out, k = t - c * k_out; s_kT[k * UBD_C + c] = hk[t]; }
    __syncthreads();
    // one 16-byte chunk (8 channels of a pixel) per thread: loads and stores of a wave are contiguous; per output
    // channel k the thread's eight weights are two ds_read_b128
    const u32x4 *pa = (const u32x4 *)a9;
    u32x4 *pg = (u32x4 *)g;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < npix * 3; t += (long)gridDim.x * blockDim.x) {
        const long p = t / 3;
        const int c8 = (int)(t - p * 3);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < k_out; ++k) {
            const float dl = dlogits[p * k_out + k];
            const f32x4 w0 = *(const f32x4 *)&s_kT[k * UBD_C + c8 * 8], w1 = *(const f32x4 *)&s_kT[k * UBD_C + c8 * 8 + 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] = fmaf(dl, w0[e], acc[e]); acc[4 + e] = fmaf(dl, w1[e], acc[4 + e]); }
        }
        const u32x4 av = pa[t];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const short a_lo = (short)(av[e] & 0xFFFFu), a_hi = (short)(av[e] >> 16);
            const unsigned lo = a_lo > 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)acc[2 * e]) : 0u;
            const unsigned hi = a_hi > 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)acc[2 * e + 1]) : 0u;
            o[e] = lo | (hi << 16);
        }
        pg[t] = o;
    }
}

// k_out == 1 (detection only, the benchmark configuration): head data gradient AND head weight gradient in one pass over A9 --
// the two kernels each read the 48 bytes per pixel (19 + 16 us at 64 images).  Thread = one 16-byte chunk (8 channels) of a
// pixel; the launch has a multiple of 3 threads, so a thread keeps its channel group for the whole grid-stride loop and carries
// eight fp32 sums hk-gradient sums (+ the bias-gradient sum in group 0).  Block reduction in a fixed order (deterministic):
// partial row [24 | 1] per block, summed by reduce_partials_kernel.
template <typename T>
__global__ __launch_bounds__(256) void head_bwd1_16_kernel(const float *__restrict__ dlogits, const unsigned short *__restrict__ a9,
                                                           const float *__restrict__ hk, unsigned short *__restrict__ g,
                                                           float *__restrict__ partials, long npix)
{
    __shared__ float s_acc[256][9];
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = (int)(t0 % 3);                                  // constant per thread: gridDim.x * 256 is a multiple of 3
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = hk[c8 * 8 + e];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, accb = 0.f;
    const u32x4 *pa = (const u32x4 *)a9;
    u32x4 *pg = (u32x4 *)g;
    for (long t = t0; t < npix * 3; t += (long)gridDim.x * blockDim.x) {
        const long p = t / 3;
        const float dl = dlogits[p];
        const u32x4 av = pa[t];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const short a_lo = (short)(av[e] & 0xFFFFu), a_hi = (short)(av[e] >> 16);
            const unsigned lo = a_lo > 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)(dl * w[2 * e])) : 0u;
            const unsigned hi = a_hi > 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)(dl * w[2 * e + 1])) : 0u;
            o[e] = lo | (hi << 16);
            acc[2 * e] = fmaf((float)__builtin_bit_cast(T, (unsigned short)(av[e] & 0xFFFFu)), dl, acc[2 * e]);
            acc[2 * e + 1] = fmaf((float)__builtin_bit_cast(T, (unsigned short)(av[e] >> 16)), dl, acc[2 * e + 1]);
        }
        pg[t] = o;
        accb += dl;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_acc[threadIdx.x][e] = acc[e];
    s_acc[threadIdx.x][8] = accb;
    __syncthreads();
    if (threadIdx.x <= UBD_C) {
        const int c = threadIdx.x, grp = c < UBD_C ? c >> 3 : 0, e = c < UBD_C ? c & 7 : 8;
        // threads of this block with channel group grp: tid = first, first + 3, ...
        const int first = (int)(((grp - (long)blockIdx.x * blockDim.x) % 3 + 3) % 3);
        float v = 0.f;
        for (int tid = first; tid < 256; tid += 3) v += s_acc[tid][e];
        partials[(size_t)blockIdx.x * (UBD_C + 1) + c] = v;
    }
}

// ------------------------------------------------------------------------------------ dilated wgrad, bf16 MFMA
// Work items as in dil_wgrad_kernel: (image, phase (ry, rx) of the d x d sub-grid, 8 x 16 tile of that sub-grid).
// LDS per buffer: X tile with halo, 10 x 18 pixels x 48 B, then the G tile, 8 x 16 pixels x 48 B (LDS-DMA, chunk c at
// byte 16 c).  One k-block = 32 sub-pixels = 2 tile rows; wave w takes k-block w of every item.
// v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, kg = lane >> 4) holds A[i][8kg .. 8kg+7] and B[8kg .. 8kg+7][i];
// ds_read_b64_tr_b16 fills 4 of the 8 k-values: in the 16-lane group kg, lane 4q+p supplies the address of LDS row
// (= pixel) q, columns 4p .. 4p+3 and lane i receives column i of the 4 rows.  Columns of the A operand are the
// flat (tap, ci) index rho = 16 mt + i, so segment p of M-tile mt starts at rho0 = 16 mt + 4p: tap t = rho0 / 24 shifts the
// pixel, ci = rho0 % 24 the channel (24 = 6 x 4: a segment never straddles a tap).  rho = 216 is the all-ones column
// (bias gradient), 217.. are zero: those segments read 8-byte constants kept behind the tile buffers.
#define W16_TH(TW) ((TW) == 16 ? 16 : 8)     // tile rows: 16 x 16 tiles (two k-blocks per wave and item: the per-item barriers and DMA
                                              // bookkeeping cost as much as one k-block of MFMAs); 8 x 8 sub-grids (dilation 16 on 128-wide maps) keep 8 rows
// TW = 16 (default) or 8 (sub-grids at most 8 columns wide, i.e. dilation 16 on 128-wide maps: half of a 16-wide tile
// would be padding).  k-block = 32 sub-pixels = 2 rows x 16 or 4 rows x 8.
// PAIR (round 4; 16-wide tiles of the fused form only): sub-grids that are exactly 8 columns wide (dilation 16 on 128-wide maps) leave half
// of a 16-wide tile empty, and the 8-wide form pays the per-item costs (barriers, DMA bookkeeping, decode) for 64 pixels.  A PAIR item
// holds the sub-grids of phases (ry, rx) and (ry, rx + 1) side by side: tile columns 0-7 / 8-15, 8 rows; the staged images get ONE zero
// column between the two (patch columns 0 | 1-8 | 9 | 10-17 | 18): the right neighbour of the first sub-grid's last column and the left
// neighbour of the second one's first column are both outside their sub-grids -- 'same' padding.
template <int TW, bool DX = false, bool PAIR = false> struct w16_cfg {
    static_assert(!PAIR || (TW == 16 && DX), "paired sub-grids: the fused 16-wide form");
    static constexpr int TH = PAIR ? 8 : W16_TH(TW);
    static constexpr int XW = PAIR ? 19 : TW + 2;
    static constexpr int XPIX = (TH + 2) * XW;                   // 324 / 100 / 190
    // DX (the data gradient of the same layer is computed from the same tiles, see dil_wgrad16_kernel): the G tile carries its
    // one-pixel halo too and has the X tile's geometry
    static constexpr int GW = DX ? XW : TW;                      // G tile row pitch in pixels
    static constexpr int GPIX = DX ? XPIX : TH * TW;             // 324 / 256 / 64 / 190
    // DMA pieces (one wave-instruction = 64 chunks of 16 bytes): the X tile is padded to whole pieces so that a piece is
    // either X or G -- its tensor's descriptor then sits in scalar registers and the lanes add a 32-bit offset
    static constexpr int XR = (XPIX * 3 + 63) / 64, GR = (GPIX * 3 + 63) / 64;   // 16 + 12 (16 with halo) / 5 + 3 pieces
    static constexpr int GOFF = XR * 1024;                   // byte offset of the G tile in a buffer
    static constexpr int ROUNDS = (XR + GR + 3) / 4;         // pieces per wave
    static constexpr int BUF_BYTES = ROUNDS * 4 * 1024;      // 28 KiB (32 with halo) / 8 KiB
    static constexpr int KROWS = 32 / TW;                    // tile rows per k-block
};
#define W16_BUF_BYTES_MAX (w16_cfg<16>::BUF_BYTES)
// geometry of a launch, worked out once on the host (as d16s_geom, fwd16.hip): six 32-bit divisions less in every block's prologue
struct w16_geom { int sh, tiles_x, tiles_y, dxp, items; unsigned m_tx, m_ty, m_d, m_dx; };
template <int TW, bool PAIR> static w16_geom w16_geometry(int n, int h, int w, int d)
{
    constexpr int TH = PAIR ? 8 : W16_TH(TW);
    w16_geom g;
    const int sw = (w + d - 1) / d;
    g.sh = (h + d - 1) / d;
    g.tiles_y = PAIR ? 1 : (g.sh + TH - 1) / TH; g.tiles_x = PAIR ? 1 : (sw + TW - 1) / TW;
    g.dxp = PAIR ? d >> 1 : d;                                            // column phases per item row (PAIR: two per item; d is even)
    g.items = n * d * g.dxp * g.tiles_y * g.tiles_x;
    auto magic = [](unsigned dv) { return dv == 1u ? 0u : 0xFFFFFFFFu / dv + 1u; };       // ceil(2^32 / dv): floor(a / dv) = umulhi(a, m) while a * dv < 2^32
    g.m_tx = magic((unsigned)g.tiles_x); g.m_ty = magic((unsigned)g.tiles_y); g.m_d = magic((unsigned)d); g.m_dx = magic((unsigned)g.dxp);
    return g;
}

// 16 x 16 two-bit helpers of the fused data gradient (the separable-layer header has its own)
template <typename T> __device__ __forceinline__ unsigned wg_pack2(float lo, float hi)
{
    typedef T t2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, t2));
}
// g (two 16-bit values) with each half zeroed where the matching half of the saved activation m (never negative) is +0 / -0
__device__ __forceinline__ unsigned wg_relu_mask2(unsigned g, unsigned m)
{
    unsigned k, r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(m & 0x7FFF7FFFu), "v"(0x00010001u));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(g), "v"(k));
    return r;
}

// DX: the DATA gradient of the layer, G' = conv(G, flipped / transposed kernels) * (X > 0), is computed from the same staged tiles
// right after the weight-gradient MFMAs of an item (its mask is the X tile the weight gradient reads anyway, its input the G tile
// with a one-pixel halo): the separate data-gradient kernel read G and X once more (100 MB per layer and 64 images) and spent most
// of its instructions on addresses.  Same arithmetic in the same order as dilconv16_kernel<T, 1> (bit-identical results).
// MSPLIT: how the four waves share the 14 x 2 accumulator tiles of the weight gradient.  false (16-wide tiles): by k-blocks -- every wave
// holds all 28 tiles (112 registers) for its k-blocks and the block adds the four sets up at the end of the launch (252 registers: two
// blocks per CU).  true (the fused 8-wide form of the dilation-16 layer): by M tiles -- wave w owns mt = w, w + 4, w + 8, w + 12 for ALL
// k-blocks: 8 tiles = 32 registers, no reduction at the end, every wave reads the G operand (155 registers, 39 KB of LDS: three blocks
// per CU).  Round 4, same-box A/Bs of the bf16 train step: M-split for the 16-wide form too, two tile buffers / two blocks: +13 us; one
// tile buffer / three blocks (163 registers, 47 KB): +4 us -- the third wave does not pay for the extra transposed reads; but it makes
// the FUSED 8-wide form (dilation 16: it lost to the two separate kernels at 252 registers, 75 us against 38 + 33) the faster one: -11 us.
template <typename T, int TW, bool DX, bool MSPLIT = (TW == 8), bool PAIR = false>
#define W16_OCC(TW, DX) (((TW) == 8) ? 3 : 2)
__global__ __launch_bounds__(256, W16_OCC(TW, DX)) void dil_wgrad16_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ gz,
                                                             float *__restrict__ partials, int n, int h, int w, int d,
                                                             const u32x4 *__restrict__ wfrag_t, unsigned short *__restrict__ gout, const rp_job prev, const w16_geom geo
#ifdef UBD_STAMPS
                                                             , unsigned long long *__restrict__ stamps
#endif
                                                             )
{
#ifdef UBD_STAMPS   // diagnostic build only: s_memtime of lane 0 of every wave at the phase boundaries of its first 8 items
#define WGSTAMP(k) do { if (stamps && iter < 8 && (threadIdx.x & 63) == 0) stamps[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + iter) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WGSTAMP(k) do {} while (0)
#endif
    using C = w16_cfg<TW, DX, PAIR>;
    static_assert(!PAIR || !MSPLIT, "paired sub-grids: accumulators split by k-blocks");
    constexpr int TILES_BYTES = (!MSPLIT && 2 * C::BUF_BYTES < 28672) ? 28672 : 2 * C::BUF_BYTES;   // two tile buffers; the k-split block reduction needs 28 KiB
    constexpr int CONST_OFF = TILES_BYTES;                                    // [0,8): {1,0,0,0}   [8,32): zeros
    constexpr int WT_OFF = CONST_OFF + 64;                                    // DX: the layer's transposed fragments [7][2][64 lanes] x 16 B
    __shared__ __attribute__((aligned(16))) char smem[TILES_BYTES + 64 + (DX ? 7 * 2 * 64 * 16 : 0)];   // ONE LDS object (see fwd16.hip)
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
    if (threadIdx.x < 16)
        ((unsigned *)(smem + CONST_OFF))[threadIdx.x] = threadIdx.x == 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)1.0f) : 0u;   // 64 bytes: {1,0,0,0} then zeros

    // A operand: byte offset of segment p of M-tile mt relative to the X-tile pixel of the output position (slot sl <-> M tile mt(sl))
    constexpr int MTW = MSPLIT ? 4 : 14;                                 // M-tile slots per wave
    auto mt_of = [&](int sl) { return MSPLIT ? wid + 4 * sl : sl; };     // wave-uniform
    int aoff[MTW];
#pragma unroll
    for (int sl = 0; sl < MTW; ++sl) {
        const int rho0 = 16 * mt_of(sl) + 4 * p;
        const int t = rho0 / UBD_C, ci = rho0 - t * UBD_C;
        aoff[sl] = ((t / 3) * C::XW + (t % 3)) * (UBD_C * 2) + ci * 2;   // mt == 13, p >= 2: constants instead (below); mt >= 14: unused
    }
    // DX: this lane's K-slice of chunk c (as dilconv16_kernel): k0 = 32c + 8 grp -> tap (4c + grp) / 3, channel group (4c + grp) % 3
    int doff[DX ? 7 : 1];
    if constexpr (DX) {
        for (int t = threadIdx.x; t < 7 * 2 * 64; t += 256) ((u32x4 *)(smem + WT_OFF))[t] = wfrag_t[t];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const int g = 4 * c + grp, t = g / 3;
            doff[c] = t < 9 ? ((t / 3 - 1) * C::GW + (t % 3 - 1)) * (UBD_C * 2) + (g - 3 * t) * 16 : -(1 << 20);   // chunk 6, k-group 3 lies beyond K: zeros
        }
    }
    // pixel of this lane inside a k-block (2 tile rows x 16 columns or 4 x 8): k = 8 grp + 4 j + qq, j = 0, 1
    const int krow = TW == 16 ? grp >> 1 : grp, kcol = TW == 16 ? 8 * (grp & 1) + qq : qq;
    const int kcolp = kcol + (PAIR ? (kcol >> 3) : 0);             // column inside the staged images (PAIR: past the zero column)

    // DMA chunk -> (tile row, tile column, 16-byte part) of this lane, per piece of this wave (piece = 4 rd + wid: X pieces first, then
    // G pieces).  Kept as ONE register per piece: the chunk's byte offset relative to the tile's pixel (1, 1) -- ((sy - 1) d w + (sx - 1) d)
    // pixels of 48 bytes + 16 part, a multiple of 16 -- in the upper 27 bits (as offset / 16) and the tile column sx (0..17) in the lower
    // five: an item then costs one wave-uniform base and, per piece, seven vector instructions (column test, offset, select) instead of the
    // sixteen of the (sy, sx, part) form -- in-kernel stamps had 1.5 k of an item's 10-12 k cycles in this segment without the DMA instructions and 2.3-3 k with
    // them; the train step did not move with the shorter form (1.136 vs 1.137 ms): the SIMD issues its two waves' instruction streams back to back,
    // and 70 vector instructions fewer per item are 1 % of them
    int relsx[C::ROUNDS];
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
        const int piece = rd * 4 + wid;
        int sy, sx, part;
        if (piece < C::XR) {
            int c = piece * 64 + lane;
            c = c < C::XPIX * 3 ? c : C::XPIX * 3 - 1;
            const int pix = c / 3; part = c - pix * 3; sy = pix / C::XW; sx = pix - sy * C::XW;      // sy, sx include the +1 halo shift
        } else {
            int cg = (piece - C::XR) * 64 + lane;
            cg = cg < C::GPIX * 3 ? cg : C::GPIX * 3 - 1;
            const int gp = cg / 3; part = cg - gp * 3;
            if constexpr (DX) { sy = gp / C::GW; sx = gp % C::GW; }         // with halo: the X tile's geometry
            else { sy = gp / TW + 1; sx = gp % TW + 1; }
        }
        if constexpr (PAIR) {
            // columns 1-8: sub-grid rx, 10-17: sub-grid rx + 1 (one pixel to the right in the map); 0, 9, 18: zeros.  The low bits hold "this chunk exists"
            const bool real = sx != 0 && sx != 9 && sx != 18;
            const int colrel = sx < 9 ? (sx - 1) * d : 1 + (sx - 10) * d;
            const int rel16 = ((sy - 1) * d * w + colrel) * 3 + part;
            relsx[rd] = (int)(((unsigned)rel16 << 5) | (real ? 1u : 0u));
        } else {
        const int rel16 = ((sy - 1) * d * w + (sx - 1) * d) * 3 + part;     // offset / 16
        relsx[rd] = (int)(((unsigned)rel16 << 5) | (unsigned)sx);
        }
    }

    const int sh = geo.sh, tiles_y = geo.tiles_y, tiles_x = geo.tiles_x, dxp = geo.dxp, items = geo.items;      // from the host (w16_geometry)
    struct item_t { int img, ry, rx, sy0, sx0; };
    // item index -> (image, phase, tile) with magic-number divisions (wave-uniform: s_mul_hi; exact while it * divisor < 2^32)
    const unsigned m_tx = geo.m_tx, m_ty = geo.m_ty, m_d = geo.m_d, m_dx = geo.m_dx;
    auto divm = [](unsigned a, unsigned dv, unsigned m) { return dv == 1u ? a : __umulhi(a, m); };
    auto decode = [&](int it) {
        item_t r;
        unsigned a = (unsigned)it, b;
        b = divm(a, (unsigned)tiles_x, m_tx); const int tx = (int)(a - b * (unsigned)tiles_x); a = b;
        b = divm(a, (unsigned)tiles_y, m_ty); const int ty = (int)(a - b * (unsigned)tiles_y); a = b;
        b = divm(a, (unsigned)dxp, m_dx); r.rx = (int)(a - b * (unsigned)dxp) * (PAIR ? 2 : 1); a = b;
        b = divm(a, (unsigned)d, m_d); r.ry = (int)(a - b * (unsigned)d);
        r.img = (int)b;
        r.sy0 = ty * C::TH; r.sx0 = tx * TW;
        return r;
    };
    const unsigned lds_smem = ubd_lds_addr(smem);
    // One buffer descriptor per image and tensor: pixels above / below the image fall out of its range by themselves, pixels left /
    // right of it get an out-of-range offset, and the LDS-DMA writes ZEROS for them -- the 'same' padding of X and the ragged
    // border of a sub-grid tile cost neither clamped addresses nor the fix-up pass + second barrier per item that most items of a
    // dilated layer needed (round-2 stamps: ~1 k of 5.5 k cycles per item).
    const unsigned img_bytes = (unsigned)h * (unsigned)w * (UBD_C * 2);
    auto dma_item = [&](const item_t &I, int bufoff) {
        const size_t imgoff = (size_t)I.img * h * w * (UBD_C * 2);
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)x + imgoff), 0, (int)img_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)gz + imgoff), 0, (int)img_bytes, 0x00020000);
        // wave-uniform: byte offset of the tile's pixel (1, 1); tile columns sx_lo .. sx_hi are inside the map (gx = rx + (sx0 + sx - 1) d in [0, w)).
        // Rows above / below the map need no test: their offsets are negative or >= img_bytes, i.e. outside the descriptor's range.
        const int gx1 = I.rx + I.sx0 * d;
        const int base = ((I.ry + I.sy0 * d) * w + gx1) * (UBD_C * 2);
        const int sx_lo = gx1 - d < 0 ? 1 : 0;
        const int sx_span = (w - 1 - gx1 + d) / d - sx_lo;                  // sx_hi - sx_lo where the tile starts inside the map (gx1 <= w - 1)
        // A map NARROWER than the dilation has column phases without a single pixel (rx >= w: e.g. a 10-wide map at dilation 16).  Their
        // tiles are all padding; without this test sx_span went negative, the unsigned compare below accepted every column, and the
        // staged tiles held OTHER columns' pixels -- the layer's weight and bias gradients came out several times too large (found by the
        // random-shape soak of round 6, 3 x 104 x 40 images; maps of the BASELINE shapes are never narrower than their dilation).
        const bool inside = gx1 < w;
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int piece = rd * 4 + wid;
            if (piece >= C::XR + C::GR) break;                        // wave-uniform
            const int v = relsx[rd];
            unsigned off;
            if constexpr (PAIR) off = (v & 1) ? (unsigned)(base + ((v >> 1) & ~15)) : 0x80000000u;      // both sub-grids are exactly 8 columns wide (host)
            else off = (inside && (unsigned)((v & 31) - sx_lo) <= (unsigned)sx_span) ? (unsigned)(base + ((v >> 1) & ~15)) : 0x80000000u;
            ubd_blds16(piece < C::XR ? rx : rg, off, lds_smem + bufoff + piece * 1024);   // asm form (common.h): hipcc drained the builtin in front of the tr reads
        }
    };

    // XCD-aware item ranges: blocks b, b + 8, ... share an XCD (and its L2) and walk one contiguous eighth of the item
    // list (image-major), so the d x d phases of an image -- which interleave inside the same cache lines -- are fetched
    // through ONE L2 instead of up to eight (d = 8: 65 -> 37 us)
    const int xcd = blockIdx.x & 7;
    const int nblk_x = ((int)gridDim.x + 7 - xcd) >> 3;
    const int chunk = (items + 7) >> 3;
    const int it_begin = xcd * chunk;
    const int it_end = it_begin + chunk < items ? it_begin + chunk : items;
    f32x4 acc[MTW][2] = {};
    int it = it_begin + (int)(blockIdx.x >> 3);
    item_t I = decode(it < it_end ? it : it_begin);
    if (it < it_end) dma_item(I, 0);
    for (int iter = 0; it < it_end; ++iter, it += nblk_x) {
        char *buf = smem + (iter & 1) * C::BUF_BYTES;
        WGSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this item's DMA (issued one item ago, as asm) has landed
        WGSTAMP(1);
        __syncthreads();                              // ... for every wave; everyone left the other buffer
        WGSTAMP(2);
        item_t Inext = I;
        if (it + nblk_x < it_end) { Inext = decode(it + nblk_x); dma_item(Inext, ((iter + 1) & 1) * C::BUF_BYTES); }
        WGSTAMP(3);
        const int rows_eff = min(C::TH, sh - I.sy0);
#pragma unroll 1
        for (int kb = MSPLIT ? 0 : wid; C::KROWS * kb < rows_eff; kb += MSPLIT ? 1 : 4) {   // wave-uniform: k-blocks whose tile rows hold real sub-pixels
            const int py = C::KROWS * kb + krow;
            const char *xb = buf + (py * C::XW + kcolp) * (UBD_C * 2);
            const char *gb = buf + C::GOFF + ((py + (DX ? 1 : 0)) * C::GW + kcolp + (DX ? 1 : 0)) * (UBD_C * 2);
            // B operand: segments of the two N tiles (co 0..15, 16..23 + zero padding)
            u32x4 b[2];
            {
                const char *g0 = gb + 8 * p;
                const char *g1 = p < 2 ? gb + 32 + 8 * p : smem + CONST_OFF + 8;
                const s16x4 v00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)g0);
                const s16x4 v01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(g0 + 4 * UBD_C * 2));
                const s16x4 v10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)g1);
                const s16x4 v11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p < 2 ? g1 + 4 * UBD_C * 2 : g1));
                b[0] = __builtin_bit_cast(u32x4, __builtin_shufflevector(v00, v01, 0, 1, 2, 3, 4, 5, 6, 7));
                b[1] = __builtin_bit_cast(u32x4, __builtin_shufflevector(v10, v11, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int sl = 0; sl < MTW; ++sl) {
                const int mt = mt_of(sl);
                if (mt < 14) {
                    const char *a0 = xb + aoff[sl];
                    const char *a1 = a0 + 4 * UBD_C * 2;
                    if (mt == 13 && p >= 2) { a0 = smem + CONST_OFF + (p == 2 ? 0 : 8); a1 = a0; }
                    const s16x4 va0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
                    const s16x4 va1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
                    const u32x4 a = __builtin_bit_cast(u32x4, __builtin_shufflevector(va0, va1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[sl][0] = mfma16<T>(a, b[0], acc[sl][0]);
                    acc[sl][1] = mfma16<T>(a, b[1], acc[sl][1]);
                }
            }
        }
        WGSTAMP(4);
        if constexpr (DX) {
            const int i16 = lane & 15;
            __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)gout + (size_t)I.img * h * w * (UBD_C * 2)), 0, (int)img_bytes, 0x00020000);
            const char *zero16 = smem + CONST_OFF + 16;
            // both halves of the fragments (output channels 0..15 / 16..23) live in registers for the duration of this phase only
            u32x4 wr0[7], wr1[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) { wr0[c] = ((const u32x4 *)(smem + WT_OFF))[(c * 2) * 64 + lane]; wr1[c] = ((const u32x4 *)(smem + WT_OFF))[(c * 2 + 1) * 64 + lane]; }
#pragma unroll 1
            for (int kb = wid; C::KROWS * kb < rows_eff; kb += 4) {
                // one MFMA column tile = 16 pixels = one tile row of a 16-wide tile, two rows of an 8-wide one
                constexpr int RSTEP = 16 / TW;
                const int prow = TW == 16 ? 0 : i16 >> 3, pcol = TW == 16 ? i16 : i16 & 7;
                const int pcolp = pcol + (PAIR ? (pcol >> 3) : 0);   // column inside the staged images
#pragma unroll 1
                for (int rr = 0; rr < C::KROWS; rr += RSTEP) {
                    const int r0 = C::KROWS * kb + rr;
                    if (r0 >= rows_eff) break;                                // wave-uniform
                    const int r = r0 + prow;
                    const char *gpix = buf + C::GOFF + ((r + 1) * C::GW + pcolp + 1) * (UBD_C * 2);
                    u32x4 a[7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) a[c] = *(const u32x4 *)(doff[c] > -(1 << 19) ? gpix + doff[c] : zero16);
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 7; ++c) {                              // weights as the A operand: D = [channel][pixel]
                        acc0 = mfma16<T>(wr0[c], a[c], acc0);
                        acc1 = mfma16<T>(wr1[c], a[c], acc1);
                    }
                    // ReLU mask = the saved activation X of this pixel (centre of the X tile), same channels as the result rows
                    const char *xpix = buf + ((r + 1) * C::XW + pcolp + 1) * (UBD_C * 2);
                    const u32x2 m0 = *(const u32x2 *)(xpix + 8 * grp);
                    const u32x2 m1 = *(const u32x2 *)(xpix + (grp < 2 ? 32 + 8 * grp : 0));
                    const u32x2 o0 = {wg_relu_mask2(wg_pack2<T>(acc0[0], acc0[1]), m0[0]), wg_relu_mask2(wg_pack2<T>(acc0[2], acc0[3]), m0[1])};
                    const u32x2 o1 = {wg_relu_mask2(wg_pack2<T>(acc1[0], acc1[1]), m1[0]), wg_relu_mask2(wg_pack2<T>(acc1[2], acc1[3]), m1[1])};
                    const int gy = I.ry + (I.sy0 + r) * d, gx = PAIR ? I.rx + (pcol >> 3) + (pcol & 7) * d : I.rx + (I.sx0 + pcol) * d;
                    const unsigned off = (gx < w && gy < h) ? (unsigned)((gy * w + gx) * (UBD_C * 2)) + 8u * grp : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b64(o0, rout, (int)off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(o1, rout, (int)(grp < 2 ? off + 32u : 0x80000000u), 0, 0);   // channels 16 + 4 grp + r exist for grp < 2
                }
            }
        }
        WGSTAMP(5);
        I = Inext;
    }
#undef WGSTAMP
    if constexpr (MSPLIT) {
        // this block's row of the partial-sum matrix: every wave writes the M tiles it owns (D layout: col = lane & 15 = co within the N
        // tile, row = 4 (lane >> 4) + r = rho within the M tile)
        float *__restrict__ prow = partials + (size_t)blockIdx.x * (217 * UBD_C);
#pragma unroll
        for (int sl = 0; sl < MTW; ++sl) {
            const int mt = mt_of(sl);
            if (mt < 14) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * grp + r, col = 16 * nt + (lane & 15);
                        if (row < 217 && col < UBD_C) prow[row * UBD_C + col] = acc[sl][nt][r];
                    }
            }
        }
    } else {
        wgrad_block_reduce(acc, (float *)smem, partials + (size_t)blockIdx.x * (217 * UBD_C), lane, wid);
    }
    __syncthreads();                                   // every wave has left the LDS
    rp_reduce_tail(prev, (float *)smem);               // the partial rows of the producer in front of this kernel (backward.hip)
}
