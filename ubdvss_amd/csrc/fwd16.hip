// 16-bit activation forward pass (UBD_BF16 / UBD_F16): BASELINE.json configs[2..4] store activations in
// bf16 / fp16; weights stay fp32 masters (converted once per call into packed 16-bit MFMA fragments),
// every accumulation is fp32, logits are fp32.
//
// Reference semantics as forward.hip (semantic_segmentation/net.py:225-252, :278-314).
//   sepconv16_kernel   L1..L3: depthwise 3x3 in fp32 on the VALU (inputs widened on load), pointwise 1x1 on the
//                      fp32 MFMA, bias + ReLU, narrowed to 16 bit through an LDS transpose so that every tile
//                      leaves as full 16-byte stores.
//   dilconv16_kernel   L4..L9: implicit GEMM on v_mfma_f32_16x16x32_{bf16,f16}: M = 16 pixels, N = 24 (2 x 16),
//                      K = 216 padded to 7 x 32.  24 channels = 3 groups of 8, so every lane's 8-element
//                      K-slice is ONE 16-byte piece of one tap's pixel (buffer_load_dwordx4, zero fill = padding);
//                      the layer's weights live in 56 VGPRs.  At 16 bit the layer is HBM-bound (AI ~46 flop/B).
//   head16_kernel      1x1 conv 24 -> K from 16-bit activations, fp32 logits.
#include "common.h"
#include "pack.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct h16;
template <> struct h16<__bf16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct h16<_Float16> {
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ unsigned short to_bits(float v) { return __builtin_bit_cast(unsigned short, (T)v); }
template <typename T> __device__ __forceinline__ float from_bits(unsigned short b) { return (float)__builtin_bit_cast(T, b); }
template <typename T> __device__ __forceinline__ void widen2(unsigned w, float &lo, float &hi)
{
    lo = from_bits<T>((unsigned short)(w & 0xFFFFu));
    hi = from_bits<T>((unsigned short)(w >> 16));
}


// acc + a.lo * b.lo + a.hi * b.hi with 16-bit inputs and fp32 accumulation (v_dot2c_f32_{bf16,f16})
template <typename T> __device__ __forceinline__ float dot2_16(unsigned a, unsigned b, float acc);
template <> __device__ __forceinline__ float dot2_16<__bf16>(unsigned a, unsigned b, float acc)
{
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), acc, false);
}
template <> __device__ __forceinline__ float dot2_16<_Float16>(unsigned a, unsigned b, float acc)
{
    typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), acc, false);
}
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    // one v_cvt_pk_bf16_f32 (bf16) / two v_cvt_f16_f32 + pack (fp16); round to nearest even like the scalar casts
    typedef T t2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, t2));
}
template <typename T> __device__ __forceinline__ float round16(float v) { return (float)(T)v; }
// ReLU of two packed 16-bit floats: negative values have negative int16 patterns (v_pk_max_i16 with 0; -0 becomes +0)
__device__ __forceinline__ unsigned relu_pk16(unsigned v)
{
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
// 0xFFFF per half whose 16-bit pattern is a positive short, else 0: max(m, 0) -> min(., 1) -> * 0xFFFF (three packed ops)
__device__ __forceinline__ unsigned mask_pk16(unsigned m)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 one = {1, 1}, ones = {0xFFFF, 0xFFFF};
    const u16x2 p = __builtin_bit_cast(u16x2, relu_pk16(m));
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_elementwise_min(p, one) * ones));
}

// ------------------------------------------------------------------------------------ pack (pack.h)
template <typename T>
__global__ void pack16_kernel(const float *__restrict__ params, unsigned *__restrict__ out, size_t off0, size_t layer_stride, int transpose,
                              pack_sep16_args sa, int with_sep)
{
    pack16_body<T>(params, out, off0, layer_stride, transpose, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
    // forward fragments: followed by the ready operands of L2 / L3 for the one-kernel stem (sep123_16.h)
    if (with_sep) pack_sep16_ready_body<T>(params, out + UBD_NUM_DIL * UBD_DIL16_FRAG_U32, sa, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// ------------------------------------------------------------------------------------ shared epilogue
// All products are issued with the weights as the MFMA A operand (D = W^T . X^T): D col = lane & 15 = pixel of the
// 16-pixel tile, row = 4*(lane>>4) + reg = output channel, i.e. every lane holds channels 4q..4q+3 (N-tile 0) and
// 16+4q..19+4q (N-tile 1, lanes q < 2) of ITS pixel: the tile leaves as two 8-byte buffer stores per lane, no LDS
// transpose.  EPI 0: y = relu(acc + bias); EPI 1: y = acc * (mask > 0) with the 16-bit mask words m0 / m1 of the same
// bytes.  Exactly two unconditional stores per call (callers count them for s_waitcnt vmcnt).
template <typename T, int EPI>
__device__ __forceinline__ void store_tile16_t(unsigned short *__restrict__ y, size_t first_pixel, int npx, int lane,
                                               f32x4 acc0, f32x4 acc1, f32x4 bA, f32x4 bB, u32x2 m0, u32x2 m1)
{
    const int i = lane & 15, q = lane >> 4;
    // EPI 0: bias added last (the order the oracle uses); ReLU is taken on the packed 16-bit values: a negative float
    // has a negative 16-bit pattern, v_pk_max_i16 with 0 clears it (and turns -0 into +0), two channels per instruction.
    if constexpr (EPI == 0) { acc0 += bA; acc1 += bB; }
    unsigned w00 = pack2<T>(acc0[0], acc0[1]), w01 = pack2<T>(acc0[2], acc0[3]);
    unsigned w10 = pack2<T>(acc1[0], acc1[1]), w11 = pack2<T>(acc1[2], acc1[3]);
    if constexpr (EPI == 0) {
        w00 = relu_pk16(w00); w01 = relu_pk16(w01); w10 = relu_pk16(w10); w11 = relu_pk16(w11);
    } else {
        // ReLU mask: the saved activation is > 0 iff its 16-bit pattern is a positive short
        w00 &= mask_pk16(m0[0]); w01 &= mask_pk16(m0[1]); w10 &= mask_pk16(m1[0]); w11 &= mask_pk16(m1[1]);
    }
    const u32x2 o0 = {w00, w01}, o1 = {w10, w11};
    const unsigned long long rp = (unsigned long long)(y + first_pixel * UBD_C);
    const unsigned rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rp);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rp >> 32));
    npx = npx < 0 ? 0 : npx;
    const unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane(npx * UBD_C * 2);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)rhi << 32) | rlo), 0, (int)bytes, 0x00020000);
    const unsigned base = (unsigned)i * (UBD_C * 2u) + 8u * (unsigned)q;       // beyond `bytes` for pixels >= npx
    __builtin_amdgcn_raw_buffer_store_b64(o0, rs, (int)base, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(o1, rs, (int)(q < 2 ? base + 32u : 0x40000000u), 0, 0);
}

// ------------------------------------------------------------------------------------ separable layers
// LDS-staged like the fp32 sepconv_kernel (forward.hip): persistent blocks walk output tiles of 16 columns x TH
// rows; the input patch crosses L2->CU once per tile.  IN_MODE 0: fp32 input, 1: uint8 input (CIN 1/3, patch
// converted to fp32 on the way into LDS), 2: 16-bit input (CIN 24): the raw 48-byte pixels are copied with
// LDS-DMA (3 x 16 B per pixel, double-buffered against the compute phase) and widened when the taps are read.
// Channel split of a 24-channel pixel over the 4 k-groups: q takes channels 4q..4q+3 (one ds_read_b64 at byte 8q)
// and 16+2q, 17+2q (one ds_read_b32 at byte 32+4q); pixel stride 48 B = 12 banks keeps 16 neighbouring pixels
// conflict-free at stride 1.  The per-lane depthwise / pointwise weights are re-gathered to that split.
template <int CIN, int STRIDE> struct sep16_cfg {
    static constexpr int TH = (CIN == UBD_C && STRIDE == 2) ? 4 : 16;         // stride 2: 9 x 33 pixel patches (14 KiB)
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = 15 * STRIDE + 3;
    static constexpr int CHUNKS = (CIN == UBD_C) ? PH * PW * 3 : 0;           // 16-byte chunks of the 16-bit patch
    static constexpr int ROUNDS = (CIN == UBD_C) ? (CHUNKS + 255) / 256 : 1;
    static constexpr int ELEMS = PH * PW * CIN;
    static constexpr int BUF_BYTES = (CIN == UBD_C) ? ROUNDS * 256 * 16 : (ELEMS + 3) / 4 * 16;
    static constexpr int STAGE_REGS = (CIN == UBD_C) ? 1 : (ELEMS + 255) / 256;
    static constexpr int NSTORE = TH / 2;                                     // buffer stores per wave per tile (2 per row tile)
};

template <int CIN, int STRIDE, int IN_MODE, typename T>
__global__ __launch_bounds__(256, (CIN == UBD_C) ? 1 : 5) void sepconv16_kernel(const void *__restrict__ xin, unsigned short *__restrict__ y,
                                                        const float *__restrict__ frag, const float *__restrict__ bias, int n,
                                                        int H, int W, int OH, int OW, int pad_lo, float pre_sub, float pre_div)
{
    using C = sep16_cfg<CIN, STRIDE>;
    constexpr int CPL = (CIN == UBD_C) ? 6 : 1;
    // 24 channels: ring of three LDS patch buffers (16 / 14 KiB each), i.e. the DMA runs TWO tiles ahead: one tile's
    // compute phase (~1 us) is shorter than the DMA latency under load.
    constexpr int NBUF = (CIN == UBD_C) ? 3 : 1;
    constexpr int NR = C::TH / 4;                                           // row tiles per wave per block tile
    // ONE LDS object: with a second __shared__ array hipcc orders every LDS read behind the LDS-DMA in flight (s_waitcnt vmcnt(0))
    __shared__ __attribute__((aligned(16))) char patch_mem[NBUF * C::BUF_BYTES];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    const float *pwfrag = frag, *dwlane = frag + UBD_SEP_FRAG_FLOATS;
    // 16-bit mode uses every convolution kernel in the activation type (oracle: net_numpy.forward act_dtype).
    // 24 channels: the depthwise taps are packed as (w, 0) / (0, w) 16-bit pairs so that ONE v_dot2c per channel and tap
    // multiplies the raw 16-bit activation pair from LDS (no widening), and the pointwise product is ONE
    // v_mfma_f32_16x16x32_{bf16,f16} per N tile: k-slot 8q + e holds channel chs(q, e) for e < 6, zero for e = 6, 7.
    // 1/3 channels: fp32 image x rounded taps on the VALU, rounded depthwise output, fp32 MFMA (K = 4).
    float dwk[9][CPL], pwf[CPL][2];
    u32x4 pwb[2];                                      // 24 channels: B operands of the two N tiles
    // 24 channels: the depthwise convolution runs on the matrix pipe.  A depthwise tap is a diagonal weight matrix, so the K = 32
    // of one 16-bit MFMA carries several taps: two taps x 16 channels for channels 0..15 (five MFMAs for nine taps; the spare
    // slot re-reads tap 8 with zero weights) and four taps x 8 channels for channels 16..23 (three MFMAs, rows permuted so that
    // lane (pixel, q) receives channels 16 + 2q, 17 + 2q): the B operand is 16 raw bytes of the LDS patch at the tap's pixel.
    // 8 ds_read_b128 + 8 MFMAs per 16-pixel row instead of 18 reads + 54 v_dot2c (the kernel is bound by vector-instruction issue).
    u32x4 wa0[5], wa1[3];
    int xo0[5], xo1[3];                                // B operand byte offsets inside the patch for tile row 0
    if constexpr (CIN == UBD_C) {
        // the fragments are packed for channel 6q'+s' (forward.hip): entry of tap t, channel ch
        auto wdw = [&](int t, int ch) { return to_bits<T>(dwlane[(t * 6 + ch % 6) * 64 + 16 * (ch / 6)]); };
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int ts = 2 * j + (q >> 1), t = ts < 9 ? ts : 8;
            const int e = i - 8 * (q & 1);                         // k = 8q + e <-> channel 8 (q & 1) + e == row i
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && e >= 0 && e < 8) w[e >> 1] = wdw(t, i) << (16 * (e & 1));
            wa0[j] = u32x4{w[0], w[1], w[2], w[3]};
            xo0[j] = ((t / 3) * C::PW + i * STRIDE + t % 3) * (UBD_C * 2) + 16 * (q & 1);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ts = 4 * j + q, t = ts < 9 ? ts : 8;         // k-group q <-> tap 4j + q, channels 16..23
            const int r = i & 3, e = 2 * (i >> 2) + r;             // row i = 4 qq + r <-> channel 16 + 2 qq + r (r < 2)
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && r < 2) w[e >> 1] = wdw(t, 16 + e) << (16 * (e & 1));
            wa1[j] = u32x4{w[0], w[1], w[2], w[3]};
            xo1[j] = ((t / 3) * C::PW + i * STRIDE + t % 3) * (UBD_C * 2) + 32;
        }
        float pw6[6][2];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);      // this lane's channels
            const int src_lane = 16 * (ch / 6) + i, ss = ch % 6;
            pw6[s][0] = pwfrag[(ss * 2 + 0) * 64 + src_lane];
            pw6[s][1] = pwfrag[(ss * 2 + 1) * 64 + src_lane];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            pwb[nt] = u32x4{pack2<T>(pw6[0][nt], pw6[1][nt]), pack2<T>(pw6[2][nt], pw6[3][nt]), pack2<T>(pw6[4][nt], pw6[5][nt]), 0u};
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk[t][0] = round16<T>(dwlane[(t * 6) * 64 + lane]);
        pwf[0][0] = round16<T>(pwfrag[0 * 64 + lane]);
        pwf[0][1] = round16<T>(pwfrag[1 * 64 + lane]);
    }
    const f32x4 bA = *(const f32x4 *)(bias + 4 * q);
    const f32x4 bB = q < 2 ? *(const f32x4 *)(bias + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int cb = (q < CIN) ? q : 0;                                       // CIN < 24: this lane's channel (weights are 0 beyond)

    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + C::TH - 1) / C::TH;
    const int total = n * tiles_y * tiles_x;
    ubd_tile_decoder tdec;                                                 // neighbouring tiles on one XCD (shared halo lines), no divisions
    tdec.init(tiles_x, tiles_y, total);
    auto tile_coords = [&](int tile, int &img, int &oy0, int &ox0) {
        int tx, ty;
        tdec.decode(tile, tx, ty, img);
        oy0 = ty * C::TH; ox0 = tx * 16;
    };
    int dma_rel[C::ROUNDS];
    if constexpr (CIN == UBD_C) {
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            int c = rd * 256 + wid * 64 + lane;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pix = c / 3, sp = c - pix * 3;
            const int pr = pix / C::PW, pc = pix - pr * C::PW;
            dma_rel[rd] = (pr * W + pc) * (UBD_C * 2) + sp * 16;
        }
    }
    auto dma_tile = [&](int tile, char *buf) {
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int ix0 = ox0 * STRIDE - pad_lo, iy0 = oy0 * STRIDE - pad_lo;
        const char *src = (const char *)xin;
        const bool interior = (iy0 >= 0) && (ix0 >= 0) && (iy0 + C::PH <= H) && (ix0 + C::PW <= W);   // block-uniform
        if (interior) {
            const char *origin = src + (((size_t)img * H + iy0) * W + ix0) * (UBD_C * 2);
#pragma unroll
            for (int rd = 0; rd < C::ROUNDS; ++rd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(origin + dma_rel[rd]),
                                                 (__attribute__((address_space(3))) void *)(buf + (rd * 256 + wid * 64) * 16), 16, 0, 0);
            return;
        }
        // border tile: the same pieces through a buffer descriptor that covers exactly the image -- rows above / below it fall out
        // of range by themselves, columns left / right of it get an out-of-range offset, and the LDS-DMA writes ZEROS for them
        // (the 'same' padding; no clamped addresses, no zero-fix pass and no extra barrier afterwards)
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(src + (size_t)img * H * W * (UBD_C * 2)), 0,
                                                                        (int)((unsigned)H * W * (UBD_C * 2)), 0x00020000);
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int cbase = rd * 256 + wid * 64;
            int c = cbase + lane;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pix = c / 3, sp = c - pix * 3;
            const int pr = pix / C::PW, pc = pix - pr * C::PW;
            const int gy = iy0 + pr, gx = ix0 + pc;
            const unsigned off = (unsigned)gx < (unsigned)W ? (unsigned)((gy * W + gx) * (UBD_C * 2) + sp * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + cbase * 16), 16, (int)off, 0, 0, 0);
        }
    };
    int ld_rel[C::STAGE_REGS];
    if constexpr (CIN != UBD_C) {
#pragma unroll
        for (int k = 0; k < C::STAGE_REGS; ++k) {
            int e = k * 256 + threadIdx.x;
            e = e < C::ELEMS ? e : C::ELEMS - 1;
            const int pix = e / CIN, ch = e - pix * CIN;
            const int pr = pix / C::PW, pc = pix - pr * C::PW;
            ld_rel[k] = (pr * W + pc) * CIN + ch;
        }
    }
    auto load_regs = [&](int tile, unsigned (&st)[C::STAGE_REGS]) {
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int ix0 = ox0 * STRIDE - pad_lo, iy0 = oy0 * STRIDE - pad_lo;
        const bool interior = (iy0 >= 0) && (ix0 >= 0) && (iy0 + C::PH <= H) && (ix0 + C::PW <= W);   // block-uniform
        if (interior) {
            const size_t origin = (((size_t)img * H + iy0) * W + ix0) * CIN;
#pragma unroll
            for (int k = 0; k < C::STAGE_REGS; ++k) {
                if constexpr (IN_MODE == 1) st[k] = ((const unsigned char *)xin)[origin + ld_rel[k]];
                else st[k] = ((const unsigned *)xin)[origin + ld_rel[k]];
            }
            return;
        }
        // border tile: every element from the clamped position (image base in scalar registers + a 32-bit offset; the size_t index
        // arithmetic under per-element branches cost five quarter-rate v_mad_u64_u32 each), all loads in flight, then the elements
        // outside the image are replaced (exactly 0 after the preprocessing below)
        const int WC = W * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)img * H * WC * ((IN_MODE == 1) ? 1 : 4);
#pragma unroll
        for (int k = 0; k < C::STAGE_REGS; ++k) {
            int e = k * 256 + threadIdx.x;
            e = e < C::ELEMS ? e : C::ELEMS - 1;
            const int pr = e / (C::PW * CIN), pf = e - pr * (C::PW * CIN);
            const int gy = min(max(iy0 + pr, 0), H - 1), gf = min(max(ix0 * CIN + pf, 0), WC - 1);
            const unsigned off = (unsigned)(gy * WC + gf);
            if constexpr (IN_MODE == 1) st[k] = img8[off];
            else st[k] = ((const unsigned *)img8)[off];
        }
#pragma unroll
        for (int k = 0; k < C::STAGE_REGS; ++k) {
            int e = k * 256 + threadIdx.x;
            e = e < C::ELEMS ? e : C::ELEMS - 1;
            const int pr = e / (C::PW * CIN), pf = e - pr * (C::PW * CIN);
            const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(ix0 * CIN + pf) < (unsigned)WC;
            st[k] = inside ? st[k] : ((IN_MODE == 1) ? 0x100u : __builtin_bit_cast(unsigned, pre_sub));
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    unsigned stage[C::STAGE_REGS];      // raw loaded bits (fp32 pattern or zero-extended byte): nothing consumes them before the LDS write
    constexpr int AHEAD = NBUF - 1;                                   // tiles in flight ahead of the one being computed
    if constexpr (CIN == UBD_C) {
        dma_tile(tile, patch_mem);
        if (AHEAD == 2 && tile + (int)gridDim.x < total) dma_tile(tile + gridDim.x, patch_mem + C::BUF_BYTES);
    } else {
        load_regs(tile, stage);
    }

    for (int it = 0;; ++it) {
        char *patch = patch_mem + ((CIN == UBD_C) ? (it % NBUF) * C::BUF_BYTES : 0);
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int nxt = tile + gridDim.x;
        const bool has_next = (CIN == UBD_C) && nxt < total;         // block-uniform; 1/3-channel tiles: one per block (see launch)
        if constexpr (CIN == UBD_C) {
            // Counted wait (vmcnt counts stores too and retires in order): this tile's DMA must have landed, everything
            // issued after it may stay in flight.  After DMA(it) each wave issued: [AHEAD == 2: the stores of tile it-2,]
            // the DMA of tile it+AHEAD-1... i.e. per later tile C::ROUNDS DMA instructions (if that tile exists) and per
            // computed tile C::NSTORE buffer stores.
            static_assert(CIN != UBD_C || AHEAD == 2, "ring of three");
            constexpr int R = C::ROUNDS, S = C::NSTORE;
            if (it == 0) { if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            else if (it == 1) { if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S + R) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory"); }
            else { if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S + R) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S) : "memory"); }
            __builtin_amdgcn_s_barrier();
            {
                const int ahead_tile = tile + AHEAD * (int)gridDim.x;
                if (ahead_tile < total) dma_tile(ahead_tile, patch_mem + ((it + AHEAD) % NBUF) * C::BUF_BYTES);
            }
        } else {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int k = 0; k < C::STAGE_REGS; ++k) {
                const int e = k * 256 + threadIdx.x;
                if (e < C::ELEMS) {                                   // raw values were in flight during the previous compute phase
                    if constexpr (IN_MODE == 1) ((float *)patch)[e] = stage[k] > 255u ? 0.f : ((float)stage[k] - pre_sub) / pre_div;
                    else ((float *)patch)[e] = (__builtin_bit_cast(float, stage[k]) - pre_sub) / pre_div;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            if (has_next) load_regs(nxt, stage);
        }

        if constexpr (CIN == UBD_C) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = STRIDE == 1 ? wid * NR + k : wid + 4 * k;
                const int oy = oy0 + r;
                const char *rowb = patch + r * (STRIDE * C::PW * UBD_C * 2);
                u32x4 b0[5], b1[3];
#pragma unroll
                for (int j = 0; j < 5; ++j) b0[j] = *(const u32x4 *)(rowb + xo0[j]);
#pragma unroll
                for (int j = 0; j < 3; ++j) b1[j] = *(const u32x4 *)(rowb + xo1[j]);
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 c0 = z4, c1 = z4;
#pragma unroll
                for (int j = 0; j < 5; ++j) c0 = h16<T>::mfma(wa0[j], b0[j], c0);
#pragma unroll
                for (int j = 0; j < 3; ++j) c1 = h16<T>::mfma(wa1[j], b1[j], c1);
                // depthwise output rounded to T (as the oracle stores it) = B operand of the pointwise product
                const u32x4 av = {pack2<T>(c0[0], c0[1]), pack2<T>(c0[2], c0[3]), pack2<T>(c1[0], c1[1]), 0u};
                const f32x4 acc0 = h16<T>::mfma(pwb[0], av, z4);     // weights as the A operand: D = [channel][pixel]
                const f32x4 acc1 = h16<T>::mfma(pwb[1], av, z4);
                int npx = OW - ox0 < 16 ? OW - ox0 : 16;
                npx = (oy < OH) ? npx : 0;
                const u32x2 nomask = {0u, 0u};
                store_tile16_t<T, 0>(y, ((size_t)img * OH + (oy < OH ? oy : 0)) * OW + ox0, npx, lane, acc0, acc1, bA, bB, nomask, nomask);
            }
        } else {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = wid + 4 * k;
            const int oy = oy0 + r;
            float dwv = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    dwv = fmaf(((const float *)patch)[((r * STRIDE + ky) * C::PW + i * STRIDE + kx) * CIN + cb], dwk[ky * 3 + kx][0], dwv);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const float dr = round16<T>(dwv);
            const f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[0][0], dr, z4, 0, 0, 0);
            const f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[0][1], dr, z4, 0, 0, 0);
            int npx = OW - ox0 < 16 ? OW - ox0 : 16;
            npx = (oy < OH) ? npx : 0;
            const u32x2 nomask = {0u, 0u};
            store_tile16_t<T, 0>(y, ((size_t)img * OH + (oy < OH ? oy : 0)) * OW + ox0, npx, lane, acc0, acc1, bA, bB, nomask, nomask);
        }
        }
        if (!has_next) break;
        tile = nxt;
    }
}

// ------------------------------------------------------------------------------------ L1 -> L2 in one kernel
// net.py:292-294: ZeroPadding2D + SeparableConv2D(24, 3x3, stride 2, relu) (L1) and SeparableConv2D(24, 3x3, 'same', relu) (L2).
// L2's 18 x 18 input patch is COMPUTED from the image instead of being copied from memory: L1's activation (201 MB per 64 images of
// 512 x 512 at 16 bit) is not read back, and an inference pass does not write it either (WRITE_A1 = false; the train step keeps it,
// every pixel stored once by the tile that owns it).  Same arithmetic as the two kernels above, operation for operation -- the
// depthwise sum in fp32 in tap order, rounded to T, fp32 MFMA, bias, rounding, ReLU on the packed pairs --, so a1 and a2 are
// bit-identical to the split pass (tests/test_gpu_forward16.py::test_fused_stem16_equals_split; UBD_STEM16=split keeps the two kernels).
// Price: L1 on the halo as well (324 / 256 pixels per tile); L1 is the cheap layer (1 or 3 input channels).
// Per tile: image patch 37 x 40 x CIN in LDS | barrier | L1 on 21 units of
// 16 patch pixels -> the patch image L2 reads (0 outside L1's map = L2's 'same' padding) | barrier | L2 as sepconv16_kernel<24, 1>.
template <int CIN> struct sep12_cfg {
    static constexpr int AP = 18;                                  // a1 patch side
    static constexpr int XP = 2 * (AP - 1) + 3;                    // image patch rows: 37
    // patch columns: the 37 that are needed start at image column 32 tx - 2 - pad; the LDS image starts at 32 tx - 4, so every patch row
    // begins on a 16-byte boundary of the image row (H, W multiples of 4) and is moved in 16-byte pieces: 40 columns
    static constexpr int XW = 40;
    static constexpr int ROWF = XW * CIN;                          // floats per patch row: 120 (RGB) / 40 (grey)
    static constexpr int RC = ROWF / 4;                            // 16-byte chunks per row
    static constexpr int CHUNKS = XP * RC;                         // 1110 / 370
    static constexpr int ROUNDS = (CHUNKS + 255) / 256;            // chunks per thread: 5 / 2
    static constexpr int A1_BYTES = 1024 * 16;                     // 324 pixels x 48 bytes + a spare corner for masked lanes
    static constexpr int XP_BYTES = CHUNKS * 16;
    static constexpr int UNITS = (AP * AP + 15) / 16;              // 21
    static constexpr int UPW = (UNITS + 3) / 4;                    // unit slots per wave: 6 (wave 0: 6 units, waves 1-3: 5 and a masked one)
};

// PLAIN: fp32 input that is fed as it is -- the image patch goes straight from memory into LDS (16-byte LDS-DMA through a buffer
// descriptor: zeros outside the image = L1's zero padding), requested a whole tile ahead into the other of two buffers (XB = 2) or,
// with one buffer (XB = 1: a fourth block fits the CU), as soon as L1 has read the patch; no staging registers.  Otherwise (uint8 and / or preprocessing): through registers at the head of the tile, converted on the way into LDS; the
// other blocks of the CU cover the load latency.
template <int CIN, int IN_MODE, bool PLAIN, bool WRITE_A1, typename T, int XB>
__global__ __launch_bounds__(256, XB == 1 ? 4 : 3) void sep12_16_kernel(const void *__restrict__ xin, unsigned short *__restrict__ a1out,
                                                         unsigned short *__restrict__ y, const float *__restrict__ frag1,
                                                         const float *__restrict__ bias1, const float *__restrict__ frag2,
                                                         const float *__restrict__ bias2, int n, int H, int W, int H2, int W2,
                                                         int pad_lo, float pre_sub, float pre_div)
{
    using C = sep12_cfg<CIN>;
    constexpr int PW = C::AP;
    static_assert(!PLAIN || IN_MODE == 0, "LDS-DMA moves fp32 pixels only");
    __shared__ __attribute__((aligned(16))) char smem[C::A1_BYTES + (PLAIN ? XB : 1) * C::XP_BYTES];
    char *a1p = smem;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;

    // L1: per-lane taps of channel q (zero beyond CIN) and the two pointwise fragments, rounded to T like every kernel of the 16-bit pass
    float dwk1[9];
    {
        const float *dwl1 = frag1 + UBD_SEP_FRAG_FLOATS;
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk1[t] = round16<T>(dwl1[(t * 6) * 64 + lane]);
    }
    const float pw1a = round16<T>(frag1[0 * 64 + lane]), pw1b = round16<T>(frag1[1 * 64 + lane]);
    const f32x4 b1A = *(const f32x4 *)(bias1 + 4 * q);
    const f32x4 b1B = q < 2 ? *(const f32x4 *)(bias1 + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int cb = (q < CIN) ? q : 0;

    // L2: tap-folded diagonal depthwise fragments + pointwise B operands (as sepconv16_kernel<24, 1>)
    u32x4 wa0[5], wa1[3], pwb[2];
    int xo0[5], xo1[3];
    {
        const float *pwfrag = frag2, *dwlane = frag2 + UBD_SEP_FRAG_FLOATS;
        auto wdw = [&](int t, int ch) { return to_bits<T>(dwlane[(t * 6 + ch % 6) * 64 + 16 * (ch / 6)]); };
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int ts = 2 * j + (q >> 1), t = ts < 9 ? ts : 8;
            const int e = i - 8 * (q & 1);
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && e >= 0 && e < 8) w[e >> 1] = wdw(t, i) << (16 * (e & 1));
            wa0[j] = u32x4{w[0], w[1], w[2], w[3]};
            xo0[j] = ((t / 3) * PW + i + t % 3) * (UBD_C * 2) + 16 * (q & 1);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ts = 4 * j + q, t = ts < 9 ? ts : 8;
            const int r = i & 3, e = 2 * (i >> 2) + r;
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && r < 2) w[e >> 1] = wdw(t, 16 + e) << (16 * (e & 1));
            wa1[j] = u32x4{w[0], w[1], w[2], w[3]};
            xo1[j] = ((t / 3) * PW + i + t % 3) * (UBD_C * 2) + 32;
        }
        float pw6[6][2];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);
            const int src_lane = 16 * (ch / 6) + i, ss = ch % 6;
            pw6[s][0] = pwfrag[(ss * 2 + 0) * 64 + src_lane];
            pw6[s][1] = pwfrag[(ss * 2 + 1) * 64 + src_lane];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            pwb[nt] = u32x4{pack2<T>(pw6[0][nt], pw6[1][nt]), pack2<T>(pw6[2][nt], pw6[3][nt]), pack2<T>(pw6[4][nt], pw6[5][nt]), 0u};
    }
    const f32x4 b2A = *(const f32x4 *)(bias2 + 4 * q);
    const f32x4 b2B = q < 2 ? *(const f32x4 *)(bias2 + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (W2 + 15) >> 4, tiles_y = (H2 + 15) >> 4;
    const int total = n * tiles_y * tiles_x;
    ubd_tile_decoder tdec;
    tdec.init(tiles_x, tiles_y, total);
    auto tile_coords = [&](int tile, int &img, int &oy0, int &ox0) {
        int tx, ty;
        tdec.decode(tile, tx, ty, img);
        oy0 = ty * 16; ox0 = tx * 16;
    };
    const int WC = W * CIN;
    // chunk c = 256 k + thread of the image patch = row c / RC, floats 4 (c % RC) .. + 3 of that row; the image's left and right
    // borders fall on chunk boundaries (W is a multiple of 4), so a chunk is inside the image or outside it as a whole
    const int dx0 = 2 - pad_lo;                                            // patch column of the first column that is needed
    const unsigned lds_xp = ubd_lds_addr(smem + C::A1_BYTES);
    auto dma_x = [&](int tile, int buf) {                                  // PLAIN
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int iy0 = (oy0 - 1) * 2 - pad_lo, if0 = (ox0 * 2 - 4) * CIN;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)xin + (size_t)img * H * WC), 0,
                                                                        (int)((unsigned)H * (unsigned)WC * 4u), 0x00020000);
        const unsigned dst = lds_xp + (unsigned)(buf * C::XP_BYTES + wid * 1024);
#pragma unroll
        for (int k = 0; k < C::ROUNDS; ++k) {
            const int c = k * 256 + (int)threadIdx.x;
            const int pr = c / C::RC, gf = if0 + 4 * (c - pr * C::RC);
            // rows above / below the image fall out of the descriptor's range by themselves (a negative offset wraps); chunks left / right
            // of it get an out-of-range offset: zeros in LDS.  Only the lanes that have a chunk write (16 bytes at dst + 16 * lane).
            const unsigned off = (unsigned)gf < (unsigned)WC ? (unsigned)(((iy0 + pr) * WC + gf) * 4) : 0x80000000u;
            if (k < C::ROUNDS - 1 || c < C::CHUNKS) ubd_blds16(rsrc, off, dst + (unsigned)(k * 4096));
        }
    };
    // !PLAIN: 4 bytes (uint8) / 16 bytes (fp32) per chunk through registers
    constexpr int SW = (IN_MODE == 1) ? 1 : 4;                             // dwords per chunk
    auto load_regs = [&](int tile, unsigned (&st)[C::ROUNDS][SW]) {
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int iy0 = (oy0 - 1) * 2 - pad_lo, if0 = (ox0 * 2 - 4) * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)img * H * WC * ((IN_MODE == 1) ? 1 : 4);
        // every chunk from the clamped position, all loads in flight, then the chunks outside the image are replaced (exactly 0 after
        // the preprocessing) -- a select per load would put every load behind its own branch and wait
#pragma unroll
        for (int k = 0; k < C::ROUNDS; ++k) {
            int c = k * 256 + (int)threadIdx.x;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pr = c / C::RC, pf = 4 * (c - pr * C::RC);
            const int gy = min(max(iy0 + pr, 0), H - 1), gf = min(max(if0 + pf, 0), WC - 4);
            const unsigned off = (unsigned)(gy * WC + gf);
            if constexpr (IN_MODE == 1) st[k][0] = *(const unsigned *)(img8 + off);
            else {
                const u32x4 v = *(const u32x4 *)(img8 + (size_t)off * 4);
                st[k][0] = v[0]; st[k][1] = v[1]; st[k][2] = v[2]; st[k][3] = v[3];
            }
        }
    };
    auto chunk_inside = [&](int k, int iy0, int if0) {
        int c = k * 256 + (int)threadIdx.x;
        c = c < C::CHUNKS ? c : C::CHUNKS - 1;
        const int pr = c / C::RC, pf = 4 * (c - pr * C::RC);
        return (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(if0 + pf) < (unsigned)WC;
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    if constexpr (PLAIN) dma_x(tile, 0);
    for (int it = 0;; ++it) {
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int nxt = tile + (int)gridDim.x;
        const bool has_next = nxt < total;                              // block-uniform
        const float *xp = (const float *)(smem + C::A1_BYTES + ((PLAIN && XB == 2) ? (it & 1) * C::XP_BYTES : 0));
        if constexpr (PLAIN) {
            // Counted wait (vmcnt counts stores too and retires in order): this tile's DMA was issued a tile ago; behind it the wave
            // issued that tile's stores: three of L1's activation when it is kept and eight of L2.
            constexpr int S = (WRITE_A1 ? 3 : 0) + 8;
            if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
            __builtin_amdgcn_s_barrier();                               // also: every wave has left the previous tile's phases (a1p and the other patch buffer are free)
            if (XB == 2 && has_next) dma_x(nxt, (it + 1) & 1);
        } else {
            unsigned stage[C::ROUNDS][SW];
            load_regs(tile, stage);
            // the previous tile's L1 phase (the readers of xp) ended at its second barrier: xp is free
            const int iy0 = (oy0 - 1) * 2 - pad_lo, if0 = (ox0 * 2 - 4) * CIN;
#pragma unroll
            for (int k = 0; k < C::ROUNDS; ++k) {
                const int c = k * 256 + (int)threadIdx.x;
                const bool inside = chunk_inside(k, iy0, if0);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float raw;
                    if constexpr (IN_MODE == 1) raw = (float)((stage[k][0] >> (8 * e)) & 0xFFu);
                    else raw = __builtin_bit_cast(float, stage[k][e]);
                    v[e] = inside ? (raw - pre_sub) / pre_div : 0.f;
                }
                if (c < C::CHUNKS) *(f32x4 *)((float *)xp + 4 * c) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();                               // also: every wave has left the previous tile's L2 phase (a1p is free)
        }

        // ---- L1 on this wave's units of 16 patch pixels (two units' taps in flight: unrolled further the kernel spills)
        // branch-free bodies (masked lanes and the slots past the last unit write to a spare corner of the patch buffer / an out-of-range
        // offset), so that the scheduler can run two units' LDS reads, FMA chains and MFMAs against each other
#pragma unroll 2
        for (int j = 0; j < C::UPW; ++j) {
            const int u = wid + 4 * j;                                   // wave-uniform; slots >= UNITS: every lane masked
            const int p = 16 * u + i;
            const bool valid = p < PW * PW;
            const int pp = valid ? p : PW * PW - 1;
            const int pr = pp / PW, pc = pp - pr * PW;
            const float *xb = xp + 2 * pr * C::ROWF + (2 * pc + dx0) * CIN + cb;
            float dwv = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) dwv = fmaf(xb[ky * C::ROWF + kx * CIN], dwk1[ky * 3 + kx], dwv);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const float dr = round16<T>(dwv);
            f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw1a, dr, z4, 0, 0, 0);
            f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw1b, dr, z4, 0, 0, 0);
            acc0 += b1A; acc1 += b1B;
            const int gy = oy0 - 1 + pr, gx = ox0 - 1 + pc;
            const bool inmap = (unsigned)gy < (unsigned)H2 && (unsigned)gx < (unsigned)W2;
            u32x2 o0 = {relu_pk16(pack2<T>(acc0[0], acc0[1])), relu_pk16(pack2<T>(acc0[2], acc0[3]))};
            u32x2 o1 = {relu_pk16(pack2<T>(acc1[0], acc1[1])), relu_pk16(pack2<T>(acc1[2], acc1[3]))};
            if (!inmap) { o0 = u32x2{0u, 0u}; o1 = u32x2{0u, 0u}; }      // outside L1's map: L2's zero padding
            const int spare = PW * PW * (UBD_C * 2) + 8 * lane;          // 512 of the 832 bytes behind the last patch pixel
            *(u32x2 *)(a1p + (valid ? pp * (UBD_C * 2) + 8 * q : spare)) = o0;
            *(u32x2 *)(a1p + ((valid && q < 2) ? pp * (UBD_C * 2) + 32 + 8 * q : spare)) = o1;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if constexpr (PLAIN && XB == 1) { if (has_next) dma_x(nxt, 0); }   // one patch buffer: free now, filled during the L2 phase

        if constexpr (WRITE_A1) {
            // the tile's own 16 x 16 pixels of L1's activation leave for memory: patch rows 1..16, columns 1..16 = 768 contiguous
            // bytes per row in LDS and in the map; three 16-byte pieces per thread (masked pieces: out-of-range offset)
            __amdgpu_buffer_rsrc_t a1rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a1out + (size_t)img * H2 * W2 * UBD_C), 0,
                                                                            (int)((unsigned)H2 * (unsigned)W2 * (UBD_C * 2u)), 0x00020000);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = k * 256 + (int)threadIdx.x;
                const int row = c / 48, cc = c - row * 48;
                const u32x4 v = *(const u32x4 *)(a1p + ((row + 1) * PW + 1) * (UBD_C * 2) + cc * 16);
                const int gy = oy0 + row, gx = ox0 + cc / 3;
                const bool in = gy < H2 && gx < W2;
                __builtin_amdgcn_raw_buffer_store_b128(v, a1rs, in ? (gy * W2 + ox0) * (UBD_C * 2) + cc * 16 : (int)0x80000000u, 0, 0);
            }
        }

        // ---- L2 on this wave's four tile rows
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = wid * 4 + k;
            const int oy = oy0 + r;
            const char *rowb = a1p + r * (PW * UBD_C * 2);
            u32x4 b0[5], b1[3];
#pragma unroll
            for (int j = 0; j < 5; ++j) b0[j] = *(const u32x4 *)(rowb + xo0[j]);
#pragma unroll
            for (int j = 0; j < 3; ++j) b1[j] = *(const u32x4 *)(rowb + xo1[j]);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 c0 = z4, c1 = z4;
#pragma unroll
            for (int j = 0; j < 5; ++j) c0 = h16<T>::mfma(wa0[j], b0[j], c0);
#pragma unroll
            for (int j = 0; j < 3; ++j) c1 = h16<T>::mfma(wa1[j], b1[j], c1);
            const u32x4 av = {pack2<T>(c0[0], c0[1]), pack2<T>(c0[2], c0[3]), pack2<T>(c1[0], c1[1]), 0u};
            const f32x4 acc0 = h16<T>::mfma(pwb[0], av, z4);
            const f32x4 acc1 = h16<T>::mfma(pwb[1], av, z4);
            int npx = W2 - ox0 < 16 ? W2 - ox0 : 16;
            npx = (oy < H2) ? npx : 0;
            const u32x2 nomask = {0u, 0u};
            store_tile16_t<T, 0>(y, ((size_t)img * H2 + (oy < H2 ? oy : 0)) * W2 + ox0, npx, lane, acc0, acc1, b2A, b2B, nomask, nomask);
        }
        if (!has_next) break;
        tile = nxt;
    }
}

#include "sep123_16.h"
#ifdef UBD_STAMPS
static unsigned long long *g_d16s_stamps = nullptr;
static int g_d16s_stamps_d = 0;
extern "C" void ubd_debug_set_stamps_d16s(void *p, int d) { g_d16s_stamps = (unsigned long long *)p; g_d16s_stamps_d = d; }
#define D16S_STAMP_ARG , (d == g_d16s_stamps_d ? g_d16s_stamps : nullptr)
static unsigned long long *g_s123_stamps = nullptr;
extern "C" void ubd_debug_set_stamps_s123(void *p) { g_s123_stamps = (unsigned long long *)p; }
#define S123_16_STAMP_ARG , g_s123_stamps
#else
#define S123_16_STAMP_ARG
#define D16S_STAMP_ARG
#endif

// ------------------------------------------------------------------------------------ dilated layers
struct a16_frags { u32x4 v[7]; u32x2 m0, m1; };

// EPI 0: y = relu(conv + bias).  EPI 1 (data gradient of the 16-bit train step): y = conv * (mask > 0), no bias; `wfrag`
// then holds the flipped / transposed kernel and `mask` the saved output of the layer below (same shape as y).
// n / d for the (wave-uniform) tile bookkeeping: m = ceil(2^32 / d) computed on the host, exact while n * d < 2^32
// (checked by the launcher); d == 1 has no 32-bit magic number.
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned d, unsigned m) { return d == 1u ? n : __umulhi(n, m); }

struct dil16_tile { int xt, rowid, yy, img; };

template <typename T, int EPI>
__global__ __launch_bounds__(256) void dilconv16_kernel(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                        const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                                                        const unsigned short *__restrict__ mask, int n, int h,
                                                        int w, int d, unsigned mg_tx, unsigned mg_h, float *__restrict__ logits3)
{
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    u32x4 wr[7][2];
#pragma unroll
    for (int c = 0; c < 7; ++c) { wr[c][0] = wfrag[(c * 2 + 0) * 64 + lane]; wr[c][1] = wfrag[(c * 2 + 1) * 64 + lane]; }
    f32x4 bA = {0.f, 0.f, 0.f, 0.f}, bB = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != 1) {
        bA = *(const f32x4 *)(bias + 4 * q);
        if (q < 2) bB = *(const f32x4 *)(bias + 16 + 4 * q);
    }
    // EPI 3 (train step, one output channel): as EPI 2, but the activation is stored as well (the backward pass reads it)
    // and the logits go to `logits3`.
    // EPI 2 (last hidden layer of an inference pass with one output channel): `mask` carries the fp32 head (24 weights +
    // bias, net.py:308-311), `y` the fp32 logits; the activation is rounded to T as if it had been stored, never written
    f32x4 hA = {0.f, 0.f, 0.f, 0.f}, hB = {0.f, 0.f, 0.f, 0.f};
    float hbias = 0.f;
    if constexpr (EPI == 2 || EPI == 3) {
        const float *head = (const float *)mask;
        hA = *(const f32x4 *)(head + 4 * q);
        if (q < 2) hB = *(const f32x4 *)(head + 16 + 4 * q);
        hbias = head[UBD_C];
    }
    // this lane's K-slice of chunk c: k0 = 32c + 8q -> tap (4c+q)/3, channel group ((4c+q)%3)*8.
    // delta[c]: byte offset of that tap / channel group relative to the centre pixel inside ONE image (the loads go through
    // a per-image buffer descriptor, so rows above / below the image fall out of range by themselves and read zeros =
    // the 'same' padding); dxc[c]: column shift, checked per lane.  Chunk 6, q = 3 lies beyond K: always out of range.
    int delta[7], dxc[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int g = 4 * c + q, t = g / 3;
        const int dy = (t / 3 - 1) * d, dx = (t % 3 - 1) * d;
        delta[c] = t < 9 ? (dy * w + dx) * (UBD_C * 2) + (g - 3 * t) * 16 : (1 << 30);
        dxc[c] = t < 9 ? dx : 0;
    }
    const int i48 = i * (UBD_C * 2);
    const unsigned tiles_x = (unsigned)(w + 15) >> 4;
    const int total = n * h * (int)tiles_x;
    const int xcd = blockIdx.x & 7;
    const int nblk_x = (gridDim.x + 7 - xcd) >> 3;
    const int chunk = (total + 7) >> 3;
    const int t_begin = xcd * chunk;
    const int t_end = (t_begin + chunk < total) ? t_begin + chunk : total;
    const int stride = nblk_x * 4;
    int tile = t_begin + (int)(blockIdx.x >> 3) * 4 + wid;
    if (tile >= t_end) return;
    const unsigned img_bytes = (unsigned)h * (unsigned)w * (unsigned)(UBD_C * 2);

    auto load = [&](a16_frags &a, dil16_tile &tc, int tl) {
        tc.rowid = (int)udiv_magic((unsigned)tl, tiles_x, mg_tx);
        tc.xt = tl - tc.rowid * (int)tiles_x;
        tc.img = (int)udiv_magic((unsigned)tc.rowid, (unsigned)h, mg_h);
        tc.yy = tc.rowid - tc.img * h;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)tc.img * h * w * UBD_C), 0, (int)img_bytes, 0x00020000);
        const int px = tc.xt * 16 + i;
        // byte offset of the centre pixel: wave-uniform part in scalar registers + the lane's constant i * 48 (the kernel is bound by
        // vector-instruction issue -- 85 VALU + 14 MFMA per 16-pixel tile in the PMC counters --, not by memory)
        const int sbase = (tc.yy * w + tc.xt * 16) * (UBD_C * 2);
        const int base = sbase + i48;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const bool xok = (unsigned)(px + dxc[c]) < (unsigned)w;
            a.v[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, xok ? base + delta[c] : (1 << 30), 0, 0);
        }
        if constexpr (EPI == 1) {                    // the mask words of the bytes this lane will store
            __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(mask + (size_t)tc.img * h * w * UBD_C), 0, (int)img_bytes, 0x00020000);
            const int moff = px < w ? base + 8 * q : (1 << 30);
            a.m0 = __builtin_amdgcn_raw_buffer_load_b64(mrsrc, moff, 0, 0);
            a.m1 = __builtin_amdgcn_raw_buffer_load_b64(mrsrc, q < 2 ? moff + 32 : (1 << 30), 0, 0);
        }
    };
    auto compute_store = [&](const a16_frags &a, const dil16_tile &tc) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            acc0 = h16<T>::mfma(wr[c][0], a.v[c], acc0);          // weights as the A operand: D = [channel][pixel]
            acc1 = h16<T>::mfma(wr[c][1], a.v[c], acc1);
        }
        const int x0 = tc.xt * 16;
        const int npx = w - x0 < 16 ? w - x0 : 16;
        if constexpr (EPI == 2 || EPI == 3) {
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(round16<T>(fmaxf(acc0[r] + bA[r], 0.f)), hA[r], part);
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(round16<T>(fmaxf(acc1[r] + bB[r], 0.f)), hB[r], part);   // hB = 0 for q >= 2
            part += __shfl_xor(part, 16, 64);                           // sum over the four channel quarters
            part += __shfl_xor(part, 32, 64);
            float *lg = EPI == 2 ? (float *)y : logits3;
            if (q == 0 && i < npx) lg[(size_t)tc.rowid * w + x0 + i] = part + hbias;
        }
        if constexpr (EPI == 3) {
            const u32x2 nomask = {0u, 0u};
            store_tile16_t<T, 0>(y, (size_t)tc.rowid * w + x0, npx, lane, acc0, acc1, bA, bB, nomask, nomask);
        } else if constexpr (EPI != 2) {
            u32x2 m0 = {0u, 0u}, m1 = {0u, 0u};
            if constexpr (EPI == 1) { m0 = a.m0; m1 = a.m1; }
            store_tile16_t<T, EPI>(y, (size_t)tc.rowid * w + x0, npx, lane, acc0, acc1, bA, bB, m0, m1);
        }
    };
    a16_frags A0, A1;
    dil16_tile T0, T1;
    const int t_last = t_end - 1;
    load(A0, T0, tile);
    for (;;) {
        int nxt = tile + stride;
        load(A1, T1, nxt < t_last ? nxt : t_last);
        compute_store(A0, T0);
        tile = nxt;
        if (tile >= t_end) break;
        nxt = tile + stride;
        load(A0, T0, nxt < t_last ? nxt : t_last);
        compute_store(A1, T1);
        tile = nxt;
        if (tile >= t_end) break;
    }
}

// ------------------------------------------------------------------------------------ dilated layer, staged in LDS
// The same product on tiles of the dilation sub-grids (items = (image, phase of the d x d sub-grid, 16 x 16 tile), as the bf16
// weight-gradient kernel): the 18 x 18 input tile arrives by descriptor-addressed LDS-DMA (ubd_blds16: rows / columns outside the
// image land as zeros = the 'same' padding), double-buffered one item ahead; a wave takes four tile rows, per row seven
// ds_read_b128 (A fragments at the tap offsets, constant per lane) + 14 MFMAs + the epilogue.  The direct kernel above spends
// 159 instructions per 16 pixels (PMC: waves wait for an issue slot half of their time), 28 of them on per-load column checks
// and most of the scalar ones on per-tile bookkeeping; here the bookkeeping is per 256 pixels.  EPI 0 (bias + ReLU) only.
// geometry of a launch, worked out once on the host: five 32-bit divisions (~40 scalar instructions each) less in every block's prologue
struct d16s_geom { int tiles_x, tiles_y, items; unsigned m_tx, m_ty, m_d; };
static d16s_geom d16s_geometry(int n, int h, int w, int d)
{
    d16s_geom g;
    const int sh = (h + d - 1) / d, sw = (w + d - 1) / d;
    g.tiles_y = (sh + 15) >> 4; g.tiles_x = (sw + 15) >> 4;
    g.items = n * d * d * g.tiles_y * g.tiles_x;
    auto magic = [](unsigned dv) { return dv == 1u ? 0u : 0xFFFFFFFFu / dv + 1u; };       // ceil(2^32 / dv): floor(a / dv) = umulhi(a, m) while a * dv < 2^32
    g.m_tx = magic((unsigned)g.tiles_x); g.m_ty = magic((unsigned)g.tiles_y); g.m_d = magic((unsigned)d);
    return g;
}
#define D16S_PIX (18 * 18)
#define D16S_PIECES ((D16S_PIX * 3 + 63) / 64)               // 16 pieces of 1 KiB (the last one runs past the tile)
#define D16S_BUF (D16S_PIECES * 1024)
template <typename T, int EPI = 0>
#ifndef D16S_OCC
#define D16S_OCC 3
#endif
#ifndef D16S_UNROLL
#define D16S_UNROLL 2
#endif
__global__ __launch_bounds__(256, D16S_OCC) void dilconv16s_kernel(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                            const u32x4 *__restrict__ wfrag, const float *__restrict__ bias, int n, int h,
                                                            int w, int d, const float *__restrict__ head, float *__restrict__ logits, const d16s_geom geo
#ifdef UBD_STAMPS
                                                            , unsigned long long *__restrict__ stamps
#endif
                                                            )
{
#ifdef UBD_STAMPS   // diagnostic build only: s_memtime of lane 0 of every wave at the phase boundaries of its first 8 items (tools/stamps_d16s.py)
#define D16STAMP(k) do { if (stamps && iter < 8 && (threadIdx.x & 63) == 0 && blockIdx.x < 768) stamps[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + iter) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    // block time line: [5] entry, [6] in front of the item loop, [7] behind it (s_memtime), [5..7] of item slot 1: the same three points on the 100-MHz clock all CUs share
#define D16BLK(k) do { if (stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 768) { const size_t b_ = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 * 8; stamps[b_ + (k)] = __builtin_amdgcn_s_memtime(); stamps[b_ + 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define D16STAMP(k) do {} while (0)
#define D16BLK(k) do {} while (0)
#endif
    D16BLK(5);
    // EPI 0: y = relu(conv + bias).  EPI 2 / 3 (last hidden layer with ONE output channel, round 4: the staged kernel takes L9 too): the
    // 1 x 1 head (24 fp32 weights + bias at `head`, net.py:308-311) is applied in the epilogue, fp32 logits to `logits`; EPI 2 (inference)
    // does not store the activation, EPI 3 (train step) does.  Same expressions as dilconv16_kernel<T, 2 / 3>: bit-identical logits.
    __shared__ __attribute__((aligned(16))) char smem[2 * D16S_BUF + 64];     // ONE LDS object; [2 BUF, +64): zeros
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    if (threadIdx.x < 16) ((unsigned *)(smem + 2 * D16S_BUF))[threadIdx.x] = 0u;
    u32x4 wr[7][2];
#pragma unroll
    for (int c = 0; c < 7; ++c) { wr[c][0] = wfrag[(c * 2 + 0) * 64 + lane]; wr[c][1] = wfrag[(c * 2 + 1) * 64 + lane]; }
    const f32x4 bA = *(const f32x4 *)(bias + 4 * q);
    const f32x4 bB = q < 2 ? *(const f32x4 *)(bias + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 hA = {0.f, 0.f, 0.f, 0.f}, hB = {0.f, 0.f, 0.f, 0.f};
    float hbias = 0.f;
    if constexpr (EPI == 2 || EPI == 3) {
        hA = *(const f32x4 *)(head + 4 * q);
        if (q < 2) hB = *(const f32x4 *)(head + 16 + 4 * q);
        hbias = head[UBD_C];
    }
    // this lane's K-slice of chunk c: k0 = 32c + 8q -> tap (4c + q) / 3, channel group (4c + q) % 3; chunk 6, q = 3 lies beyond K
    int doff[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int g = 4 * c + q, t = g / 3;
        doff[c] = t < 9 ? ((t / 3 - 1) * 18 + (t % 3 - 1)) * (UBD_C * 2) + (g - 3 * t) * 16 : -(1 << 20);
    }
    // DMA chunk -> (tile row, tile column, 16-byte part) of this lane, per piece of this wave (piece = 4 rd + wid)
    int cinfo[D16S_PIECES / 4];
#pragma unroll
    for (int rd = 0; rd < D16S_PIECES / 4; ++rd) {
        int c = (rd * 4 + wid) * 64 + lane;
        c = c < D16S_PIX * 3 ? c : D16S_PIX * 3 - 1;
        const int pix = c / 3, sy = pix / 18;
        cinfo[rd] = sy | ((pix - sy * 18) << 8) | ((c - pix * 3) << 16);
    }
    const int tiles_y = geo.tiles_y, tiles_x = geo.tiles_x, items = geo.items;     // from the host (d16s_geometry; in the kernel they were 64-bit division loops first, then five 32-bit divisions)
    struct item_t { int img, ry, rx, sy0, sx0; };
    const unsigned m_tx = geo.m_tx, m_ty = geo.m_ty, m_d = geo.m_d;
    auto divm = [](unsigned a, unsigned dv, unsigned m) { return dv == 1u ? a : __umulhi(a, m); };
    auto decode = [&](int it) {
        item_t r;
        unsigned a = (unsigned)it, b;
        b = divm(a, (unsigned)tiles_x, m_tx); const int tx = (int)(a - b * (unsigned)tiles_x); a = b;
        b = divm(a, (unsigned)tiles_y, m_ty); const int ty = (int)(a - b * (unsigned)tiles_y); a = b;
        b = divm(a, (unsigned)d, m_d); r.rx = (int)(a - b * (unsigned)d); a = b;
        b = divm(a, (unsigned)d, m_d); r.ry = (int)(a - b * (unsigned)d);
        r.img = (int)b;
        r.sy0 = ty * 16; r.sx0 = tx * 16;
        return r;
    };
    const unsigned lds_smem = ubd_lds_addr(smem);
    const unsigned img_bytes = (unsigned)h * (unsigned)w * (UBD_C * 2);
    auto dma_item = [&](const item_t &I, int bufoff) {
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)x + (size_t)I.img * h * w * (UBD_C * 2)), 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int rd = 0; rd < D16S_PIECES / 4; ++rd) {
            const int ci = cinfo[rd];
            const int gy = I.ry + (I.sy0 + (ci & 0xFF) - 1) * d;      // < 0 or >= h: the offset leaves the descriptor's range
            const int gx = I.rx + (I.sx0 + ((ci >> 8) & 0xFF) - 1) * d;
            const unsigned off = (unsigned)gx < (unsigned)w ? (unsigned)((gy * w + gx) * (UBD_C * 2)) + (unsigned)((ci >> 16) & 0xFF) * 16u : 0x80000000u;
            ubd_blds16(rx, off, lds_smem + bufoff + (rd * 4 + wid) * 1024);
        }
    };
    // XCD-aware item ranges (the d x d phases of an image interleave inside the same cache lines: one L2 per image)
    const int xcd = blockIdx.x & 7;
    const int nblk_x = ((int)gridDim.x + 7 - xcd) >> 3;
    const int chunk = (items + 7) >> 3;
    const int it_begin = xcd * chunk;
    const int it_end = it_begin + chunk < items ? it_begin + chunk : items;
    int it = it_begin + (int)(blockIdx.x >> 3);
    item_t I = decode(it < it_end ? it : it_begin);
    if (it < it_end) dma_item(I, 0);
    const char *zero16 = smem + 2 * D16S_BUF;
    D16BLK(6);
    for (int iter = 0; it < it_end; ++iter, it += nblk_x) {
        const char *buf = smem + (iter & 1) * D16S_BUF;
        // this item's tile has landed; the eight output stores of the previous item (younger than its DMA) stay in flight
        // (per row: two activation stores, EPI 2 / 3 one logit store)
        constexpr int NST = 4 * (EPI == 2 ? 1 : (EPI == 3 ? 3 : 2));
        D16STAMP(0);
        if (iter > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        D16STAMP(1);
        __syncthreads();                              // ... for every wave; everyone left the other buffer
        D16STAMP(2);
        item_t Inext = I;
        if (it + nblk_x < it_end) { Inext = decode(it + nblk_x); dma_item(Inext, ((iter + 1) & 1) * D16S_BUF); }
        D16STAMP(3);
        __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)y + (size_t)I.img * h * w * (UBD_C * 2)), 0, (int)img_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rlog = __builtin_amdgcn_make_buffer_rsrc((void *)((EPI == 2 || EPI == 3) ? (char *)(logits + (size_t)I.img * h * w) : (char *)y), 0, (int)((unsigned)h * (unsigned)w * 4u), 0x00020000);
#pragma unroll D16S_UNROLL
        for (int rr = 0; rr < 4; ++rr) {              // fixed trip count: rows outside the sub-grid only lose their stores
            const int r = 4 * wid + rr;
            const char *gpix = buf + ((r + 1) * 18 + i + 1) * (UBD_C * 2);
            u32x4 a[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) a[c] = *(const u32x4 *)(doff[c] > -(1 << 19) ? gpix + doff[c] : zero16);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 7; ++c) {             // weights as the A operand: D = [channel][pixel]
                acc0 = h16<T>::mfma(wr[c][0], a[c], acc0);
                acc1 = h16<T>::mfma(wr[c][1], a[c], acc1);
            }
            const int gy = I.ry + (I.sy0 + r) * d, gx = I.rx + (I.sx0 + i) * d;
            if constexpr (EPI == 2 || EPI == 3) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(round16<T>(fmaxf(acc0[e] + bA[e], 0.f)), hA[e], part);
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(round16<T>(fmaxf(acc1[e] + bB[e], 0.f)), hB[e], part);   // hB = 0 for q >= 2
                part += __shfl_xor(part, 16, 64);                       // sum over the four channel quarters
                part += __shfl_xor(part, 32, 64);
                // an unconditional buffer store (out-of-range offset for the lanes that have nothing to write): the count of stores per item is what
                // the counted vmcnt wait at the top of the next item relies on
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part + hbias), rlog, (q == 0 && gx < w && gy < h) ? (gy * w + gx) * 4 : (int)0x80000000u, 0, 0);
            }
            if constexpr (EPI != 2) {
            acc0 += bA; acc1 += bB;                   // bias added last (the order the oracle uses), ReLU on the packed values
            const u32x2 o0 = {relu_pk16(pack2<T>(acc0[0], acc0[1])), relu_pk16(pack2<T>(acc0[2], acc0[3]))};
            const u32x2 o1 = {relu_pk16(pack2<T>(acc1[0], acc1[1])), relu_pk16(pack2<T>(acc1[2], acc1[3]))};
            const unsigned off = (gx < w && gy < h) ? (unsigned)((gy * w + gx) * (UBD_C * 2)) + 8u * q : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b64(o0, rout, (int)off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(o1, rout, (int)(q < 2 ? off + 32u : 0x80000000u), 0, 0);   // channels 16 + 4q + r exist for q < 2
            }
        }
        D16STAMP(4);
        I = Inext;
    }
    D16BLK(7);
#undef D16STAMP
#undef D16BLK
}

// ------------------------------------------------------------------------------------ head
template <typename T>
__global__ __launch_bounds__(256) void head16_kernel(const unsigned short *__restrict__ x, float *__restrict__ logits,
                                                     const float *__restrict__ hk, const float *__restrict__ hb, long npix, int k_out)
{
    __shared__ float s_k[UBD_C * (UBD_MAX_CLASSES + 1)];
    __shared__ float s_b[UBD_MAX_CLASSES + 1];
    for (int t = threadIdx.x; t < UBD_C * k_out; t += blockDim.x) s_k[t] = hk[t];
    for (int t = threadIdx.x; t < k_out; t += blockDim.x) s_b[t] = hb[t];
    __syncthreads();
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const u32x4 *px = (const u32x4 *)(x + p * UBD_C);
        float v[UBD_C];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const u32x4 t = px[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) widen2<T>(t[e], v[c * 8 + 2 * e], v[c * 8 + 2 * e + 1]);
        }
        for (int ko = 0; ko < k_out; ++ko) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < UBD_C; ++c) acc = fmaf(v[c], s_k[c * k_out + ko], acc);
            logits[p * k_out + ko] = acc + s_b[ko];
        }
    }
}

// ------------------------------------------------------------------------------------ host
void ubd_fwd16_layout_compute(int n, int H, int W, int training, ubd_fwd16_layout *L)
{
    size_t off = 0;
    L->off_wfrag32 = off; off += ubd_align_up((size_t)UBD_FWD_DIRECT_FLOATS * sizeof(float), 256);
    L->off_wfrag16 = off; off += ubd_align_up(((size_t)UBD_NUM_DIL * UBD_DIL16_FRAG_U32 + UBD_SEP16_READY_U32) * sizeof(unsigned), 256);   // + ready operands of L2 / L3
    const size_t a = ubd_align_up((size_t)n * (H / 2) * (W / 2) * UBD_C * 2, 256);
    const size_t b = ubd_align_up((size_t)n * (H / 4) * (W / 4) * UBD_C * 2, 256);
    L->off_a1 = off; off += a;
    L->off_a2 = off; off += a;
    if (training) {
        for (int k = 0; k < 7; ++k) { L->off_acts[k] = off; off += b; }
    } else {
        const size_t b0 = off, b1 = off + b;
        off += 2 * b;
        for (int k = 0; k < 7; ++k) L->off_acts[k] = (k & 1) ? b1 : b0;
    }
    L->total = off;
}

int ubd_pack16_workspace(ubd_handle *h, const float *params, char *ws, size_t ws_bytes, hipStream_t st)
{
    ubd_fwd16_layout L;
    ubd_fwd16_layout_compute(1, 4, 4, 0, &L);
    UBD_REQUIRE(ws_bytes >= L.off_a1, "ubd_pack_weights: workspace too small");
    ubd_launch_pack_direct(h, params, (float *)(ws + L.off_wfrag32), st);
    ubd_launch_pack16(h, params, (unsigned *)(ws + L.off_wfrag16), 0, st);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t ubd_forward16_workspace_bytes(int n, int H, int W)
{
    ubd_fwd16_layout L;
    ubd_fwd16_layout_compute(n, H, W, 0, &L);
    return L.total;
}

template <int CIN, int STRIDE, int IN_MODE, typename T>
static void launch_sep16(const ubd_handle *h, const void *x, unsigned short *y, const float *frag, const float *bias, int n,
                         int H, int W, int OH, int OW, int pad_lo, float sub, float div, hipStream_t st)
{
    using C = sep16_cfg<CIN, STRIDE>;
    const long tiles = (long)n * ((OH + C::TH - 1) / C::TH) * ((OW + 15) / 16);
    const int per_cu = (CIN == UBD_C) ? 3 : 5;                 // LDS-limited residency
    long grid = (long)h->num_cus * per_cu;
    if (grid > tiles || CIN != UBD_C) grid = tiles;      // 1/3 channels: one tile per block
    hipLaunchKernelGGL((sepconv16_kernel<CIN, STRIDE, IN_MODE, T>), dim3(grid), dim3(256), 0, st, x, y, frag, bias, n, H, W, OH, OW, pad_lo, sub, div);
}

template <int CIN, int IN_MODE, typename T>
static void launch_sep12(const ubd_handle *h, const void *x, unsigned short *a1, unsigned short *a2, const float *frag1, const float *bias1,
                         const float *frag2, const float *bias2, int n, int H, int W, int pad_lo, float sub, float div, bool write_a1,
                         hipStream_t st)
{
    const int H2 = H / 2, W2 = W / 2;
    const long tiles = (long)n * ((H2 + 15) / 16) * ((W2 + 15) / 16);
    long grid = (long)h->num_cus * 3;                          // three blocks per CU (registers; LDS 50 KiB with two patch buffers)
    if (grid > tiles) grid = tiles;
    long grid4 = (long)h->num_cus * 4;
    if (grid4 > tiles) grid4 = tiles;
    // inference: one patch buffer and four blocks per CU (34 KiB, 120 registers); train step: two buffers, three blocks (0.172 -> 0.169 ms for
    // cfg5 / 1.200 -> 1.192 ms for the bf16 train step against the other assignment)
    int xb = write_a1 ? 2 : 1;
#define UBD_SEP12_LAUNCH(PLAIN, WR)                                                                                                   \
    do { if (xb == 1) hipLaunchKernelGGL((sep12_16_kernel<CIN, IN_MODE, PLAIN, WR, T, 1>), dim3(grid4), dim3(256), 0, st, x, a1, a2, frag1, bias1, frag2,    \
                       bias2, n, H, W, H2, W2, pad_lo, sub, div); else hipLaunchKernelGGL((sep12_16_kernel<CIN, IN_MODE, PLAIN, WR, T, 2>), dim3(grid), dim3(256), 0, st, x, a1, a2, frag1, bias1, frag2,    \
                       bias2, n, H, W, H2, W2, pad_lo, sub, div); } while (0)
    // fp32 pixels fed as they are, offsets inside one image below 2^31: LDS-DMA
    const bool plain = IN_MODE == 0 && sub == 0.f && div == 1.f && (size_t)H * W * CIN * 4 < (1ull << 31) && ((uintptr_t)x & 15) == 0;   // 16-byte DMA pieces: an offset view of a tensor takes the register-staged variant
    if constexpr (IN_MODE == 0) {
        if (plain) {
            if (write_a1) UBD_SEP12_LAUNCH(true, true); else UBD_SEP12_LAUNCH(true, false);
            return;
        }
    }
    xb = 2;                                                    // register-staged input: the three-blocks-per-CU build (128 registers would spill); one patch buffer either way
    if (write_a1) UBD_SEP12_LAUNCH(false, true); else UBD_SEP12_LAUNCH(false, false);
#undef UBD_SEP12_LAUNCH
}

template <int CIN, int IN_MODE, typename T>
static void launch_sep123(const ubd_handle *h, const void *x, unsigned short *a1, unsigned short *a2, unsigned short *a3, const float *frag1,
                          const float *bias1, const unsigned *ready23, const float *bias2, const float *bias3, int n, int H, int W, int pad_lo,
                          float sub, float div, bool write_a12, hipStream_t st)
{
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const long tiles = (long)n * ((H4 + 7) / 8) * ((W4 + 7) / 8);
    long grid = (long)h->num_cus * 3;                          // three blocks per CU (54 KB of LDS each)
    if (grid > tiles) grid = tiles;
#define UBD_SEP123_LAUNCH(PLAIN, WR)                                                                                                        \
    hipLaunchKernelGGL((sep123_16_kernel<CIN, IN_MODE, PLAIN, WR, T>), dim3(grid), dim3(256), 0, st, x, a1, a2, a3, frag1, bias1, (const u32x4 *)ready23, \
                       bias2, bias3, n, H, W, H2, W2, H4, W4, pad_lo, sub, div S123_16_STAMP_ARG)
    // fp32 pixels fed as they are, offsets inside one image below 2^31: LDS-DMA
    const bool plain = IN_MODE == 0 && sub == 0.f && div == 1.f && (size_t)H * W * CIN * 4 < (1ull << 31) && ((uintptr_t)x & 15) == 0;   // 16-byte DMA pieces: an offset view of a tensor takes the register-staged variant
    if constexpr (IN_MODE == 0) {
        if (plain) {
            if (write_a12) UBD_SEP123_LAUNCH(true, true); else UBD_SEP123_LAUNCH(true, false);
            return;
        }
    }
    if (write_a12) UBD_SEP123_LAUNCH(false, true); else UBD_SEP123_LAUNCH(false, false);
#undef UBD_SEP123_LAUNCH
}

static unsigned magic_u32(unsigned d) { return d <= 1u ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

template <typename T>
static void launch_dil16(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, const void *mask, int d,
                         const void *in, void *out, int n, int H4, int W4, hipStream_t st, float *logits3)
{
    const unsigned tiles_x = (unsigned)(W4 + 15) / 16;
    const long tiles = (long)n * H4 * tiles_x;
    int grid = ubd_grid_for(tiles, h->num_cus, 4, 4);
    grid = (grid + 7) / 8 * 8;
    const unsigned mg_tx = magic_u32(tiles_x), mg_h = magic_u32((unsigned)H4);
    // forward layers whose dilation sub-grids are wider than 8 pixels: the LDS-staged kernel (UBD_DILCONV16=direct keeps the direct one)
    const int sw = (W4 + d - 1) / d, sh = (H4 + d - 1) / d;
    if ((epi == 0 || epi == 2 || epi == 3) && sw > 8 && !h->direct_dil16) {
        const long items = (long)n * d * d * ((sh + 15) / 16) * ((sw + 15) / 16);
        int g2 = h->num_cus * D16S_OCC;
        const d16s_geom geo = d16s_geometry(n, H4, W4, d);
        if (g2 > items) g2 = (int)items;
        g2 = (g2 + 7) / 8 * 8;                                     // the item ranges are cut per XCD: all eight need a block
        if (epi == 0)
            hipLaunchKernelGGL((dilconv16s_kernel<T, 0>), dim3(g2), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                               (const u32x4 *)frag, bias, n, H4, W4, d, (const float *)nullptr, (float *)nullptr, geo D16S_STAMP_ARG);
        else if (epi == 2)      // out = fp32 logits, mask = fp32 head: the activation is not stored
            hipLaunchKernelGGL((dilconv16s_kernel<T, 2>), dim3(g2), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)nullptr,
                               (const u32x4 *)frag, bias, n, H4, W4, d, (const float *)mask, (float *)out, geo D16S_STAMP_ARG);
        else                    // out = activation, logits3 = fp32 logits
            hipLaunchKernelGGL((dilconv16s_kernel<T, 3>), dim3(g2), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                               (const u32x4 *)frag, bias, n, H4, W4, d, (const float *)mask, logits3, geo D16S_STAMP_ARG);
    } else if (epi == 0)
        hipLaunchKernelGGL((dilconv16_kernel<T, 0>), dim3(grid), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                           (const u32x4 *)frag, bias, (const unsigned short *)nullptr, n, H4, W4, d, mg_tx, mg_h, (float *)nullptr);
    else if (epi == 2)      // out = fp32 logits (n, H4, W4, 1), mask = fp32 head (24 weights + bias)
        hipLaunchKernelGGL((dilconv16_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                           (const u32x4 *)frag, bias, (const unsigned short *)mask, n, H4, W4, d, mg_tx, mg_h, (float *)nullptr);
    else if (epi == 3)      // out = activation, logits3 = fp32 logits, mask = fp32 head
        hipLaunchKernelGGL((dilconv16_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                           (const u32x4 *)frag, bias, (const unsigned short *)mask, n, H4, W4, d, mg_tx, mg_h, logits3);
    else
        hipLaunchKernelGGL((dilconv16_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)out,
                           (const u32x4 *)frag, bias, (const unsigned short *)mask, n, H4, W4, d, mg_tx, mg_h, (float *)nullptr);
}

// 16-bit dense dilated layer (epi 0: forward, epi 1: data gradient with ReLU mask); element type from the handle
void ubd_launch_dilconv16(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, const void *mask, int d,
                          const void *in, void *out, int n, int H4, int W4, hipStream_t st, float *logits3)
{
    if (h->cfg.dtype == UBD_BF16) launch_dil16<__bf16>(h, epi, frag, bias, mask, d, in, out, n, H4, W4, st, logits3);
    else launch_dil16<_Float16>(h, epi, frag, bias, mask, d, in, out, n, H4, W4, st, logits3);
}

// packed 16-bit fragments of all six dilated layers (transpose = 1: data-gradient kernels)
pack_sep16_args ubd_pack_sep16_args(const ubd_handle *h)
{
    pack_sep16_args sa;
    for (int l = 0; l < 2; ++l) { sa.off_dw[l] = h->off_sep_dw[1 + l]; sa.off_pw[l] = h->off_sep_pw[1 + l]; }
    return sa;
}

// transpose = 0: `out` is the forward fragment region (dilated fragments + the ready operands of L2 / L3 behind them)
void ubd_launch_pack16(const ubd_handle *h, const float *params, unsigned *out, int transpose, hipStream_t st)
{
    const pack_sep16_args sa = ubd_pack_sep16_args(h);
    if (h->cfg.dtype == UBD_BF16)
        hipLaunchKernelGGL((pack16_kernel<__bf16>), dim3(48), dim3(256), 0, st, params, out, h->off_dil_k[0], h->off_dil_k[1] - h->off_dil_k[0], transpose, sa, transpose == 0);
    else
        hipLaunchKernelGGL((pack16_kernel<_Float16>), dim3(48), dim3(256), 0, st, params, out, h->off_dil_k[0], h->off_dil_k[1] - h->off_dil_k[0], transpose, sa, transpose == 0);
}

template <typename T>
static int forward16_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n,
                          int H, int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st, bool inference)
{
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    float *wfrag = (float *)(ws + L.off_wfrag32);
    unsigned *wfrag16 = (unsigned *)(ws + L.off_wfrag16);
    unsigned short *a1 = (unsigned short *)(ws + L.off_a1), *a2 = (unsigned short *)(ws + L.off_a2);
    const bool prepacked = (in_dtype & UBD_IN_PREPACKED) != 0;
    in_dtype &= ~UBD_IN_PREPACKED;
    if (!prepacked) {
        ubd_launch_pack_direct(h, params, wfrag, st);                  // fp32 depthwise / pointwise fragments
        ubd_launch_pack16(h, params, wfrag16, 0, st);
    }
    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const float *sf0 = wfrag, *sf1 = wfrag + per_sep, *sf2 = wfrag + 2 * per_sep;
    const int pad_s2 = h->cfg.fml_compatible ? 1 : 0;
    float sub = 0.f, div = 1.f;
    if (preprocessing == UBD_PRE_MOBILENET) { sub = 127.5f; div = 127.5f; }
    const bool u8 = in_dtype == UBD_IN_U8;
    unsigned short *cur = (unsigned short *)(ws + L.off_acts[0]);
    if (h->split_stem16 == 0) {
        // L1 -> L2 -> L3 in one kernel (sep123_16.h): neither a1 nor a2 is read back; the train step keeps both for the backward pass
        const float *b0 = params + h->off_sep_b[0], *b1 = params + h->off_sep_b[1], *b2 = params + h->off_sep_b[2];
        const unsigned *ready23 = wfrag16 + (size_t)UBD_NUM_DIL * UBD_DIL16_FRAG_U32;
        if (h->cfg.c_in == 1) {
            if (u8) launch_sep123<1, 1, T>(h, images, a1, a2, cur, sf0, b0, ready23, b1, b2, n, H, W, pad_s2, sub, div, !inference, st);
            else launch_sep123<1, 0, T>(h, images, a1, a2, cur, sf0, b0, ready23, b1, b2, n, H, W, pad_s2, sub, div, !inference, st);
        } else {
            if (u8) launch_sep123<3, 1, T>(h, images, a1, a2, cur, sf0, b0, ready23, b1, b2, n, H, W, pad_s2, sub, div, !inference, st);
            else launch_sep123<3, 0, T>(h, images, a1, a2, cur, sf0, b0, ready23, b1, b2, n, H, W, pad_s2, sub, div, !inference, st);
        }
    } else if (h->split_stem16 == 1) {
        if (h->cfg.c_in == 1) {
            if (u8) launch_sep16<1, 2, 1, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
            else launch_sep16<1, 2, 0, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
        } else {
            if (u8) launch_sep16<3, 2, 1, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
            else launch_sep16<3, 2, 0, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
        }
        launch_sep16<UBD_C, 1, 2, T>(h, a1, a2, sf1, params + h->off_sep_b[1], n, H2, W2, H2, W2, 1, 0.f, 1.f, st);
    } else {
        // L1 -> L2 in one kernel; the train step keeps L1's activation (the backward pass reads it)
        const float *b0 = params + h->off_sep_b[0], *b1 = params + h->off_sep_b[1];
        if (h->cfg.c_in == 1) {
            if (u8) launch_sep12<1, 1, T>(h, images, a1, a2, sf0, b0, sf1, b1, n, H, W, pad_s2, sub, div, !inference, st);
            else launch_sep12<1, 0, T>(h, images, a1, a2, sf0, b0, sf1, b1, n, H, W, pad_s2, sub, div, !inference, st);
        } else {
            if (u8) launch_sep12<3, 1, T>(h, images, a1, a2, sf0, b0, sf1, b1, n, H, W, pad_s2, sub, div, !inference, st);
            else launch_sep12<3, 0, T>(h, images, a1, a2, sf0, b0, sf1, b1, n, H, W, pad_s2, sub, div, !inference, st);
        }
    }
    if (h->split_stem16 != 0)
        launch_sep16<UBD_C, 2, 2, T>(h, a2, cur, sf2, params + h->off_sep_b[2], n, H2, W2, H4, W4, pad_s2, 0.f, 1.f, st);
    // inference with a single output channel: the head rides in the epilogue of L9 and L9's activation is never written
    const bool fuse_head = inference && h->k_out == 1 && h->off_head_b == h->off_head_k + UBD_C;
    for (int k = 0; k < UBD_NUM_DIL; ++k) {
        unsigned short *nxt = (unsigned short *)(ws + L.off_acts[k + 1]);
        if (!inference && h->k_out == 1 && k == UBD_NUM_DIL - 1 && h->off_head_b == h->off_head_k + UBD_C) {
            // train step: L9 stores its activation (the backward pass reads it) and applies the head in the same epilogue
            ubd_launch_dilconv16(h, 3, wfrag16 + (size_t)k * UBD_DIL16_FRAG_U32, params + h->off_dil_b[k], params + h->off_head_k, UBD_DILATIONS[k], cur, nxt, n, H4, W4, st, logits);
            UBD_CHECK_HIP(hipGetLastError());
            return 0;
        }
        if (fuse_head && k == UBD_NUM_DIL - 1) {
            ubd_launch_dilconv16(h, 2, wfrag16 + (size_t)k * UBD_DIL16_FRAG_U32, params + h->off_dil_b[k], params + h->off_head_k, UBD_DILATIONS[k], cur, logits, n, H4, W4, st);
            UBD_CHECK_HIP(hipGetLastError());
            return 0;
        }
        ubd_launch_dilconv16(h, 0, wfrag16 + (size_t)k * UBD_DIL16_FRAG_U32, params + h->off_dil_b[k], nullptr, UBD_DILATIONS[k], cur, nxt, n, H4, W4, st);
        cur = nxt;
    }
    const long npix = (long)n * H4 * W4;
    int hgrid = (int)((npix + 255) / 256);
    if (hgrid > h->num_cus * 8) hgrid = h->num_cus * 8;
    hipLaunchKernelGGL((head16_kernel<T>), dim3(hgrid), dim3(256), 0, st, cur, logits, params + h->off_head_k, params + h->off_head_b, npix, h->k_out);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

int ubd_forward16_layout(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H,
                         int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st, bool inference)
{
    UBD_REQUIRE((size_t)n * (H / 4) * (W / 4) * UBD_C * 2 < 0xFFFFFFFFull, "ubd_forward: batch too large for 32-bit buffer offsets; split the batch");
    // dilconv16_kernel: per-image 30-bit byte offsets and magic-number division of the tile index
    UBD_REQUIRE((size_t)(H / 4) * (W / 4) * UBD_C * 2 < (1ull << 30), "ubd_forward: image too large for the 16-bit path (%d x %d)", H, W);
    UBD_REQUIRE((unsigned long long)n * (H / 4) * ((W / 4 + 15) / 16) * ((W / 4 + 15) / 16) < (1ull << 32) &&
                (unsigned long long)n * (H / 4) * (H / 4) < (1ull << 32), "ubd_forward: batch too large for the 16-bit path; split the batch");
    if (h->cfg.dtype == UBD_BF16) return forward16_impl<__bf16>(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st, inference);
    return forward16_impl<_Float16>(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st, inference);
}

int ubd_forward16(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H, int W,
                  float *logits, char *ws, size_t ws_bytes, hipStream_t st)
{
    UBD_REQUIRE(ws_bytes >= ubd_forward16_workspace_bytes(n, H, W), "ubd_forward: workspace too small for the 16-bit path");
    ubd_fwd16_layout L;
    ubd_fwd16_layout_compute(n, H, W, 0, &L);
    return ubd_forward16_layout(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st, true);
}
