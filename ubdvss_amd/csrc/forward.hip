// Forward pass of the ubdvss dilated FCN on gfx950 (MI355X), fp32 path.
//
// Reference semantics: semantic_segmentation/net.py:225-252 (conv_bn) and :278-314
// (_build_dilated_conv_model); SURVEY.md section 9.1.  NHWC activations, Keras-ordered
// flat fp32 parameters.
//
// Kernel plan (one launch per layer, every kernel MFMA-based, 64-wide waves):
//   pack_weights_kernel   flat params -> per-lane MFMA B-fragments (tiny, once per call)
//   sepconv_kernel        L1..L3: depthwise 3x3 on the VALU directly in MFMA A-operand
//                         layout (lane = (pixel i, k-quarter q)), pointwise 1x1 as
//                         v_mfma_f32_16x16x4_f32, bias + ReLU epilogue.  HBM-bound.
//   dilconv_kernel        L4..L9: implicit GEMM M = 16 pixels, N = 24 (padded to 2x16),
//                         K = 216, the whole layer's weights resident in 108 VGPRs per
//                         lane, A fragments loaded straight from global memory with
//                         buffer loads (hardware zero fill = 'same' padding), register
//                         double buffering across persistent tiles.  fp32-MFMA-bound.
//   head_kernel           1x1 conv 24 -> 1+n_classes, no activation.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------
// Packed weight fragments.
//  dilated layer L, tap t (ky*3+kx), sub-step j (0..5), N-tile nt (0..1), lane:
//     ci = j < 4 ? 4*q + j : 16 + 2*q + (j-4)      (q = lane>>4)
//     co = (lane & 15) + 16*nt                      (zero for co >= 24)
//     frag[((t*6 + j)*2 + nt)*64 + lane] = k[ky][kx][ci][co]
//  separable layer s: channel of (lane, step): CIN==24 ? 6*q + step : (step==0 && q<CIN ? q : none)
//     pwfrag[(step*2 + nt)*64 + lane] = pw[ch][co]
//     dwlane[(tap*6 + step)*64 + lane] = dw[tap][ch]
// ------------------------------------------------------------------------------------
#include "pack.h"
__global__ void pack_weights_kernel(const float *__restrict__ params, float *__restrict__ wfrag, pack_args a)
{
    pack_weights_body(params, wfrag, a, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// ------------------------------------------------------------------------------------
// Epilogue shared by sepconv and dilconv: D layout of v_mfma_f32_16x16x4_f32 is
// col = lane & 15 (output channel), row = 4*(lane>>4) + reg (pixel in the 16-pixel tile).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void store_tile_relu(float *__restrict__ y, size_t row_base_elems, int x0, int ow,
                                                int lane, f32x4 acc0, f32x4 acc1, float b0, float b1)
{
    // Branch-free: buffer stores whose per-lane offset is pushed out of range for lanes that must not write.
    const int co = lane & 15, q = lane >> 4;
    // wave-uniform tile base; readfirstlane makes the uniformity provable, otherwise hipcc wraps every buffer
    // store in a waterfall loop (cdna_hip_programming.md T20)
    const unsigned long long rp = (unsigned long long)(y + (row_base_elems + (size_t)x0) * UBD_C);
    const unsigned rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rp);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rp >> 32));
    float *rowp = (float *)(((unsigned long long)rhi << 32) | rlo);
    int npx = ow - x0 < 16 ? ow - x0 : 16;                                    // valid pixels in this tile
    npx = npx < 0 ? 0 : npx;
    const unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane(npx * UBD_C * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, (int)bytes, 0x00020000);
    const unsigned base = (unsigned)(4 * q) * (UBD_C * 4u) + (unsigned)co * 4u;
    const unsigned base1 = co < 8 ? base + 64u : 0x40000000u;                // channels 16..23 only from lanes co < 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc0[r] + b0, 0.f)), rs, (int)(base + r * 96u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc1[r] + b1, 0.f)), rs, (int)(base1 + r * 96u), 0, 0);
    }
}

// Epilogue for products issued as D = W^T . X^T (A operand = weights, B operand = activations): the D layout then is
// col = lane & 15 = PIXEL of the 16-pixel tile, row = 4*(lane>>4) + reg = OUTPUT CHANNEL, so every lane holds four
// consecutive channels of its pixel and the tile leaves as two 16-byte buffer stores per lane (channels 4q..4q+3, and
// 16+4q..19+4q from the lanes q < 2) instead of eight scattered dword stores.  bA / bB: the matching bias vectors.
__device__ __forceinline__ void store_tile_relu_t(float *__restrict__ y, size_t row_base_elems, int x0, int ow,
                                                  int lane, f32x4 acc0, f32x4 acc1, f32x4 bA, f32x4 bB)
{
    const int i = lane & 15, q = lane >> 4;
    const unsigned long long rp = (unsigned long long)(y + (row_base_elems + (size_t)x0) * UBD_C);
    const unsigned rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rp);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rp >> 32));
    float *rowp = (float *)(((unsigned long long)rhi << 32) | rlo);
    int npx = ow - x0 < 16 ? ow - x0 : 16;                                    // valid pixels in this tile
    npx = npx < 0 ? 0 : npx;
    const unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane(npx * UBD_C * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, (int)bytes, 0x00020000);
    const unsigned base = (unsigned)i * (UBD_C * 4u) + 16u * (unsigned)q;     // beyond `bytes` for pixels >= npx
    const unsigned base1 = q < 2 ? base + 64u : 0x40000000u;
    f32x4 o0, o1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o0[r] = fmaxf(acc0[r] + bA[r], 0.f); o1[r] = fmaxf(acc1[r] + bB[r], 0.f); }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rs, (int)base, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rs, (int)base1, 0, 0);
}

// same with the bias already in the accumulators (it rode in as the MFMA's C operand): ReLU in one instruction per value
__device__ __forceinline__ void store_tile_relu_nb(float *__restrict__ y, size_t row_base_elems, int x0, int ow,
                                                   int lane, f32x4 acc0, f32x4 acc1, bool drop = false /* this lane's pixel is not stored */)
{
    const int i = lane & 15, q = lane >> 4;
    const unsigned long long rp = (unsigned long long)(y + (row_base_elems + (size_t)x0) * UBD_C);
    const unsigned rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rp);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rp >> 32));
    float *rowp = (float *)(((unsigned long long)rhi << 32) | rlo);
    int npx = ow - x0 < 16 ? ow - x0 : 16;
    npx = npx < 0 ? 0 : npx;
    const unsigned bytes = (unsigned)__builtin_amdgcn_readfirstlane(npx * UBD_C * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, (int)bytes, 0x00020000);
    const unsigned base = drop ? 0x40000000u : (unsigned)i * (UBD_C * 4u) + 16u * (unsigned)q;
    const unsigned base1 = (q < 2 && !drop) ? base + 64u : 0x40000000u;
    const float cap = __builtin_inff();
    f32x4 o0, o1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o0[r] = ubd_relu_cap(acc0[r], cap); o1[r] = ubd_relu_cap(acc1[r], cap); }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rs, (int)base, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rs, (int)base1, 0, 0);
}

// Backward epilogue: no bias; multiply by the ReLU mask of the layer below (its saved output > 0).
__device__ __forceinline__ void store_tile_masked(float *__restrict__ y, const float (&mk)[8],
                                                  size_t row_base_elems, int x0, int ow, int lane, f32x4 acc0, f32x4 acc1)
{
    const int co = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int px = x0 + 4 * q + r;
        if (px < ow) {
            const size_t e = (row_base_elems + (size_t)px) * UBD_C;
            y[e + co] = mk[r] > 0.f ? acc0[r] : 0.f;
            if (co < 8) y[e + 16 + co] = mk[4 + r] > 0.f ? acc1[r] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------
// Separable 3x3 conv (+bias+ReLU), LDS-staged.
// Block = 4 waves = one output tile of 16 columns x TH rows.  The input patch
// ((TH-1)*S+3) x (15*S+3) pixels is copied ONCE from global memory into LDS (zero padding and the
// fused uint8 / mobilenet preprocessing applied on the way), so every input byte crosses L2->CU once
// per tile instead of once per tap.  Each wave then produces TH/4 row tiles of 16 pixels:
// lane = (i = lane&15 : pixel, q = lane>>4 : channel group); depthwise 3x3 on the VALU directly in
// the MFMA A-operand layout, pointwise 1x1 as v_mfma_f32_16x16x4_f32.
// LDS pixel stride: 26 dwords for 24 channels (104 B keeps 8-byte alignment and makes the
// ds_read_b64 of 16 neighbouring pixels conflict-free), CIN dwords for 1/3 channels.
// ------------------------------------------------------------------------------------
template <int CIN, int STRIDE> struct sep_cfg {
    static constexpr int TH = (CIN == UBD_C) ? (STRIDE == 2 ? 4 : 8) : 16;   // 24 channels: 9 x 33 (stride 2) / 10 x 18 (stride 1) pixel patches
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = 15 * STRIDE + 3;
    static constexpr int PS = CIN;                                   // LDS pixel stride in dwords
    static constexpr int CHUNKS = (CIN == UBD_C) ? PH * PW * 6 : 0;  // 16-byte chunks of the 24-channel patch
    static constexpr int ELEMS = PH * PW * CIN;
    static constexpr int BUF_FLOATS = (CIN == UBD_C) ? (CHUNKS + 255) / 256 * 256 * 4 : (ELEMS + 3) / 4 * 4;
    static constexpr int STAGE_REGS = (CIN == UBD_C) ? 1 : (ELEMS + 255) / 256;   // per-thread prefetch registers (CIN < 24)
};

// Persistent: each block walks tiles blockIdx.x, +gridDim.x, ...; the patch of tile t+1 is fetched while
// tile t is computed (24 channels: LDS-DMA into the other half of a double buffer; 1/3 channels: loads held
// in registers across the compute phase, written to LDS afterwards).  Per-lane weights are loaded once.
template <int CIN, int STRIDE, int IN_U8>
__global__ __launch_bounds__(256, (CIN == UBD_C) ? (STRIDE == 1 ? 3 : 2) : 5) void sepconv_kernel(const void *__restrict__ xin, float *__restrict__ y,
                                                      const float *__restrict__ frag,  // pwfrag then dwlane
                                                      const float *__restrict__ bias, int n, int H, int W, int OH,
                                                      int OW, int pad_lo, float pre_sub, float pre_div)
{
    using C = sep_cfg<CIN, STRIDE>;
    constexpr int CPL = (CIN == UBD_C) ? 6 : 1;   // channels per lane
    constexpr int NBUF = (CIN == UBD_C) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float patch_mem[NBUF * C::BUF_FLOATS];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform in an SGPR
    const int i = lane & 15, q = lane >> 4;
    const float *pwfrag = frag;
    const float *dwlane = frag + UBD_SEP_FRAG_FLOATS;

    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + C::TH - 1) / C::TH;
    const int total = n * tiles_y * tiles_x;

    // per-lane weights
    // 24 channels: lane (i, q) owns channels {4q..4q+3, 16+2q, 17+2q} (one ds_read_b128 + one ds_read_b64 per tap); the
    // fragments are packed for channel 6q'+s', so fetch the entries of this lane's channels
    float dwk[9][CPL];
    float pwf[CPL][2];
#pragma unroll
    for (int s = 0; s < CPL; ++s) {
        const int ch = (CIN == UBD_C) ? (s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4)) : 0;
        const int src_lane = (CIN == UBD_C) ? 16 * (ch / 6) + i : lane, ss = (CIN == UBD_C) ? ch % 6 : s;
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk[t][s] = dwlane[(t * 6 + ss) * 64 + src_lane];
        pwf[s][0] = pwfrag[(ss * 2 + 0) * 64 + src_lane];
        pwf[s][1] = pwfrag[(ss * 2 + 1) * 64 + src_lane];
    }
    const f32x4 bA = *(const f32x4 *)(bias + 4 * q);
    const f32x4 bB = q < 2 ? *(const f32x4 *)(bias + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int cb = (q < CIN) ? q : 0;                                  // 1/3 channels

    ubd_tile_decoder tdec;                                                 // neighbouring tiles on one XCD (shared halo lines), no divisions
    tdec.init(tiles_x, tiles_y, total);
    auto tile_coords = [&](int tile, int &img, int &oy0, int &ox0) {
        int tx, ty;
        tdec.decode(tile, tx, ty, img);
        oy0 = ty * C::TH; ox0 = tx * 16;
    };
    // 24 channels: LDS-DMA (global_load_lds_dwordx4): 64 x 16 B per wave instruction land linearly in LDS, no
    // VGPR round trip.  LDS slot c = pix*6 + sp holds chunk part = (sp + 3*f) % 6 of patch pixel pix,
    // f = (patch column >> 3) & 1: rotating every other octet of columns by half a pixel (12 dwords) makes the
    // ds_read_b128 (channels 4q..4q+3) and ds_read_b64 (channels 16+2q, 17+2q) of 16 neighbouring pixels
    // bank-conflict free at stride 1 (pixel pitch 24 dwords: 24 i mod 64 repeats after 8 pixels).  Out-of-image
    // pixels are fetched from a clamped address and zeroed afterwards (border tiles only).
    constexpr int ROUNDS = (CIN == UBD_C) ? (C::CHUNKS + 255) / 256 : 1;
    int dma_rel[ROUNDS];           // interior tiles: byte offset of this lane's chunk relative to the patch origin
    if constexpr (CIN == UBD_C) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            int c = rd * 256 + wid * 64 + lane;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pix = c / 6, sp = c - pix * 6;
            const int pr = pix / C::PW, pc = pix - pr * C::PW;
            int part = sp + 3 * ((pc >> 3) & 1);
            part = part >= 6 ? part - 6 : part;
            dma_rel[rd] = ((pr * W + pc) * UBD_C + part * 4) * 4;
        }
    }
    auto dma_tile = [&](int tile, float *buf) {
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int ix0 = ox0 * STRIDE - pad_lo, iy0 = oy0 * STRIDE - pad_lo;
        const char *src = (const char *)xin;
        const bool interior = (iy0 >= 0) && (ix0 >= 0) && (iy0 + C::PH <= H) && (ix0 + C::PW <= W);   // block-uniform
        if (interior) {
            const char *origin = src + (((size_t)img * H + iy0) * W + ix0) * (UBD_C * sizeof(float));
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(origin + dma_rel[rd]),
                                                 (__attribute__((address_space(3))) void *)(buf + (rd * 256 + wid * 64) * 4), 16, 0, 0);
            return;
        }
        // border tile: the same pieces through a buffer descriptor that covers exactly the image -- rows above / below it fall out
        // of range by themselves, columns left / right of it get an out-of-range offset, and the LDS-DMA writes ZEROS for them
        // (the 'same' padding; no clamped addresses, no zero-fix pass and no extra barrier afterwards)
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(src + (size_t)img * H * W * (UBD_C * sizeof(float))), 0,
                                                                        (int)((unsigned)H * W * (unsigned)(UBD_C * sizeof(float))), 0x00020000);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int cbase = rd * 256 + wid * 64;                   // wave-uniform first chunk of this piece
            int c = cbase + lane;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pix = c / 6, sp = c - pix * 6;
            const int pr = pix / C::PW, pc = pix - pr * C::PW;
            int part = sp + 3 * ((pc >> 3) & 1);
            part = part >= 6 ? part - 6 : part;
            const int gy = iy0 + pr, gx = ix0 + pc;
            const unsigned off = (unsigned)gx < (unsigned)W ? (unsigned)(((gy * W + gx) * UBD_C + part * 4) * (int)sizeof(float)) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + cbase * 4), 16, (int)off, 0, 0, 0);
        }
    };
    // 1 / 3 channels: plain loads into registers (converted / preprocessed), stored to LDS later
    int ld_rel[C::STAGE_REGS];     // interior tiles: element offset of this thread's patch elements
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
                if constexpr (IN_U8) st[k] = ((const unsigned char *)xin)[origin + ld_rel[k]];
                else st[k] = ((const unsigned *)xin)[origin + ld_rel[k]];
            }
            return;
        }
        // border tile: every element from the clamped position (image base in scalar registers + a 32-bit offset; the size_t index
        // arithmetic under per-element branches cost five quarter-rate v_mad_u64_u32 each), all loads in flight, then the elements
        // outside the image are replaced (exactly 0 after the preprocessing below)
        const int WC = W * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)img * H * WC * ((IN_U8) ? 1 : 4);
#pragma unroll
        for (int k = 0; k < C::STAGE_REGS; ++k) {
            int e = k * 256 + threadIdx.x;
            e = e < C::ELEMS ? e : C::ELEMS - 1;
            const int pr = e / (C::PW * CIN), pf = e - pr * (C::PW * CIN);
            const int gy = min(max(iy0 + pr, 0), H - 1), gf = min(max(ix0 * CIN + pf, 0), WC - 1);
            const unsigned off = (unsigned)(gy * WC + gf);
            if constexpr (IN_U8) st[k] = img8[off];
            else st[k] = ((const unsigned *)img8)[off];
        }
#pragma unroll
        for (int k = 0; k < C::STAGE_REGS; ++k) {
            int e = k * 256 + threadIdx.x;
            e = e < C::ELEMS ? e : C::ELEMS - 1;
            const int pr = e / (C::PW * CIN), pf = e - pr * (C::PW * CIN);
            const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(ix0 * CIN + pf) < (unsigned)WC;
            st[k] = inside ? st[k] : ((IN_U8) ? 0x100u : __builtin_bit_cast(unsigned, pre_sub));
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    unsigned stage[C::STAGE_REGS];      // raw loaded bits (fp32 pattern or zero-extended byte): nothing consumes them before the LDS write
    if constexpr (CIN == UBD_C) dma_tile(tile, patch_mem);
    else load_regs(tile, stage);

    for (int it = 0;; ++it) {
        float *patch = patch_mem + ((CIN == UBD_C) ? (it & 1) * C::BUF_FLOATS : 0);
        int img, oy0, ox0;
        tile_coords(tile, img, oy0, ox0);
        const int nxt = tile + gridDim.x;
        const bool has_next = (CIN == UBD_C) && nxt < total;         // block-uniform; 1/3-channel tiles: one per block (see launch)
        if constexpr (CIN == UBD_C) {
            // This tile's DMA must have landed.  vmcnt counts stores too (CDNA4) and __syncthreads() would drain
            // them all (~2 us of store latency per tile): every wave issues exactly NSTORE buffer stores per tile
            // AFTER the next tile's DMA, so "all but the NSTORE youngest" retires the DMA and nothing else.
            constexpr int NSTORE = (C::TH / 4) * 2;
            if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NSTORE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NSTORE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else { static_assert(NSTORE == 2, "counted vmcnt"); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();    // + everyone left the other buffer
            if (has_next) dma_tile(nxt, patch_mem + ((it + 1) & 1) * C::BUF_FLOATS);
        } else {
            __builtin_amdgcn_s_barrier();                            // previous tile's readers are done (no memory drain)
#pragma unroll
            for (int k = 0; k < C::STAGE_REGS; ++k) {
                const int e = k * 256 + threadIdx.x;
                if (e < C::ELEMS) {                                   // raw values were in flight during the previous compute phase
                    if constexpr (IN_U8) patch[e] = stage[k] > 255u ? 0.f : ((float)stage[k] - pre_sub) / pre_div;
                    else patch[e] = (__builtin_bit_cast(float, stage[k]) - pre_sub) / pre_div;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): LDS writes done
            __builtin_amdgcn_s_barrier();
            if (has_next) load_regs(nxt, stage);                     // in flight during the compute phase
        }

        if constexpr (CIN == UBD_C && STRIDE == 1) {
            // ---- compute, stride 1: wave `wid` owns RW consecutive rows and slides over RW + 2 patch rows: every
            //      patch row is read from LDS once (3 x (b128 + b64)) and feeds up to three output rows
            constexpr int RW = C::TH / 4;
            const int rb = wid * RW;
            float dwv[RW][6];
#pragma unroll
            for (int o = 0; o < RW; ++o)
#pragma unroll
                for (int s = 0; s < 6; ++s) dwv[o][s] = 0.f;
#pragma unroll
            for (int yy = 0; yy < RW + 2; ++yy) {
                f32x4 v4[3];
                f32x2 v2[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int pcol = i + kx;
                    const float *p = patch + ((rb + yy) * C::PW + pcol) * C::PS;
                    // 16-byte chunk c of the pixel sits in slot (c + 3f) % 6, f = (patch column >> 3) & 1
                    const int rot = 3 * ((pcol >> 3) & 1);
                    int s4 = q + rot, s2 = 4 + (q >> 1) + rot;
                    s4 = s4 >= 6 ? s4 - 6 : s4; s2 = s2 >= 6 ? s2 - 6 : s2;
                    v4[kx] = *(const f32x4 *)(p + 4 * s4);
                    v2[kx] = *(const f32x2 *)(p + 4 * s2 + 2 * (q & 1));
                }
#pragma unroll
                for (int o = 0; o < RW; ++o) {
                    const int ky = yy - o;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int t = ky * 3 + kx;
                        dwv[o][0] = fmaf(v4[kx][0], dwk[t][0], dwv[o][0]);
                        dwv[o][1] = fmaf(v4[kx][1], dwk[t][1], dwv[o][1]);
                        dwv[o][2] = fmaf(v4[kx][2], dwk[t][2], dwv[o][2]);
                        dwv[o][3] = fmaf(v4[kx][3], dwk[t][3], dwv[o][3]);
                        dwv[o][4] = fmaf(v2[kx][0], dwk[t][4], dwv[o][4]);
                        dwv[o][5] = fmaf(v2[kx][1], dwk[t][5], dwv[o][5]);
                    }
                }
                if (yy >= 2) {                               // output row yy - 2 is complete
                    const int o = yy - 2, oy = oy0 + rb + o;
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 6; ++s) {            // weights as the A operand: D = [channel][pixel] (see store_tile_relu_t)
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][0], dwv[o][s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][1], dwv[o][s], acc1, 0, 0, 0);
                    }
                    store_tile_relu_t(y, ((size_t)img * OH + oy) * OW, ox0, oy < OH ? OW : 0, lane, acc0, acc1, bA, bB);
                }
            }
        } else {
        // ---- compute: wave `wid` owns rows wid, wid+4, ...
        for (int r = wid; r < C::TH; r += 4) {                       // fixed trip count: rows past the image only mask their stores
            const int oy = oy0 + r;
            float dwv[CPL];
#pragma unroll
            for (int s = 0; s < CPL; ++s) dwv[s] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int pcol = i * STRIDE + kx;
                    const float *p = patch + ((r * STRIDE + ky) * C::PW + pcol) * C::PS;
                    const int t = ky * 3 + kx;
                    if constexpr (CIN == UBD_C) {
                        // 16-byte chunk c of the pixel sits in slot (c + 3f) % 6, f = (patch column >> 3) & 1
                        const int rot = 3 * ((pcol >> 3) & 1);
                        int s4 = q + rot, s2 = 4 + (q >> 1) + rot;
                        s4 = s4 >= 6 ? s4 - 6 : s4; s2 = s2 >= 6 ? s2 - 6 : s2;
                        const f32x4 v4 = *(const f32x4 *)(p + 4 * s4);
                        const f32x2 v2 = *(const f32x2 *)(p + 4 * s2 + 2 * (q & 1));
                        dwv[0] = fmaf(v4[0], dwk[t][0], dwv[0]);
                        dwv[1] = fmaf(v4[1], dwk[t][1], dwv[1]);
                        dwv[2] = fmaf(v4[2], dwk[t][2], dwv[2]);
                        dwv[3] = fmaf(v4[3], dwk[t][3], dwv[3]);
                        dwv[4] = fmaf(v2[0], dwk[t][4], dwv[4]);
                        dwv[5] = fmaf(v2[1], dwk[t][5], dwv[5]);
                    } else {
                        dwv[0] = fmaf(p[cb], dwk[t][0], dwv[0]);     // dwk is zero for lanes without a channel
                    }
                }
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < CPL; ++s) {              // weights as the A operand: D = [channel][pixel] (see store_tile_relu_t)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][0], dwv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf[s][1], dwv[s], acc1, 0, 0, 0);
            }
            store_tile_relu_t(y, ((size_t)img * OH + oy) * OW, ox0, oy < OH ? OW : 0, lane, acc0, acc1, bA, bB);
        }
        }
        if (!has_next) break;
        tile = nxt;
    }
}

#ifdef UBD_STAMPS   // diagnostic build only: device buffer that receives in-kernel s_memtime stamps
static unsigned long long *g_ubd_stamps = nullptr;
extern "C" void ubd_debug_set_stamps(void *p) { g_ubd_stamps = (unsigned long long *)p; }
#endif
#include "pp_lds.h"
#include "stem23.h"
#include "stem123.h"

// ------------------------------------------------------------------------------------
// Dense dilated 3x3 conv 24 -> 24 (+bias+ReLU), fp32 MFMA, weights resident in VGPRs.
// ------------------------------------------------------------------------------------
struct a_frags {
    f32x4 v4[9];
    f32x2 v2[9];
};

__device__ __forceinline__ void dil_load(a_frags &a, __amdgpu_buffer_rsrc_t rsrc, unsigned oob, int tile,
                                         int tiles_x, int h, int w, int d, int lane)
{
    const int i = lane & 15, q = lane >> 4;
    const int xt = (int)((unsigned)tile % (unsigned)tiles_x);
    const int rowid = (int)((unsigned)tile / (unsigned)tiles_x);       // = img*h + y
    const int yy = (int)((unsigned)rowid % (unsigned)h);
    const int px = xt * 16 + i;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + (ky - 1) * d;
        const bool rok = (iy >= 0) && (iy < h);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = px + (kx - 1) * d;
            const bool ok = rok && (ix >= 0) && (ix < w);
            const unsigned byte_off = (unsigned)((rowid + (ky - 1) * d) * w + ix) * (unsigned)(UBD_C * 4);
            const unsigned o4 = ok ? byte_off + 16u * q : oob;
            const unsigned o2 = ok ? byte_off + 64u + 8u * q : oob;
            u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o4, 0, 0);
            u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o2, 0, 0);
            a.v4[ky * 3 + kx] = __builtin_bit_cast(f32x4, r4);
            a.v2[ky * 3 + kx] = __builtin_bit_cast(f32x2, r2);
        }
    }
}

// EPI 0: y = relu(conv + bias) (forward).  EPI 1: y = conv * (mask_src > 0) (data gradient: `wfrag`
// then holds the spatially flipped, channel-transposed kernel and `bias` is the saved activation
// whose ReLU mask applies).
template <int EPI>
__global__ __launch_bounds__(256, 2) void dilconv_f32_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                             const float *__restrict__ wfrag,
                                                             const float *__restrict__ bias, int n, int h, int w,
                                                             int d, unsigned in_bytes)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    float wr[9][6][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wr[t][j][0] = wfrag[((t * 6 + j) * 2 + 0) * 64 + lane];
            wr[t][j][1] = wfrag[((t * 6 + j) * 2 + 1) * 64 + lane];
        }
    float b0 = 0.f, b1 = 0.f;
    if constexpr (EPI == 0) { b0 = bias[i]; b1 = (i < 8) ? bias[16 + i] : 0.f; }

    const int tiles_x = (w + 15) >> 4;
    const int total = n * h * tiles_x;
    // XCD-aware split: blocks b and b+8 share an XCD (and its L2); give each XCD group one
    // contiguous eighth of the tile range so that halo rows are re-read from the same L2.
    const int xcd = blockIdx.x & 7;
    const int nblk_x = (gridDim.x + 7 - xcd) >> 3;        // blocks in this XCD group
    const int chunk = (total + 7) >> 3;
    const int t_begin = xcd * chunk;
    const int t_end = (t_begin + chunk < total) ? t_begin + chunk : total;
    const int stride = nblk_x * 4;
    int tile = t_begin + (int)(blockIdx.x >> 3) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile >= t_end) return;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)in_bytes, 0x00020000);
    const unsigned oob = in_bytes;   // offset >= num_records -> hardware returns 0

    auto compute_store = [&](const a_frags &a, int tl) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float mk[8];
        if constexpr (EPI == 1) {      // ReLU-mask source of this tile's outputs: issued now, consumed after the MFMAs
            const int co = lane & 15, q = lane >> 4;
            const int xt0 = (int)((unsigned)tl % (unsigned)tiles_x);
            const int rid = (int)((unsigned)tl / (unsigned)tiles_x);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int opx = xt0 * 16 + 4 * q + r;
                const size_t e = ((size_t)rid * w + (size_t)(opx < w ? opx : 0)) * UBD_C;
                mk[r] = bias[e + co];
                mk[4 + r] = bias[e + 16 + (co & 7)];
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v4[t][j], wr[t][j][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v4[t][j], wr[t][j][1], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v2[t][j], wr[t][4 + j][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v2[t][j], wr[t][4 + j][1], acc1, 0, 0, 0);
            }
        }
        const int xt = (int)((unsigned)tl % (unsigned)tiles_x);
        const int rowid = (int)((unsigned)tl / (unsigned)tiles_x);     // = img*h + y
        if constexpr (EPI == 0) store_tile_relu(y, (size_t)rowid * w, xt * 16, w, lane, acc0, acc1, b0, b1);
        else store_tile_masked(y, mk, (size_t)rowid * w, xt * 16, w, lane, acc0, acc1);
    };

    // Register double buffering.  The prefetch is UNCONDITIONAL (the tile index is clamped on the last
    // iteration): a conditional prefetch makes hipcc's s_waitcnt insertion assume the worst-case number of
    // outstanding loads at the join and wait for the NEXT tile's loads before this tile's MFMAs.
    a_frags A0, A1;
    const int t_last = t_end - 1;
    dil_load(A0, rsrc, oob, tile, tiles_x, h, w, d, lane);
    for (;;) {
        int nxt = tile + stride;
        dil_load(A1, rsrc, oob, nxt < t_last ? nxt : t_last, tiles_x, h, w, d, lane);
        compute_store(A0, tile);
        tile = nxt;
        if (tile >= t_end) break;
        nxt = tile + stride;
        dil_load(A0, rsrc, oob, nxt < t_last ? nxt : t_last, tiles_x, h, w, d, lane);
        compute_store(A1, tile);
        tile = nxt;
        if (tile >= t_end) break;
    }
}

// ------------------------------------------------------------------------------------
// Head: 1x1 conv 24 -> k_out, no activation (net.py:311).  Thread per pixel.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ x, float *__restrict__ logits,
                                                   const float *__restrict__ hk, const float *__restrict__ hb,
                                                   long npix, int k_out)
{
    __shared__ float s_k[UBD_C * (UBD_MAX_CLASSES + 1)];
    __shared__ float s_b[UBD_MAX_CLASSES + 1];
    for (int t = threadIdx.x; t < UBD_C * k_out; t += blockDim.x) s_k[t] = hk[t];
    for (int t = threadIdx.x; t < k_out; t += blockDim.x) s_b[t] = hb[t];
    __syncthreads();
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const f32x4 *px = (const f32x4 *)(x + p * UBD_C);
        float v[UBD_C];
#pragma unroll
        for (int c4 = 0; c4 < 6; ++c4) {
            f32x4 t = px[c4];
            v[c4 * 4 + 0] = t[0]; v[c4 * 4 + 1] = t[1]; v[c4 * 4 + 2] = t[2]; v[c4 * 4 + 3] = t[3];
        }
        for (int ko = 0; ko < k_out; ++ko) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < UBD_C; ++c) acc = fmaf(v[c], s_k[c * k_out + ko], acc);
            logits[p * k_out + ko] = acc + s_b[ko];
        }
    }
}

// ------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------
static size_t act_bytes(const ubd_handle *, long n, long hh, long ww) { return (size_t)(n * hh * ww) * UBD_C * sizeof(float); }

void ubd_fwd_layout_compute(const ubd_handle *h, int n, int H, int W, int training, ubd_fwd_layout *L)
{
    const size_t wf = (size_t)UBD_FWD_FRAG_FLOATS * sizeof(float);
    size_t off = 0;
    L->off_wfrag = off; off += ubd_align_up(wf, 256);
    L->off_tickets = off; off += 256;
    const size_t a = ubd_align_up(act_bytes(h, n, H / 2, W / 2), 256);
    const size_t b = ubd_align_up(act_bytes(h, n, H / 4, W / 4), 256);
    L->off_a1 = off; off += a;
    L->off_a2 = off; off += a;
    if (training) {
        for (int k = 0; k < 7; ++k) { L->off_acts[k] = off; off += b; }
        L->off_b[0] = L->off_acts[0]; L->off_b[1] = L->off_acts[1];
    } else {
        L->off_b[0] = off; off += b;
        L->off_b[1] = off; off += b;
        for (int k = 0; k < 7; ++k) L->off_acts[k] = L->off_b[k & 1];
    }
    L->total = off;
}

extern "C" size_t ubd_forward_workspace_bytes(const ubd_handle *h, int n, int height, int width)
{
    if (h && h->cfg.dtype != UBD_F32) return ubd_forward16_workspace_bytes(n, height, width);
    ubd_fwd_layout L;
    ubd_fwd_layout_compute(h, n, height, width, 0, &L);
    return L.total;
}

int ubd_grid_for(long waves_needed, int num_cus, int waves_per_block, int blocks_per_cu)
{
    long blocks = (waves_needed + waves_per_block - 1) / waves_per_block;
    long cap = (long)num_cus * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int CIN, int STRIDE>
static void launch_sep(const ubd_handle *h, const void *x, int in_u8, float *y, const float *frag, const float *bias,
                       int n, int H, int W, int OH, int OW, int pad_lo, float sc, float sh, hipStream_t st)
{
    const int th = sep_cfg<CIN, STRIDE>::TH;
    const int tiles = n * ((OH + th - 1) / th) * ((OW + 15) / 16);
    const int per_cu = (CIN == UBD_C) ? (STRIDE == 1 ? 3 : 2) : 5;   // register / LDS-limited residency
    int grid = h->num_cus * per_cu;
    if (grid > tiles || CIN != UBD_C) grid = tiles;      // 1/3 channels: one tile per block, residency (not a register prefetch) hides the load latency
    if (in_u8)
        hipLaunchKernelGGL((sepconv_kernel<CIN, STRIDE, 1>), dim3(grid), dim3(256), 0, st, x, y, frag, bias, n, H, W, OH, OW, pad_lo, sc, sh);
    else
        hipLaunchKernelGGL((sepconv_kernel<CIN, STRIDE, 0>), dim3(grid), dim3(256), 0, st, x, y, frag, bias, n, H, W, OH, OW, pad_lo, sc, sh);
}

void ubd_launch_pack_direct(const ubd_handle *h, const float *params, float *wfrag, hipStream_t st)
{
    pack_args pa;
    for (int s = 0; s < 3; ++s) { pa.off_sep_dw[s] = h->off_sep_dw[s]; pa.off_sep_pw[s] = h->off_sep_pw[s]; }
    for (int k = 0; k < UBD_NUM_DIL; ++k) pa.off_dil_k[k] = h->off_dil_k[k];
    pa.c_in = h->cfg.c_in;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, st, params, wfrag, pa);
}

static void launch_pack(const ubd_handle *h, const float *params, float *wfrag, hipStream_t st)
{
    pack_args pa;
    for (int s = 0; s < 3; ++s) { pa.off_sep_dw[s] = h->off_sep_dw[s]; pa.off_sep_pw[s] = h->off_sep_pw[s]; }
    for (int k = 0; k < UBD_NUM_DIL; ++k) pa.off_dil_k[k] = h->off_dil_k[k];
    pa.c_in = h->cfg.c_in;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, st, params, wfrag, pa);
    if (h->use_wino) ubd_launch_pack_wino(h, params, wfrag + UBD_FWD_DIRECT_FLOATS, 0, st);
    if (h->wino_x6) ubd_launch_pack_wino6(h, params, (unsigned *)(wfrag + UBD_FWD_WINO6_OFF), st);
}

// frag: this layer's 6912 packed floats; aux: bias (epi 0) or mask source activation (epi 1)
void ubd_launch_dilconv(const ubd_handle *h, int epi, const float *frag, const float *aux, int dilation,
                        const float *in, float *out, int n, int H4, int W4, hipStream_t st)
{
    const unsigned in_bytes = (unsigned)((size_t)n * H4 * W4 * UBD_C * 4);
    const long tiles = (long)n * H4 * ((W4 + 15) / 16);
    int grid = ubd_grid_for(tiles, h->num_cus, 4, 2);     // 250 VGPRs -> two waves per SIMD hide each other's waits
    grid = (grid + 7) / 8 * 8;
    if (epi == 0)
        hipLaunchKernelGGL(dilconv_f32_kernel<0>, dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, dilation, in_bytes);
    else
        hipLaunchKernelGGL(dilconv_f32_kernel<1>, dim3(grid), dim3(256), 0, st, in, out, frag, aux, n, H4, W4, dilation, in_bytes);
}

static void launch_dil(const ubd_handle *h, const float *params, const float *wfrag, int k, const float *in, float *out,
                       int n, int H4, int W4, hipStream_t st)
{
    if (h->wino_x6) {
        ubd_launch_dilconv_wino6(h, 0, (const unsigned *)(wfrag + UBD_FWD_WINO6_OFF) + (size_t)k * UBD_WINO6_FRAG_U32, params + h->off_dil_b[k],
                                 UBD_DILATIONS[k], in, out, n, H4, W4, st);
        return;
    }
    if (h->use_wino) {
        ubd_launch_dilconv_wino(h, 0, wfrag + UBD_FWD_DIRECT_FLOATS + (size_t)k * UBD_WINO_FRAG_FLOATS, params + h->off_dil_b[k], UBD_F32,
                                UBD_DILATIONS[k], in, out, n, H4, W4, st);
        return;
    }
    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const float *dfrag = wfrag + 3 * per_sep;
    ubd_launch_dilconv(h, 0, dfrag + (size_t)k * UBD_DIL_FRAG_FLOATS, params + h->off_dil_b[k], UBD_DILATIONS[k], in, out, n, H4, W4, st);
}

extern "C" int ubd_pack_weights(ubd_handle *h, const float *params, void *workspace, size_t workspace_bytes, void *stream)
{
    UBD_REQUIRE(h && params && workspace, "ubd_pack_weights: null argument");
    if (h->cfg.dtype != UBD_F32) return ubd_pack16_workspace(h, params, (char *)workspace, workspace_bytes, (hipStream_t)stream);
    ubd_fwd_layout L;
    ubd_fwd_layout_compute(h, 1, 4, 4, 0, &L);
    UBD_REQUIRE(workspace_bytes >= L.off_a1, "ubd_pack_weights: workspace too small");
    launch_pack(h, params, (float *)((char *)workspace + L.off_wfrag), (hipStream_t)stream);
    UBD_CHECK_HIP(hipMemsetAsync((char *)workspace + L.off_tickets, 0, 256, (hipStream_t)stream));   // the fused stem's self-resetting counters
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int ubd_dilated_layer(ubd_handle *h, const float *params, int layer, const void *in, void *out, int n,
                                 int map_h, int map_w, const void *workspace, void *stream)
{
    UBD_REQUIRE(h && params && in && out && workspace, "ubd_dilated_layer: null argument");
    UBD_REQUIRE(layer >= 0 && layer < UBD_NUM_DIL, "ubd_dilated_layer: layer %d out of range", layer);
    UBD_REQUIRE(h->cfg.dtype == UBD_F32, "ubd_dilated_layer: only UBD_F32 in this build");
    UBD_REQUIRE((size_t)n * map_h * map_w * UBD_C * 4 <= (1ull << 30), "ubd_dilated_layer: tensor too large (activation bytes must not exceed 2^30)");
    launch_dil(h, params, (const float *)workspace, layer, (const float *)in, (float *)out, n, map_h, map_w, (hipStream_t)stream);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

// Runs L1..L9 + head.  acts[0..8] receive the hidden activations (L1..L9 outputs).
static bool fused_stem_applies(const ubd_handle *h, int n, int H)
{
    // the fused stem kernels give every CU whole strips of tiles (n * H4 / 4 of them): they need ~2 strips per CU to fill the chip
    // (a single 512 x 512 image has 32); smaller launches take the three separate kernels unless UBD_STEM forces a variant
    const int H4 = H / 4;
    const long stem_strips = (long)n * ((H4 + s23_cfg::TH3 - 1) / s23_cfg::TH3);
    const bool stem_big = h->fuse_force || stem_strips >= 2L * h->num_cus;
    return stem_big && h->fuse_stem == 2 && h->cfg.fml_compatible != 0;
}
// launches too small for strips (one image: 32 strips for 256 CUs): the same kernel with ONE tile as its work unit (stem123.h, COLD) instead of
// three separate launches; UBD_STEM=cold123 forces it at any size (tests)
static bool cold_stem_applies(const ubd_handle *h, int n, int H)
{
    if (h->cfg.fml_compatible == 0) return false;
    if (h->fuse_stem == 3) return true;
    return h->fuse_stem == 2 && !h->fuse_force && !fused_stem_applies(h, n, H);
}
bool ubd_forward_uses_fused_stem(const ubd_handle *h, int n, int H, int W) { (void)W; return h->cfg.dtype == UBD_F32 && fused_stem_applies(h, n, H); }

int ubd_forward_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                     int n, int H, int W, float *logits, char *ws, const ubd_fwd_layout &L, hipStream_t st, bool inference,
                     const pp_lds_args *pp_job)
{
    UBD_REQUIRE(h->cfg.dtype == UBD_F32, "ubd_forward: only UBD_F32 activations are implemented in this build");
    const bool prepacked = (in_dtype & UBD_IN_PREPACKED) != 0;
    in_dtype &= ~UBD_IN_PREPACKED;
    UBD_REQUIRE(n > 0 && H > 0 && W > 0 && (H % 4) == 0 && (W % 4) == 0, "ubd_forward: height and width must be positive multiples of 4 (got %d x %d)", H, W);
    UBD_REQUIRE(in_dtype == UBD_IN_F32 || in_dtype == UBD_IN_U8, "ubd_forward: bad in_dtype %d", in_dtype);
    UBD_REQUIRE(!(in_dtype == UBD_IN_U8 && h->cfg.c_in == UBD_C), "ubd_forward: u8 input needs c_in 1 or 3");
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    // the Winograd kernel marks out-of-image rows / columns with 2^30 offset terms: a quarter-resolution activation must
    // stay within 2^30 bytes (11.2 M pixels, e.g. 682 images of 512 x 512)
    UBD_REQUIRE((size_t)n * H4 * W4 * UBD_C * 4 <= (1ull << 30), "ubd_forward: batch too large for 32-bit buffer offsets; split the batch");
    float *wfrag = (float *)(ws + L.off_wfrag);
    float *a1 = (float *)(ws + L.off_a1), *a2 = (float *)(ws + L.off_a2);

    if (!prepacked) {
        launch_pack(h, params, wfrag, st);
        // strip tickets of the one-kernel stem ([0]) and its check-out counter ([16]): the kernel leaves both at zero, so they are
        // zeroed whenever the caller does not vouch for the workspace -- whichever stem variant THIS call runs (the next, prepacked
        // call on the same workspace may qualify for the one-kernel stem when this one does not, e.g. 8 x 512^2 then 32 x 256^2)
        UBD_CHECK_HIP(hipMemsetAsync(ws + L.off_tickets, 0, 256, st));
    }

    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const float *sf0 = wfrag, *sf1 = wfrag + per_sep, *sf2 = wfrag + 2 * per_sep;
    const int pad_s2 = h->cfg.fml_compatible ? 1 : 0;
    // (image - 127.5) / 127.5 (net.py:217-218); identity otherwise.  sc = subtrahend, sh = divisor.
    float sc = 0.f, sh = 1.f;
    if (preprocessing == UBD_PRE_MOBILENET) { sc = 127.5f; sh = 127.5f; }
    const int u8 = in_dtype == UBD_IN_U8;
    float *cur = (float *)(ws + L.off_acts[0]);
    const long stem_strips = (long)n * ((H4 + s23_cfg::TH3 - 1) / s23_cfg::TH3);
    const bool stem_big = h->fuse_force || stem_strips >= 2L * h->num_cus;
    const bool fuse_all = inference && fused_stem_applies(h, n, H);
    UBD_REQUIRE(!pp_job || fuse_all, "ubd_forward: a postprocess job needs the fused stem kernel (internal error)");
    const bool cold_all = inference && !fuse_all && !pp_job && cold_stem_applies(h, n, H);
    if (cold_all) {
        // L1 -> L2 -> L3 in one kernel, one cold-started tile per work unit (stem123.h COLD): tiles of a row 15 L3 columns apart
        const long tiles = (long)n * ((H4 + s23_cfg::TH3 - 1) / s23_cfg::TH3) * (W4 <= 16 ? 1 : 1 + (W4 - 16 + 14) / 15);
        int grid = h->num_cus;
        if (grid > tiles) grid = (int)tiles;
        pp_lds_args pj;
        memset(&pj, 0, sizeof(pj));
        const float *b0 = params + h->off_sep_b[0], *b1 = params + h->off_sep_b[1], *b2 = params + h->off_sep_b[2];
        int *ticket = (int *)(ws + L.off_tickets);               // not touched by this form
#ifdef UBD_STAMPS
#define S123C_STAMP_ARG , (unsigned long long *)nullptr
#else
#define S123C_STAMP_ARG
#endif
#define UBD_LAUNCH_S123C(CINV, U8V, PLV) hipLaunchKernelGGL((stem123_kernel<CINV, U8V, PLV, true>), dim3(grid), dim3(s23_cfg::NT), 0, st, images, cur, sf0, b0, sf1, b1, sf2, b2, n, H, W, H2, W2, H4, W4, sc, sh, ticket, pj S123C_STAMP_ARG)
        const bool plain = !u8 && sc == 0.f && sh == 1.f && (size_t)H * W * h->cfg.c_in * 4 < (1ull << 30) && ((uintptr_t)images & 15) == 0;
        if (h->cfg.c_in == 1) { if (u8) UBD_LAUNCH_S123C(1, 1, 0); else if (plain) UBD_LAUNCH_S123C(1, 0, 1); else UBD_LAUNCH_S123C(1, 0, 0); }
        else { if (u8) UBD_LAUNCH_S123C(3, 1, 0); else if (plain) UBD_LAUNCH_S123C(3, 0, 1); else UBD_LAUNCH_S123C(3, 0, 0); }
#undef UBD_LAUNCH_S123C
    } else if (fuse_all) {
        // L1 -> L2 -> L3 in one kernel (stem123.h): neither a1 nor a2 is touched
        const int strips = n * ((H4 + s23_cfg::TH3 - 1) / s23_cfg::TH3);
        int grid = h->num_cus;
        if (grid > strips) grid = strips;
        pp_lds_args pj;
        memset(&pj, 0, sizeof(pj));                              // n = 0: no postprocess job rides along
        if (pp_job) pj = *pp_job;
        const float *b0 = params + h->off_sep_b[0], *b1 = params + h->off_sep_b[1], *b2 = params + h->off_sep_b[2];
        int *ticket = (int *)(ws + L.off_tickets);               // zeroed with the weight pack above; the kernel resets them itself
#ifdef UBD_STAMPS
#define S123_STAMP_ARG , g_ubd_stamps
#else
#define S123_STAMP_ARG
#endif
#define UBD_LAUNCH_S123(CINV, U8V, PLV) hipLaunchKernelGGL((stem123_kernel<CINV, U8V, PLV>), dim3(grid), dim3(s23_cfg::NT), 0, st, images, cur, sf0, b0, sf1, b1, sf2, b2, n, H, W, H2, W2, H4, W4, sc, sh, ticket, pj S123_STAMP_ARG)
        const bool plain = !u8 && sc == 0.f && sh == 1.f && (size_t)H * W * h->cfg.c_in * 4 < (1ull << 30) && ((uintptr_t)images & 15) == 0;   // fp32 fed as it is: 16-byte LDS-DMA path (offsets of one image in 30 bits, 16-byte aligned base)
        if (h->cfg.c_in == 1) { if (u8) UBD_LAUNCH_S123(1, 1, 0); else if (plain) UBD_LAUNCH_S123(1, 0, 1); else UBD_LAUNCH_S123(1, 0, 0); }
        else { if (u8) UBD_LAUNCH_S123(3, 1, 0); else if (plain) UBD_LAUNCH_S123(3, 0, 1); else UBD_LAUNCH_S123(3, 0, 0); }
#undef UBD_LAUNCH_S123
    } else if (h->cfg.c_in == 1)
        launch_sep<1, 2>(h, images, u8, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sc, sh, st);
    else
        launch_sep<3, 2>(h, images, u8, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sc, sh, st);
    if (fuse_all || cold_all) {
    } else if (inference && stem_big && h->fuse_stem) {
        // L2 -> L3 in one kernel: L2's activation stays in LDS (stem23.h); the a2 buffer is not touched
        const int strips = n * ((H4 + s23_cfg::TH3 - 1) / s23_cfg::TH3);   // a block walks whole row strips of tiles
        int grid = h->num_cus;                                   // one 8-wave block per CU (120 KB of LDS)
        if (grid > strips) grid = strips;
#ifdef UBD_STAMPS
#define S23_STAMP_ARG , g_ubd_stamps
#else
#define S23_STAMP_ARG
#endif
        if (pad_s2)
            hipLaunchKernelGGL(stem23_kernel<true>, dim3(grid), dim3(s23_cfg::NT), 0, st, a1, cur, sf1, params + h->off_sep_b[1], sf2, params + h->off_sep_b[2],
                               n, H2, W2, H4, W4, pad_s2 S23_STAMP_ARG);
        else
            hipLaunchKernelGGL(stem23_kernel<false>, dim3(grid), dim3(s23_cfg::NT), 0, st, a1, cur, sf1, params + h->off_sep_b[1], sf2, params + h->off_sep_b[2],
                               n, H2, W2, H4, W4, pad_s2 S23_STAMP_ARG);
    } else {
        launch_sep<UBD_C, 1>(h, a1, 0, a2, sf1, params + h->off_sep_b[1], n, H2, W2, H2, W2, 1, 0.f, 1.f, st);
        launch_sep<UBD_C, 2>(h, a2, 0, cur, sf2, params + h->off_sep_b[2], n, H2, W2, H4, W4, pad_s2, 0.f, 1.f, st);
    }

    // inference with a single output channel: the head rides in the epilogue of L9 and L9's activation is never written
    const bool fuse_head = inference && h->use_wino && h->k_out == 1 && h->off_head_b == h->off_head_k + UBD_C;
    for (int k = 0; k < UBD_NUM_DIL; ++k) {
        float *nxt = (float *)(ws + L.off_acts[k + 1]);
        if (fuse_head && k == UBD_NUM_DIL - 1 && h->wino_x6) {
            ubd_launch_dilconv_wino6(h, 2, (const unsigned *)(wfrag + UBD_FWD_WINO6_OFF) + (size_t)k * UBD_WINO6_FRAG_U32, params + h->off_dil_b[k],
                                     UBD_DILATIONS[k], cur, logits, n, H4, W4, st, params + h->off_head_k);
            UBD_CHECK_HIP(hipGetLastError());
            return 0;
        }
        if (fuse_head && k == UBD_NUM_DIL - 1) {
            ubd_launch_dilconv_wino(h, 2, wfrag + UBD_FWD_DIRECT_FLOATS + (size_t)k * UBD_WINO_FRAG_FLOATS, params + h->off_dil_b[k], UBD_F32,
                                    UBD_DILATIONS[k], cur, logits, n, H4, W4, st, params + h->off_head_k);
            UBD_CHECK_HIP(hipGetLastError());
            return 0;
        }
        launch_dil(h, params, wfrag, k, cur, nxt, n, H4, W4, st);
        cur = nxt;
    }
    const long npix = (long)n * H4 * W4;
    int hgrid = (int)((npix + 255) / 256);
    if (hgrid > h->num_cus * 8) hgrid = h->num_cus * 8;
    hipLaunchKernelGGL(head_kernel, dim3(hgrid), dim3(256), 0, st, cur, logits, params + h->off_head_k, params + h->off_head_b, npix, h->k_out);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int ubd_forward(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                           int n, int height, int width, float *logits, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    UBD_REQUIRE(h && params && images && logits && workspace, "ubd_forward: null argument");
    if (h->cfg.dtype != UBD_F32) {
        UBD_REQUIRE(n > 0 && height > 0 && width > 0 && (height % 4) == 0 && (width % 4) == 0, "ubd_forward: height and width must be positive multiples of 4 (got %d x %d)", height, width);
        UBD_REQUIRE((in_dtype & ~UBD_IN_PREPACKED) == UBD_IN_F32 || (in_dtype & ~UBD_IN_PREPACKED) == UBD_IN_U8, "ubd_forward: bad in_dtype %d", in_dtype);
        return ubd_forward16(h, params, images, in_dtype, preprocessing, n, height, width, logits, (char *)workspace, workspace_bytes, (hipStream_t)stream);
    }
    ubd_fwd_layout L;
    ubd_fwd_layout_compute(h, n, height, width, 0, &L);
    UBD_REQUIRE(workspace_bytes >= L.total, "ubd_forward: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    return ubd_forward_impl(h, params, images, in_dtype, preprocessing, n, height, width, logits, (char *)workspace, L, (hipStream_t)stream, true);
}

// ubd_forward of one batch + ubd_postprocess of ANOTHER (earlier) batch's logits, enqueued together: when the forward pass runs
// the one-kernel stem, the first blocks of that kernel do the postprocess (one image each, pp_lds.h) before they join the stem's
// strip queue -- no second stream, no events, no extra launch; otherwise the two calls are simply made one after the other.
extern "C" int ubd_forward_postprocess(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                                       int n, int height, int width, float *logits, void *workspace, size_t workspace_bytes,
                                       const float *pp_logits, int pp_n, int pp_map_h, int pp_map_w, float logit_threshold, int scale,
                                       float min_area, int32_t *binary_map, int32_t *quads, int32_t *classes, int32_t *counts, int cap,
                                       void *pp_workspace, size_t pp_workspace_bytes, void *stream)
{
    UBD_REQUIRE(h && params && images && logits && workspace, "ubd_forward_postprocess: null argument");
    UBD_REQUIRE(pp_logits != logits, "ubd_forward_postprocess: the logits being postprocessed must not be the buffer this call writes");
    if (h->cfg.dtype == UBD_F32 && n > 0 && height > 0 && (height % 4) == 0 && ubd_forward_uses_fused_stem(h, n, height, width)) {
        pp_lds_args job;
        const int fits = ubd_pp_fill_job(h, pp_logits, pp_n, pp_map_h, pp_map_w, logit_threshold, scale, min_area, binary_map, quads, classes,
                                         counts, cap, pp_workspace, pp_workspace_bytes,
                                         s23_cfg::NT, &job);
        if (fits < 0) return 1;
        if (fits == 1) {
            ubd_fwd_layout L;
            ubd_fwd_layout_compute(h, n, height, width, 0, &L);
            UBD_REQUIRE(workspace_bytes >= L.total, "ubd_forward: workspace too small (%zu < %zu)", workspace_bytes, L.total);
            return ubd_forward_impl(h, params, images, in_dtype, preprocessing, n, height, width, logits, (char *)workspace, L, (hipStream_t)stream, true, &job);
        }
    }
    int rc = ubd_postprocess(h, pp_logits, pp_n, pp_map_h, pp_map_w, logit_threshold, scale, min_area, binary_map, quads, classes, counts, cap,
                             pp_workspace, pp_workspace_bytes, stream);
    if (rc) return rc;
    return ubd_forward(h, params, images, in_dtype, preprocessing, n, height, width, logits, workspace, workspace_bytes, stream);
}
