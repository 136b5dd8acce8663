// Separable-layer backward of the UBD_BF16 train step (included by backward.hip after sep_bwd_kernel).
//
// Same math as sep_bwd_kernel + sep_dx_kernel, but the gradient tensor G = dL/dZ of layers L1 and L2 is never written
// to memory: the kernel of layer L builds its G tile in LDS from the depthwise-output gradient dDW of the layer ABOVE
// (a 3x3 depthwise transposed convolution, masked with L's saved output), then runs the usual per-tile work.
// dDW tensors are stored in bf16.  Per 64-image batch this removes the G1/G2 round trips (2 x 403 MB written and
// read as fp32) and halves the dDW traffic.
//
//   GSRC 0  (L3)  G tile = bf16 tensor G3 written by the dilated data-gradient kernel
//   GSRC 1  (L1)  G = dwT(dDW of L2; stride 1, pad 1)            * (a1 > 0)
//   GSRC 2  (L2)  G = dwT(dDW of L3; stride 2, pad_up top/left)  * (a2 > 0)
//
// The ReLU masks are read as BITS (round 6): one 32-bit word per pixel -- bit d = (channel 2d of the saved activation > 0), bit 16 + d =
// (channel 2d + 1 > 0), d = 0..11: the two halves of the word mirror the two halves of the 16-bit pairs, so a pair's 0 / 1 factors are
// (word >> d) & 0x00010001 -- 4 bytes per pixel instead of the 48 of the activation itself.  The words are a by-product of the kernel that runs before: a 24-channel kernel
// reads the layer's INPUT activation as its X patch anyway (L3's kernel reads a2, L2's reads a1) and writes the words of the pixels
// its tile owns.  Per 64-image batch: -188 MB read in L2's kernel and in L1's, +17 MB written in L3's and in L2's.
//
// Phase 0: LDS-DMA of the input patch (24 channels; 1/3 channels go through registers one tile ahead), of the raw bf16 D
// tile (G3 tile, or dDW-above tile with halo) and of the ReLU-bit tile (one word per pixel of this layer's output; GSRC 1, 2);
// out-of-map pixels read as zeros (buffer range).  D and mask of the NEXT tile are requested right after phase 1; the stride-1
// 24-channel layer double-buffers its X patch and requests the next one under phase 2.
// Phase 1 (L1, L2): the G tile [pixel][24] in T from the raw tiles (lane = pixel column i, channels {4q..4q+3, 16+2q,
// 17+2q}); every G tensor of the bf16 train step is a bf16 tensor, these two just never leave LDS.  L1 (stride-1
// transposed conv): on the matrix pipe, taps folded into the K of the MFMA (diagonal weight matrices); L2 (stride 2): v_dot2c.
// Phase 2, two row tiles (= 32 pixels = one k-block of the 16-bit MFMA) per wave: dDW = G pw^T as ONE
// v_mfma_f32_16x16x32 per tile of channels (K = output channel, B operand = 16 bytes of a G row; the result lands in
// the depthwise lane layout); the depthwise recompute (24 channels: tap-folded MFMAs; 1/3 channels: VALU) rounded to T into
// a per-wave [32 pixels][channels] LDS image; dpw / db = DW^T G with K = pixel, both operands read with
// ds_read_b64_tr_b16 (as dil_wgrad16_kernel), an all-ones column gives the bias gradient; the depthwise-kernel gradient
// (24 channels) as the diagonals of X_t^T dDW, 14 MFMAs per k-block whose accumulators are owned by waves (see M2 below).
#pragma once

// NW = waves per block: the 1/3-channel layer needs few registers, so 6 waves share one tile's LDS (3 waves per SIMD at two
// blocks per CU); the 24-channel layers hold 54 + 16 accumulators per lane and run 4 waves per block.
#ifndef SEPB16_G2_MFMA
#define SEPB16_G2_MFMA 1      // L2's G tile on the matrix pipe (0: the v_dot2c form)
#endif
template <int CIN, int STRIDE, int GSRC, int XDMA = 0> struct sepb16_cfg {
    static constexpr int NW = 4;
    static constexpr int NT = NW * 64;
    // 1/3-channel layer (L1): 8-row tiles since round 4 -- with 16 rows the block took 56 240 bytes of LDS = 44 granules of 1280 bytes and 180
    // registers: two blocks per CU; with 8 rows 30 448 bytes / 156 registers: three (bf16 train step 1.141-1.157 -> 1.129-1.140 ms in a same-box A/B;
    // 8 rows at two blocks per CU: 1.175-1.180; four blocks: 25 spilled registers)
#ifndef SEPB16_TH1
#define SEPB16_TH1 8
#endif
#ifndef SEPB16_L1_BLOCKS
#define SEPB16_L1_BLOCKS 3
#endif
    static constexpr int TH = (CIN == UBD_C) ? 8 : SEPB16_TH1;
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = 15 * STRIDE + 3;
    static constexpr int XPIX = PH * PW;
    static constexpr int GPIX = TH * 16;
    static constexpr int DROWS = GSRC == 0 ? TH : (GSRC == 1 ? TH + 2 : TH / 2 + 2);
    static constexpr int DCOLS = GSRC == 0 ? 16 : (GSRC == 1 ? 18 : 10);
    static constexpr int DPIX = DROWS * DCOLS;
    // DMA regions (16-byte chunks, 3 per bf16 pixel; each region is a whole number of 1 KiB wave-instructions):
    // [X patch (24 ch only) x XBUF] [D tile] [ReLU-bit tile: one word per G pixel (GSRC 1, 2), 4-byte DMA, 64 words per wave-instruction]
    // XBUF 2 (the stride-1 24-channel layer, 8-row tiles): the NEXT tile's X patch is fetched under phase 2 into the other
    // buffer -- with one buffer the patch was requested at the top of the tile and the wave sat out a memory round trip
    // there (7.7 k of 25 k cycles per tile in the stamps: the CU had ~16 KB in flight on average, 2-3 TB/s chip-wide).
    static constexpr int XBUF = (CIN == UBD_C && STRIDE == 1) ? 2 : 1;
    static constexpr int XCHUNKS = (CIN == UBD_C) ? XPIX * 3 : 0;
    static constexpr int XI = (XCHUNKS + 63) / 64, DI = (DPIX * 3 + 63) / 64;        // wave-instructions
    static constexpr int MI = (GSRC != 0) ? (GPIX + 63) / 64 : 0;
    static constexpr int XK = (XI + NW - 1) / NW, DK = (DI + NW - 1) / NW, MK = (MI + NW - 1) / NW;   // per wave
    static constexpr int OFF_D = XBUF * XI * 1024;                                  // byte offsets inside the DMA area
    static constexpr int OFF_M = OFF_D + DI * 1024;
    static constexpr int DMA_BYTES = OFF_M + ((MI * 256 + 1023) / 1024) * 1024;
    static constexpr int XREGS = (CIN == UBD_C || XDMA) ? 1 : (XPIX * CIN + NT - 1) / NT;                  // staged input elements per thread
    // XDMA (1/3-channel fp32 input that needs no preprocessing, image rows a whole number of 16-byte chunks): the patch rows are
    // fetched by LDS-DMA from the 16-byte boundary at or below their first float (skew 0..3 floats: pad_lo channels back from a
    // 16-byte-aligned tile origin), double-buffered; XROWC chunks per row
    static constexpr int XROWC = (PW * CIN + (CIN == 3 ? 1 : 3) + 3) / 4;
    static constexpr int XROW = XDMA ? XROWC * 4 : PW * CIN;                        // row pitch of the fp32 patch (floats)
    static constexpr int XD_CHUNKS = PH * XROWC, XDI = (XD_CHUNKS + 63) / 64, XDK = (XDI + NW - 1) / NW;
    static constexpr int XF32_BYTES = (CIN == UBD_C) ? 0 : (XDMA ? 2 * XDI * 1024 : (XPIX * CIN + 3) / 4 * 16);   // fp32 patch of 1/3-channel inputs
    static constexpr int OFF_DMA = XF32_BYTES;
    static constexpr int OFF_G = OFF_DMA + DMA_BYTES;                  // G tile in T (GSRC 0: the D tile itself)
    static constexpr int OFF_SDW = OFF_G + (GSRC == 0 ? 0 : GPIX * UBD_C * 2);   // per-wave depthwise-output images
    static constexpr int SDW_W = (CIN == UBD_C) ? UBD_C : 4;           // columns per pixel ([ch.., 1, 0..] for 1/3 channels)
    static constexpr int SDW_BYTES = 32 * SDW_W * 2;                   // two row tiles per wave
    static constexpr int OFF_SDD = OFF_SDW + NW * SDW_BYTES;       // per-wave dDW images [32 pixels][24] in T (24-channel layers)
    static constexpr int OFF_UT = OFF_SDD + (CIN == UBD_C ? NW * SDW_BYTES : 0);   // taps of the layer above [3][4][24], packed 16-bit pairs (kx = 3: zeros; GSRC 2)
    static constexpr int OFF_CONST = OFF_UT + 12 * UBD_C * 4;          // [0,8): {1,0,0,0} in T   [8,32): zeros ([16,32): a 16-byte zero operand)
    static constexpr int LDS_BYTES = OFF_CONST + 32;
    // blocks per CU = waves per SIMD: three when the LDS clearly allows it (a grid that is not fully resident runs in two
    // uneven waves of blocks) and the kernel fits 168 VGPRs (24 channels), else two
#ifndef SEPB16_XD_BLOCKS
#define SEPB16_XD_BLOCKS 4      // round 6: with the ReLU masks as bits the DMA-fed L1 kernel needs 125 registers: four blocks per CU hide more of its exposed DMA wait (1.3-1.7 k of 6 k cycles per tile in the stamps)
#endif
    static constexpr int BLOCKS_PER_CU = LDS_BYTES > 78 * 1024 ? 1 : ((CIN == UBD_C && 3 * LDS_BYTES <= 150 * 1024) ? 3 : (CIN != UBD_C ? (XDMA ? SEPB16_XD_BLOCKS : SEPB16_L1_BLOCKS) : 2));
    static constexpr int PART = 9 * CIN + CIN * UBD_C + UBD_C;
};

// acc + a.lo * b.lo + a.hi * b.hi, 16-bit inputs, fp32 accumulation (v_dot2c_f32_{bf16,f16})
template <typename T> __device__ __forceinline__ float dot2b(unsigned a, unsigned b, float acc);
template <> __device__ __forceinline__ float dot2b<__bf16>(unsigned a, unsigned b, float acc)
{
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), acc, false);
}
template <> __device__ __forceinline__ float dot2b<_Float16>(unsigned a, unsigned b, float acc)
{
    typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), acc, false);
}

template <typename T> __device__ __forceinline__ unsigned pack2b(float lo, float hi)
{
    typedef T t2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, t2));
}

template <typename T> __device__ __forceinline__ void widen2b(unsigned w, float &lo, float &hi)
{
    lo = (float)__builtin_bit_cast(T, (unsigned short)(w & 0xFFFFu));
    hi = (float)__builtin_bit_cast(T, (unsigned short)(w >> 16));
}

// g (two 16-bit values: channel pair d) with each half kept where the pixel's ReLU word has bit d / bit 16 + d set
__device__ __forceinline__ unsigned relu_bits2(unsigned g, unsigned word, int d)
{
    const unsigned k = (word >> d) & 0x00010001u;    // 0 / 1 factor per half
    unsigned r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(g), "v"(k));
    return r;
}
// 0 / 1 per half of a pair of saved 16-bit activations (never negative: > 0 iff the 15 magnitude bits are non-zero)
__device__ __forceinline__ unsigned relu_pair_bits(unsigned m)
{
    unsigned k;                                      // asm: hipcc rewrites the vector form into two compares, two selects and a v_perm
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(m & 0x7FFF7FFFu), "v"(0x00010001u));
    return k;
}

template <int D, int NMAX> using sepb16_magic = ubd_magic24<D, NMAX>;      // (common.h)

// Staging of a tile by LDS-DMA: chunk -> (pixel, part) -> (row, column) with two magic-number divisions, then ONE buffer-addressed
// DMA per 1 KiB piece through a descriptor that covers exactly the image (ubd_blds16, common.h): rows above / below the image fall
// out of its range by themselves, columns left / right of it get an out-of-range offset, and the hardware writes ZEROS for them --
// the 'same' padding and the ragged borders need no clamped addresses, no zero-fix pass and no extra barrier on border tiles.
// ~12 vector instructions per piece instead of ~40 (the staging was 4 k of the L2 tile's 23 k cycles in the stamps of round 2; a
// table of the chunk coordinates in LDS was slower still -- 8 k: the reads queue behind phase 2 of the CU's other blocks).
template <int NK, int NINSTR, int NCHUNKS, int COLS, int NW>
__device__ __forceinline__ void sepb16_stage_ar(const char *__restrict__ tensor, int img, int th, int tw, int y0, int x0,
                                                unsigned lds_dst, int lane, int wid)
{
    using M3 = sepb16_magic<3, NINSTR * 64>;
    using MC = sepb16_magic<COLS, NCHUNKS / 3 + 1>;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(tensor + (size_t)img * th * tw * (UBD_C * 2)), 0,
                                                                    (int)((unsigned)th * tw * (UBD_C * 2)), 0x00020000);   // wave-uniform
    asm volatile("" : "+v"(lane));         // opaque per call: hipcc otherwise keeps every piece's (row, column, part) in registers across the tile loop and spills
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int instr = k * NW + wid;
        if (instr >= NINSTR) break;                                       // wave-uniform
        unsigned c = (unsigned)(instr * 64 + lane);
        c = c < (unsigned)NCHUNKS ? c : (unsigned)(NCHUNKS - 1);
        const unsigned pix = __umul24(c, M3::m) >> M3::sh, part = c - 3u * pix;
        const unsigned pr = __umul24(pix, MC::m) >> MC::sh, pc = pix - (unsigned)COLS * pr;
        const int gy = y0 + (int)pr, gx = x0 + (int)pc;
        const unsigned off = (unsigned)gx < (unsigned)tw ? (unsigned)((gy * tw + gx) * (UBD_C * 2)) + part * 16u : 0x80000000u;
        ubd_blds16(rsrc, off, lds_dst + instr * 1024);
    }
}

// the ReLU-bit tile [TH][16] words of image img at (y0, x0): 4-byte LDS-DMA, 64 words (four tile rows) per wave-instruction
template <int NK, int NINSTR, int NW>
__device__ __forceinline__ void sepb16_stage_bits(const unsigned *__restrict__ bits, int img, int th, int tw, int y0, int x0, unsigned lds_dst, int lane, int wid)
{
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(bits + (size_t)img * th * tw), 0, (int)((unsigned)th * tw * 4u), 0x00020000);   // wave-uniform
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int instr = k * NW + wid;
        if (instr >= NINSTR) break;                                       // wave-uniform
        const int w = instr * 64 + lane, gy = y0 + (w >> 4), gx = x0 + (w & 15);
        const unsigned off = (unsigned)gx < (unsigned)tw ? (unsigned)((gy * tw + gx) * 4) : 0x80000000u;   // rows below the map: past the range by themselves
        ubd_blds4(rsrc, off, lds_dst + instr * 256);
    }
}

// IN_MODE (1/3-channel layers): 0 fp32 input through registers (preprocessed here), 1 uint8 input through registers, 2 fp32 input by LDS-DMA
template <int CIN, int STRIDE, int IN_MODE, int GSRC, typename T>
__global__ __launch_bounds__(256, (sepb16_cfg<CIN, STRIDE, GSRC, IN_MODE == 2>::BLOCKS_PER_CU >= 3) ? (sepb16_cfg<CIN, STRIDE, GSRC, IN_MODE == 2>::BLOCKS_PER_CU) : 2) void sepb16_kernel(const void *__restrict__ xin, const unsigned short *__restrict__ D,
                                                        const unsigned *__restrict__ mbits, unsigned *__restrict__ xbits, unsigned short *__restrict__ dDW,
                                                        const float *__restrict__ dw_own, const float *__restrict__ pw_own,
                                                        const float *__restrict__ dw_up, float *__restrict__ partials, int n, int H,
                                                        int W, int OH, int OW, int pad_lo, int DH, int DW_, int pad_up, float pre_sub,
                                                        float pre_div, const rp_job prev
#ifdef UBD_STAMPS
                                                        , unsigned long long *__restrict__ stamps
#endif
                                                        )
{
#ifdef UBD_STAMPS   // diagnostic build only: s_memtime of lane 0 of every wave at the phase boundaries of its first 8 tiles
    int stamp_it = 0;
#define SBSTAMP(k) do { if (stamps && stamp_it < 8 && (threadIdx.x & 63) == 0) stamps[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + stamp_it) * 12 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SBSTAMP(k) do {} while (0)
#endif
    constexpr bool XDMA = (IN_MODE == 2);
    constexpr int IN_U8 = (IN_MODE == 1);
    static_assert(!XDMA || CIN != UBD_C, "the 24-channel patch is a 16-bit tensor");
    using C = sepb16_cfg<CIN, STRIDE, GSRC, XDMA>;
    constexpr int CPL = (CIN == UBD_C) ? 6 : 1;
    constexpr int NT_A = (CIN == UBD_C) ? 2 : 1;
    constexpr int MT_PW = (CIN == UBD_C) ? 2 : 1;
    constexpr int HR_UNROLL = (CIN == UBD_C) ? 1 : 2;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];             // ONE LDS object
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    float *xf32 = (float *)lds;                                                  // 1/3-channel patch (fp32)
    char *dma = lds + C::OFF_DMA;
    const char *draw = dma + C::OFF_D, *mraw = dma + C::OFF_M;                   // D tile, ReLU-bit tile (one word per G pixel)
    char *g16 = (GSRC == 0) ? dma + C::OFF_D : lds + C::OFF_G;              // G tile [pixel][24] in T
    char *sdw = lds + C::OFF_SDW + wid * C::SDW_BYTES;                      // this wave's [32 pixels][SDW_W] depthwise outputs
    char *sdd = lds + C::OFF_SDD + wid * C::SDW_BYTES;                      // ... and dDW values (24-channel layers)
    const char *c_ones = lds + C::OFF_CONST, *c_zero = lds + C::OFF_CONST + 8;
    if (threadIdx.x < 8) ((unsigned *)(lds + C::OFF_CONST))[threadIdx.x] = threadIdx.x == 0 ? (unsigned)__builtin_bit_cast(unsigned short, (T)1.0f) : 0u;
    if constexpr (CIN != UBD_C) {                                           // [ch.., 1, 0..] rows: the constant columns are written once
        for (int t = lane; t < 32 * C::SDW_W; t += 64) {
            const int col = t % C::SDW_W;
            ((unsigned short *)sdw)[t] = col == CIN ? __builtin_bit_cast(unsigned short, (T)1.0f) : (unsigned short)0;
        }
    }
    const unsigned lds_dma = ubd_lds_addr(dma);
    const unsigned lds_x = ubd_lds_addr(lds);
    __amdgpu_buffer_rsrc_t ddw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dDW, 0, dDW ? (int)((unsigned)n * OH * OW * (UBD_C * 2)) : 0, 0x00020000);   // <= 2^31 bytes (ubd.h size limits)
    unsigned *utp = (unsigned *)(lds + C::OFF_UT);

    // Lane (i, q) owns channels chs(s) = {4q .. 4q+3, 16+2q, 17+2q} of pixel column i (24-channel layers) or channel q
    // (1/3 channels).  Depthwise taps live in LDS tables ([tap][24], read as b128 + b64 broadcast per k-group): keeping
    // 2 x 54 of them in VGPRs limits the kernel to two waves per SIMD.
    if constexpr (GSRC != 0 && !((GSRC == 2) && (SEPB16_G2_MFMA != 0)))
        for (int t = threadIdx.x; t < 12 * UBD_C; t += C::NT) {
            const int ch = t % UBD_C, kk = t / UBD_C, kx = kk & 3, ky = kk >> 2;
            // 16-bit tap in the half of the dword that matches the channel's position in its pair: ONE v_dot2c then
            // multiplies the raw activation pair from LDS (the other product is x * 0)
            const unsigned wb = kx < 3 ? (unsigned)__builtin_bit_cast(unsigned short, (T)dw_up[(ky * 3 + kx) * UBD_C + ch]) : 0u;
            utp[t] = (ch & 1) ? (wb << 16) : wb;
        }
    float dwk1[9];                                                               // 1/3-channel layers: this lane's taps
    if constexpr (CIN != UBD_C) {
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk1[t] = q < CIN ? (float)(T)dw_own[t * CIN + q] : 0.f;
    }
    // GSRC 1 (the 1/3-channel kernels; stride-1 transposed depthwise conv): the G tile is built on the matrix pipe.  Per row
    // of 16 pixels G^T[co][pixel] = sum_taps diag(w_tap) D^T: M = output channel, N = pixel, and because the weight matrix
    // of a depthwise tap is diagonal, the K = 32 of one v_mfma_f32_16x16x32 carries SEVERAL taps: two taps x 16 channels
    // for channels 0..15 (five MFMAs for the nine taps), four taps x 8 channels for channels 16..23 (three MFMAs, rows
    // permuted so that lane (pixel, q) receives channels 16 + 2q, 17 + 2q).  B operand = 16 bytes of the raw D tile at the
    // tap's pixel, one ds_read_b128 per MFMA; 8 reads + 8 MFMAs per row instead of 18 reads + 54 v_dot2c (phase 1 was
    // 6.2 k of the tile's 14.3 k cycles in the stamps).
    constexpr bool UREG = (CIN != UBD_C) && (GSRC == 1);
    u32x4 ga0[UREG ? 5 : 1], ga1[UREG ? 3 : 1];               // A operands: lane (m = i, kg = q) holds k = 8q .. 8q+7
    int gb0[UREG ? 5 : 1], gb1[UREG ? 3 : 1];                 // B operand byte offsets inside the D tile for row 0 (row r: + r * DCOLS * 48)
    if constexpr (UREG) {
        auto wup = [&](int t, int ch) { return (unsigned)__builtin_bit_cast(unsigned short, (T)dw_up[t * UBD_C + ch]); };
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int ts = 2 * j + (q >> 1), t = ts < 9 ? ts : 8;  // this k-group's tap; the spare slot re-reads tap 8 with zero
            const int e = i - 8 * (q & 1);                         // weights (data this pixel sums anyway: no foreign NaN can enter)
            unsigned w[4] = {0u, 0u, 0u, 0u};                      // k = 8q + e <-> channel 8 (q & 1) + e == row i
            if (ts < 9 && e >= 0 && e < 8) w[e >> 1] = wup(t, i) << (16 * (e & 1));
            ga0[j] = u32x4{w[0], w[1], w[2], w[3]};
            const int ky = t / 3, kx = t - 3 * ky;
            gb0[j] = ((2 - ky) * C::DCOLS + (i + 2 - kx)) * 48 + 16 * (q & 1);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ts = 4 * j + q, t = ts < 9 ? ts : 8;         // k-group q <-> tap 4j + q, channels 16..23
            const int qq = i >> 2, r = i & 3, e = 2 * qq + r;       // row i = 4 qq + r <-> channel 16 + 2 qq + r (r < 2)
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && r < 2) w[e >> 1] = wup(t, 16 + e) << (16 * (e & 1));
            ga1[j] = u32x4{w[0], w[1], w[2], w[3]};
            const int ky = t / 3, kx = t - 3 * ky;
            gb1[j] = ((2 - ky) * C::DCOLS + (i + 2 - kx)) * 48 + 32;
        }
    }
    // GSRC 2 (L2: stride-2 transposed depthwise conv of L3's dDW): the same construction (round 4; it was 24 v_dot2c + 16 LDS reads per
    // row).  A G pixel receives only the taps whose row / column parity matches its own: the rows of a wave (wid + 4 kr, oy0 even) all
    // have ONE parity, so a wave folds either the six taps ky in {0, 2} or the three taps ky = 1 into its MFMAs -- three + two (two + one)
    // per row of 16 pixels --, and a lane whose column parity does not match a tap's kx supplies the 16-byte ZERO operand as its B
    // column for that tap.  The diagonal A operands hold one non-zero 16-bit value per lane: kept compact (one register per MFMA) and
    // expanded with four selects where they are used -- the kernel has no 20 registers to spare at three blocks per CU.
    constexpr bool UREG2 = (GSRC == 2) && (SEPB16_G2_MFMA != 0);
    unsigned g2d0[UREG2 ? 3 : 1], g2d1[UREG2 ? 2 : 1];            // the non-zero value of this lane's A fragment, already in its half of the dword
    int g2b0[UREG2 ? 3 : 1], g2b1[UREG2 ? 2 : 1];                 // B operand: byte offset inside the D tile for the wave's first row, < 0: the zero operand
    const int g2par = (wid + pad_up) & 1;                         // wave-uniform: 0 -> taps ky in {0, 2}, 1 -> ky = 1
    if constexpr (UREG2) {
        const int ntap = g2par ? 3 : 6;
        auto wup2 = [&](int ky, int kx, int ch) { return (unsigned)__builtin_bit_cast(unsigned short, (T)dw_up[(ky * 3 + kx) * UBD_C + ch]); };
        auto tap_of = [&](int ts, int &ky, int &kx) { ky = g2par ? 1 : (ts < 3 ? 0 : 2); kx = ts % 3; };
        auto boff = [&](int ky, int kx, int chunk_bytes) {
            if (((i + pad_up - kx) & 1) != 0) return -1;             // this pixel column does not receive the tap
            const int dr = ((wid + pad_up - ky) >> 1) + 1, dc = ((i + pad_up - kx) >> 1) + 1;
            return (dr * C::DCOLS + dc) * 48 + chunk_bytes;
        };
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ts = 2 * j + (q >> 1), e = i - 8 * (q & 1);      // k = 8q + e <-> channel 8 (q & 1) + e == row i
            int ky, kx;
            tap_of(ts < ntap ? ts : 0, ky, kx);
            g2d0[j] = (ts < ntap && e >= 0 && e < 8) ? wup2(ky, kx, i) << (16 * (e & 1)) : 0u;
            g2b0[j] = ts < ntap ? boff(ky, kx, 16 * (q & 1)) : -1;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ts = 4 * j + q, rr = i & 3, e = 2 * (i >> 2) + rr;   // row i = 4 qq + rr <-> channel 16 + 2 qq + rr (rr < 2)
            int ky, kx;
            tap_of(ts < ntap ? ts : 0, ky, kx);
            g2d1[j] = (ts < ntap && rr < 2) ? wup2(ky, kx, 16 + e) << (16 * (e & 1)) : 0u;
            g2b1[j] = ts < ntap ? boff(ky, kx, 32) : -1;
        }
    }
    // 24-channel layers, phase 2 on the matrix pipe (M2).  The depthwise recompute uses the same tap-folded diagonal
    // operands as the G tile above, with this layer's own taps.  The depthwise-kernel gradient ddw[t][c] = sum_p X[p + t][c]
    // dDW[p][c] is the diagonal of X_t^T dDW with K = pixel: one MFMA per tap for channels 0..15 (A = X shifted by the tap,
    // B = dDW, both read transposed from their [pixel][24] LDS images), and for channels 16..23 two taps share an MFMA (rows
    // 0..7 / 8..15 of A, B = those eight dDW channels twice): 14 MFMAs per 32 pixels whose diagonals are the 216 sums --
    // instead of 108 half-empty v_dot2c per 16 pixels (phase 2 was issue-bound: ~210 instructions per 16-pixel row).
    constexpr bool M2 = (CIN == UBD_C);
    u32x4 wa0[M2 ? 5 : 1], wa1[M2 ? 3 : 1];
    int xo0[M2 ? 5 : 1], xo1[M2 ? 3 : 1];                     // B operand byte offsets inside the X patch for tile row 0
    // The 14 ddw accumulators are OWNED by waves (unit u = tap 0..8 for channels 0..15, 9..13 = tap pairs for channels
    // 16..23; wave w owns units w, w + 4, w + 8, w + 12): every wave runs its units over ALL k-blocks of the tile after a
    // block barrier, so a wave holds 4 accumulators instead of 14 (56 VGPRs do not fit beside the rest at 3 waves per SIMD)
    // and its sums are complete -- no cross-wave reduction at the end.
    f32x4 accdw[M2 ? 4 : 1] = {};
    int uo[M2 ? 4 : 1];                                        // per slot: tap offset + this lane's segment inside the pixel
    if constexpr (M2) {
        static_assert(!M2 || C::TH == 2 * C::NW, "one k-block (two tile rows) per wave");
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int u = wid + 4 * sl, p4 = lane & 3;
            if (u < 9) uo[sl] = 8 * p4 + ((u / 3) * C::PW + u % 3) * 48;
            else {
                const int jj = u - 9, t0 = jj < 3 ? 3 * jj : (jj == 3 ? 2 : 8);
                uo[sl] = 32 + 8 * (p4 & 1) + (p4 >= 2 ? (jj == 3 ? C::PW : 1) * 48 : 0) + ((t0 / 3) * C::PW + t0 % 3) * 48;
            }
        }
        auto wown = [&](int t, int ch) { return (unsigned)__builtin_bit_cast(unsigned short, (T)dw_own[t * UBD_C + ch]); };
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int ts = 2 * j + (q >> 1), t = ts < 9 ? ts : 8;
            const int e = i - 8 * (q & 1);
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && e >= 0 && e < 8) w[e >> 1] = wown(t, i) << (16 * (e & 1));
            wa0[j] = u32x4{w[0], w[1], w[2], w[3]};
            const int ky = t / 3, kx = t - 3 * ky;
            xo0[j] = (ky * C::PW + i * STRIDE + kx) * 48 + 16 * (q & 1);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ts = 4 * j + q, t = ts < 9 ? ts : 8;
            const int qq4 = i >> 2, r = i & 3, e = 2 * qq4 + r;
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (ts < 9 && r < 2) w[e >> 1] = wown(t, 16 + e) << (16 * (e & 1));
            wa1[j] = u32x4{w[0], w[1], w[2], w[3]};
            const int ky = t / 3, kx = t - 3 * ky;
            xo1[j] = (ky * C::PW + i * STRIDE + kx) * 48 + 32;
        }
    }
    // A operand of the dDW product (K = output channel co, 24 padded to 32): lane (m = i, kg = q) holds
    // pw[ch(m, tile)][8q .. 8q+7] in T (zero for q = 3); the result rows 4q + r then are this lane's own channels:
    // tile 0 -> 4q + r, tile 1 (r < 2) -> 16 + 2q + r; 1/3 channels: tile 0 only, row 4q -> channel q
    u32x4 apwb[NT_A];
#pragma unroll
    for (int tl = 0; tl < NT_A; ++tl) {
        int ch;
        if constexpr (CIN == UBD_C) ch = tl == 0 ? i : ((i & 3) < 2 ? 16 + 2 * (i >> 2) + (i & 3) : -1);
        else ch = ((i & 3) == 0 && (i >> 2) < CIN) ? (i >> 2) : -1;
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w8[e] = (ch >= 0 && q < 3) ? pw_own[ch * UBD_C + 8 * q + e] : 0.f;
        apwb[tl] = u32x4{pack2b<T>(w8[0], w8[1]), pack2b<T>(w8[2], w8[3]), pack2b<T>(w8[4], w8[5]), pack2b<T>(w8[6], w8[7])};
    }
    const bool ch_ok = (CIN == UBD_C) || (q < CIN);
    const int cb = (q < CIN) ? q : 0;                                            // 1/3 channels
    // GSRC 2 (stride-2 transposed conv): the taps that reach this lane's pixel column have kx = par, par + 2
    const int par = (i + pad_up) & 1;

    // 1/3 channels: element e = threadIdx.x + 256 k of the [PH][PW * CIN] patch as (row << 16 | column-in-floats); the
    // address is image base (scalar registers) + a 32-bit element offset -- the size_t index arithmetic of round 1 cost five
    // quarter-rate v_mad_u64_u32 per element on border tiles
    int xpk[C::XREGS];
    if constexpr (CIN != UBD_C && !XDMA) {
#pragma unroll
        for (int k = 0; k < C::XREGS; ++k) {
            int e = k * C::NT + (int)threadIdx.x;
            e = e < C::XPIX * CIN ? e : C::XPIX * CIN - 1;
            const int pr = e / (C::PW * CIN);
            xpk[k] = (pr << 16) | (e - pr * (C::PW * CIN));
        }
    }

    float ddw[M2 ? 1 : 9][CPL];                                 // 1/3 channels: per-lane sums (VALU)
#pragma unroll
    for (int t = 0; t < (M2 ? 1 : 9); ++t)
#pragma unroll
        for (int s = 0; s < CPL; ++s) ddw[t][s] = 0.f;
    f32x4 accpw[MT_PW][2] = {};

    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + C::TH - 1) / C::TH;
    const int total = n * tiles_y * tiles_x;
    // Tile geometry and the staging steps.  Pipeline per tile: [X patch DMA (24 ch)] -> wait -> phase 1 (D, mask -> G tile)
    // -> D / mask DMA of the NEXT tile (their LDS regions are free again) and, for 1/3 channels, the next tile's input
    // loads into registers -> phase 2.  Only the 24-channel X patch (single-buffered) is fetched with exposed latency.
    struct geom { int img, oy0, ox0, iy0, ix0, dy0, dx0; bool xborder; };
    ubd_tile_decoder tdec;
    tdec.init(tiles_x, tiles_y, total);
    auto tile_geom = [&](int tile) {
        geom g;
        int tx, ty;
        tdec.decode(tile, tx, ty, g.img);                                  // neighbouring tiles on one XCD (shared halo lines)
        g.oy0 = ty * C::TH; g.ox0 = tx * 16;
        g.ix0 = g.ox0 * STRIDE - pad_lo; g.iy0 = g.oy0 * STRIDE - pad_lo;
        g.dy0 = GSRC == 0 ? g.oy0 : (GSRC == 1 ? g.oy0 - 1 : (g.oy0 >> 1) - 1);       // origin of the D tile in the D tensor
        g.dx0 = GSRC == 0 ? g.ox0 : (GSRC == 1 ? g.ox0 - 1 : (g.ox0 >> 1) - 1);
        g.xborder = (g.iy0 < 0) || (g.ix0 < 0) || (g.iy0 + C::PH > H) || (g.ix0 + C::PW > W);
        return g;
    };
    // D tile and, for GSRC 1 / 2, the ReLU-mask tile (this layer's saved output under the G tile; used once per pixel in
    // phase 1, so both regions are free for the next tile's data as soon as phase 1 is over)
    auto stage_dm = [&](const geom &g) {
        sepb16_stage_ar<C::DK, C::DI, C::DPIX * 3, C::DCOLS, C::NW>((const char *)D, g.img, DH, DW_, g.dy0, g.dx0, lds_dma + C::OFF_D, lane, wid);
        if constexpr (GSRC != 0)
            sepb16_stage_bits<C::MK, C::MI, C::NW>(mbits, g.img, OH, OW, g.oy0, g.ox0, lds_dma + C::OFF_M, lane, wid);
    };
    auto stage_x = [&](const geom &g, int buf) {
        if constexpr (CIN == UBD_C)
            sepb16_stage_ar<C::XK, C::XI, C::XCHUNKS, C::PW, C::NW>((const char *)xin, g.img, H, W, g.iy0, g.ix0, lds_dma + buf * (C::XI * 1024), lane, wid);
    };
    // 1/3-channel input: raw bits (fp32 pattern or zero-extended byte; 0x100 / pre_sub bits = "outside", exactly 0 after the
    // preprocessing) held in registers across phase 2
    unsigned xreg[C::XREGS];
    const int xskew = (-(CIN * pad_lo)) & 3;                            // XDMA: floats between the 16-byte boundary and the patch's first float
    auto stage_xd = [&](const geom &g, int buf) {
        if constexpr (XDMA) {
            using MR = sepb16_magic<C::XROWC, C::XDI * 64>;
            const int WC = W * CIN;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)xin + (size_t)g.img * H * WC * 4), 0,
                                                                            (int)((unsigned)H * WC * 4), 0x00020000);   // wave-uniform; rows above / below the image fall out of its range
            const int a0f = g.ix0 * CIN - xskew;                          // a multiple of 4 (tile origins are multiples of 32 pixels)
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int k = 0; k < C::XDK; ++k) {
                const int instr = k * C::NW + wid;
                if (instr >= C::XDI) break;                               // wave-uniform
                const unsigned c = (unsigned)(instr * 64 + ln);
                const unsigned pr = __umul24(c, MR::m) >> MR::sh, pc = c - (unsigned)C::XROWC * pr;
                const int gy = g.iy0 + (int)pr, f0 = a0f + 4 * (int)pc;
                // chunks left / right of the image row (never partial: the row is a whole number of chunks) and the slack of the last piece: zeros
                const unsigned off = (c < (unsigned)C::XD_CHUNKS && (unsigned)f0 < (unsigned)WC) ? (unsigned)((gy * WC + f0) * 4) : 0x80000000u;
                ubd_blds16(rsrc, off, lds_x + buf * (C::XDI * 1024) + instr * 1024);
            }
        }
    };
    auto load_x = [&](const geom &g) {
        const int WC = W * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)g.img * H * WC * (IN_U8 ? 1 : 4);   // wave-uniform
        const int fx0 = g.ix0 * CIN;
        if (!g.xborder) {                                             // block-uniform
            const unsigned o0 = (unsigned)(g.iy0 * WC + fx0);
#pragma unroll
            for (int k = 0; k < C::XREGS; ++k) {
                const unsigned off = o0 + (unsigned)__umul24(xpk[k] >> 16, WC) + (unsigned)(xpk[k] & 0xFFFF);
                if constexpr (IN_U8) xreg[k] = img8[off];
                else xreg[k] = ((const unsigned *)img8)[off];
            }
        } else {                                                      // clamped address; the elements outside the image are replaced
#pragma unroll                                                        // where the registers are consumed (a select here would wait for the loads)
            for (int k = 0; k < C::XREGS; ++k) {
                const int gy = g.iy0 + (xpk[k] >> 16), gf = fx0 + (xpk[k] & 0xFFFF);
                const unsigned off = (unsigned)(min(max(gy, 0), H - 1) * WC + min(max(gf, 0), WC - 1));
                if constexpr (IN_U8) xreg[k] = img8[off];
                else xreg[k] = ((const unsigned *)img8)[off];
            }
        }
    };

    // GSRC 0 reads its G tile (= the staged D tile) throughout phase 2, so there the D tile cannot be prefetched under
    // phase 2: it is fetched together with the X patch at the top of the tile
    constexpr bool D_AHEAD = (GSRC != 0);
    constexpr bool X_AHEAD = (C::XBUF == 2);
    int xb = 0;                                                         // X buffer of the current tile
    int tile = blockIdx.x;
    if (tile < total) {
        const geom g0 = tile_geom(tile);
        if constexpr (D_AHEAD) stage_dm(g0);
        if constexpr (XDMA) stage_xd(g0, 0);
        else if constexpr (CIN != UBD_C) load_x(g0);
        if constexpr (X_AHEAD) stage_x(g0, 0);
    }
    for (; tile < total; tile += gridDim.x) {
        const geom g = tile_geom(tile);
        const int img = g.img, oy0 = g.oy0, ox0 = g.ox0, ix0 = g.ix0, iy0 = g.iy0;
        const bool xborder = g.xborder;
        const char *xraw = dma + xb * (C::XI * 1024);                  // 24-channel patch (bf16)
        SBSTAMP(0);
        // previous tile's phase 2 is done: X patch / xf32 are free.  XDMA: nothing is written before the barrier behind the wait
        // below, and the next tile's patch goes into the other buffer, last read two barriers ago
        if constexpr (!XDMA) __syncthreads();
        SBSTAMP(1);
        if constexpr (!D_AHEAD) stage_dm(g);
        if constexpr (CIN == UBD_C) {
            if constexpr (!X_AHEAD) stage_x(g, 0);
        } else if constexpr (!XDMA) {
            const bool plain = !IN_U8 && pre_sub == 0.f && pre_div == 1.f;      // already preprocessed fp32 input: a copy (no 12-instruction division)
            if (xborder) {                                             // block-uniform: outside the image = exactly 0 after the preprocessing
#pragma unroll
                for (int k = 0; k < C::XREGS; ++k) {
                    const int gy = iy0 + (xpk[k] >> 16), gf = ix0 * CIN + (xpk[k] & 0xFFFF);
                    const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gf < (unsigned)(W * CIN);
                    xreg[k] = inside ? xreg[k] : (IN_U8 ? 0x100u : __builtin_bit_cast(unsigned, pre_sub));
                }
            }
#pragma unroll
            for (int k = 0; k < C::XREGS; ++k) {
                const int e = k * C::NT + (int)threadIdx.x;
                if (e < C::XPIX * CIN) {
                    if constexpr (IN_U8) xf32[e] = xreg[k] > 255u ? 0.f : ((float)xreg[k] - pre_sub) / pre_div;
                    else xf32[e] = plain ? __builtin_bit_cast(float, xreg[k]) : (__builtin_bit_cast(float, xreg[k]) - pre_sub) / pre_div;
                }
            }
        }
        // The DMA is issued as asm (ubd_glds16, common.h): hipcc neither waits for it here nor drains the NEXT tile's D
        // DMA in front of phase 2's LDS accesses (it did with the builtin: every tile waited a full fetch latency there).
        SBSTAMP(2);
        // this tile's DMA has landed.  X_AHEAD kernels: it was requested in front of the previous tile's phase 2, whose four dDW
        // stores per wave are younger and need not be waited for (the counter retires in order; round 2 stamps: 1.1 k cycles
        // of every 10.6 k-cycle tile went into draining them)
        if (X_AHEAD && tile != (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // ... for every wave; LDS writes visible
        SBSTAMP(3);
        // ---- ReLU bits of the X pixels this tile owns (24-channel layers: X is the saved output of the layer below, whose kernel runs
        //      next and needs nothing else of it).  Owned = the TH*STRIDE x 16*STRIDE pixels at patch (pad_lo, pad_lo); pixels outside the
        //      map are zero in the patch.  The store is older than the next tile's DMA requests: the vmcnt(4) accounting stays.
        if constexpr (CIN == UBD_C) {
            if (xbits) {                                                // kernel-uniform
                constexpr int OWNW = 16 * STRIDE, OWN = C::TH * STRIDE * OWNW;
#pragma unroll
                for (int k = 0; k < (OWN + C::NT - 1) / C::NT; ++k) {
                    const int p = k * C::NT + (int)threadIdx.x;
                    if (OWN % C::NT != 0 && p >= OWN) break;            // wave-uniform (OWN is a multiple of 64)
                    const int rr = p / OWNW, cc = p % OWNW;
                    const char *px = xraw + ((rr + pad_lo) * C::PW + cc + pad_lo) * 48;
                    const u32x4 v0 = *(const u32x4 *)px, v1 = *(const u32x4 *)(px + 16), v2 = *(const u32x4 *)(px + 32);
                    unsigned word = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        word |= (relu_pair_bits(v0[d]) << d) | (relu_pair_bits(v1[d]) << (4 + d)) | (relu_pair_bits(v2[d]) << (8 + d));
                    const int y = oy0 * STRIDE + rr, x = ox0 * STRIDE + cc;
                    if (y < H && x < W) xbits[((size_t)img * H + y) * W + x] = word;
                }
            }
        }
        // ---- phase 1: G tile in T (GSRC 0: the staged G3 tile is used as it is)
        if constexpr (GSRC != 0) {
#pragma unroll
        for (int kr = 0; kr < C::TH / C::NW; ++kr) {
            const int r = wid + C::NW * kr;
            float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const unsigned mword = *(const unsigned *)(mraw + (r * 16 + i) * 4);   // this pixel's ReLU bits; the lane's channels: 4q .. 4q+3, 16+2q, 17+2q
            if constexpr (UREG) {
                const char *rowb = draw + r * (C::DCOLS * 48);
                u32x4 b0[5], b1[3];
#pragma unroll
                for (int j = 0; j < 5; ++j) b0[j] = *(const u32x4 *)(rowb + gb0[j]);
#pragma unroll
                for (int j = 0; j < 3; ++j) b1[j] = *(const u32x4 *)(rowb + gb1[j]);
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 5; ++j) c0 = mfma16<T>(ga0[j], b0[j], c0);
#pragma unroll
                for (int j = 0; j < 3; ++j) c1 = mfma16<T>(ga1[j], b1[j], c1);
                acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3]; acc[4] = c1[0]; acc[5] = c1[1];
            } else if constexpr (UREG2) {
                const char *rowb = draw + kr * (2 * C::DCOLS * 48);        // row wid + 4 kr: two D rows further down
                const char *zero16 = lds + C::OFF_CONST + 16;
                const int e0 = i - 8 * (q & 1), s0 = (e0 >= 0 && e0 < 8) ? (e0 >> 1) : -1;
                const int s1 = (i & 3) < 2 ? (i >> 2) : -1;
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                u32x4 b0[3], b1[2];
#pragma unroll
                for (int j = 0; j < 3; ++j) b0[j] = *(const u32x4 *)(g2b0[j] >= 0 ? rowb + g2b0[j] : zero16);
#pragma unroll
                for (int j = 0; j < 2; ++j) b1[j] = *(const u32x4 *)(g2b1[j] >= 0 ? rowb + g2b1[j] : zero16);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j == 2 && g2par) break;                         // wave-uniform: three taps fill two MFMAs (channels 0..15)
                    const u32x4 wa = {s0 == 0 ? g2d0[j] : 0u, s0 == 1 ? g2d0[j] : 0u, s0 == 2 ? g2d0[j] : 0u, s0 == 3 ? g2d0[j] : 0u};
                    c0 = mfma16<T>(wa, b0[j], c0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && g2par) break;                         // ... and one (channels 16..23)
                    const u32x4 wa = {s1 == 0 ? g2d1[j] : 0u, s1 == 1 ? g2d1[j] : 0u, s1 == 2 ? g2d1[j] : 0u, s1 == 3 ? g2d1[j] : 0u};
                    c1 = mfma16<T>(wa, b1[j], c1);
                }
                acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3]; acc[4] = c1[0]; acc[5] = c1[1];
            } else {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (((r + pad_up - ky) & 1) != 0) continue;         // wave-uniform: row parity (oy0 is even)
                const int dr = ((r + pad_up - ky) >> 1) + 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kx = par + 2 * j, dc = ((i + pad_up - par) >> 1) + 1 - j;   // kx = 3: zero row of the table
                    const char *pd = draw + (dr * C::DCOLS + dc) * 48;
                    const u32x2 a = *(const u32x2 *)(pd + 8 * q);
                    const unsigned b = *(const unsigned *)(pd + 32 + 4 * q);
                    const unsigned *pu = utp + (ky * 4 + kx) * UBD_C;
                    const u32x4 w4 = *(const u32x4 *)(pu + 4 * q);
                    const u32x2 w2 = *(const u32x2 *)(pu + 16 + 2 * q);
                    acc[0] = dot2b<T>(a[0], w4[0], acc[0]); acc[1] = dot2b<T>(a[0], w4[1], acc[1]);
                    acc[2] = dot2b<T>(a[1], w4[2], acc[2]); acc[3] = dot2b<T>(a[1], w4[3], acc[3]);
                    acc[4] = dot2b<T>(b, w2[0], acc[4]);    acc[5] = dot2b<T>(b, w2[1], acc[5]);
                }
            }
            }

            u32x2 g4 = {pack2b<T>(acc[0], acc[1]), pack2b<T>(acc[2], acc[3])};
            unsigned g2 = pack2b<T>(acc[4], acc[5]);
            // ReLU mask: the pair's two bits become the 0 / 1 factors of a packed integer multiply
            g4[0] = relu_bits2(g4[0], mword, 2 * q); g4[1] = relu_bits2(g4[1], mword, 2 * q + 1); g2 = relu_bits2(g2, mword, 8 + q);   // pairs 2q, 2q + 1, 8 + q
            char *pg = g16 + (r * 16 + i) * 48;
            *(u32x2 *)(pg + 8 * q) = g4;
            *(unsigned *)(pg + 32 + 4 * q) = g2;
        }
        }
        SBSTAMP(4);
        __syncthreads();
        SBSTAMP(5);
        if (tile + (int)gridDim.x < total) {                           // block-uniform: next tile's D / mask (and 1/3-channel input)
            const geom gn = tile_geom(tile + gridDim.x);
            if constexpr (D_AHEAD) stage_dm(gn);
            if constexpr (XDMA) stage_xd(gn, xb ^ 1);
            else if constexpr (CIN != UBD_C) load_x(gn);
            if constexpr (X_AHEAD) stage_x(gn, xb ^ 1);
        }
        SBSTAMP(6);

        // ---- phase 2: two row tiles (one k-block of 32 pixels) per step
        const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;   // roles in the transposed reads (bwd16.h)
        const float *xt = XDMA ? (const float *)(lds + xb * (C::XDI * 1024)) + xskew : xf32;   // fp32 patch of this tile (1/3 channels)
#pragma unroll 1
        for (int rp = 0; rp < C::TH / (2 * C::NW); ++rp) {
            const int r0 = wid + C::NW * (2 * rp), r1 = r0 + C::NW;
            if constexpr (M2) {
#pragma unroll 1
            for (int hr = 0; hr < 2; ++hr) {
                const int r = hr ? r1 : r0;
                const int oy = oy0 + r, ox = ox0 + i;
                // dDW[pixel i][ch] = sum_co G[i][co] pw[ch][co] (rows = this lane's channels), stored and kept in T
                u32x4 gb = *(const u32x4 *)(g16 + (r * 16 + i) * 48 + 16 * (q < 3 ? q : 0));
                if (q == 3) gb = u32x4{0u, 0u, 0u, 0u};
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 dA = mfma16<T>(apwb[0], gb, z4), dB = mfma16<T>(apwb[1], gb, z4);
                // depthwise recompute: sum_taps diag(w_tap) X^T, taps folded into K (8 reads + 8 MFMAs)
                const char *rowb = xraw + r * (STRIDE * C::PW * 48);
                u32x4 b0[5], b1[3];
#pragma unroll
                for (int j = 0; j < 5; ++j) b0[j] = *(const u32x4 *)(rowb + xo0[j]);
#pragma unroll
                for (int j = 0; j < 3; ++j) b1[j] = *(const u32x4 *)(rowb + xo1[j]);
                f32x4 c0 = z4, c1 = z4;
#pragma unroll
                for (int j = 0; j < 5; ++j) c0 = mfma16<T>(wa0[j], b0[j], c0);
#pragma unroll
                for (int j = 0; j < 3; ++j) c1 = mfma16<T>(wa1[j], b1[j], c1);
                const u32x2 d4 = {pack2b<T>(dA[0], dA[1]), pack2b<T>(dA[2], dA[3])};
                const unsigned d2 = pack2b<T>(dB[0], dB[1]);
                char *pdd = sdd + (hr * 16 + i) * 48;
                *(u32x2 *)(pdd + 8 * q) = d4;
                *(unsigned *)(pdd + 32 + 4 * q) = d2;
                {
                    // 8 bytes at channel 4q, 4 bytes at channel 16 + 2q.  Buffer stores: pixels outside the map get an offset past the
                    // tensor and are dropped, so every wave issues exactly FOUR stores per tile -- the wait at the top of the next
                    // tile counts on that (vmcnt(4): the DMA has landed, the stores may still be on their way)
                    const unsigned po = (oy < OH && ox < OW) ? (unsigned)(((img * OH + oy) * OW + ox) * (UBD_C * 2)) : 0xFFFFFF00u;
                    __builtin_amdgcn_raw_buffer_store_b64(d4, ddw_rsrc, (int)(po + 8u * q), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(d2, ddw_rsrc, (int)(po + 32u + 4u * q), 0, 0);
                }
                char *ps = sdw + (hr * 16 + i) * 48;                   // depthwise output in T (as the forward pass stored it)
                *(u32x2 *)(ps + 8 * q) = u32x2{pack2b<T>(c0[0], c0[1]), pack2b<T>(c0[2], c0[3])};
                *(unsigned *)(ps + 32 + 4 * q) = pack2b<T>(c1[0], c1[1]);
            }
            } else {
#pragma unroll HR_UNROLL
            for (int hr = 0; hr < 2; ++hr) {                           // 1/3 channels: VALU tap pass (27 products per pixel)
                const int r = hr ? r1 : r0;
                // dDW[pixel i][channel q] = sum_co G[i][co] pw[q][co] (row 4q of the product), then ONE pass over the taps feeds
                // both the depthwise recompute (for dpw) and the depthwise-kernel gradient
                u32x4 gb = *(const u32x4 *)(g16 + (r * 16 + i) * 48 + 16 * (q < 3 ? q : 0));
                if (q == 3) gb = u32x4{0u, 0u, 0u, 0u};
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const float ddwv = mfma16<T>(apwb[0], gb, z4)[0];
                float dwv = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int t = ky * 3 + kx;
                        const float v = xt[(r * STRIDE + ky) * C::XROW + (i * STRIDE + kx) * CIN + cb];
                        dwv = fmaf(v, dwk1[t], dwv);                    // dwk1 is zero for lanes without a channel
                        ddw[t][0] = fmaf(v, ddwv, ddw[t][0]);
                    }
                // depthwise output in T (as the forward pass stored it) into this wave's [32 pixels][channel, 1, 0 ..] image
                if (ch_ok) ((unsigned short *)sdw)[(hr * 16 + i) * C::SDW_W + cb] = __builtin_bit_cast(unsigned short, (T)dwv);
            }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // dpw[ch][co] += sum over the 32 pixels DW[pixel][ch] G[pixel][co]; an all-ones column of the A operand gives db.
            // Transposed reads: lane 4qq+pp of group grp supplies the address of pixel k = 8 grp + 4 j + qq, segment pp.
            {
                u32x4 am[MT_PW], bn[2];
#pragma unroll
                for (int mt = 0; mt < MT_PW; ++mt) {
                    s16x4 h[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = 8 * grp + 4 * j + qq;
                        const char *pa;
                        if constexpr (CIN == UBD_C) pa = mt == 0 ? sdw + k * 48 + 8 * pp : (pp < 2 ? sdw + k * 48 + 32 + 8 * pp : (pp == 2 ? c_ones : c_zero));
                        else pa = pp == 0 ? sdw + k * (C::SDW_W * 2) : c_zero;
                        h[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pa);
                    }
                    am[mt] = __builtin_bit_cast(u32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    s16x4 h[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = 8 * grp + 4 * j + qq;
                        const char *pg = g16 + (((k < 16 ? r0 : r1) * 16 + (k & 15)) * 48);
                        const char *pb = nt == 0 ? pg + 8 * pp : (pp < 2 ? pg + 32 + 8 * pp : c_zero);
                        h[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pb);
                    }
                    bn[nt] = __builtin_bit_cast(u32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int mt = 0; mt < MT_PW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) accpw[mt][nt] = mfma16<T>(am[mt], bn[nt], accpw[mt][nt]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (M2) {
            // ddw: B = dDW^T of a k-block (channels 0..15; channels 16..23 twice), A = X^T shifted by the unit's tap; pixel k of
            // k-block kb sits at patch pixel ((k < 16 ? kb : kb + NW) * STRIDE + ky) * PW + (k & 15) * STRIDE + kx.  Channels
            // 16..23: tap pairs (0,1) (3,4) (6,7) (2,5) (8,-): segments pp < 2 read the first tap, pp >= 2 the second (one pixel
            // to the right, or one patch row down for (2,5); the partner of tap 8 reads finite padding and is never used).
            SBSTAMP(7);
            __syncthreads();                                           // every wave's dDW image is complete
            SBSTAMP(8);
            int ka[2];                                                 // this lane's pixel of read j, relative to the k-block's first row
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = 8 * grp + 4 * j + qq;
                ka[j] = ((k < 16 ? 0 : C::NW * STRIDE * C::PW) + (k & 15) * STRIDE) * 48;
            }
#pragma unroll 1
            for (int kb = 0; kb < C::NW; ++kb) {
                const char *sddk = lds + C::OFF_SDD + kb * C::SDW_BYTES;
                u32x4 bd[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    s16x4 h[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = 8 * grp + 4 * j + qq;
                        const char *pb = sddk + k * 48 + (nt == 0 ? 8 * pp : 32 + 8 * (pp & 1));
                        h[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pb);
                    }
                    bd[nt] = __builtin_bit_cast(u32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
                }
                const char *xr0 = xraw + kb * (STRIDE * C::PW * 48);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    if (sl == 3 && wid >= 2) break;                    // wave-uniform: units 14, 15 do not exist
                    s16x4 h[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        h[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(xr0 + ka[j] + uo[sl]));
                    const bool low = sl < 2 || (sl == 2 && wid == 0);  // wave-uniform: unit < 9 -> channels 0..15
                    const u32x4 bsel = low ? bd[0] : bd[1];
                    accdw[sl] = mfma16<T>(__builtin_bit_cast(u32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7)), bsel, accdw[sl]);
                }
            }
        }
        SBSTAMP(9);
#ifdef UBD_STAMPS
        ++stamp_it;
#endif
        if constexpr (X_AHEAD || XDMA) xb ^= 1;
    }
#undef SBSTAMP
    // ---- flush: wave-sequential reduction of the per-lane sums into this block's row of the partial-sum matrix
    //      row layout: [9*CIN depthwise | CIN*24 pointwise | 24 bias]
    __syncthreads();
    float *red = (float *)lds;
    for (int t = threadIdx.x; t < C::PART; t += blockDim.x) red[t] = 0.f;
    __syncthreads();
    if constexpr (!M2) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = ddw[t][0];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            ddw[t][0] = v;
        }
    }
    for (int ph = 0; ph < C::NW; ++ph) {
        if (wid == ph) {
            if constexpr (M2) {
                // diagonals of this wave's ddw accumulators: column i, row i sits in lane (i, q = i >> 2), element i & 3
                if (q == (i >> 2)) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int u = wid + 4 * sl;
                        const float v = (i & 3) == 0 ? accdw[sl][0] : ((i & 3) == 1 ? accdw[sl][1] : ((i & 3) == 2 ? accdw[sl][2] : accdw[sl][3]));
                        if (u < 9) red[u * UBD_C + i] += v;
                        else if (u < 14) {
                            const int jj = u - 9, t0 = jj < 3 ? 3 * jj : (jj == 3 ? 2 : 8), t1 = jj < 3 ? 3 * jj + 1 : (jj == 3 ? 5 : -1);
                            const int t = i < 8 ? t0 : t1;
                            if (t >= 0) red[t * UBD_C + 16 + (i & 7)] += v;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (i == 0 && ch_ok) red[t * CIN + cb] += ddw[t][0];
            }
#pragma unroll
            for (int mt = 0; mt < MT_PW; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * q + r, col = i + 16 * nt;
                        if (col < UBD_C && row <= CIN) red[9 * CIN + row * UBD_C + col] += accpw[mt][nt][r];   // row CIN = bias
                    }
        }
        __syncthreads();
    }
    float *prow = partials + (size_t)blockIdx.x * C::PART;
    for (int t = threadIdx.x; t < C::PART; t += blockDim.x) prow[t] = red[t];
    __syncthreads();                                   // the LDS image is free
    rp_reduce_tail(prev, (float *)lds);                // the partial rows of the producer in front of this kernel (backward.hip)
}
