// L1 -> L2 -> L3 of the 16-bit pass in ONE kernel (included by fwd16.hip; round 4).
//
// net.py:292-296: ZeroPadding2D + SeparableConv2D(24, 3x3, stride 2, relu) (L1), SeparableConv2D(24, 3x3, 'same', relu) (L2),
// ZeroPadding2D + SeparableConv2D(24, 3x3, stride 2, relu) (L3).  sep12_16_kernel (fwd16.hip) already computes L2's input patch
// from the image; L3 then read L2's activation back from memory (201 MB per 64 images of 512 x 512, 101 MB per 8 images of
// 1024 x 1024 -- and an inference pass wrote those bytes only to read them once).  Here the tile of L2 outputs stays in LDS and
// L3 is computed from it:
//     8 x 8 L3 outputs  <-  17 x 17 L2 outputs (LDS)  <-  19 x 19 L1 outputs (LDS)  <-  39 x 48 image pixels (LDS)
// Neither a1 nor a2 is read from memory; an inference pass writes neither (the train step stores both, every pixel once, by the
// tile that owns it: the backward pass reads them).  Same arithmetic as sep12_16_kernel + sepconv16_kernel<24, 2>, operation for
// operation (depthwise of L1 on the VALU in tap order, tap-folded diagonal MFMAs for L2 / L3, rounding to T where the split pass
// stores T): a1, a2, a3 are bit-identical to the split pass (tests/test_gpu_forward16.py; UBD_STEM16=fused12 / split keep it).
// Price: L1 on 361 and L2 on 289 pixels per 256 owned ones.  L2's tile is walked as 19 units of 16 pixels: its 17 rows (columns 0-15), its
// 17th column (rows 0-15) and the corner pixel (the B operand of a diagonal MFMA is a per-lane LDS address, so a unit need not be a
// row); L3's 64 outputs are one unit per wave.
// LDS: 17.7 + 14.6 + 0.8 + 20.6 KB (RGB fp32 image patch) = 53.6 KB = 42 of the CU's 128 granules of 1280 bytes: three blocks per CU.
// Per tile: image patch in LDS | barrier | L1 -> a1 patch (0 outside L1's map = L2's 'same' padding) | barrier | request the next
// tile's image patch | L2 -> L2 tile (0 outside L2's map = L3's padding) | barrier | L3 -> memory.
#pragma once

template <int CIN> struct sep123_cfg {
    static constexpr int AP = 19;                                  // a1 patch side
    static constexpr int TP = 17;                                  // L2 tile side
    static constexpr int XP = 2 * (AP - 1) + 3;                    // image patch rows: 39
    // patch columns: the 39 that are needed start at image column 32 tx - 2 - 3 pad; the LDS image starts at 32 tx - 4 - 4 pad, so every
    // patch row begins on a 16-byte boundary of the image row (H, W multiples of 4) and is moved in 16-byte pieces: 44 columns
    static constexpr int XW = 44;
    static constexpr int ROWF = XW * CIN;                          // floats per patch row: 132 (RGB) / 44 (grey)
    static constexpr int RC = ROWF / 4;                            // 16-byte chunks per row
    static constexpr int CHUNKS = XP * RC;                         // 1287 / 429
    static constexpr int ROUNDS = (CHUNKS + 255) / 256;            // chunks per thread: 6 / 2
    // The two LDS images the diagonal MFMAs read are PLANAR (round 4): the three 16-byte chunks of a pixel (channels 0-7 / 8-15 / 16-23) live
    // in three planes of 16 bytes per pixel, a multiple of 256 bytes apart.  ds_read_b128 serves a wave in four groups of 16 lanes that are NOT
    // consecutive ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md): a group mixes eight pixels of k-group q with the other eight pixels of
    // k-group q + 1, which read DIFFERENT chunks; with pixel-major 48-byte pixels five of a group's sixteen lanes collide for every
    // alignment (PMC: SQ_LDS_BANK_CONFLICT = 52 % of SQ_LDS_IDX_ACTIVE, the LDS array busy 68 % of the kernel).  In the planar image a
    // lane's bank depends on its pixel only: a group whose two k-groups read the same tap (five of a unit's eight reads) is conflict-free
    // as long as the unit's sixteen pixels have distinct indices mod 16 -- true for a row of 16 and for a column (pitch 19: 3 i mod 16).
    // Both images are written by the kernel itself, so the layout is free (a DMA-filled tile is not: fwd16.hip dilconv16s).
    static constexpr int A1_PLANE = (AP * AP * 16 + 255) / 256 * 256;      // 5888
    static constexpr int T2_PLANE = (TP * TP * 16 + 255) / 256 * 256;      // 4864
    static constexpr int A1_BYTES = 3 * A1_PLANE;                  // 17664
    static constexpr int T2_BYTES = 3 * T2_PLANE;                  // 14592
    static constexpr int SPARE_BYTES = 512;                        // where masked lanes write (8 bytes per lane)
    static constexpr int BIAS_BYTES = 3 * UBD_C * 4;               // the three layers' biases
    static constexpr int XP_BYTES = CHUNKS * 16;                   // 20592 / 6864
    static constexpr int UNITS1 = (AP * AP + 15) / 16;             // 23
    static constexpr int UPW1 = (UNITS1 + 3) / 4;                  // 6
    static constexpr int UNITS2 = TP + 2;                          // 17 rows of 16 pixels, the 17th column as a unit, the corner pixel: 19
    static constexpr int UPW2 = (UNITS2 + 3) / 4;                  // 5
    // gfx950 hands LDS out in granules of 1280 bytes (128 per CU): three blocks per CU need <= 42 granules = 53760 bytes each
    static constexpr int SMEM = A1_BYTES + T2_BYTES + SPARE_BYTES + BIAS_BYTES + XP_BYTES;   // 53648 / 39920
    static_assert(SMEM <= 42 * 1280, "three blocks per CU");
};

// The operands of a 24-channel separable layer (tap-folded diagonal depthwise fragments, pointwise A operands) come READY per lane
// from the workspace (pack.h pack_sep16_ready_body: what sepconv16_kernel<24, S> builds in its prologue).  Both layers' sets do not
// fit the registers of three waves per SIMD (2 x 40), and loading a set per phase queues the loads behind the tile's stores (vmcnt
// retires in order: in-kernel stamps had 35 % of the train step's tile period in that wait).  A diagonal fragment, however, holds ONE
// non-zero 16-bit value per lane: the kernel keeps each layer COMPACT -- eight values already shifted into their half + the six
// pointwise dwords = 14 registers -- and expands the 40-register form at the head of the layer's phase with 32 selects.
struct sep123_compact { unsigned d0[5], d1[3]; u32x2 pa[2]; unsigned pb[2]; };
__device__ __forceinline__ sep123_compact sep123_load_compact(const u32x4 *__restrict__ ready, int lane)
{
    sep123_compact c;
#pragma unroll
    for (int j = 0; j < 5; ++j) { const u32x4 v = ready[j * 64 + lane]; c.d0[j] = v[0] | v[1] | v[2] | v[3]; }       // one non-zero half: kept in place
#pragma unroll
    for (int j = 0; j < 3; ++j) { const u32x4 v = ready[(5 + j) * 64 + lane]; c.d1[j] = v[0] | v[1] | v[2] | v[3]; }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { const u32x4 v = ready[(8 + nt) * 64 + lane]; c.pa[nt] = u32x2{v[0], v[1]}; c.pb[nt] = v[2]; }
    return c;
}
// the byte offsets of the B operands are those of the TAPS (row pitch `pitch` pixels, 16 bytes per pixel) and of the chunk's PLANE: the
// caller adds its pixel's own
__device__ __forceinline__ void sep123_expand(sep123_compact c, int pitch, int plane, int lane, u32x4 (&wa0)[5], u32x4 (&wa1)[3],
                                              u32x4 (&pwb)[2], int (&xo0)[5], int (&xo1)[3])
{
    // opaque copies: the expansion is loop-invariant, and hoisted out of the tile loop it would pin both 40-register sets
#pragma unroll
    for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(c.d0[j]));
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(c.d1[j]));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { asm volatile("" : "+v"(c.pa[nt])); asm volatile("" : "+v"(c.pb[nt])); }
    const int i = lane & 15, q = lane >> 4;
    const int e0 = i - 8 * (q & 1);                                  // wa0: k-slot e0 (valid 0..7) -> dword e0 >> 1
    const int s0 = (e0 >= 0 && e0 < 8) ? (e0 >> 1) : -1;
    const int s1 = (i & 3) < 2 ? (i >> 2) : -1;                      // wa1: k-slot 2 (i >> 2) + (i & 3) -> dword i >> 2
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        wa0[j] = u32x4{s0 == 0 ? c.d0[j] : 0u, s0 == 1 ? c.d0[j] : 0u, s0 == 2 ? c.d0[j] : 0u, s0 == 3 ? c.d0[j] : 0u};
        const int ts = 2 * j + (q >> 1), t = ts < 9 ? ts : 8;
        xo0[j] = ((t / 3) * pitch + t % 3) * 16 + (q & 1) * plane;          // planar image: 16 bytes per pixel, chunk = plane
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        wa1[j] = u32x4{s1 == 0 ? c.d1[j] : 0u, s1 == 1 ? c.d1[j] : 0u, s1 == 2 ? c.d1[j] : 0u, s1 == 3 ? c.d1[j] : 0u};
        const int ts = 4 * j + q, t = ts < 9 ? ts : 8;
        xo1[j] = ((t / 3) * pitch + t % 3) * 16 + 2 * plane;
    }
    pwb[0] = u32x4{c.pa[0][0], c.pa[0][1], c.pb[0], 0u};
    pwb[1] = u32x4{c.pa[1][0], c.pa[1][1], c.pb[1], 0u};
}

// one 16-pixel unit of a 24-channel separable layer: depthwise on the matrix pipe (8 MFMAs), rounded to T, pointwise (2 MFMAs), bias,
// rounding, ReLU on the packed pairs -- the arithmetic of sepconv16_kernel / store_tile16_t<T, 0>.  pix: this lane's pixel in the source
// LDS image (byte address of its top-left tap); returns the lane's eight / four output channels as packed pairs.
template <typename T>
__device__ __forceinline__ void sep123_unit(const char *pix, const u32x4 (&wa0)[5], const u32x4 (&wa1)[3], const u32x4 (&pwb)[2],
                                            const int (&xo0)[5], const int (&xo1)[3], f32x4 bA, f32x4 bB, u32x2 &o0, u32x2 &o1)
{
    u32x4 b0[5], b1[3];
#pragma unroll
    for (int j = 0; j < 5; ++j) b0[j] = *(const u32x4 *)(pix + xo0[j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) b1[j] = *(const u32x4 *)(pix + xo1[j]);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = z4, c1 = z4;
#pragma unroll
    for (int j = 0; j < 5; ++j) c0 = h16<T>::mfma(wa0[j], b0[j], c0);
#pragma unroll
    for (int j = 0; j < 3; ++j) c1 = h16<T>::mfma(wa1[j], b1[j], c1);
    const u32x4 av = {pack2<T>(c0[0], c0[1]), pack2<T>(c0[2], c0[3]), pack2<T>(c1[0], c1[1]), 0u};
    f32x4 acc0 = h16<T>::mfma(pwb[0], av, z4);
    f32x4 acc1 = h16<T>::mfma(pwb[1], av, z4);
    acc0 += bA; acc1 += bB;
    o0 = u32x2{relu_pk16(pack2<T>(acc0[0], acc0[1])), relu_pk16(pack2<T>(acc0[2], acc0[3]))};
    o1 = u32x2{relu_pk16(pack2<T>(acc1[0], acc1[1])), relu_pk16(pack2<T>(acc1[2], acc1[3]))};
}

// WRITE_A12: the train step keeps L1's and L2's activations (the backward pass reads them)
template <int CIN, int IN_MODE, bool PLAIN, bool WRITE_A12, typename T>
#ifndef S123_OCC
#define S123_OCC 3
#endif
__global__ __launch_bounds__(256, S123_OCC) void sep123_16_kernel(
    const void *__restrict__ xin, unsigned short *__restrict__ a1out, unsigned short *__restrict__ a2out, unsigned short *__restrict__ y,
    const float *__restrict__ frag1, const float *__restrict__ bias1, const u32x4 *__restrict__ ready23, const float *__restrict__ bias2,
    const float *__restrict__ bias3, int n, int H, int W, int H2, int W2, int H4, int W4, int pad_lo,
    float pre_sub, float pre_div
#ifdef UBD_STAMPS
    , unsigned long long *__restrict__ stamps
#endif
    )
{
#ifdef UBD_STAMPS   // diagnostic build only: s_memtime of lane 0 of every wave at the phase boundaries of its first 16 tiles
#define S3STAMP(k) do { if (stamps && it < 16 && (threadIdx.x & 63) == 0 && blockIdx.x < 768) stamps[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + it) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S3STAMP(k) do {} while (0)
#endif
    using C = sep123_cfg<CIN>;
    constexpr int AP = C::AP, TP = C::TP;
    static_assert(!PLAIN || IN_MODE == 0, "LDS-DMA moves fp32 pixels only");
    __shared__ __attribute__((aligned(16))) char smem[C::SMEM];          // ONE LDS object (see fwd16.hip)
    char *a1p = smem, *t2p = smem + C::A1_BYTES, *sparep = smem + C::A1_BYTES + C::T2_BYTES;
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
    const int cb = (q < CIN) ? q : 0;
    const int spare_off = 8 * lane;

    const int tiles_x = (W4 + 7) >> 3, tiles_y = (H4 + 7) >> 3;
    const int total = n * tiles_y * tiles_x;
    ubd_tile_decoder tdec;
    tdec.init(tiles_x, tiles_y, total);
    // tile -> image, first L3 row / column; the L2 tile starts at (2 oy3 - pad, 2 ox3 - pad), the a1 patch one pixel up / left of that
    auto tile_coords = [&](int tile, int &img, int &oy3, int &ox3) {
        int tx, ty;
        tdec.decode(tile, tx, ty, img);
        oy3 = ty * 8; ox3 = tx * 8;
    };
    const int WC = W * CIN;
    const int dx0 = 2 + pad_lo;                                            // patch column of the first column that is needed
    const unsigned lds_xp = ubd_lds_addr(smem + C::A1_BYTES + C::T2_BYTES + C::SPARE_BYTES + C::BIAS_BYTES);
    const float *xp = (const float *)(smem + C::A1_BYTES + C::T2_BYTES + C::SPARE_BYTES + C::BIAS_BYTES);
    float *biasp = (float *)(smem + C::A1_BYTES + C::T2_BYTES + C::SPARE_BYTES);
    if (threadIdx.x < 3 * UBD_C) biasp[threadIdx.x] = threadIdx.x < UBD_C ? bias1[threadIdx.x] : (threadIdx.x < 2 * UBD_C ? bias2[threadIdx.x - UBD_C] : bias3[threadIdx.x - 2 * UBD_C]);   // visible after the first barrier
    auto dma_x = [&](int tile) {                                           // PLAIN
        int img, oy3, ox3;
        tile_coords(tile, img, oy3, ox3);
        const int iy0 = (2 * oy3 - pad_lo - 1) * 2 - pad_lo, if0 = (ox3 * 4 - 4 - 4 * pad_lo) * CIN;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)xin + (size_t)img * H * WC), 0,
                                                                        (int)((unsigned)H * (unsigned)WC * 4u), 0x00020000);
        const unsigned dst = lds_xp + (unsigned)(wid * 1024);
#pragma unroll
        for (int k = 0; k < C::ROUNDS; ++k) {
            const int c = k * 256 + (int)threadIdx.x;
            const int pr = c / C::RC, gf = if0 + 4 * (c - pr * C::RC);
            // rows above / below the image fall out of the descriptor's range by themselves (a negative offset wraps); chunks left / right
            // of it get an out-of-range offset: zeros in LDS.  Only the lanes that have a chunk write (16 bytes at dst + 16 * lane).
            const unsigned off = (unsigned)gf < (unsigned)WC ? (unsigned)((ubd_mul24(iy0 + pr, WC) + gf) * 4) : 0x80000000u;
            if (k < C::ROUNDS - 1 || c < C::CHUNKS) ubd_blds16(rsrc, off, dst + (unsigned)(k * 4096));
        }
    };
    // !PLAIN: 4 bytes (uint8) / 16 bytes (fp32) per chunk through registers
    constexpr int SW = (IN_MODE == 1) ? 1 : 4;                             // dwords per chunk
    auto load_regs = [&](int iy0, int if0, int img, unsigned (&st)[C::ROUNDS][SW]) {
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)img * H * WC * ((IN_MODE == 1) ? 1 : 4);
        // every chunk from the clamped position, all loads in flight; the chunks outside the image are replaced afterwards
#pragma unroll
        for (int k = 0; k < C::ROUNDS; ++k) {
            int c = k * 256 + (int)threadIdx.x;
            c = c < C::CHUNKS ? c : C::CHUNKS - 1;
            const int pr = c / C::RC, pf = 4 * (c - pr * C::RC);
            const int gy = min(max(iy0 + pr, 0), H - 1), gf = min(max(if0 + pf, 0), WC - 4);
            const unsigned off = (unsigned)(ubd_mul24(gy, WC) + gf);
            if constexpr (IN_MODE == 1) st[k][0] = *(const unsigned *)(img8 + off);
            else {
                const u32x4 v = *(const u32x4 *)(img8 + (size_t)off * 4);
                st[k][0] = v[0]; st[k][1] = v[1]; st[k][2] = v[2]; st[k][3] = v[3];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    if constexpr (PLAIN) dma_x(tile);

    // L2 / L3 operands: compact for the whole launch, expanded per phase (the two expanded sets are never live together)
    const sep123_compact k2 = sep123_load_compact(ready23, lane), k3 = sep123_load_compact(ready23 + 10 * 64, lane);
    for (int it = 0;; ++it) {
        int img, oy3, ox3;
        tile_coords(tile, img, oy3, ox3);
        const int oy2 = 2 * oy3 - pad_lo, ox2 = 2 * ox3 - pad_lo;        // first row / column of the L2 tile
        const int nxt = tile + (int)gridDim.x;
        const bool has_next = nxt < total;                              // block-uniform
        S3STAMP(0);
        if constexpr (PLAIN) {
            // Counted wait (vmcnt counts stores too and retires in order): this tile's DMA was issued during the previous tile, and
            // behind it the wave issued that tile's stores: six 16-byte pieces of a1 / a2 when they are kept, two of L3
            constexpr int S = (WRITE_A12 ? 6 : 0) + 2;
            if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
            __builtin_amdgcn_s_barrier();                               // also: every wave has left the previous tile's L3 phase
        } else {
            unsigned stage[C::ROUNDS][SW];
            const int iy0 = (oy2 - 1) * 2 - pad_lo, if0 = (ox3 * 4 - 4 - 4 * pad_lo) * CIN;
            load_regs(iy0, if0, img, stage);
            // the previous tile's L1 phase (the readers of xp) ended at its second barrier: xp is free
#pragma unroll
            for (int k = 0; k < C::ROUNDS; ++k) {
                int c = k * 256 + (int)threadIdx.x;
                const bool live = c < C::CHUNKS;
                c = live ? c : C::CHUNKS - 1;
                const int pr = c / C::RC, pf = 4 * (c - pr * C::RC);
                const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(if0 + pf) < (unsigned)WC;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float raw;
                    if constexpr (IN_MODE == 1) raw = (float)((stage[k][0] >> (8 * e)) & 0xFFu);
                    else raw = __builtin_bit_cast(float, stage[k][e]);
                    v[e] = inside ? (raw - pre_sub) / pre_div : 0.f;
                }
                if (live) *(f32x4 *)((float *)xp + 4 * c) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();                               // also: every wave has left the previous tile's L3 phase
        }

        S3STAMP(1);
        // ---- L1 on this wave's units of 16 patch pixels
        {
            const f32x4 b1A = *(const f32x4 *)(biasp + 4 * q);
            const f32x4 b1B = q < 2 ? *(const f32x4 *)(biasp + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int j = 0; j < C::UPW1; ++j) {
                const int u = wid + 4 * j;                               // wave-uniform; slots >= UNITS1: every lane masked
                const int p = 16 * u + i;
                const bool valid = p < AP * AP;
                const int pp = valid ? p : AP * AP - 1;
                const int pr = ubd_div24<AP, AP * AP>(pp), pc = pp - ubd_mul24(pr, AP);          // 24-bit multiplies (common.h)
                const float *xb = xp + ubd_mul24(pr, 2 * C::ROWF) + ubd_mul24(2 * pc + dx0, CIN) + cb;
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
                const int gy = oy2 - 1 + pr, gx = ox2 - 1 + pc;
                const bool inmap = (unsigned)gy < (unsigned)H2 && (unsigned)gx < (unsigned)W2;
                u32x2 o0 = {relu_pk16(pack2<T>(acc0[0], acc0[1])), relu_pk16(pack2<T>(acc0[2], acc0[3]))};
                u32x2 o1 = {relu_pk16(pack2<T>(acc1[0], acc1[1])), relu_pk16(pack2<T>(acc1[2], acc1[3]))};
                if (!inmap) { o0 = u32x2{0u, 0u}; o1 = u32x2{0u, 0u}; }  // outside L1's map: L2's zero padding
                // channels 4q .. 4q + 3 = half (q & 1) of chunk q >> 1; channels 16 + 4q .. (q < 2) = half q of chunk 2
                *(u32x2 *)(valid ? a1p + (q >> 1) * C::A1_PLANE + pp * 16 + 8 * (q & 1) : sparep + spare_off) = o0;
                *(u32x2 *)((valid && q < 2) ? a1p + 2 * C::A1_PLANE + pp * 16 + 8 * q : sparep + spare_off) = o1;
            }
        }
        S3STAMP(2);
        u32x4 wa0[5], wa1[3], pwb[2];
        int xo0[5], xo1[3];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        S3STAMP(3);
        if constexpr (PLAIN) { if (has_next) dma_x(nxt); }                 // the image patch is free: filled during the L2 / L3 phases
        S3STAMP(4);

        if constexpr (WRITE_A12) {
            // the tile's own 16 x 16 pixels of L1's activation (map rows 2 oy3 .., columns 2 ox3 ..: patch rows / columns 1 + pad ..) leave
            // for memory: 768 contiguous bytes per row of the map (gathered from the three planes); three 16-byte pieces per thread
            __amdgpu_buffer_rsrc_t a1rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a1out + (size_t)img * H2 * W2 * UBD_C), 0,
                                                                            (int)((unsigned)H2 * (unsigned)W2 * (UBD_C * 2u)), 0x00020000);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = k * 256 + (int)threadIdx.x;
                const int row = c / 48, cc = c - row * 48;
                const int cpx = cc / 3, cch = cc - 3 * cpx;                          // pixel of the row, chunk
                const u32x4 v = *(const u32x4 *)(a1p + cch * C::A1_PLANE + (ubd_mul24(row + 1 + pad_lo, AP) + 1 + pad_lo + cpx) * 16);
                const int gy = 2 * oy3 + row, gx = 2 * ox3 + cc / 3;
                const bool in = gy < H2 && gx < W2;
                const int gpx = ubd_mul24(gy, W2) + 2 * ox3;                         // pixel index inside the image: x 48 as shifts (it may exceed 24 bits)
                __builtin_amdgcn_raw_buffer_store_b128(v, a1rs, in ? (gpx << 5) + (gpx << 4) + cc * 16 : (int)0x80000000u, 0, 0);
            }
        }

        // ---- L2 on this wave's units of 16 consecutive pixels of the 17 x 17 tile
        {
            sep123_expand(k2, AP, C::A1_PLANE, lane, wa0, wa1, pwb, xo0, xo1);
            const f32x4 b2A = *(const f32x4 *)(biasp + UBD_C + 4 * q);
            const f32x4 b2B = q < 2 ? *(const f32x4 *)(biasp + UBD_C + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef S123_UNROLL2
#define S123_UNROLL2 1
#endif
#pragma unroll S123_UNROLL2
            for (int j = 0; j < C::UPW2; ++j) {
                const int u = wid + 4 * j;                               // wave-uniform: 0..16 rows, 17 the last column, 18 the corner, 19 nothing
                const int pr = u < TP ? u : (u == TP ? i : TP - 1);
                const int pc = u < TP ? i : TP - 1;
                const bool valid = u < TP || (u == TP && i < TP - 1) || (u == TP + 1 && i == 0);
                const int pp = ubd_mul24(pr, TP) + pc;
                u32x2 o0, o1;
                sep123_unit<T>(a1p + (ubd_mul24(pr, AP) + pc) * 16, wa0, wa1, pwb, xo0, xo1, b2A, b2B, o0, o1);
                const int gy = oy2 + pr, gx = ox2 + pc;
                const bool inmap = (unsigned)gy < (unsigned)H2 && (unsigned)gx < (unsigned)W2;
                if (!inmap) { o0 = u32x2{0u, 0u}; o1 = u32x2{0u, 0u}; }  // outside L2's map: L3's zero padding
                *(u32x2 *)(valid ? t2p + (q >> 1) * C::T2_PLANE + pp * 16 + 8 * (q & 1) : sparep + spare_off) = o0;
                *(u32x2 *)((valid && q < 2) ? t2p + 2 * C::T2_PLANE + pp * 16 + 8 * q : sparep + spare_off) = o1;
            }
        }
        S3STAMP(5);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        S3STAMP(6);

        if constexpr (WRITE_A12) {
            // the tile's own 16 x 16 pixels of L2's activation: tile rows / columns pad .. pad + 15
            __amdgpu_buffer_rsrc_t a2rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a2out + (size_t)img * H2 * W2 * UBD_C), 0,
                                                                            (int)((unsigned)H2 * (unsigned)W2 * (UBD_C * 2u)), 0x00020000);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = k * 256 + (int)threadIdx.x;
                const int row = c / 48, cc = c - row * 48;
                const int cpx = cc / 3, cch = cc - 3 * cpx;
                const u32x4 v = *(const u32x4 *)(t2p + cch * C::T2_PLANE + (ubd_mul24(row + pad_lo, TP) + pad_lo + cpx) * 16);
                const int gy = 2 * oy3 + row, gx = 2 * ox3 + cc / 3;
                const bool in = gy < H2 && gx < W2;
                const int gpx = ubd_mul24(gy, W2) + 2 * ox3;
                __builtin_amdgcn_raw_buffer_store_b128(v, a2rs, in ? (gpx << 5) + (gpx << 4) + cc * 16 : (int)0x80000000u, 0, 0);
            }
        }

        // ---- L3: one unit per wave = tile rows 2 wid, 2 wid + 1 of the 8 x 8 outputs
#ifndef S123_NO_L3
        {
            sep123_expand(k3, TP, C::T2_PLANE, lane, wa0, wa1, pwb, xo0, xo1);
            const f32x4 b3A = *(const f32x4 *)(biasp + 2 * UBD_C + 4 * q);
            const f32x4 b3B = q < 2 ? *(const f32x4 *)(biasp + 2 * UBD_C + 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int r3 = 2 * wid + (i >> 3), c3 = i & 7;
            u32x2 o0, o1;
            sep123_unit<T>(t2p + (ubd_mul24(2 * r3, TP) + 2 * c3) * 16, wa0, wa1, pwb, xo0, xo1, b3A, b3B, o0, o1);
            __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)(y + (size_t)img * H4 * W4 * UBD_C), 0,
                                                                           (int)((unsigned)H4 * (unsigned)W4 * (UBD_C * 2u)), 0x00020000);
            const int gy = oy3 + r3, gx = ox3 + c3;
            const bool in = gy < H4 && gx < W4;
            const int gpx3 = ubd_mul24(gy, W4) + gx;
            const unsigned off = in ? (unsigned)((gpx3 << 5) + (gpx3 << 4)) + 8u * (unsigned)q : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b64(o0, yrs, (int)off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(o1, yrs, (int)(q < 2 ? off + 32u : 0x80000000u), 0, 0);
        }
#endif
        S3STAMP(7);
        if (!has_next) break;
        tile = nxt;
    }
}
