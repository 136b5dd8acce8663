// Fused L2 -> L3 of the stem for inference (included by forward.hip after sepconv_kernel).
//
// Reference semantics: net.py:294 (SeparableConv2D 24->24, stride 1, 'same', ReLU) followed by net.py:295-296
// (ZeroPadding2D((1,0),(1,0)) if fml + SeparableConv2D 24->24, stride 2, ReLU).  L3 is a stride-2 layer over L2's
// output, so L2's activation (the largest tensor of the pass: 201 MB written + 201 MB read per batch of 32) never
// has to exist in memory: a block computes the L2 outputs its L3 tile needs into LDS and consumes them in place.
// Saves 40 % of the stem's HBM bytes; costs 19/16 of L2's arithmetic (one halo row per 8 rows, one halo column per
// 32 columns).  Training keeps the separate kernels (the backward pass needs L2's output).
//
// Tile: 4 x 16 L3 outputs  <-  9 x 33 L2 outputs ("positions" (pr, pc))  <-  11 x 35 L1 outputs (the a1 patch):
//   L2 position (pr, pc) = L2 pixel (R0 + pr, C0 + pc), R0 = 2*oy0 - pad_lo, C0 = 2*ox0 - pad_lo (pad_lo = 1: fml);
//   it reads a1 patch pixels (pr + ky, pc + kx); L3 output (r, i) reads positions (2r + ky, 2i + kx).
//   32 of the 33 columns are two 16-pixel MFMA tiles (positions pad_lo + 16*half + i); the 33rd (position 0 if fml,
//   32 otherwise) is done as ONE more MFMA tile whose 16 "pixels" are the 9 rows of that column.
// Phases per tile (ONE persistent block of 8 waves per CU):
//   0  a1 patch by LDS-DMA (global_load_lds_dwordx4, the layout of sepconv_kernel: chunk-rotated 24-dword pixels): two
//      buffers, the patch of tile t+2 is requested by waves 4-7 as soon as phase A of tile t has released its buffer
//      (while waves 0-3 run phase B), so a fetch has a whole tile period to land; asm-issued + counted vmcnt
//   A  L2: depthwise on the VALU in the MFMA operand layout (wave = one column half x 3, 2 or 1 consecutive rows, sliding
//      over the patch rows), pointwise = 12 fp32 MFMAs per 16 pixels, bias + ReLU, zero outside the L2 map (that is
//      L3's zero padding), result to LDS with a 28-dword pixel pitch (conflict-free b128 reads at stride 2)
//   B  L3 (waves 0-3): wave = one output row; depthwise stride 2 from the LDS image, pointwise MFMA, bias + ReLU, 16-byte
//      stores.
// The kernel is bound by instruction issue, not by HBM (PMC + in-kernel stamps): tile coordinates advance
// incrementally (no divisions in the loop), LDS read addresses are per-lane constants + immediates, the bias rides in the
// MFMA accumulator, and the L3-padding select only runs on tiles that touch the border of L2's map.
#pragma once

struct s23_cfg {
    static constexpr int TH3 = 4;
    static constexpr int LR = 2 * TH3 + 1, LC = 33;            // L2 positions
    static constexpr int PH = LR + 2, PW = LC + 2;             // a1 patch
    static constexpr int CHUNKS = PH * PW * 6;
    static constexpr int NT = 512, NW = NT / 64;               // threads / waves per block
    static constexpr int ROUNDS = (CHUNKS + NT - 1) / NT;
    static constexpr int BUF_FLOATS = ROUNDS * NT * 4;
    static constexpr int LP = 28;                              // L2-image pixel pitch (dwords)
    static constexpr int L2_FLOATS = LR * LC * LP;
    static constexpr int W3PW_FLOATS = 2 * 64 * 8;             // L3 pointwise fragments: [nt][lane][s (6, +2 pad)]
    static constexpr int W3DW_FLOATS = 4 * 9 * 8;              // L3 depthwise taps per channel quarter: [q][tap][6 (+2 pad)]
    static constexpr int CARRY_FLOATS = 2 * LR * LP;           // position 32 of the previous tile, two generations
    static constexpr int SMEM_FLOATS = 2 * BUF_FLOATS + L2_FLOATS + W3PW_FLOATS + W3DW_FLOATS + CARRY_FLOATS;
};

// CARRY: fml padding (pad_lo = 1): the 33rd L2 column is inherited from the tile to the left; otherwise (TF 'same' at stride 2,
// pad_lo = 0) it is the column to the RIGHT of the tile and is computed as one more MFMA unit.
template <bool CARRY>
__global__ __launch_bounds__(s23_cfg::NT, 1) void stem23_kernel(const float *__restrict__ a1, float *__restrict__ y,
                                                        const float *__restrict__ frag2, const float *__restrict__ bias2,
                                                        const float *__restrict__ frag3, const float *__restrict__ bias3,
                                                        int n, int H2, int W2, int H4, int W4, int pad_lo
#ifdef UBD_STAMPS
                                                        , unsigned long long *__restrict__ stamps
#endif
                                                        )
{
#ifdef UBD_STAMPS   // diagnostic build only (tools/build_diag.sh): s_memtime at the phase boundaries, lane 0 of every wave, 16 tiles
#define UBD_STAMP(k) do { if (stamps && it < 16 && lane == 0) stamps[(((size_t)blockIdx.x * 8 + wid) * 16 + it) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UBD_STAMP(k) do {} while (0)
#endif
    using C = s23_cfg;
    __shared__ __attribute__((aligned(16))) float smem[C::SMEM_FLOATS];                    // ONE LDS object
    float *l2 = smem + 2 * C::BUF_FLOATS;
    float *w3pw = l2 + C::L2_FLOATS, *w3dw = w3pw + C::W3PW_FLOATS;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;

    // per-lane weights: lane (i, q) owns channels {4q..4q+3, 16+2q, 17+2q} (fragments are packed for channel 6q'+s', see
    // sepconv_kernel).  L2's (the hot phase) live in 66 VGPRs; L3's are read from LDS tables in phase B.
    float dwk2[9][6], pwf2[6][2];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);
        const int src_lane = 16 * (ch / 6) + i, ss = ch % 6;
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk2[t][s] = frag2[UBD_SEP_FRAG_FLOATS + (t * 6 + ss) * 64 + src_lane];
        pwf2[s][0] = frag2[(ss * 2 + 0) * 64 + src_lane]; pwf2[s][1] = frag2[(ss * 2 + 1) * 64 + src_lane];
    }
    for (int e = threadIdx.x; e < C::W3PW_FLOATS; e += C::NT) {
        const int nt = e >> 9, ln = (e >> 3) & 63, s = e & 7, lq = ln >> 4, li = ln & 15;
        const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
        w3pw[e] = s < 6 ? frag3[((ch % 6) * 2 + nt) * 64 + 16 * (ch / 6) + li] : 0.f;
    }
    for (int e = threadIdx.x; e < C::W3DW_FLOATS; e += C::NT) {
        const int lq = e / 72, r = e - lq * 72, t = r >> 3, s = r & 7;
        const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
        w3dw[e] = s < 6 ? frag3[UBD_SEP_FRAG_FLOATS + (t * 6 + ch % 6) * 64 + 16 * (ch / 6)] : 0.f;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // biases in the D layout (lane = pixel i, registers = channels 4q.. / 16+4q..): the MFMA accumulators start from them
    const f32x4 b2A = *(const f32x4 *)(bias2 + 4 * q), b2B = q < 2 ? *(const f32x4 *)(bias2 + 16 + 4 * q) : z4;
    const f32x4 b3A = *(const f32x4 *)(bias3 + 4 * q), b3B = q < 2 ? *(const f32x4 *)(bias3 + 16 + 4 * q) : z4;

    // ---- tile sequence of this block: whole row STRIPS of tiles (strip = (image, tile row); logical strips blockIdx.x,
    // + gridDim.x, ... mapped XCD-aware like ubd_xcd_tile), the tiles of a strip from left to right.  Walking a strip in x
    // order lets a tile inherit its leftmost L2 column (position 0 = column 2*ox0 - 1 with the fml padding) from the tile
    // before it -- that column is the previous tile's position 32 -- instead of computing it as a 19th MFMA unit.
    const int tiles_x = (W4 + 15) >> 4, tiles_y = (H4 + C::TH3 - 1) / C::TH3;
    const int strips = n * tiles_y;
    const int G = (int)gridDim.x;
    struct tpos { int tx, ty, img, ls; };                                          // ls: logical strip index (>= strips: past the end)
    auto strip_pos = [&](int ls, int tx) {
        tpos p;
        const int sidx = ubd_xcd_tile(ls < strips ? ls : strips - 1, strips);
        p.ty = (int)((unsigned)sidx % (unsigned)tiles_y);
        p.img = (int)((unsigned)sidx / (unsigned)tiles_y);
        p.tx = tx; p.ls = ls;
        return p;
    };
    auto advance = [&](tpos p) {
        if (p.tx + 1 < tiles_x) { ++p.tx; return p; }
        return strip_pos(p.ls + G, 0);
    };

    // ---- LDS-DMA of one a1 patch = 37 pieces of 1 KiB (64 chunks of 16 B).  In the steady state waves 4-7 issue them (10, 9,
    //      9, 9 pieces, ~170 cycles each) while waves 0-3 run phase B; the prologue splits them over all eight waves.
    constexpr int PIECES = (C::CHUNKS + 63) / 64;
    constexpr int MAXP = 10;
    const int my_first = wid >= 4 ? (wid == 4 ? 0 : 10 + 9 * (wid - 5)) : 0;
    const int my_count = wid >= 4 ? (wid == 4 ? 10 : 9) : 0;
    auto chunk_src = [&](int piece, int iy0, int ix0, int img, bool interior) {
        int c = piece * 64 + lane;
        c = c < C::CHUNKS ? c : C::CHUNKS - 1;
        const int pix = (int)((unsigned)c / 6u), sp = c - pix * 6;
        const int pr = (int)((unsigned)pix / (unsigned)C::PW), pc = pix - pr * C::PW;
        int part = sp + 3 * ((pc >> 3) & 1);
        part = part >= 6 ? part - 6 : part;
        int gy = iy0 + pr, gx = ix0 + pc;
        if (!interior) {
            gy = gy < 0 ? 0 : (gy >= H2 ? H2 - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= W2 ? W2 - 1 : gx);
        }
        return (const char *)a1 + ((((size_t)img * H2 + gy) * W2 + gx) * UBD_C + part * 4) * sizeof(float);
    };
    unsigned dma_rel[MAXP];                                                         // interior tiles: byte offset from the patch origin (>= 0)
#pragma unroll
    for (int k = 0; k < MAXP; ++k) dma_rel[k] = (unsigned)(chunk_src(my_first + (k < my_count ? k : 0), 0, 0, 0, true) - (const char *)a1);
    const unsigned lds_base = ubd_lds_addr(smem);
    auto dma_tile = [&](tpos p, float *patch, bool steady) {
        const unsigned lds_patch = lds_base + (unsigned)(patch - smem) * 4u;        // wave-uniform
        const int iy0 = 2 * p.ty * C::TH3 - pad_lo - 1, ix0 = 32 * p.tx - pad_lo - 1;   // a1 pixel of patch (0, 0)
        const bool interior = (iy0 >= 0) && (ix0 >= 0) && (iy0 + C::PH <= H2) && (ix0 + C::PW <= W2);   // block-uniform
        if (!steady) {                                                              // prologue: piece = wid, wid + 8, ...
            for (int piece = wid; piece < PIECES; piece += C::NW) ubd_glds16_at(chunk_src(piece, iy0, ix0, p.img, interior), lds_patch + (unsigned)piece * 1024u);
            return;
        }
        const char *origin = (const char *)a1 + (((size_t)p.img * H2 + iy0) * W2 + ix0) * (UBD_C * sizeof(float));
        if (interior) {                                                             // scalar base + 32-bit lane offset
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                if (k < my_count) ubd_glds16_sbase(origin, dma_rel[k], lds_patch + (unsigned)(my_first + k) * 1024u);   // wave-uniform count
        } else {
#pragma unroll 1
            for (int k = 0; k < my_count; ++k)
                ubd_glds16_at(chunk_src(my_first + k, iy0, ix0, p.img, false), lds_patch + (unsigned)(my_first + k) * 1024u);
        }
    };

    // ---- per-lane LDS read offsets (floats) of phase A: patch pixel (rb + yy, pos + kx), 16-byte chunk c in slot
    //      (c + 3f) % 6, f = (patch column >> 3) & 1; constant for the whole launch
    // rows of the 9: each column half is split {0-2, 3-4, 5-6, 7-8} over its four waves when the 33rd column is inherited
    // (fml padding); otherwise half 1 is split {0-2, 3-5, 6-8} over waves 1, 3, 5 and wave 7 computes the 33rd column (about
    // two rows' worth: 18 LDS reads with 2-way conflicts)
    constexpr bool carry = CARRY;
    const int half = wid & 1, rg = wid >> 1;
    const int rb = (half == 0 || carry) ? (rg == 0 ? 0 : 2 * rg + 1) : (rg < 3 ? 3 * rg : 0);
    const int rw = (half == 0 || carry) ? (rg == 0 ? 3 : 2) : (rg < 3 ? 3 : 0);
    const int pos = pad_lo + 16 * half + i;                                         // this lane's position column
    auto slot_off = [&](int prow, int pcol, int &o4, int &o2) {
        const int rot = 3 * ((pcol >> 3) & 1);
        int s4 = q + rot, s2 = 4 + (q >> 1) + rot;
        s4 = s4 >= 6 ? s4 - 6 : s4; s2 = s2 >= 6 ? s2 - 6 : s2;
        o4 = (prow * C::PW + pcol) * UBD_C + 4 * s4;
        o2 = (prow * C::PW + pcol) * UBD_C + 4 * s2 + 2 * (q & 1);
    };
    int ro4[3], ro2[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) slot_off(rb, pos + kx, ro4[kx], ro2[kx]);
    const int posx = pad_lo ? 0 : 32;                                               // the 33rd column (wave 7)
    const int ri = i < C::LR ? i : C::LR - 1;
    int co4[3], co2[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) slot_off(ri, posx + kx, co4[kx], co2[kx]);
    const int l2w = (rb * C::LC + pos) * C::LP + 4 * q;                             // phase A write offset of (row rb, this lane)
    const int l2r = (2 * (wid & 3) * C::LC + 2 * i) * C::LP;                        // phase B read offset of tap (0, 0)

    float *carry_buf = w3dw + C::W3DW_FLOATS;                                       // [2][LR][LP]: position 32 of the previous tile
    if ((int)blockIdx.x >= strips) return;
    __syncthreads();                                                                // weight tables written
    tpos cur = strip_pos(blockIdx.x, 0);
    tpos nx1 = advance(cur);
    dma_tile(cur, smem, false);
    if (nx1.ls < strips) dma_tile(nx1, smem + C::BUF_FLOATS, false);
    for (int it = 0;; ++it) {
        const float *patch = smem + (it & 1) * C::BUF_FLOATS;
        const int img = cur.img, oy0 = cur.ty * C::TH3, ox0 = cur.tx * 16;
        const tpos nx2 = advance(nx1);
        const bool has_next = nx1.ls < strips, has_next2 = has_next && nx2.ls < strips;   // block-uniform
        const int R0 = 2 * oy0 - pad_lo, C0 = 2 * ox0 - pad_lo;                     // L2 pixel of position (0, 0)
        // This tile's patch has landed.  The DMA is issued as asm (ubd_glds16): hipcc neither drains it in front of the LDS
        // writes of phase A nor waits for it anywhere -- these counted waits are the only ones.  Outstanding on waves 4-7,
        // oldest first: [DMA of this tile: k pieces] [DMA of the next tile: k pieces, issued during the previous phase B];
        // k = 10 (wave 4) or 9.  Waves 0-3 only have their L3 stores outstanding, which nobody waits for.
        UBD_STAMP(0);
        if (it < 2 || !has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // prologue DMAs / nothing issued behind this patch
        else if (wid == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (wid > 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        UBD_STAMP(1);
        __builtin_amdgcn_s_barrier();                                               // ... for everyone; the L2 image is free again
        UBD_STAMP(2);
        const int iy0 = R0 - 1, ix0 = C0 - 1;
        const bool border = (iy0 < 0) || (ix0 < 0) || (iy0 + C::PH > H2) || (ix0 + C::PW > W2);
        if (border) {                                                               // block-uniform: L2's 'same' zero padding
            float *pw_ = smem + (it & 1) * C::BUF_FLOATS;
            for (int pix = threadIdx.x; pix < C::PH * C::PW; pix += C::NT) {
                const int pr = pix / C::PW, pc = pix - pr * C::PW;
                const int gy = iy0 + pr, gx = ix0 + pc;
                if (gy < 0 || gy >= H2 || gx < 0 || gx >= W2) {
                    f32x4 *z = (f32x4 *)(pw_ + pix * UBD_C);
#pragma unroll
                    for (int k = 0; k < 6; ++k) z[k] = z4;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();                                           // raw barrier: the DMA in flight is not drained
        }
        // L3's zero padding: L2 positions outside L2's map must read as 0 (block-uniform: does this tile have any?)
        const bool mask_needed = (R0 < 0) || (C0 < 0) || (R0 + C::LR > H2) || (C0 + C::LC > W2);
        UBD_STAMP(3);

        // ---- phase A: L2 on the 9 x 33 positions; wave = (column half, rows {0-2, 3-4, 5-6, 7-8})
        {
            float dwv[3][6];
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int s = 0; s < 6; ++s) dwv[o][s] = 0.f;
#pragma unroll
            for (int yy = 0; yy < 5; ++yy) {
                if (yy < rw + 2) {                                                  // wave-uniform
                    f32x4 v4[3];
                    f32x2 v2[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        v4[kx] = *(const f32x4 *)(patch + ro4[kx] + yy * (C::PW * UBD_C));
                        v2[kx] = *(const f32x2 *)(patch + ro2[kx] + yy * (C::PW * UBD_C));
                    }
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const int ky = yy - o;
                        if (ky < 0 || ky > 2) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int t = ky * 3 + kx;
                            dwv[o][0] = fmaf(v4[kx][0], dwk2[t][0], dwv[o][0]);
                            dwv[o][1] = fmaf(v4[kx][1], dwk2[t][1], dwv[o][1]);
                            dwv[o][2] = fmaf(v4[kx][2], dwk2[t][2], dwv[o][2]);
                            dwv[o][3] = fmaf(v4[kx][3], dwk2[t][3], dwv[o][3]);
                            dwv[o][4] = fmaf(v2[kx][0], dwk2[t][4], dwv[o][4]);
                            dwv[o][5] = fmaf(v2[kx][1], dwk2[t][5], dwv[o][5]);
                        }
                    }
                    if (yy >= 2) {                                                  // position row rb + yy - 2 is complete
                        const int o = yy - 2;
                        f32x4 acc0 = b2A, acc1 = b2B;
#pragma unroll
                        for (int s = 0; s < 6; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][0], dwv[o][s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][1], dwv[o][s], acc1, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc0[r] = fmaxf(acc0[r], 0.f); acc1[r] = fmaxf(acc1[r], 0.f); }
                        if (mask_needed) {
                            const bool ok = (unsigned)(C0 + pos) < (unsigned)W2 && (unsigned)(R0 + rb + o) < (unsigned)H2;
                            if (!ok) { acc0 = z4; acc1 = z4; }
                        }
                        float *dst = l2 + l2w + o * (C::LC * C::LP);
                        *(f32x4 *)dst = acc0;
                        if (q < 2) *(f32x4 *)(dst + 16) = acc1;
                        if (CARRY && half == 1 && i == 15) {                        // position 32: the next tile's position 0
                            float *cd = carry_buf + ((it & 1) * C::LR + rb + o) * C::LP + 4 * q;
                            *(f32x4 *)cd = acc0;
                            if (q < 2) *(f32x4 *)(cd + 16) = acc1;
                        }
                    }
                }
            }
            if constexpr (CARRY) if (wid == C::NW - 1) {
                // position 0 (column 2*ox0 - 1): the previous tile's position 32, or L3's zero padding at the left image edge.
                // The previous generation of the carry buffer is not written during this tile; the L2 image's column 0 is only
                // read in phase B.
                if (lane < C::LR * 6) {
                    const int row = lane / 6, ch4 = lane - row * 6;
                    f32x4 v = z4;
                    if (cur.tx > 0) v = *(const f32x4 *)(carry_buf + (((it + 1) & 1) * C::LR + row) * C::LP + 4 * ch4);
                    *(f32x4 *)(l2 + (row * C::LC) * C::LP + 4 * ch4) = v;
                }
            }
            if constexpr (!CARRY) if (wid == C::NW - 1) {
                // the 33rd column: lane i <-> position row i (9 of 16 lanes carry a pixel)
                float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v4 = *(const f32x4 *)(patch + co4[kx] + ky * (C::PW * UBD_C));
                        const f32x2 v2 = *(const f32x2 *)(patch + co2[kx] + ky * (C::PW * UBD_C));
                        const int t = ky * 3 + kx;
                        dv[0] = fmaf(v4[0], dwk2[t][0], dv[0]); dv[1] = fmaf(v4[1], dwk2[t][1], dv[1]);
                        dv[2] = fmaf(v4[2], dwk2[t][2], dv[2]); dv[3] = fmaf(v4[3], dwk2[t][3], dv[3]);
                        dv[4] = fmaf(v2[0], dwk2[t][4], dv[4]); dv[5] = fmaf(v2[1], dwk2[t][5], dv[5]);
                    }
                f32x4 acc0 = b2A, acc1 = b2B;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][0], dv[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][1], dv[s], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc0[r] = fmaxf(acc0[r], 0.f); acc1[r] = fmaxf(acc1[r], 0.f); }
                const bool ok = (unsigned)(C0 + posx) < (unsigned)W2 && (unsigned)(R0 + ri) < (unsigned)H2;
                if (!ok) { acc0 = z4; acc1 = z4; }
                if (i < C::LR) {
                    float *dst = l2 + (ri * C::LC + posx) * C::LP + 4 * q;
                    *(f32x4 *)dst = acc0;
                    if (q < 2) *(f32x4 *)(dst + 16) = acc1;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                         // lgkmcnt(0): the L2 image is written
        UBD_STAMP(4);
        __builtin_amdgcn_s_barrier();                                               // ... by everyone; this tile's patch buffer is free
        UBD_STAMP(5);

        if (wid < 4) {
            // ---- phase B, waves 0-3: L3 output row oy0 + wid
            const int oy = oy0 + wid;
            float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float *p = l2 + l2r + (ky * C::LC + kx) * C::LP;
                    const f32x4 v4 = *(const f32x4 *)(p + 4 * q);
                    const f32x2 v2 = *(const f32x2 *)(p + 16 + 2 * q);
                    const float *wt = w3dw + (q * 9 + ky * 3 + kx) * 8;
                    const f32x4 w4 = *(const f32x4 *)wt;
                    const f32x2 w2 = *(const f32x2 *)(wt + 4);
                    dv[0] = fmaf(v4[0], w4[0], dv[0]); dv[1] = fmaf(v4[1], w4[1], dv[1]);
                    dv[2] = fmaf(v4[2], w4[2], dv[2]); dv[3] = fmaf(v4[3], w4[3], dv[3]);
                    dv[4] = fmaf(v2[0], w2[0], dv[4]); dv[5] = fmaf(v2[1], w2[1], dv[5]);
                }
            const f32x4 pa0 = *(const f32x4 *)(w3pw + lane * 8), pa1 = *(const f32x4 *)(w3pw + 512 + lane * 8);
            const f32x2 pb0 = *(const f32x2 *)(w3pw + lane * 8 + 4), pb1 = *(const f32x2 *)(w3pw + 512 + lane * 8 + 4);
            f32x4 acc0 = b3A, acc1 = b3B;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa0[s] : pb0[s - 4], dv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa1[s] : pb1[s - 4], dv[s], acc1, 0, 0, 0);
            }
            store_tile_relu_t(y, ((size_t)img * H4 + oy) * W4, ox0, oy < H4 ? W4 : 0, lane, acc0, acc1, z4, z4);   // bias already in
        }
        UBD_STAMP(6);
        // ---- waves 4-7: the patch of tile t + 2, into the buffer phase A has just released
        if (wid >= 4 && has_next2) dma_tile(nx2, smem + (it & 1) * C::BUF_FLOATS, true);
        UBD_STAMP(7);
        if (!has_next) break;
        cur = nx1;
        nx1 = nx2;
    }
#undef UBD_STAMP
}
