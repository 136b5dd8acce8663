// The whole stem L1 -> L2 -> L3 of an inference pass in ONE kernel (fml padding; included by forward.hip after stem23.h).
//
// Reference semantics: net.py:292-296 -- ZeroPadding2D((1,0),(1,0)) + SeparableConv2D(24, 3x3, stride 2, relu) (L1),
// SeparableConv2D(24, 3x3, 'same', relu) (L2), ZeroPadding2D + SeparableConv2D(stride 2, relu) (L3) -- and the fused
// (x - 127.5) / 127.5 of net.py:217-218 for uint8 input.  stem23.h keeps L2's output in LDS; this kernel also computes the
// L1 outputs an L2 tile needs from the IMAGE, so neither L1's nor L2's activation (2 x 201 MB written + 2 x 201 MB read per
// batch of 32 at fp32) ever exists in memory: the stem reads 100 MB (25 MB for uint8) and writes 50 MB.  Price: L1 is
// recomputed on the 11 x 34 halo patch of every tile (1.46x its arithmetic -- L1 is the cheap layer: 3 input channels) and L2 on
// 9 x 32 (1.125x).  The per-tile phases:
//   0a  the tile's 23 x 69 x C_in input patch goes to LDS: fp32 input that is fed as it is (PLAIN) by 4-byte LDS-DMA through a
//       buffer descriptor (zeros outside the image = L1's zero padding), requested a whole phase A ahead; uint8 / preprocessed
//       input through registers (requested at the start of phase A, converted right after it)
//   0b  L1 on the 374 patch pixels as 24 units of 16 (flat index -> (row, col)): depthwise on the VALU (lane = pixel x input
//       channel), pointwise = 2 fp32 MFMAs, bias + ReLU, 0 outside L1's map (= L2's zero padding), into the a1 patch image
//       (the chunk-rotated layout stem23's phase A reads)
//   A   L2 on 9 x 32 positions (as stem23.h, CARRY variant; wave 7 first moves the inherited 33rd L2 column into place);
//   B   L3 (waves 0-3), 16-byte stores.
// Two block barriers per tile, one 8-wave block per CU: phase B of tile t shares its barrier interval with phase 0b of tile
// t + 1 (waves 4-7, which have no L3 row, take twice the L1 units).  Training keeps the separate kernels (it needs a1 and a2).
#pragma once

// a1 patch image of this kernel: pixel pitch 24 dwords = six 16-byte chunks; chunk c of the pixel in patch column `col` sits in
// slot c ^ bit2(col) (c < 4) / 4 + ((c - 4) ^ bit3(col)) (c = 4, 5).  Found by enumerating swizzles under the LDS bank map of
// MI355X_MICROARCH.md (16-lane groups of ds_read_b128, 32-lane groups of ds_read_b64, 8-lane groups of ds_write_b128): phase
// A's three b128 + three b64 reads per row are conflict-free (4 + 2 LDS cycles), L1's chunk stores cost 8.7 + 8 cycles per unit
// instead of 16 + 8 -- 1024 LDS cycles per tile against 1596 for the rotate-by-three-slots layout of stem23.h (which it keeps:
// its image is written by LDS-DMA).  PMC: SQ_LDS_BANK_CONFLICT of the kernel in profiles/r03_pmc_stem123_fp32.txt.
__device__ __forceinline__ int s123_slot(int c, int col) { return c < 4 ? (c ^ ((col >> 2) & 1)) : 4 + ((c - 4) ^ ((col >> 3) & 1)); }

template <int CIN> struct s123_cfg {
    using B = s23_cfg;
    static constexpr int NT = B::NT, NW = B::NW;
    static constexpr int AC = B::LC + 1;                           // a1 patch columns that are needed: 1 .. 34 of the 35
    static constexpr int XH = 2 * B::PH + 1, XW = 2 * AC + 1;      // input patch 23 x 69
    static constexpr int XE = XH * XW * CIN;                       // elements
    static constexpr int XREGS = (XE + NT - 1) / NT;
    // LDS image of the patch: one more pixel column on the left (image column 64 tx - 4) makes every patch row start on a
    // 16-byte boundary of the image row (H, W multiples of 4), so a row is ONE 16-byte LDS-DMA piece of XCH lanes; rows are XS
    // floats apart (a whole number of 16-byte chunks).  The 23 x 69 patch proper sits at columns 1 .. 69.
    static constexpr int XCH = ((XW + 1) * CIN + 3) / 4;           // 16-byte chunks per row: 53 (RGB) / 18 (grey)
    static constexpr int XS = XCH * 4;                             // row pitch in floats
    static constexpr int XP_FLOATS = XH * XS;
    static constexpr int A1_FLOATS = B::PH * B::PW * UBD_C;        // a1 patch, stem23's layout
    static constexpr int NPIX = B::PH * AC;                        // 374 L1 outputs per tile
    static constexpr int UNITS = (NPIX + 15) / 16;                 // 24
    static constexpr int UPW = (UNITS + NW - 1) / NW;              // units per wave
    static constexpr int W1_FLOATS = 64 * 12 + 64 + 4;              // L1 per-lane weights (9 depthwise taps, 2 pointwise, pad) + biases of L1 / L3 (2 x 32) + the ring of strip ids
    static constexpr int LUT_FLOATS = 256;                          // uint8 input: the 256 preprocessed values (one lookup per element instead of a subtract + an exact division)
    static constexpr int STEM_FLOATS = A1_FLOATS + B::L2_FLOATS + XP_FLOATS + B::W3PW_FLOATS + B::W3DW_FLOATS + B::CARRY_FLOATS + W1_FLOATS + LUT_FLOATS;
    static constexpr int PP_FLOATS = (PP_LDS_MAX_BYTES + 3) / 4;   // the postprocess job some blocks run first (pp_lds.h) uses the same LDS
    static constexpr int SMEM_FLOATS = STEM_FLOATS > PP_FLOATS ? STEM_FLOATS : PP_FLOATS;
    static_assert(XCH <= 64 && (A1_FLOATS % 4) == 0 && (B::L2_FLOATS % 4) == 0, "patch rows are single 16-byte-aligned DMA pieces");
};

// PLAIN: fp32 input that is fed as it is (no preprocessing): the patch goes straight from memory into LDS by 4-byte LDS-DMA through a
// buffer descriptor (zeros outside the image = L1's padding), requested right after phase 0b of the previous tile -- no staging
// registers and no phase 0a.
// COLD (round 6, launches too small to give every CU strips of tiles -- one image): the work unit is ONE tile, started cold.  A tile
// inside a row has nobody to inherit L2 position 0 from, so it does not produce its first L3 column: the tiles of a row sit 15 L3
// columns apart (tile k > 0 at column 15 k covers columns 15 k + 1 .. 15 k + 15; tile 0 covers 0 .. 15 -- the image edge is its
// padding), nine tiles for 128 columns instead of eight.  Everything else -- patch, L1, L2, L3 of a tile -- is the code of the strip
// walk, so every output is computed by the same instructions on the same values: bit-identical (tests/test_gpu_forward.py).  Tiles
// are dealt out statically (block b takes b, b + grid, ...): no tickets, no ring.  One 512 x 512 image: 288 tiles on 256 CUs, one
// launch of ~8 us instead of three of 5.6 + 6.6 + 7.2 us.
template <int CIN, int IN_U8, int PLAIN, bool COLD = false>
__global__ __launch_bounds__(s23_cfg::NT, 1) void stem123_kernel(const void *__restrict__ xin, float *__restrict__ y,
                                                                const float *__restrict__ frag1, const float *__restrict__ bias1,
                                                                const float *__restrict__ frag2, const float *__restrict__ bias2,
                                                                const float *__restrict__ frag3, const float *__restrict__ bias3,
                                                                int n, int H, int W, int H2, int W2, int H4, int W4,
                                                                float pre_sub, float pre_div, int *__restrict__ ticket, pp_lds_args pj
#ifdef UBD_STAMPS
                                                                , unsigned long long *__restrict__ stamps
#endif
                                                                )
{
    using C = s23_cfg;
    using X = s123_cfg<CIN>;
    __shared__ __attribute__((aligned(16))) float smem[X::SMEM_FLOATS];                     // ONE LDS object
    // ---- the postprocess of an EARLIER batch rides along (ubd_forward_postprocess): block b first does image b, b + grid, ... of
    // that job -- threshold, components, boxes, class vote, lists, all inside the block (pp_lds.h) -- and then joins the strip
    // queue below, where the blocks that had no image have meanwhile taken its share (tickets).  One launch instead of a second
    // stream with two events per step: round 3 measured 10 us of idle forward stream behind every event record and a placement
    // race between the postprocess blocks and this kernel's whole-CU blocks.
#ifdef UBD_STAMPS   // diagnostic build only: the block's time line on the 100-MHz clock all CUs share (tools/stamps_stem_blocks.py)
#define S123_BLOCK_STAMP(k) do { if (stamps && threadIdx.x == 0 && (k) < 32) stamps[(size_t)gridDim.x * 8 * 16 * 8 + (size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S123_BLOCK_STAMP(k) do {} while (0)
#endif
    S123_BLOCK_STAMP(0);
    if (pj.n > 0) {
        for (int im = (int)blockIdx.x; im < pj.n; im += (int)gridDim.x) {
            pp_image_lds<C::NT, true>((int *)smem, pj, im);
            __syncthreads();
        }
    }
    S123_BLOCK_STAMP(1);
    // The block's first D strips are its own without asking: strips b D .. b D + D - 1 (256 blocks drawing their first ticket from
    // one counter at the same moment took 2.6 us of every block's prologue in the stamps).  A block that had a postprocess job comes
    // up late and alone: it draws tickets like everybody does later -- and the first tickets handed out are the slots those blocks
    // did not take (ticket_ls below).  Requested before anything else (an atomic round trip to the far L2), consumed after the
    // weight loads below have been issued.
    const int Dc = ((W4 + 15) >> 4) >= 3 ? 1 : 4 - ((W4 + 15) >> 4);
    const int njob = pj.n > 0 ? (pj.n < (int)gridDim.x ? pj.n : (int)gridDim.x) : 0;
    auto ticket_ls = [&](int k) { return k < njob * Dc ? k : k + ((int)gridDim.x - njob) * Dc; };   // ticket -> logical strip
    int t0_early = 0;
    if (!COLD && threadIdx.x == 0 && (int)blockIdx.x < njob) t0_early = __hip_atomic_fetch_add(ticket, Dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float *a1p = smem;
    float *l2 = a1p + X::A1_FLOATS;
    float *xp = l2 + C::L2_FLOATS;
    float *w3pw = xp + X::XP_FLOATS, *w3dw = w3pw + C::W3PW_FLOATS;
    float *carry_buf = w3dw + C::W3DW_FLOATS;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;

    // ---- weights.  L2's 66 per-lane values live in VGPRs (hot phase); L1's eleven too; L3's come from LDS tables.
    float dwk2[9][6], pwf2[6][2];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);
        const int src_lane = 16 * (ch / 6) + i, ss = ch % 6;
#pragma unroll
        for (int t = 0; t < 9; ++t) dwk2[t][s] = frag2[UBD_SEP_FRAG_FLOATS + (t * 6 + ss) * 64 + src_lane];
        pwf2[s][0] = frag2[(ss * 2 + 0) * 64 + src_lane]; pwf2[s][1] = frag2[(ss * 2 + 1) * 64 + src_lane];
    }
    // L1's eleven per-lane values (lane (i, q): input channel q, zero weights for q >= C_in) sit in an LDS table and are
    // re-read at the start of every phase 0b: kept in VGPRs across phase A they cost 11 spilled registers
    float *w1t = carry_buf + C::CARRY_FLOATS;
    float *bt = w1t + 64 * 12;                                     // [0,32): L1's bias (24 + zeros), [32,64): L3's
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 b2A = *(const f32x4 *)(bias2 + 4 * q), b2B = q < 2 ? *(const f32x4 *)(bias2 + 16 + 4 * q) : z4;

    // ---- tile sequence: whole row strips of tiles, left to right (stem23.h).  Strips are handed out dynamically, every one
    // of them a ticket from ONE global counter (zeroed by the host per pass; a strip is eight tiles = ~20 us of work, so a
    // launch draws a few thousand tickets and the counter is never contended).  This kernel needs a whole CU per block, and
    // the pipelined postprocess of the previous batch holds 32 CUs for ~60-130 us right when it starts: with a static split
    // the 32 blocks that cannot be placed sit on an eighth of the strips (step 0.43 -> 0.46 ms); with tickets they come up
    // late, find (almost) nothing left and exit.  A block claims as few strips ahead as the one-tile lookahead allows (D
    // below: one for rows of three or more tiles) -- every strip claimed early by a late block ends after everybody else.
    // The ids live in a four-entry LDS ring: entry j + D is requested when strip j starts and published at the end of that
    // tile, at least one full tile (three barriers) before anybody reads it.
    const int cold_kx = W4 <= 16 ? 1 : 1 + (W4 - 16 + 14) / 15;                     // COLD: tiles per row, 15 columns apart
    const int tiles_x = COLD ? 1 : (W4 + 15) >> 4, tiles_y = (H4 + C::TH3 - 1) / C::TH3;
    const int strips = COLD ? n * tiles_y * cold_kx : n * tiles_y;                  // COLD: a "strip" is one tile
    const int D = tiles_x >= 3 ? 1 : 4 - tiles_x;                                   // strips claimed ahead of the current one
    int *ring = (int *)(bt + 64);                                                   // logical strip id of the block's strip ordinal j at [j & 3]
    float *lut = bt + 64 + 4;                                                       // uint8 input: ((float)b - pre_sub) / pre_div for b = 0 .. 255, filled with the weight tables
    struct tpos { int tx, ty, img, ord, ls; };                                      // ord: ordinal of the strip in this block's sequence
    auto strip_pos = [&](int ord, int tx) {
        tpos p;
        if constexpr (COLD) {                                                        // tile b, b + grid, ...: (image, tile row, tile of the row)
            const int ls = (int)blockIdx.x + ord * (int)gridDim.x;
            const int sidx = ubd_xcd_tile(ls < strips ? ls : strips - 1, strips);
            const int rowi = (int)((unsigned)sidx / (unsigned)cold_kx);
            p.tx = sidx - rowi * cold_kx;
            p.ty = (int)((unsigned)rowi % (unsigned)tiles_y);
            p.img = (int)((unsigned)rowi / (unsigned)tiles_y);
            p.ord = ord; p.ls = ls;
            return p;
        }
        const int ls = __builtin_amdgcn_readfirstlane(ring[ord & 3]);
        const int sidx = ubd_xcd_tile(ls < strips ? ls : strips - 1, strips);
        p.ty = (int)((unsigned)sidx % (unsigned)tiles_y);
        p.img = (int)((unsigned)sidx / (unsigned)tiles_y);
        p.tx = tx; p.ord = ord; p.ls = ls;
        return p;
    };
    auto advance = [&](tpos p) {
        if (!COLD && p.tx + 1 < tiles_x) { ++p.tx; return p; }
        return strip_pos(p.ord + 1, 0);
    };
    auto xo3 = [&](const tpos &p) { return (COLD ? 15 : 16) * p.tx; };              // first L3 column of the tile

    // ---- input patch of a tile: rows 4*oy0 - 5 .. + 22, columns 4*ox0 - 3 .. + 68, raw bits into registers (fp32 pattern
    //      or zero-extended byte; 0x100 / pre_sub bits = "outside the image", exactly 0 after the preprocessing)
    unsigned xreg[X::XREGS];
    // element e = threadIdx.x + 512 k of the patch in row-major order [23][69 * C_in]: (row, column * C_in + channel) advance by
    // (512 / RW, 512 % RW) with a carry -- no divisions in the loop
    constexpr int RWF = X::XW * CIN;                                                 // floats per patch row
    const int e0_row = (int)threadIdx.x / RWF, e0_col = (int)threadIdx.x - e0_row * RWF;
    // Addresses: image base in scalar registers + a 32-bit element offset (the size_t index arithmetic of the first version cost
    // ~20 vector instructions per element: phase 0a was 2.5 k of the tile's 11.5 k cycles in the stamps).  Interior tiles
    // (block-uniform) load unconditionally; border tiles load from the clamped position and the elements outside the image
    // are replaced where the registers are consumed (a select next to the load would wait for it).
    auto tile_interior = [&](tpos p) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, ix0 = 4 * xo3(p) - 3;
        return (iy0 >= 0) && (ix0 >= 0) && (iy0 + X::XH <= H) && (ix0 + X::XW <= W);
    };
    auto load_x = [&](tpos p) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (4 * xo3(p) - 3) * CIN, WC = W * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)p.img * H * WC * (IN_U8 ? 1 : 4);   // wave-uniform
        const bool interior = tile_interior(p);                                      // block-uniform
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            int gy = iy0 + pr, gf = fx0 + pcf;                                       // gf = gx * C_in + channel
            if (!interior) { gy = min(max(gy, 0), H - 1); gf = min(max(gf, 0), WC - 1); }
            else if (k == X::XREGS - 1) gy = min(gy, H - 1);                         // the last round runs past the patch (never stored)
            const unsigned off = (unsigned)__umul24(gy, WC) + (unsigned)gf;
            if constexpr (IN_U8) xreg[k] = img8[off];
            else xreg[k] = ((const unsigned *)img8)[off];
            pr += C::NT / RWF; pcf += C::NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };
    auto fix_border = [&](tpos p) {                                                  // outside the image = exactly 0 after the preprocessing
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (4 * xo3(p) - 3) * CIN, WC = W * CIN;
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(fx0 + pcf) < (unsigned)WC;
            xreg[k] = inside ? xreg[k] : (IN_U8 ? 0x100u : __builtin_bit_cast(unsigned, pre_sub));
            pr += C::NT / RWF; pcf += C::NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };

    const unsigned lds_xp = ubd_lds_addr(xp);
    auto dma_x = [&](tpos p) {                                                       // PLAIN: the patch of tile p, one 16-byte LDS-DMA piece per patch row
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (4 * xo3(p) - 4) * CIN, WC = W * CIN;   // one column left of the patch: 16-byte aligned (COLD: 60 k - 4 columns)
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)xin + (size_t)p.img * H * WC * 4), 0,
                                                                        (int)((unsigned)H * WC * 4u), 0x00020000);
        // Row r of the patch = XCH chunks of 16 bytes from image row iy0 + r: a wave-uniform term plus 16 * lane.  Rows above /
        // below the image fall out of the descriptor's range by themselves (a negative term wraps); chunks left / right of
        // the image (they never straddle its border: W is a multiple of 4) get an out-of-range offset -- zeros in LDS either
        // way = L1's zero padding.  23 pieces per tile (the first version moved 4 bytes per lane: 75 flat pieces with a
        // division and a range check per lane; in-kernel stamps: 1.3 k of the tile's 9.7 k cycles went into issuing them).
        unsigned off0 = (unsigned)lane * 16u;
        off0 = (unsigned)(fx0 + 4 * lane) < (unsigned)WC ? off0 : 0x80000000u;      // image bytes < 2^30 (host): out of range with any row term
        if (lane < X::XCH) {
#pragma unroll
            for (int k = 0; k < (X::XH + C::NW - 1) / C::NW; ++k) {
                const int row = k * C::NW + wid;
                if (row >= X::XH) break;                                             // wave-uniform
                const int term = ((iy0 + row) * WC + fx0) * 4;
                ubd_blds16(rsrc, (unsigned)term + off0, lds_xp + (unsigned)(row * X::XS * 4));
            }
        }
    };

    // ---- phase 0b: this wave's L1 units (flat pixel p = 16 u + i of the 11 x 34 needed a1 pixels), constant per launch.
    // Phase 0b of tile t + 1 shares a barrier interval with phase B of tile t (below), which only waves 0-3 run: they take two
    // units each, waves 4-7 four.
    constexpr int UPW = 4;
    int u_rc[UPW];                                                 // a1 patch (row << 8 | column 1..34) of this lane's pixel, -1: none (the read / write offsets are rebuilt from it: registers)
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = wid >= 4 ? (wid - 4) + 4 * k : (k < 2 ? 16 + wid + 4 * k : X::UNITS);
        const int p = u * 16 + i;
        const bool live = u < X::UNITS && p < X::NPIX;
        const int ar = (live ? p : 0) / X::AC, ac = (live ? p : 0) - ar * X::AC + 1;
        u_rc[k] = live ? ((ar << 8) | ac) : -1;
    }
    const int qc = q < CIN ? q : CIN - 1;                         // lanes without a channel (zero weights) read what their neighbours read: an LDS broadcast, not a second address on the same banks
    // ---- phase A constants (stem23.h, CARRY variant)
    const int half = wid & 1, rg = wid >> 1;
    const int rb = rg == 0 ? 0 : 2 * rg + 1, rw = rg == 0 ? 3 : 2;
    const int pos = 1 + 16 * half + i;
    int ro4[3], ro2[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int pcol = pos + kx;
        ro4[kx] = (rb * C::PW + pcol) * UBD_C + 4 * s123_slot(q, pcol);
        ro2[kx] = (rb * C::PW + pcol) * UBD_C + 4 * s123_slot(4 + (q >> 1), pcol) + 2 * (q & 1);
    }
    const int l2w = (rb * C::LC + pos) * C::LP + 4 * q;
    const int l2r = (2 * (wid & 3) * C::LC + 2 * i) * C::LP;

    if (!COLD && threadIdx.x == 0) {                                                 // the block's first D strips (requested at the top)
        const bool had_job = (int)blockIdx.x < njob;
#pragma unroll
        for (int j = 0; j < 3; ++j) ring[j] = had_job ? ticket_ls(t0_early + j) : (int)blockIdx.x * D + j;   // only the first D are this block's: the rest are overwritten before use
    }
    __syncthreads();
    S123_BLOCK_STAMP(3);
    tpos cur = strip_pos(0, 0);
    // The counter resets itself: every block checks out through a second counter, and the last one out -- by then nobody draws
    // tickets any more -- zeroes both for the next launch on this workspace (stream order; the host zeroes them only before
    // the first launch on a workspace: the per-pass memset was a 5 us kernel of its own between two forward passes).
    auto check_out = [&]() {
        if constexpr (COLD) return;                                                  // no tickets drawn
        __syncthreads();
        if (threadIdx.x == 0) {
            const int left = __hip_atomic_fetch_add(ticket + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == (int)gridDim.x - 1) {
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket + 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (cur.ls >= strips) { S123_BLOCK_STAMP(2); check_out(); return; }              // block-uniform: nothing left
    tpos nx1 = advance(cur);
    int pending = 0;                                                                 // ticket in flight (thread 0)
#ifdef UBD_STAMPS   // diagnostic build only: s_memtime at the phase boundaries, lane 0 of every wave, first 16 tiles of the block
#define S123_STAMP(k) do { if (stamps && it < 16 && lane == 0) stamps[(((size_t)blockIdx.x * 8 + wid) * 16 + it) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S123_STAMP(k) do {} while (0)
#endif
    // ---- phase 0a of tile p (non-PLAIN): the patch, requested a phase ago into registers, goes to LDS (preprocessed)
    auto convert_x = [&](tpos p) {
        const bool plain = !IN_U8 && pre_sub == 0.f && pre_div == 1.f;              // already preprocessed fp32 input: a copy
        if (!tile_interior(p)) fix_border(p);                                        // block-uniform
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            const int e = k * C::NT + (int)threadIdx.x;
            if (e < X::XE) {
                float *dst = xp + pr * X::XS + CIN + pcf;                            // the LDS image keeps one unused pixel column on the left (dma_x)
                if constexpr (IN_U8) *dst = xreg[k] > 255u ? 0.f : lut[xreg[k] & 255u];
                else *dst = plain ? __builtin_bit_cast(float, xreg[k]) : (__builtin_bit_cast(float, xreg[k]) - pre_sub) / pre_div;
            }
            pr += C::NT / RWF; pcf += C::NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };
    // ---- phase 0b of tile p: L1 -> a1 patch image.  A unit is a serial chain (nine LDS taps -> nine dependent FMAs -> two MFMAs
    // -> clamp -> two LDS stores); run one after the other at two waves per SIMD the units cost ~800 cycles each, nearly all
    // of it latency (in-kernel stamps).  So the stages are batched over the wave's units: all tap reads first (l1_taps), then
    // the FMA chains side by side, the MFMAs back to back, the epilogues -- and the waves that also own an L3 row put that
    // row between their tap reads and the rest (l1_finish), so the reads are long complete when they are consumed.
    auto l1_taps = [&](int k, float (&tap)[9]) {
        const int rc0 = u_rc[k] < 0 ? 1 : u_rc[k];                                   // lanes without a pixel compute on pixel (0, 1) and store nothing
        const int ar0 = rc0 >> 8, ac0 = rc0 & 255;
        const int u_rd = (2 * ar0) * X::XS + (2 * (ac0 - 1) + 1) * CIN + qc;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) tap[ky * 3 + kx] = xp[u_rd + ky * X::XS + kx * CIN];
    };
    auto l1_finish = [&](tpos p, auto nunits, auto &tap, int k0) {                  // units k0 .. k0 + NU - 1 of the wave
        constexpr int NU = decltype(nunits)::value;
        const int A0y = 2 * p.ty * C::TH3 - 2, A0x = 2 * xo3(p) - 2;                 // a1 pixel of patch (0, 0)
        float dwk1[9], pwf1[2];
        {
            const f32x4 wa = *(const f32x4 *)(w1t + lane * 12), wb = *(const f32x4 *)(w1t + lane * 12 + 4), wc = *(const f32x4 *)(w1t + lane * 12 + 8);
            dwk1[0] = wa[0]; dwk1[1] = wa[1]; dwk1[2] = wa[2]; dwk1[3] = wa[3]; dwk1[4] = wb[0]; dwk1[5] = wb[1]; dwk1[6] = wb[2]; dwk1[7] = wb[3];
            dwk1[8] = wc[0]; pwf1[0] = wc[1]; pwf1[1] = wc[2];
        }
        const f32x4 b1A = *(const f32x4 *)(bt + 4 * q), b1B = *(const f32x4 *)(bt + 16 + 4 * q);      // zeros beyond channel 23
        float dv[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) dv[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < NU; ++k) dv[k] = fmaf(tap[k][t], dwk1[t], dv[k]);    // NU independent chains
        f32x4 acc0[NU], acc1[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            acc0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf1[0], dv[k], b1A, 0, 0, 0);
            acc1[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf1[1], dv[k], b1B, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int rc = u_rc[k0 + k];
            const int ar = rc >> 8, ac = rc & 255;
            const bool inside = rc >= 0 && (unsigned)(A0y + ar) < (unsigned)H2 && (unsigned)(A0x + ac) < (unsigned)W2;
            const float cap = inside ? __builtin_inff() : 0.f;                       // outside L1's map: L2's zero padding
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[k][r] = ubd_relu_cap(acc0[k][r], cap); acc1[k][r] = ubd_relu_cap(acc1[k][r], cap); }
            if (rc >= 0) {
                float *dst = a1p + (ar * C::PW + ac) * UBD_C + 4 * s123_slot(q, ac);  // chunk q; chunk 4 + q from the lanes q < 2
                *(f32x4 *)dst = acc0[k];
                if (q < 2) {                                                         // chunk 4 + q: slot (4 + q + 3f) % 6
                    *(f32x4 *)(a1p + (ar * C::PW + ac) * UBD_C + 4 * s123_slot(4 + q, ac)) = acc1[k];
                }
            }
        }
    };
    auto phase0b = [&](tpos p) {                                                     // the whole phase at once (first tile of a block)
        if (wid < 4) {                                                               // wave-uniform: waves 0-3 own two units, waves 4-7 four
            float tap[2][9];
            l1_taps(0, tap[0]); l1_taps(1, tap[1]);
            l1_finish(p, std::integral_constant<int, 2>{}, tap, 0);
        } else {                                                                     // two pairs: four units' taps at once do not fit the register file
            float tap[2][9];
            l1_taps(0, tap[0]); l1_taps(1, tap[1]);
            l1_finish(p, std::integral_constant<int, 2>{}, tap, 0);
            l1_taps(2, tap[0]); l1_taps(3, tap[1]);
            l1_finish(p, std::integral_constant<int, 2>{}, tap, 2);
        }
    };

    // ---- first tile: patch -> LDS, L1.  The LDS weight tables are filled while the patch is on its way (and not at all by a block
    // that found no strip left).  L1's eleven per-lane values, the biases of L1 / L3, L3's pointwise and depthwise tables:
    if constexpr (PLAIN) dma_x(cur); else load_x(cur);
    {   // every thread's (up to) six table entries: all loads first, then the stores -- one memory round trip (as loops of
        // load -> store the fills took 1.8 us of the prologue)
        static_assert(64 * 12 <= 2 * C::NT && C::W3PW_FLOATS <= 2 * C::NT && C::W3DW_FLOATS <= C::NT, "table entries per thread");
        auto w1_entry = [&](int e) {
            const int ln = e / 12, k = e - ln * 12;
            return k < 9 ? frag1[UBD_SEP_FRAG_FLOATS + (k * 6) * 64 + ln] : (k < 11 ? frag1[(k - 9) * 64 + ln] : 0.f);
        };
        auto w3pw_entry = [&](int e) {
            const int nt = e >> 9, ln = (e >> 3) & 63, s = e & 7, lq = ln >> 4, li = ln & 15;
            const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
            return s < 6 ? frag3[((ch % 6) * 2 + nt) * 64 + 16 * (ch / 6) + li] : 0.f;
        };
        auto w3dw_entry = [&](int e) {
            const int lq = e / 72, r = e - lq * 72, t = r >> 3, s = r & 7;
            const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
            return s < 6 ? frag3[UBD_SEP_FRAG_FLOATS + (t * 6 + ch % 6) * 64 + 16 * (ch / 6)] : 0.f;
        };
        const int t = (int)threadIdx.x;
        const int e1 = t + C::NT < 64 * 12 ? t + C::NT : t, e3 = t + C::NT < C::W3PW_FLOATS ? t + C::NT : t, ed = t < C::W3DW_FLOATS ? t : 0;
        const float v0 = w1_entry(t), v1 = w1_entry(e1), v2 = w3pw_entry(t), v3 = w3pw_entry(e3), v4 = w3dw_entry(ed);
        const float v5 = (t & 31) < UBD_C ? (t < 32 ? bias1 : bias3)[t & 31] : 0.f;
        w1t[t] = v0; w1t[e1] = v1; w3pw[t] = v2; w3pw[e3] = v3; w3dw[ed] = v4;
        if (t < 64) bt[t] = v5;
        if constexpr (IN_U8) { if (t < 256) lut[t] = ((float)t - pre_sub) / pre_div; }     // the expression the conversion applied per element: same bits
    }
    if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
        if constexpr (IN_U8) __syncthreads();                                        // the lookup table is complete
        convert_x(cur);
    }
    __syncthreads();
    S123_BLOCK_STAMP(29);
    phase0b(cur);
    __syncthreads();
    S123_BLOCK_STAMP(30);
    // Per tile t (two block barriers): [request the input patch of t + 1] -> phase A(t) -> [patch of t + 1 complete in LDS] -> barrier
    // -> phase B(t) on waves 0-3 beside phase 0b(t + 1) (waves 4-7 take twice the units) -> barrier.  Phase 0b of the next tile
    // touches nothing phase B reads (the inherited L2 column is moved at the start of phase A instead), so the four waves that
    // have no L3 row do not idle through phase B.
    for (int it = 0;; ++it) {
        S123_STAMP(0);
        const bool new_strip = !COLD && cur.tx == 0;                                 // block-uniform
        if (new_strip) S123_BLOCK_STAMP(4 + cur.ord);
        if (new_strip && threadIdx.x == 0) pending = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the raw ticket: arithmetic on it here would wait for the round trip
        const int img = cur.img, oy0 = cur.ty * C::TH3, ox0 = xo3(cur);
        const bool has_next = nx1.ls < strips;                                       // block-uniform
        const int R0 = 2 * oy0 - 1, C0 = 2 * ox0 - 1;                                // L2 pixel of position (0, 0)
        if (has_next) { if constexpr (PLAIN) dma_x(nx1); else load_x(nx1); }        // phase 0b(t) has read the patch: the next tile's may land
        if (wid == C::NW - 1 && lane < C::LR * 6) {
            // L2 position 0 (column 2*ox0 - 1): the previous tile's position 32, or L3's zero padding at the left image edge
            const int row = lane / 6, ch4 = lane - row * 6;
            f32x4 v = z4;
            if (!COLD && cur.tx > 0) v = *(const f32x4 *)(carry_buf + (((it + 1) & 1) * C::LR + row) * C::LP + 4 * ch4);   // COLD: nobody to inherit from -- the tile's first L3 column is not stored
            *(f32x4 *)(l2 + (row * C::LC) * C::LP + 4 * ch4) = v;
        }
        S123_STAMP(1);

        // ---- phase A: L2 on positions (0..8, 1..32)
        const bool mask_needed = (R0 < 0) || (R0 + C::LR > H2) || (C0 + C::LC > W2);
        {
            float dwv[3][6];
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int s = 0; s < 6; ++s) dwv[o][s] = 0.f;
#pragma unroll
            for (int yy = 0; yy < 5; ++yy) {
                if (yy < rw + 2) {                                                   // wave-uniform
                    f32x4 v4[3];
                    f32x2 v2[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        v4[kx] = *(const f32x4 *)(a1p + ro4[kx] + yy * (C::PW * UBD_C));
                        v2[kx] = *(const f32x2 *)(a1p + ro2[kx] + yy * (C::PW * UBD_C));
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
                    if (yy >= 2) {
                        const int o = yy - 2;
                        f32x4 acc0 = b2A, acc1 = b2B;
#pragma unroll
                        for (int s = 0; s < 6; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][0], dwv[o][s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[s][1], dwv[o][s], acc1, 0, 0, 0);
                        }
                        float cap = __builtin_inff();
                        if (mask_needed) {
                            const bool ok = (unsigned)(C0 + pos) < (unsigned)W2 && (unsigned)(R0 + rb + o) < (unsigned)H2;
                            cap = ok ? cap : 0.f;                                    // L3's zero padding
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc0[r] = ubd_relu_cap(acc0[r], cap); acc1[r] = ubd_relu_cap(acc1[r], cap); }
                        float *dst = l2 + l2w + o * (C::LC * C::LP);
                        *(f32x4 *)dst = acc0;
                        if (q < 2) *(f32x4 *)(dst + 16) = acc1;
                        if (half == 1 && i == 15) {                                  // position 32: the next tile's position 0
                            float *cd = carry_buf + ((it & 1) * C::LR + rb + o) * C::LP + 4 * q;
                            *(f32x4 *)cd = acc0;
                            if (q < 2) *(f32x4 *)(cd + 16) = acc1;
                        }
                    }
                }
            }
        }
        S123_STAMP(2);
        if (has_next) {
            if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next patch has landed (requested a whole phase A ago)
            else convert_x(nx1);
        }
        S123_STAMP(3);
        __syncthreads();                                                             // L2 tile and the next input patch are complete
        S123_STAMP(4);

        // ---- phase B (waves 0-3: L3 output row oy0 + wid) beside phase 0b of the next tile (all waves)
        if (wid < 4) {
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
            f32x4 acc0 = *(const f32x4 *)(bt + 32 + 4 * q), acc1 = *(const f32x4 *)(bt + 48 + 4 * q);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa0[s] : pb0[s - 4], dv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa1[s] : pb1[s - 4], dv[s], acc1, 0, 0, 0);
            }
            store_tile_relu_nb(y, ((size_t)img * H4 + oy) * W4, ox0, oy < H4 ? W4 : 0, lane, acc0, acc1, COLD && cur.tx > 0 && i == 0);   // bias already in
            S123_STAMP(5);
            if (!has_next) break;
            float tap[2][9];
            l1_taps(0, tap[0]); l1_taps(1, tap[1]);
            l1_finish(nx1, std::integral_constant<int, 2>{}, tap, 0);
        } else {
            S123_STAMP(5);
            if (!has_next) break;
            float tap[2][9];
            l1_taps(0, tap[0]); l1_taps(1, tap[1]);
            l1_finish(nx1, std::integral_constant<int, 2>{}, tap, 0);
            l1_taps(2, tap[0]); l1_taps(3, tap[1]);
            l1_finish(nx1, std::integral_constant<int, 2>{}, tap, 2);
        }
        S123_STAMP(6);
        if (new_strip && threadIdx.x == 0) ring[(cur.ord + D) & 3] = ticket_ls(pending);       // visible after the barrier below and the next tile's
        __syncthreads();                                                             // a1 patch of the next tile complete; phase B is over
        S123_STAMP(7);
        cur = nx1;
        nx1 = advance(nx1);
    }
    S123_BLOCK_STAMP(2);
    check_out();
}
