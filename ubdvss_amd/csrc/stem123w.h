// The one-kernel inference stem L1 -> L2 -> L3 (stem123.h) re-cut for FOUR waves per SIMD: one 1024-thread block per CU, every
// wave within 128 registers (included by forward.hip after stem123.h; same tile, same LDS images, same arithmetic -- the outputs
// are bit-identical to stem123_kernel's).
//
// Reference semantics: net.py:292-296 (ZeroPadding2D + SeparableConv2D stride 2 / SeparableConv2D 'same' / ZeroPadding2D +
// SeparableConv2D stride 2, each + bias + ReLU) and the fused (x - 127.5) / 127.5 of net.py:217-218 for uint8 input.
//
// Why (round 5; measurements in profiles/r05_experiment_notes.txt, micro-benchmarks tools/ubench/l2unit.hip and mfma_fill.hip):
//   * on gfx950 the fp32 MFMA (v_mfma_f32_16x16x4_f32) and the fp32 vector ALU do NOT overlap on a SIMD -- their times add, whichever
//     waves issue them -- and the fp32 MFMA itself issues every 32 cycles from ONE wave but every ~25 / ~20 cycles per SIMD from two /
//     four waves.  So a SIMD's time per tile is (MFMAs x 20..32) + (vector instructions x ~2) and everything else is stall;
//   * stem123_kernel (two fat waves per SIMD, 256 registers, spills) keeps its SIMDs busy ~45 % of a 9.2 k-cycle tile: serial
//     LDS -> FMA -> MFMA -> clamp -> LDS chains, two block barriers per tile, L3 / L1 phases that are latency- and LDS-bound.
// Here:
//   * sixteen waves of <= 128 registers.  L2's depthwise weights live in 18 registers instead of 54: lane (i, q) keeps the taps
//     4g + (i & 3) of its six channels and every FMA takes its weight through a DPP quad broadcast (v_fmac_f32_dpp
//     quad_perm:[j,j,j,j]; the four lanes of a quad belong to the same channel group q).  A DPP FMA issues at about half the
//     rate of a plain one (measured), still the cheaper side of the trade;
//   * ONE block barrier per tile.  In step t every wave first computes one L2 row unit of tile t (rows 0-7 x two halves), then,
//     by role: waves 0-9 the L1 units of tile t + 1 (LDS-bound work under the younger waves' MFMAs), waves 10-11 L2 row 8, waves
//     12-15 the L3 rows of tile t - 1 (from the other of two L2 images).  The a1 patch is single-buffered: a wave counts itself
//     into an LDS counter once its L2 unit has read its taps, and the L1 units wait for that count (row 8's readers have a second
//     counter for the L1 units that write a1 rows 8-10);
//   * the input patch is double-buffered and requested two tiles ahead, so it is simply there when the step starts;
//   * everything else (tile walk in ticketed row strips, carried 33rd L2 column, input patch by LDS-DMA or through registers,
//     the postprocess job of an earlier batch in the first blocks) is stem123.h's, with the thread count as a parameter.
#pragma once

struct s123w_cfg { static constexpr int NT = 1024, NW = 16; };

template <int CIN> struct s123w_x {
    using B = s23_cfg;
    using W = s123w_cfg;
    static constexpr int AC = B::LC + 1;                           // a1 patch columns that are needed: 1 .. 34 of the 35
    static constexpr int XH = 2 * B::PH + 1, XW = 2 * AC + 1;      // input patch 23 x 69
    static constexpr int XE = XH * XW * CIN;
    static constexpr int XREGS = (XE + W::NT - 1) / W::NT;
    static constexpr int XCH = ((XW + 1) * CIN + 3) / 4;           // 16-byte chunks per LDS patch row (one aligned column on the left: stem123.h)
    static constexpr int XS = XCH * 4;
    static constexpr int XP_FLOATS = XH * XS;
    static constexpr int A1_FLOATS = B::PH * B::PW * UBD_C;
    static constexpr int NPIX = B::PH * AC;                        // 374 L1 outputs per tile
    static constexpr int UNITS = (NPIX + 15) / 16;                 // 24
    static constexpr int W2_FLOATS = 64 * 12 + 32;                  // L2's pointwise fragments per lane + its bias (24 + zeros): read per unit, not kept in registers
    static constexpr int W1_FLOATS = 64 * 12 + 64 + 16;             // L1 per-lane weights, biases of L1 / L3, ring of strip ids (8) + LDS counters
    static constexpr int LUT_FLOATS = 256;
    static constexpr int STEM_FLOATS = A1_FLOATS + 2 * B::L2_FLOATS + 2 * XP_FLOATS + B::W3PW_FLOATS + B::W3DW_FLOATS + B::CARRY_FLOATS + W1_FLOATS + W2_FLOATS + LUT_FLOATS;
    static constexpr int PP_FLOATS = (PP_LDS_MAX_BYTES + 3) / 4;
    static constexpr int SMEM_FLOATS = STEM_FLOATS > PP_FLOATS ? STEM_FLOATS : PP_FLOATS;
    static_assert(XCH <= 64 && (A1_FLOATS % 4) == 0 && (B::L2_FLOATS % 4) == 0, "patch rows are single 16-byte-aligned DMA pieces");
    static_assert(UNITS == 24 && W::NW == 16, "the L1 units of a tile are dealt to the waves 0-9");
    static_assert(STEM_FLOATS * 4 <= 160 * 1024, "LDS");
};

// acc += w[quad lane J] * x: the weight comes from lane J of the lane's quad (DPP quad broadcast, no extra instruction).  The DPP
// operand is a launch constant register (never written inside the tile loop: the two wait states a VALU write -> DPP read needs
// are checked on the ISA by tools/lint_dpp.py).
__device__ __forceinline__ void s123w_fmac_qp(float &acc, float w, float x, int j)
{
    switch (j) {
    case 0: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    case 1: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    case 2: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    default: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    }
}

template <int CIN, int IN_U8, int PLAIN>
__global__ __launch_bounds__(s123w_cfg::NT) void stem123w_kernel(const void *__restrict__ xin, float *__restrict__ y,
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
    using X = s123w_x<CIN>;
    constexpr int NT = s123w_cfg::NT, NW = s123w_cfg::NW;
    __shared__ __attribute__((aligned(16))) float smem[X::SMEM_FLOATS];                     // ONE LDS object
#ifdef UBD_STAMPS
#define S123W_BLOCK_STAMP(k) do { if (stamps && threadIdx.x == 0 && (k) < 32) stamps[(size_t)gridDim.x * NW * 16 * 8 + (size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S123W_BLOCK_STAMP(k) do {} while (0)
#endif
    S123W_BLOCK_STAMP(0);
    // ---- the postprocess of an EARLIER batch rides along (ubd_forward_postprocess; stem123.h)
    if (pj.n > 0) {
        for (int im = (int)blockIdx.x; im < pj.n; im += (int)gridDim.x) {
            pp_image_lds<NT, true>((int *)smem, pj, im);
            __syncthreads();
        }
    }
    S123W_BLOCK_STAMP(1);
    const int Dc = ((W4 + 15) >> 4) >= 4 ? 1 : 5 - ((W4 + 15) >> 4);                 // strips claimed ahead: the input patch is requested TWO tiles ahead
    const int njob = pj.n > 0 ? (pj.n < (int)gridDim.x ? pj.n : (int)gridDim.x) : 0;
    auto ticket_ls = [&](int k) { return k < njob * Dc ? k : k + ((int)gridDim.x - njob) * Dc; };   // ticket -> logical strip
    int t0_early = 0;
    if (threadIdx.x == 0 && (int)blockIdx.x < njob) t0_early = __hip_atomic_fetch_add(ticket, Dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float *a1p = smem;
    float *l2 = a1p + X::A1_FLOATS;
    float *xp = l2 + 2 * C::L2_FLOATS;                            // two L2 images: L3 of tile t - 1 runs beside L2 of tile t
    float *w3pw = xp + 2 * X::XP_FLOATS, *w3dw = w3pw + C::W3PW_FLOATS;
    float *carry_buf = w3dw + C::W3DW_FLOATS;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;

    // ---- L2's weights: depthwise taps 4g + (i & 3) of channel c(q, s) in w2[s][g] (18 registers; quad broadcast per FMA),
    //      pointwise fragments in 12 registers, the bias vectors in the accumulator layout
    float w2[6][3];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);
        const int src_lane = 16 * (ch / 6) + i, ss = ch % 6;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int t = 4 * g + (i & 3);
            w2[s][g] = t < 9 ? frag2[UBD_SEP_FRAG_FLOATS + (t * 6 + ss) * 64 + src_lane] : 0.f;
        }
    }
    float *w1t = carry_buf + C::CARRY_FLOATS;
    float *bt = w1t + 64 * 12;                                     // [0,32): L1's bias (24 + zeros), [32,64): L3's
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // ---- tile sequence: whole row strips of tiles handed out by tickets (stem123.h)
    const int tiles_x = (W4 + 15) >> 4, tiles_y = (H4 + C::TH3 - 1) / C::TH3;
    const int strips = n * tiles_y;
    const int D = tiles_x >= 4 ? 1 : 5 - tiles_x;
    int *ring = (int *)(bt + 64);
    int *cnt = ring + 8;                                           // [0]: L2 units of rows 0-7 that have read their taps, [1]: of row 8 (both only ever count up)
    float *w2t = bt + 64 + 16;                                     // [lane][12]: pwf2[s][nt] at 2 s + nt; then L2's bias [32]
    float *lut = w2t + X::W2_FLOATS;
    struct tpos { int tx, ty, img, ord, ls; };
    auto strip_pos = [&](int ord, int tx) {
        tpos p;
        const int ls = __builtin_amdgcn_readfirstlane(ring[ord & 7]);
        const int sidx = ubd_xcd_tile(ls < strips ? ls : strips - 1, strips);
        p.ty = (int)((unsigned)sidx % (unsigned)tiles_y);
        p.img = (int)((unsigned)sidx / (unsigned)tiles_y);
        p.tx = tx; p.ord = ord; p.ls = ls;
        return p;
    };
    auto advance = [&](tpos p) {
        if (p.tx + 1 < tiles_x) { ++p.tx; return p; }
        return strip_pos(p.ord + 1, 0);
    };

    // ---- input patch of a tile through registers (uint8 / preprocessed input; stem123.h)
    unsigned xreg[X::XREGS];
    constexpr int RWF = X::XW * CIN;
    const int e0_row = (int)threadIdx.x / RWF, e0_col = (int)threadIdx.x - e0_row * RWF;
    auto tile_interior = [&](tpos p) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, ix0 = 64 * p.tx - 3;
        return (iy0 >= 0) && (ix0 >= 0) && (iy0 + X::XH <= H) && (ix0 + X::XW <= W);
    };
    auto load_x = [&](tpos p) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (64 * p.tx - 3) * CIN, WC = W * CIN;
        const unsigned char *img8 = (const unsigned char *)xin + (size_t)p.img * H * WC * (IN_U8 ? 1 : 4);
        const bool interior = tile_interior(p);
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            int gy = iy0 + pr, gf = fx0 + pcf;
            if (!interior) { gy = min(max(gy, 0), H - 1); gf = min(max(gf, 0), WC - 1); }
            else if (k == X::XREGS - 1) gy = min(gy, H - 1);
            const unsigned off = (unsigned)__umul24(gy, WC) + (unsigned)gf;
            if constexpr (IN_U8) xreg[k] = img8[off];
            else xreg[k] = ((const unsigned *)img8)[off];
            pr += NT / RWF; pcf += NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };
    auto fix_border = [&](tpos p) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (64 * p.tx - 3) * CIN, WC = W * CIN;
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(fx0 + pcf) < (unsigned)WC;
            xreg[k] = inside ? xreg[k] : (IN_U8 ? 0x100u : __builtin_bit_cast(unsigned, pre_sub));
            pr += NT / RWF; pcf += NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };
    auto convert_x = [&](tpos p, int buf) {
        const bool plain = !IN_U8 && pre_sub == 0.f && pre_div == 1.f;
        if (!tile_interior(p)) fix_border(p);
        int pr = e0_row, pcf = e0_col;
#pragma unroll
        for (int k = 0; k < X::XREGS; ++k) {
            const int e = k * NT + (int)threadIdx.x;
            if (e < X::XE) {
                float *dst = xp + buf * X::XP_FLOATS + pr * X::XS + CIN + pcf;
                if constexpr (IN_U8) *dst = xreg[k] > 255u ? 0.f : lut[xreg[k] & 255u];
                else *dst = plain ? __builtin_bit_cast(float, xreg[k]) : (__builtin_bit_cast(float, xreg[k]) - pre_sub) / pre_div;
            }
            pr += NT / RWF; pcf += NT % RWF;
            if (pcf >= RWF) { pcf -= RWF; ++pr; }
        }
    };

    // ---- PLAIN: the patch of a tile by 16-byte LDS-DMA, one piece per patch row (zeros outside the image; stem123.h)
    const unsigned lds_xp = ubd_lds_addr(xp);
    auto dma_x = [&](tpos p, int buf, int ln) {
        const int iy0 = 4 * p.ty * C::TH3 - 5, fx0 = (64 * p.tx - 4) * CIN, WC = W * CIN;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)xin + (size_t)p.img * H * WC * 4), 0,
                                                                        (int)((unsigned)H * WC * 4u), 0x00020000);
        unsigned off0 = (unsigned)ln * 16u;
        off0 = (unsigned)(fx0 + 4 * ln) < (unsigned)WC ? off0 : 0x80000000u;
        if (ln < X::XCH) {
#pragma unroll
            for (int k = 0; k < (X::XH + NW - 1) / NW; ++k) {
                const int row = k * NW + wid;
                if (row >= X::XH) break;                                             // wave-uniform
                const int term = ((iy0 + row) * WC + fx0) * 4;
                ubd_blds16(rsrc, (unsigned)term + off0, lds_xp + (unsigned)((buf * X::XP_FLOATS + row * X::XS) * 4));
            }
        }
    };

    // ---- L1 units: 16 flat pixels of the 11 x 34 needed a1 pixels each; units 0-16 touch a1 rows 0-7 only, units 17-23 rows 8-10.
    //      First tile of a block: waves 0-7 take units wid and wid + 16, waves 8-15 unit wid.  Steady state: the ten waves 6-15
    //      (k = wid - 6) take unit k, unit k + 10 (k < 7) and one of the late units 14 + k (k >= 3).
    // Everything that is derived from the lane index is recomputed per tile from an OPAQUE copy of it (ln): kept across the tile loop
    // those address registers were spilled, and a scratch reload at the top of a tile costs a memory round trip (in-kernel stamps).
    auto unit_rc = [&](int u, int li) {                                              // a1 patch (row << 8 | column 1..34) of this lane's pixel, -1: none
        const int p = u * 16 + li;
        const bool live = u < X::UNITS && p < X::NPIX;
        const int pz = live ? p : 0;
        const int ar = ubd_div24<X::AC, 16 * 24>(pz), ac = pz - ar * X::AC + 1;
        return live ? ((ar << 8) | ac) : -1;
    };
    const int half = wid & 1, rb = wid >> 1;                                         // L2: every wave = (half, L2 row wid >> 1) for rows 0-7; afterwards waves 4 and 5 take row 8
    const int pos = 1 + 16 * half + i;
    int ro4[3], ro2[3];                                                              // tap offsets in row 0 of the a1 patch (the unit's row is a wave-uniform term)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int pcol = pos + kx;
        ro4[kx] = pcol * UBD_C + 4 * s123_slot(q, pcol);
        ro2[kx] = pcol * UBD_C + 4 * s123_slot(4 + (q >> 1), pcol) + 2 * (q & 1);
    }

    if (threadIdx.x == 0) {
        const bool had_job = (int)blockIdx.x < njob;
#pragma unroll
        for (int j = 0; j < 5; ++j) ring[j] = had_job ? ticket_ls(t0_early + j) : (int)blockIdx.x * D + j;   // only the first D are this block's: the rest are overwritten before use
        cnt[0] = 0; cnt[1] = 0; cnt[2] = 0; cnt[3] = 0;
    }
    __syncthreads();
    S123W_BLOCK_STAMP(3);
    tpos cur = strip_pos(0, 0);
    auto check_out = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int left = __hip_atomic_fetch_add(ticket + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == (int)gridDim.x - 1) {
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket + 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (cur.ls >= strips) { S123W_BLOCK_STAMP(2); check_out(); return; }
    tpos nx1 = advance(cur);
    tpos nx2 = advance(nx1);
    int pending = 0;
#ifdef UBD_STAMPS
#define S123W_STAMP(k) do { if (stamps && it < 16 && lane == 0) stamps[(((size_t)blockIdx.x * NW + wid) * 16 + it) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S123W_STAMP(k) do {} while (0)
#endif
    // LDS counters: arrive = one ds_add behind the wave's LDS reads (the LDS executes a wave's instructions in order: when the add
    // is performed the reads have been); wait = poll until the count has reached `target`
    const unsigned lds_cnt = ubd_lds_addr(cnt);
    auto cnt_arrive = [&](int which) {                                               // `which` is a literal at every call
        if (lane == 0) {
            if (which == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(lds_cnt), "v"(1) : "memory");
            else asm volatile("ds_add_u32 %0, %1 offset:4" :: "v"(lds_cnt), "v"(1) : "memory");
        }
    };
    auto cnt_wait = [&](int which, int target) {
        for (;;) {
            const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(4);
        }
        asm volatile("" ::: "memory");
    };

    // ---- L1 (net.py:292-293): weights once per wave and step; per unit nine taps -> depthwise FMAs -> two MFMAs -> clamp (0 outside
    //      L1's map = L2's padding); the results wait in registers until the a1 patch may be overwritten
    struct l1w { f32x4 wa, wb, wc, bA, bB; };
    auto l1_weights = [&](int ln) {
        l1w w;
        w.wa = *(const f32x4 *)(w1t + ln * 12); w.wb = *(const f32x4 *)(w1t + ln * 12 + 4); w.wc = *(const f32x4 *)(w1t + ln * 12 + 8);
        w.bA = *(const f32x4 *)(bt + 4 * (ln >> 4)); w.bB = *(const f32x4 *)(bt + 16 + 4 * (ln >> 4));
        return w;
    };
    auto l1_compute = [&](tpos p, const float *xb, const l1w &w, int rc, int ln, f32x4 &acc0, f32x4 &acc1) {
        asm volatile("" : "+v"(rc));                                                 // the offsets derived from it are recomputed here: hoisted out of the tile loop they were spilled
        const int lq = ln >> 4, lqc = lq < CIN ? lq : CIN - 1;                       // lanes without a channel (zero weights) read what their neighbours read: an LDS broadcast
        const int rc0 = rc < 0 ? 1 : rc;                                             // lanes without a pixel compute on pixel (0, 1) and store nothing
        const int ar = rc0 >> 8, ac = rc0 & 255;
        const int u_rd = (2 * ar) * X::XS + (2 * (ac - 1) + 1) * CIN + lqc;
        float tap[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) tap[ky * 3 + kx] = xb[u_rd + ky * X::XS + kx * CIN];
        float dv = 0.f;
        dv = fmaf(tap[0], w.wa[0], dv); dv = fmaf(tap[1], w.wa[1], dv); dv = fmaf(tap[2], w.wa[2], dv); dv = fmaf(tap[3], w.wa[3], dv);
        dv = fmaf(tap[4], w.wb[0], dv); dv = fmaf(tap[5], w.wb[1], dv); dv = fmaf(tap[6], w.wb[2], dv); dv = fmaf(tap[7], w.wb[3], dv);
        dv = fmaf(tap[8], w.wc[0], dv);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.wc[1], dv, w.bA, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.wc[2], dv, w.bB, 0, 0, 0);
        const int A0y = 2 * p.ty * C::TH3 - 2, A0x = 32 * p.tx - 2;                  // a1 pixel of patch (0, 0)
        const bool inside = rc >= 0 && (unsigned)(A0y + ar) < (unsigned)H2 && (unsigned)(A0x + ac) < (unsigned)W2;
        const float cap = inside ? __builtin_inff() : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] = ubd_relu_cap(acc0[r], cap); acc1[r] = ubd_relu_cap(acc1[r], cap); }
    };
    auto l1_store = [&](int rc, int ln, f32x4 acc0, f32x4 acc1) {
        asm volatile("" : "+v"(rc));
        if (rc >= 0) {
            const int lq = ln >> 4, ar = rc >> 8, ac = rc & 255;
            *(f32x4 *)(a1p + (ar * C::PW + ac) * UBD_C + 4 * s123_slot(lq, ac)) = acc0;
            if (lq < 2) *(f32x4 *)(a1p + (ar * C::PW + ac) * UBD_C + 4 * s123_slot(4 + lq, ac)) = acc1;
        }
    };

    // ---- one L2 row unit (net.py:294): positions (row, 1 + 16 half .. 16 + 16 half) of tile p -> L2 image `l2b`.  The head -- tap reads,
    //      the first tap row's FMAs, the count into cnt[which] behind the last read -- runs at raised priority: the L1 units of the next
    //      tile wait for ALL sixteen waves' counts, and the youngest waves are otherwise starved through their first FMAs
    auto l2_unit = [&](float *l2b, int row, int it, tpos p, int which, int ln) {
        const int li = i, lq = q;
        const float *ap = a1p + row * (C::PW * UBD_C);
        f32x4 v4[3][3];
        f32x2 v2[3][3];
        float dwv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto rd = [&](int ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                v4[ky][kx] = *(const f32x4 *)(ap + ro4[kx] + ky * (C::PW * UBD_C));
                v2[ky][kx] = *(const f32x2 *)(ap + ro2[kx] + ky * (C::PW * UBD_C));
            }
        };
        auto fm = [&](int ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx, g = t >> 2, j = t & 3;
                s123w_fmac_qp(dwv[0], w2[0][g], v4[ky][kx][0], j);
                s123w_fmac_qp(dwv[1], w2[1][g], v4[ky][kx][1], j);
                s123w_fmac_qp(dwv[2], w2[2][g], v4[ky][kx][2], j);
                s123w_fmac_qp(dwv[3], w2[3][g], v4[ky][kx][3], j);
                s123w_fmac_qp(dwv[4], w2[4][g], v2[ky][kx][0], j);
                s123w_fmac_qp(dwv[5], w2[5][g], v2[ky][kx][1], j);
            }
        };
        rd(0); rd(1);
        f32x4 cv = z4;
        if (half == 0 && ln < 6 && p.tx > 0) cv = *(const f32x4 *)(carry_buf + (((it + 1) & 1) * C::LR + row) * C::LP + 4 * ln);   // position 0 = the previous tile's position 32
        asm volatile("" ::: "memory");
        fm(0);
        rd(2);
        asm volatile("" ::: "memory");
        cnt_arrive(which);
        if (half == 0 && ln < 6) *(f32x4 *)(l2b + (row * C::LC) * C::LP + 4 * ln) = cv;   // (zeros at the left image edge: L3's padding)
        const f32x4 pw0 = *(const f32x4 *)(w2t + ln * 12), pw1 = *(const f32x4 *)(w2t + ln * 12 + 4), pw2 = *(const f32x4 *)(w2t + ln * 12 + 8);
        f32x4 acc0 = *(const f32x4 *)(w2t + 64 * 12 + 4 * lq), acc1 = *(const f32x4 *)(w2t + 64 * 12 + 16 + 4 * lq);   // the bias (zeros beyond channel 23)
        fm(1); fm(2);
        const float pwf2[12] = {pw0[0], pw0[1], pw0[2], pw0[3], pw1[0], pw1[1], pw1[2], pw1[3], pw2[0], pw2[1], pw2[2], pw2[3]};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[2 * s], dwv[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pwf2[2 * s + 1], dwv[s], acc1, 0, 0, 0);
        }
        const int R0 = 2 * p.ty * C::TH3 - 1, C0 = 32 * p.tx - 1;                    // L2 pixel of position (0, 0)
        const bool ok = (unsigned)(C0 + pos) < (unsigned)W2 && (unsigned)(R0 + row) < (unsigned)H2;
        const float cap = ok ? __builtin_inff() : 0.f;                               // outside L2's map: L3's zero padding
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] = ubd_relu_cap(acc0[r], cap); acc1[r] = ubd_relu_cap(acc1[r], cap); }
        float *dst = l2b + (row * C::LC + pos) * C::LP + 4 * lq;
        *(f32x4 *)dst = acc0;
        if (lq < 2) *(f32x4 *)(dst + 16) = acc1;
        if (half == 1 && li == 15) {                                                 // position 32: the next tile's position 0
            float *cd = carry_buf + ((it & 1) * C::LR + row) * C::LP + 4 * lq;
            *(f32x4 *)cd = acc0;
            if (lq < 2) *(f32x4 *)(cd + 16) = acc1;
        }
    };

    // ---- L3 (net.py:295-296) output row wid & 3 of tile p from L2 image `l2b`
    auto l3_unit = [&](const float *l2b, tpos p, int ln) {
        const int li = ln & 15, lq = ln >> 4;
        const int l2r = (2 * (wid & 3) * C::LC + 2 * li) * C::LP;
        const int oy = p.ty * C::TH3 + (wid & 3);
        float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float *pp = l2b + l2r + (ky * C::LC + kx) * C::LP;
                const f32x4 v4 = *(const f32x4 *)(pp + 4 * lq);
                const f32x2 v2 = *(const f32x2 *)(pp + 16 + 2 * lq);
                const float *wt = w3dw + (lq * 9 + ky * 3 + kx) * 8;
                const f32x4 w4 = *(const f32x4 *)wt;
                const f32x2 wv2 = *(const f32x2 *)(wt + 4);
                dv[0] = fmaf(v4[0], w4[0], dv[0]); dv[1] = fmaf(v4[1], w4[1], dv[1]);
                dv[2] = fmaf(v4[2], w4[2], dv[2]); dv[3] = fmaf(v4[3], w4[3], dv[3]);
                dv[4] = fmaf(v2[0], wv2[0], dv[4]); dv[5] = fmaf(v2[1], wv2[1], dv[5]);
            }
        const f32x4 pa0 = *(const f32x4 *)(w3pw + ln * 8), pa1 = *(const f32x4 *)(w3pw + 512 + ln * 8);
        const f32x2 pb0 = *(const f32x2 *)(w3pw + ln * 8 + 4), pb1 = *(const f32x2 *)(w3pw + 512 + ln * 8 + 4);
        f32x4 acc0 = *(const f32x4 *)(bt + 32 + 4 * lq), acc1 = *(const f32x4 *)(bt + 48 + 4 * lq);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa0[s] : pb0[s - 4], dv[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 4 ? pa1[s] : pb1[s - 4], dv[s], acc1, 0, 0, 0);
        }
        store_tile_relu_nb(y, ((size_t)p.img * H4 + oy) * W4, p.tx * 16, oy < H4 ? W4 : 0, ln, acc0, acc1);      // bias already in
    };

    // ---- first tiles: patches of tiles 0 and 1 -> LDS, weight tables, L1 of tile 0
    const bool has1 = nx1.ls < strips;
    if constexpr (PLAIN) { dma_x(cur, 0, lane); if (has1) dma_x(nx1, 1, lane); } else load_x(cur);
    {
        static_assert(64 * 12 <= NT && C::W3PW_FLOATS <= NT && C::W3DW_FLOATS <= NT, "one table entry per thread");
        const int t = (int)threadIdx.x;
        const int e1 = t < 64 * 12 ? t : 0, ed = t < C::W3DW_FLOATS ? t : 0;
        float v0, v2, v4;
        {
            const int ln = e1 / 12, k = e1 - ln * 12;
            v0 = k < 9 ? frag1[UBD_SEP_FRAG_FLOATS + (k * 6) * 64 + ln] : (k < 11 ? frag1[(k - 9) * 64 + ln] : 0.f);
        }
        {
            const int nt = t >> 9, ln = (t >> 3) & 63, s = t & 7, lq = ln >> 4, li = ln & 15;
            const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
            v2 = s < 6 ? frag3[((ch % 6) * 2 + nt) * 64 + 16 * (ch / 6) + li] : 0.f;
        }
        {
            const int lq = ed / 72, r = ed - lq * 72, tp = r >> 3, s = r & 7;
            const int ch = s < 4 ? 4 * lq + s : 16 + 2 * lq + (s - 4);
            v4 = s < 6 ? frag3[UBD_SEP_FRAG_FLOATS + (tp * 6 + ch % 6) * 64 + 16 * (ch / 6)] : 0.f;
        }
        const float v5 = (t & 31) < UBD_C ? (t < 32 ? bias1 : bias3)[t & 31] : 0.f;
        float v6;
        {
            const int ln = e1 / 12, k = e1 - ln * 12, s6 = k >> 1, nt = k & 1, lq = ln >> 4, li = ln & 15;
            const int ch = s6 < 4 ? 4 * lq + s6 : 16 + 2 * lq + (s6 - 4);
            v6 = frag2[((ch % 6) * 2 + nt) * 64 + 16 * (ch / 6) + li];
        }
        const float v7 = t < UBD_C ? bias2[t < UBD_C ? t : 0] : 0.f;
        if (t < 64 * 12) { w1t[t] = v0; w2t[t] = v6; }
        if (t < 32) w2t[64 * 12 + t] = v7;
        w3pw[t] = v2;
        if (t < C::W3DW_FLOATS) w3dw[t] = v4;
        if (t < 64) bt[t] = v5;
        if constexpr (IN_U8) { if (t < 256) lut[t] = ((float)t - pre_sub) / pre_div; }
    }
    if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
        if constexpr (IN_U8) __syncthreads();
        convert_x(cur, 0);
        if (has1) { load_x(nx1); convert_x(nx1, 1); }
    }
    __syncthreads();
    S123W_BLOCK_STAMP(29);
    {
        const l1w w = l1_weights(lane);
        f32x4 a0, a1;
        const int rc0 = unit_rc(wid, i);
        l1_compute(cur, xp, w, rc0, lane, a0, a1);
        l1_store(rc0, lane, a0, a1);
        if (wid < X::UNITS - NW) {                                                   // wave-uniform
            const int rc1 = unit_rc(wid + 16, i);
            l1_compute(cur, xp, w, rc1, lane, a0, a1);
            l1_store(rc1, lane, a0, a1);
        }
    }
    __syncthreads();
    S123W_BLOCK_STAMP(30);

    // Per tile t ONE block barrier:
    //   [request the input patch of tile t + 2] -> every wave: one L2 row unit of tile t (rows 0-7), counted into cnt[0] behind its tap
    //   reads -> by role: waves 0-3 (the oldest: first through their L2 unit) the L3 rows of tile t - 1 from the other L2 image | waves
    //   4, 5: L2 row 8 (cnt[1]) | waves 6-15: the L1 units of tile t + 1 (its patch landed a step ago): the first one is computed at once
    //   and stored when cnt[0] says that every wave has read its taps; the units that write a1 rows 8-10 wait for cnt[1]
    //   -> [patch of t + 2 landed] -> barrier.
    tpos prv = cur;
    int it = 0;
    for (;; ++it) {
        S123W_STAMP(0);
        int ln = lane;
        asm volatile("" : "+v"(ln));                                                 // per-tile copy of the lane index for the DMA offsets (kept across the loop they were spilled)
        const bool new_strip = cur.tx == 0;
        if (new_strip) S123W_BLOCK_STAMP(4 + cur.ord);
        if (new_strip && threadIdx.x == 0) pending = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool has_next = nx1.ls < strips, has_next2 = nx2.ls < strips;         // block-uniform
        float *l2cur = l2 + (it & 1) * C::L2_FLOATS;
        if (has_next2) { if constexpr (PLAIN) dma_x(nx2, it & 1, ln); else load_x(nx2); }   // buffer it & 1 held tile t's patch: L1 of tile t read it a step ago
        S123W_STAMP(1);
        l2_unit(l2cur, rb, it, cur, 0, ln);
        S123W_STAMP(2);
        if (wid < 4) { if (it > 0) l3_unit(l2 + ((it + 1) & 1) * C::L2_FLOATS, prv, ln); }
        else if (wid < 6) l2_unit(l2cur, 8, it, cur, 1, ln);
        S123W_STAMP(3);
        if (has_next) {
            // the 24 L1 units of tile t + 1 are DEALT from a queue (an LDS counter): whoever is through its own units takes the next
            // one -- the oldest waves win every issue slot and would otherwise sit at the barrier while the youngest still work
            const float *xb = xp + ((it + 1) & 1) * X::XP_FLOATS;
            const l1w w = l1_weights(ln);
            bool waited0 = false, waited1 = false;
            for (;;) {
                int u = 0;
                if (ln == 0) u = __hip_atomic_fetch_add(cnt + 2 + (it & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                u = __builtin_amdgcn_readfirstlane(u);
                if (u >= X::UNITS) break;
                const int rc = unit_rc(u, ln & 15);
                f32x4 a0, a1;
                l1_compute(nx1, xb, w, rc, ln, a0, a1);
                if (!waited0) { cnt_wait(0, 16 * (it + 1)); waited0 = true; }       // every tap read of rows 0-7 is done: a1 rows 0-7 may be overwritten
                if (u >= 17 && !waited1) { cnt_wait(1, 2 * (it + 1)); waited1 = true; }   // a1 rows 8-10: row 8's L2 units have read them
                l1_store(rc, ln, a0, a1);
            }
        }
        S123W_STAMP(6);
        if (has_next2) {
            if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else convert_x(nx2, it & 1);
        }
        if (new_strip && threadIdx.x == 0) ring[(cur.ord + D) & 7] = ticket_ls(pending);
        if (threadIdx.x == 0) cnt[2 + ((it + 1) & 1)] = 0;                           // the next step's unit queue (this step's is still being drawn from)
        __syncthreads();                                                             // a1 patch of tile t + 1, L2 image of tile t, input patch of tile t + 2
        S123W_STAMP(7);
        if (!has_next) break;
        prv = cur;
        cur = nx1;
        nx1 = nx2;
        nx2 = advance(nx2);
    }
    if (wid < 4) l3_unit(l2 + (it & 1) * C::L2_FLOATS, cur, lane);                   // L3 of the block's last tile
    S123W_BLOCK_STAMP(2);
    check_out();
}
