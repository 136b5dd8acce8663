// Shared host/device definitions of libubd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ubd.h"

#define UBD_C 24              // n_filters (net.py:289)
#define UBD_NUM_DIL 6         // dense dilated layers L4..L9 (net.py:298-304)

struct ubd_comm;
struct ubd_handle {
    ubd_config cfg;
    ubd_comm *comm;           // RCCL communicator + communication stream (comm.hip), nullptr: single GPU
    int k_out;                // 1 + n_classes
    // offsets (in floats) into the flat Keras-ordered parameter vector
    size_t off_sep_dw[3], off_sep_pw[3], off_sep_b[3];
    size_t off_dil_k[UBD_NUM_DIL], off_dil_b[UBD_NUM_DIL];
    size_t off_head_k, off_head_b;
    size_t n_params;
    int num_cus;
    int pp_lds_attr_set;      // pp_front_lds_kernel's dynamic-LDS limit has been raised on this handle's device
    int fuse_force;           // UBD_STEM named a fused variant explicitly: use it at any launch size
    int split_headbwd;        // UBD_HEADBWD=split: bf16 train step with classes: head data gradient and head weight gradient as two kernels (diagnostics / tests)
    int no_pair_dilbwd;       // UBD_DILBWD=pair8: narrow sub-grids (dilation 16 on 128-wide maps) keep the 8-wide tiles instead of pairs in 16-wide ones (diagnostics / tests)
    int split_dilbwd;         // UBD_DILBWD=split: bf16 dilated backward as two kernels per layer (diagnostics / tests)
    int split_stem16;         // UBD_STEM16=split: 16-bit pass with separate L1 and L2 kernels (diagnostics / tests)
    int pp_global, pp_split, pp_poison, pp_serial_tail, pp_threads_512;   // UBD_PP_* test hooks, read ONCE in ubd_create (never in the launch path)
    int sepb_x_regs;          // UBD_SEPB16_X=regs: bf16 backward of L1 stages its fp32 input patch through registers even where LDS-DMA applies (diagnostics / tests)
    int split_sepbwd32;       // UBD_SEPBWD=split: fp32 train step with the stand-alone data-gradient kernels of the separable layers (sep_dx_kernel) instead of G tiles built inside the weight-gradient kernels (diagnostics / tests)
    int chain_reduce;         // bf16 train step: weight-gradient kernels total the previous producer's partial rows at their end (default; UBD_REDUCE=batched: the two stand-alone launches)
    int direct_dil16;         // UBD_DILCONV16=direct: 16-bit forward dilated layers with the direct (unstaged) kernel (diagnostics / tests)
    int fuse_stem;            // inference stem: 2 = L1 -> L2 -> L3 in one kernel (default with fml padding), 1 = L2 -> L3 fused, 0 = three kernels (UBD_STEM=fused123|fused|unfused)
    int use_wino;             // 1: Winograd F(2x2,3x3) dilated layers (default), 0: direct implicit GEMM (UBD_DILCONV=direct)
    int loss_chain;           // UBD_LOSS=chain: the loss as its five dependent launches (the batch-global mode's form) instead of the one-launch kernel (diagnostics / tests)
    int wino_x6;              // forward Winograd products as three-way bf16 split products on the bf16 MFMA (wino6.hip; default), 0: on the fp32 MFMA (UBD_DILCONV=wino32)
};

static const int UBD_DILATIONS[UBD_NUM_DIL] = {1, 2, 4, 8, 16, 1};

// XCD-aware tile order for kernels whose blocks walk logical indices L = blockIdx.x, + gridDim.x, ...: consecutive
// blocks sit on different XCDs (8 of them, each with its own L2), so neighbouring tiles -- which share halo rows and
// cache lines -- would be fetched into two L2s.  Logical index L is mapped to tile (L & 7) * total / 8 + (L >> 3): the
// blocks of one XCD walk one contiguous eighth of the tiles.  A bijection for any grid size; identity when the tile
// count is not a multiple of 8.
#if defined(__HIPCC__)
__device__ __forceinline__ int ubd_xcd_tile(int L, int total)
{
    return (total & 7) == 0 ? (L & 7) * (total >> 3) + (L >> 3) : L;
}
#endif

#if defined(__HIPCC__)
// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at lds_dst + 16 * lane, no VGPR round trip) written as the
// instruction itself.  hipcc treats the builtin form as a pending LDS write and drains it (s_waitcnt vmcnt(0)) before
// the next ds_write -- and before LDS reads it cannot disambiguate (ds_read_b64_tr_b16) --, which exposes the whole
// fetch latency in every kernel that touches LDS while the next tile's DMA is in flight.  The asm form is outside hipcc's
// bookkeeping: the caller retires it with its own counted `s_waitcnt vmcnt(N)` + barrier before the staged bytes are read
// (cdna_hip_programming.md, "What hipcc does not do").  lds_dst: wave-uniform LDS byte address (readfirstlane'd here).
// lds_dst: wave-uniform LDS byte address (ubd_lds_addr of the block's LDS object + an offset kept in scalar registers; the
// generic-pointer form below costs a 64-bit VGPR pair per piece when hipcc hoists it out of a loop)
__device__ __forceinline__ void ubd_glds16_at(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// same with a wave-uniform 64-bit base in scalar registers + a 32-bit per-lane byte offset (no 64-bit vector address arithmetic)
__device__ __forceinline__ void ubd_glds16_sbase(const void *base_uniform, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst) : "memory");
}
// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds): lanes whose byte offset lies outside the descriptor's
// range land as ZEROS in LDS -- 'same' padding and ragged tile borders need no clamped address and no fix-up pass afterwards
// (checked on gfx950 by tools/ubench/buflds.hip: offsets past the end and wrapped negative offsets both zero-fill)
__device__ __forceinline__ void ubd_blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
// the 4-byte form (64 lanes x 4 B land at lds_dst + 4 * lane): rows of fp32 pixels that are only 4-byte aligned
__device__ __forceinline__ void ubd_blds4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
// ReLU of a matrix-pipe result in ONE instruction, with an optional zero mask: clamp(x, 0, cap), cap = +inf (keep) or 0 (the
// position is padding: result 0).  fmaxf(x, 0) on an MFMA output costs two v_max (hipcc first canonicalises a value it
// cannot prove quiet: v_max x, x) plus a v_cndmask for the mask; v_med3_f32 needs neither and returns the same value for
// every non-NaN x (NaN -> 0, as fmaxf).
__device__ __forceinline__ float ubd_relu_cap(float x, float cap) { return __builtin_amdgcn_fmed3f(x, 0.f, cap); }
__device__ __forceinline__ unsigned ubd_lds_addr(const void *lds_generic)
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds_generic);
}
__device__ __forceinline__ void ubd_glds16(const void *gsrc, const void *lds_generic) { ubd_glds16_at(gsrc, ubd_lds_addr(lds_generic)); }
#endif

#if defined(__HIPCC__)
// Tile index -> (tx, ty, img) without integer divisions in the tile loop.  hipcc expands an unsigned division by a
// run-time divisor into ~25 scalar instructions; the persistent stem / backward kernels decode two tile indices per tile
// and were spending 75 of the 93 scalar instructions per 16-pixel row tile on them (PMC).  Division by the
// launch constant d is n * m >> 32 with m = floor((2^32 - 1) / d) + 1, exact while n * d < 2^32 (checked once; tile
// counts are < 2^24 and tiles_x, tiles_y < 2^8 for every supported shape) -- otherwise the real division is used.
// floor(n / D) for 0 <= n < NMAX as (n * m) >> sh with n * m < 2^24 (one v_mul_u32_u24 + one shift); checked at compile time
template <int D, int NMAX> struct ubd_magic24 {
    static constexpr int find_sh()
    {
        for (int sh = 8; sh < 24; ++sh) {
            const long m = (1L << sh) / D + 1;
            if (m * NMAX >= (1L << 24)) continue;
            bool ok = true;
            for (long n = 0; n < NMAX && ok; ++n) ok = ((n * m) >> sh) == n / D;
            if (ok) return sh;
        }
        return -1;
    }
    static constexpr int sh = find_sh();
    static_assert(sh > 0, "no 24-bit magic number for this divisor / range");
    static constexpr unsigned m = (1u << sh) / D + 1;
};

// v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 issue at a QUARTER of the rate of the 24-bit multiplies (16 cycles per wave instead of 4): index arithmetic
// whose operands provably fit 24 bits says so (hipcc cannot know the ranges and takes the 32-bit forms -- four of them per 16-pixel unit in the L1 loop of
// the 16-bit stem, ~15 % of that loop's issue cycles; round 4)
__device__ __forceinline__ int ubd_mul24(int a, int b) { return __mul24(a, b); }
template <int D, int NMAX> __device__ __forceinline__ int ubd_div24(int n) { return (int)(__umul24((unsigned)n, ubd_magic24<D, NMAX>::m) >> ubd_magic24<D, NMAX>::sh); }
struct ubd_tile_decoder {
    unsigned tiles_x, tiles_y, mx, my, total;
    bool fast;
    __device__ __forceinline__ void init(int tx_, int ty_, int total_)
    {
        tiles_x = (unsigned)tx_; tiles_y = (unsigned)ty_; total = (unsigned)total_;
        mx = 0xFFFFFFFFu / tiles_x + 1u;
        my = 0xFFFFFFFFu / tiles_y + 1u;
        const unsigned dmax = tiles_x > tiles_y ? tiles_x : tiles_y;
        fast = (unsigned long long)total * dmax < 0x100000000ull;
    }
    __device__ __forceinline__ unsigned div_x(unsigned v) const { return fast ? (tiles_x == 1u ? v : __umulhi(v, mx)) : v / tiles_x; }
    __device__ __forceinline__ unsigned div_y(unsigned v) const { return fast ? (tiles_y == 1u ? v : __umulhi(v, my)) : v / tiles_y; }
    // logical index L (blockIdx.x + k * gridDim.x) -> XCD-aware tile -> coordinates
    __device__ __forceinline__ void decode(int L, int &tx, int &ty, int &img) const
    {
        const unsigned tile = (unsigned)ubd_xcd_tile(L, (int)total);
        const unsigned r = div_x(tile);
        tx = (int)(tile - r * tiles_x);
        const unsigned im = div_y(r);
        ty = (int)(r - im * tiles_y);
        img = (int)im;
    }
};
#endif

void ubd_set_error(const char *fmt, ...);

#define UBD_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ubd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,             \
                          hipGetErrorString(_e));                                        \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

#define UBD_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ubd_set_error(__VA_ARGS__);                                                  \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

static inline size_t ubd_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- workspace carving (all offsets 256-byte aligned) -------------------------------
// packed MFMA weight fragments, see forward.hip
#define UBD_DIL_FRAG_FLOATS (9 * 6 * 2 * 64)     // per dilated layer
#define UBD_SEP_FRAG_FLOATS (6 * 2 * 64)         // pointwise fragments per separable layer
#define UBD_SEP_DW_FLOATS (9 * 6 * 64)           // per-lane depthwise taps per separable layer
#define UBD_WINO_FRAG_FLOATS (16 * 6 * 64 * 2)   // Winograd-domain weights per dilated layer (wino.hip)
#define UBD_FWD_DIRECT_FLOATS (3 * (UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS) + UBD_NUM_DIL * UBD_DIL_FRAG_FLOATS)
#define UBD_WINO6_FRAG_U32 (16 * 2 * 3 * 64 * 4)  // the same weights as three bf16 pieces per value (wino6.hip): [xi 16][nt 2][piece 3][lane 64] x 4 dwords
#define UBD_FWD_WINO6_OFF (UBD_FWD_DIRECT_FLOATS + UBD_NUM_DIL * UBD_WINO_FRAG_FLOATS)      // in floats (= dwords) from the start of the packed weights
#define UBD_FWD_FRAG_FLOATS (UBD_FWD_WINO6_OFF + UBD_NUM_DIL * UBD_WINO6_FRAG_U32)

struct ubd_fwd_layout {
    size_t off_tickets;   // int32 [64]: strip ticket counter of the one-kernel inference stem (zeroed per pass)
    size_t off_wfrag;     // packed weights
    size_t off_a1, off_a2;   // (n, H/2, W/2, 24) activations: L1 out, L2 out
    size_t off_b[2];         // (n, H/4, W/4, 24) ping-pong for L3..L9 outputs (inference)
    size_t off_acts[7];      // training: L3..L9 outputs kept
    size_t total;
};

// ---- cross-file internals ------------------------------------------------------------------
bool ubd_comm_fused(const ubd_handle *h);
bool ubd_comm_global_loss(const ubd_handle *h);
int ubd_comm_rank(const ubd_handle *h);
enum { UBD_RED_F64 = 0, UBD_RED_I32 = 1, UBD_RED_U32 = 2 };
int ubd_comm_allreduce_raw(ubd_handle *h, void *buf, size_t count, int kind, hipStream_t st);      // in-place SUM
int ubd_comm_allgather_u32(ubd_handle *h, const unsigned *send_one, unsigned *recv_world, hipStream_t st);
extern "C" int ubd_comm_world(const ubd_handle *h);
int ubd_comm_begin_tail(ubd_handle *h, float *grads, hipStream_t st);
int ubd_comm_finish(ubd_handle *h, float *grads, hipStream_t st);
void ubd_fwd_layout_compute(const ubd_handle *h, int n, int H, int W, int training, ubd_fwd_layout *L);
struct pp_lds_args;       // pp_lds.h: one-launch postprocess job (the fused stem kernel can carry one for a previous batch)
int ubd_forward_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                     int n, int H, int W, float *logits, char *ws, const ubd_fwd_layout &L, hipStream_t st, bool inference = false,
                     const pp_lds_args *pp_job = nullptr);
// fills `job` for ubd_postprocess's arguments when the whole postprocess fits the one-launch in-block form with `threads`
// threads per block (returns 1), else returns 0; negative: argument error (ubd_last_error)
int ubd_pp_fill_job(ubd_handle *hd, const float *logits, int n, int map_h, int map_w, float logit_threshold, int scale, float min_area,
                    int32_t *binary_map, int32_t *quads, int32_t *classes, int32_t *counts, int cap, void *workspace,
                    size_t workspace_bytes, int threads, pp_lds_args *job);
bool ubd_forward_uses_fused_stem(const ubd_handle *h, int n, int H, int W);
void ubd_launch_dilconv(const ubd_handle *h, int epi, const float *frag, const float *aux, int dilation,
                        const float *in, float *out, int n, int H4, int W4, hipStream_t st);
int ubd_grid_for(long waves_needed, int num_cus, int waves_per_block, int blocks_per_cu);
int ubd_loss_impl(const float *logits, int k_out, const int32_t *y_true, long npix, float *loss, float *dlogits,
                  char *ws, hipStream_t st, ubd_handle *h = nullptr, bool prezeroed = false);
size_t ubd_loss_zero_bytes(void);
extern "C" size_t ubd_loss_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w);
void ubd_launch_pack_wino(const ubd_handle *h, const float *params, float *out, int transpose, hipStream_t st);
void ubd_launch_dilconv_wino(const ubd_handle *h, int epi, const float *frag, const void *aux, int aux_dtype, int dilation,
                             const float *in, float *out, int n, int H4, int W4, hipStream_t st, const float *head = nullptr);
void ubd_launch_pack_wino6(const ubd_handle *h, const float *params, unsigned *out, hipStream_t st);
void ubd_launch_dilconv_wino6(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, int dilation,
                              const float *in, float *out, int n, int H4, int W4, hipStream_t st, const float *head = nullptr);
void ubd_launch_pack_direct(const ubd_handle *h, const float *params, float *wfrag, hipStream_t st);
size_t ubd_forward16_workspace_bytes(int n, int H, int W);
int ubd_pack16_workspace(ubd_handle *h, const float *params, char *ws, size_t ws_bytes, hipStream_t st);
int ubd_forward16(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H, int W,
                  float *logits, char *ws, size_t ws_bytes, hipStream_t st);
#define UBD_SEP16_READY_U32 (2 * 10 * 64 * 4)      // L2, L3 of the 16-bit pass: per-lane ready operands [layer 2][slot 10][lane 64] x 4 dwords (pack.h pack_sep16_ready_body)
#define UBD_DIL16_FRAG_U32 (7 * 2 * 64 * 4)          // 16-bit dilated layer: [chunk 7][nt 2][lane 64] x 4 dwords (8 halves)
void ubd_launch_dilconv16(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, const void *mask, int d,
                          const void *in, void *out, int n, int H4, int W4, hipStream_t st, float *logits3 = nullptr);
void ubd_launch_pack16(const ubd_handle *h, const float *params, unsigned *out, int transpose, hipStream_t st);
struct ubd_fwd16_layout { size_t off_wfrag32, off_wfrag16, off_a1, off_a2, off_acts[7], total; };
void ubd_fwd16_layout_compute(int n, int H, int W, int training, ubd_fwd16_layout *L);
int ubd_forward16_layout(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H,
                         int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st, bool inference = false);
