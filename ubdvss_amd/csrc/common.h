// Shared host/device definitions of libubd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ubd.h"

#define UBD_C 24              // n_filters (net.py:289)
#define UBD_NUM_DIL 6         // dense dilated layers L4..L9 (net.py:298-304)

struct ubd_handle {
    ubd_config cfg;
    int k_out;                // 1 + n_classes
    // offsets (in floats) into the flat Keras-ordered parameter vector
    size_t off_sep_dw[3], off_sep_pw[3], off_sep_b[3];
    size_t off_dil_k[UBD_NUM_DIL], off_dil_b[UBD_NUM_DIL];
    size_t off_head_k, off_head_b;
    size_t n_params;
    int num_cus;
    int pp_lds_attr_set;      // pp_front_lds_kernel's dynamic-LDS limit has been raised on this handle's device
    int use_wino;             // 1: Winograd F(2x2,3x3) dilated layers (default), 0: direct implicit GEMM (UBD_DILCONV=direct)
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
#define UBD_FWD_FRAG_FLOATS (UBD_FWD_DIRECT_FLOATS + UBD_NUM_DIL * UBD_WINO_FRAG_FLOATS)

struct ubd_fwd_layout {
    size_t off_wfrag;     // packed weights
    size_t off_a1, off_a2;   // (n, H/2, W/2, 24) activations: L1 out, L2 out
    size_t off_b[2];         // (n, H/4, W/4, 24) ping-pong for L3..L9 outputs (inference)
    size_t off_acts[7];      // training: L3..L9 outputs kept
    size_t total;
};

// ---- cross-file internals ------------------------------------------------------------------
void ubd_fwd_layout_compute(const ubd_handle *h, int n, int H, int W, int training, ubd_fwd_layout *L);
int ubd_forward_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                     int n, int H, int W, float *logits, char *ws, const ubd_fwd_layout &L, hipStream_t st, bool inference = false);
void ubd_launch_dilconv(const ubd_handle *h, int epi, const float *frag, const float *aux, int dilation,
                        const float *in, float *out, int n, int H4, int W4, hipStream_t st);
int ubd_grid_for(long waves_needed, int num_cus, int waves_per_block, int blocks_per_cu);
int ubd_loss_impl(const float *logits, int k_out, const int32_t *y_true, long npix, float *loss, float *dlogits,
                  char *ws, hipStream_t st);
extern "C" size_t ubd_loss_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w);
void ubd_launch_pack_wino(const ubd_handle *h, const float *params, float *out, int transpose, hipStream_t st);
void ubd_launch_dilconv_wino(const ubd_handle *h, int epi, const float *frag, const void *aux, int aux_dtype, int dilation,
                             const float *in, float *out, int n, int H4, int W4, hipStream_t st, const float *head = nullptr);
void ubd_launch_pack_direct(const ubd_handle *h, const float *params, float *wfrag, hipStream_t st);
size_t ubd_forward16_workspace_bytes(int n, int H, int W);
int ubd_pack16_workspace(ubd_handle *h, const float *params, char *ws, size_t ws_bytes, hipStream_t st);
int ubd_forward16(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H, int W,
                  float *logits, char *ws, size_t ws_bytes, hipStream_t st);
#define UBD_DIL16_FRAG_U32 (7 * 2 * 64 * 4)          // 16-bit dilated layer: [chunk 7][nt 2][lane 64] x 4 dwords (8 halves)
void ubd_launch_dilconv16(const ubd_handle *h, int epi, const unsigned *frag, const float *bias, const void *mask, int d,
                          const void *in, void *out, int n, int H4, int W4, hipStream_t st, float *logits3 = nullptr);
void ubd_launch_pack16(const ubd_handle *h, const float *params, unsigned *out, int transpose, hipStream_t st);
struct ubd_fwd16_layout { size_t off_wfrag32, off_wfrag16, off_a1, off_a2, off_acts[7], total; };
void ubd_fwd16_layout_compute(int n, int H, int W, int training, ubd_fwd16_layout *L);
int ubd_forward16_layout(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H,
                         int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st, bool inference = false);
