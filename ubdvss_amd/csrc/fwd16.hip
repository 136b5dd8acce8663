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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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

#define UBD_DIL16_FRAG_U32 (7 * 2 * 64 * 4)          // per dilated layer: [chunk 7][nt 2][lane 64] x 4 dwords (8 halves)

// ------------------------------------------------------------------------------------ pack
// 16-bit B fragments of the dilated layers: lane (n = lane&15, q = lane>>4), chunk c, element j:
//   k = 32c + 8q + j (flat (tap, ci) index, zero for k >= 216), co = n + 16 nt (zero for co >= 24)
template <typename T>
__global__ void pack16_kernel(const float *__restrict__ params, unsigned *__restrict__ out, size_t off0, size_t layer_stride)
{
    const int total = UBD_NUM_DIL * UBD_DIL16_FRAG_U32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int L = idx / UBD_DIL16_FRAG_U32;
        int r = idx % UBD_DIL16_FRAG_U32;
        const int dw = r & 3, lane = (r >> 2) & 63, cn = r >> 8, nt = cn & 1, c = cn >> 1;
        const int q = lane >> 4, co = (lane & 15) + 16 * nt;
        const float *wk = params + off0 + (size_t)L * layer_stride;
        unsigned short h[2];
        for (int e = 0; e < 2; ++e) {
            const int k = 32 * c + 8 * q + 2 * dw + e;
            float v = 0.f;
            if (k < 216 && co < UBD_C) v = wk[(size_t)k * UBD_C + co];
            h[e] = to_bits<T>(v);
        }
        out[idx] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    }
}

// ------------------------------------------------------------------------------------ shared epilogue
// D layout: col = lane&15 (channel), row = 4*(lane>>4) + reg (pixel).  y = relu(acc + bias), narrowed to T and
// written through this wave's 768-byte LDS tile so that the 16 pixels x 48 B leave as 16-byte stores.
template <typename T>
__device__ __forceinline__ void store_tile16(unsigned short *__restrict__ y, size_t first_pixel, int npx, int lane,
                                             unsigned short *__restrict__ stile, f32x4 acc0, f32x4 acc1, float b0, float b1)
{
    const int co = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        stile[(4 * q + r) * UBD_C + co] = to_bits<T>(fmaxf(acc0[r] + b0, 0.f));
        if (co < 8) stile[(4 * q + r) * UBD_C + 16 + co] = to_bits<T>(fmaxf(acc1[r] + b1, 0.f));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (lane * 16 < npx * UBD_C * 2) {
        const u32x4 v = *(const u32x4 *)((const char *)stile + lane * 16);
        *(u32x4 *)((char *)(y + first_pixel * UBD_C) + lane * 16) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------ separable layers
// IN_MODE 0: fp32 input, 1: uint8 input (CIN 1/3 only), 2: 16-bit input (CIN 24)
template <int CIN, int STRIDE, int IN_MODE, typename T>
__global__ __launch_bounds__(256) void sepconv16_kernel(const void *__restrict__ xin, unsigned short *__restrict__ y,
                                                        const float *__restrict__ frag, const float *__restrict__ bias, int n,
                                                        int H, int W, int OH, int OW, int pad_lo, float pre_sub, float pre_div)
{
    constexpr int CPL = (CIN == UBD_C) ? 6 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short s_tile[4][16 * UBD_C];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    const float *pwfrag = frag, *dwlane = frag + UBD_SEP_FRAG_FLOATS;
    float dwk[9][CPL], pwf[CPL][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CPL; ++s) dwk[t][s] = dwlane[(t * 6 + s) * 64 + lane];
#pragma unroll
    for (int s = 0; s < CPL; ++s) { pwf[s][0] = pwfrag[(s * 2 + 0) * 64 + lane]; pwf[s][1] = pwfrag[(s * 2 + 1) * 64 + lane]; }
    const float b0 = bias[i], b1 = (i < 8) ? bias[16 + i] : 0.f;
    const bool ch_ok = (CIN == UBD_C) || (q < CIN);
    const int cb = (CIN == UBD_C) ? 6 * q : q;

    const int tiles_x = (OW + 15) >> 4;
    const int total = n * OH * tiles_x;
    const int nwaves = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wid; tile < total; tile += nwaves) {
        const int xt = (int)((unsigned)tile % (unsigned)tiles_x);
        const int rowid = (int)((unsigned)tile / (unsigned)tiles_x);
        const int oy = (int)((unsigned)rowid % (unsigned)OH);
        const int img = (int)((unsigned)rowid / (unsigned)OH);
        const int x0 = xt * 16, ox = x0 + i;
        float dwv[CPL];
#pragma unroll
        for (int s = 0; s < CPL; ++s) dwv[s] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * STRIDE + ky - pad_lo;
            const bool rok = (iy >= 0) && (iy < H);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx;
                const int ix = ox * STRIDE + kx - pad_lo;
                const bool ok = rok && ch_ok && (ix >= 0) && (ix < W) && (ox < OW);
                const size_t e = (((size_t)img * H + (size_t)(rok ? iy : 0)) * W + (size_t)(ok ? ix : 0)) * CIN + cb;
                if constexpr (CIN == UBD_C) {
                    unsigned w0 = 0, w1 = 0, w2 = 0;
                    if (ok) {
                        const unsigned *p = (const unsigned *)((const unsigned short *)xin + e);     // 12 B, 4-byte aligned
                        w0 = p[0]; w1 = p[1]; w2 = p[2];
                    }
                    float v[6];
                    widen2<T>(w0, v[0], v[1]); widen2<T>(w1, v[2], v[3]); widen2<T>(w2, v[4], v[5]);
#pragma unroll
                    for (int s = 0; s < 6; ++s) dwv[s] = fmaf(v[s], dwk[t][s], dwv[s]);
                } else {
                    float v = 0.f;
                    if (ok) {
                        if constexpr (IN_MODE == 1) v = ((float)((const unsigned char *)xin)[e] - pre_sub) / pre_div;
                        else v = (((const float *)xin)[e] - pre_sub) / pre_div;
                    }
                    dwv[0] = fmaf(v, dwk[t][0], dwv[0]);
                }
            }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dwv[s], pwf[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dwv[s], pwf[s][1], acc1, 0, 0, 0);
        }
        const int npx = OW - x0 < 16 ? OW - x0 : 16;
        store_tile16<T>(y, ((size_t)img * OH + oy) * OW + x0, npx, lane, s_tile[wid], acc0, acc1, b0, b1);
    }
}

// ------------------------------------------------------------------------------------ dilated layers
struct a16_frags { u32x4 v[7]; };

template <typename T>
__global__ __launch_bounds__(256) void dilconv16_kernel(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                        const u32x4 *__restrict__ wfrag, const float *__restrict__ bias, int n, int h,
                                                        int w, int d, unsigned in_bytes)
{
    __shared__ __attribute__((aligned(16))) unsigned short s_tile[4][16 * UBD_C];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    u32x4 wr[7][2];
#pragma unroll
    for (int c = 0; c < 7; ++c) { wr[c][0] = wfrag[(c * 2 + 0) * 64 + lane]; wr[c][1] = wfrag[(c * 2 + 1) * 64 + lane]; }
    const float b0 = bias[i], b1 = (i < 8) ? bias[16 + i] : 0.f;
    // this lane's K-slice of chunk c: k0 = 32c + 8q -> tap (4c+q)/3, channel group ((4c+q)%3)*8
    int dyc[7], dxc[7], cic[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int g = 4 * c + q, t = g / 3;
        cic[c] = (g - 3 * t) * 16;                   // byte offset of the 8-channel group inside the 48-byte pixel
        dyc[c] = t < 9 ? (t / 3 - 1) * d : (1 << 28);  // chunk 6, q = 3: beyond K -> always out of range
        dxc[c] = (t % 3 - 1) * d;
    }
    const int tiles_x = (w + 15) >> 4;
    const int total = n * h * tiles_x;
    const int xcd = blockIdx.x & 7;
    const int nblk_x = (gridDim.x + 7 - xcd) >> 3;
    const int chunk = (total + 7) >> 3;
    const int t_begin = xcd * chunk;
    const int t_end = (t_begin + chunk < total) ? t_begin + chunk : total;
    const int stride = nblk_x * 4;
    int tile = t_begin + (int)(blockIdx.x >> 3) * 4 + wid;
    if (tile >= t_end) return;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)in_bytes, 0x00020000);
    const unsigned oob = in_bytes;

    auto load = [&](a16_frags &a, int tl) {
        const int xt = (int)((unsigned)tl % (unsigned)tiles_x);
        const int rowid = (int)((unsigned)tl / (unsigned)tiles_x);
        const int yy = (int)((unsigned)rowid % (unsigned)h);
        const int px = xt * 16 + i;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const int iy = yy + dyc[c], ix = px + dxc[c];
            const bool ok = (iy >= 0) && (iy < h) && (ix >= 0) && (ix < w);
            const unsigned off = ((unsigned)(rowid + dyc[c]) * (unsigned)w + (unsigned)ix) * (unsigned)(UBD_C * 2) + (unsigned)cic[c];
            a.v[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? off : oob), 0, 0);
        }
    };
    auto compute_store = [&](const a16_frags &a, int tl) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            acc0 = h16<T>::mfma(a.v[c], wr[c][0], acc0);
            acc1 = h16<T>::mfma(a.v[c], wr[c][1], acc1);
        }
        const int xt = (int)((unsigned)tl % (unsigned)tiles_x);
        const int rowid = (int)((unsigned)tl / (unsigned)tiles_x);
        const int x0 = xt * 16;
        const int npx = w - x0 < 16 ? w - x0 : 16;
        store_tile16<T>(y, (size_t)rowid * w + x0, npx, lane, s_tile[wid], acc0, acc1, b0, b1);
    };
    a16_frags A0, A1;
    const int t_last = t_end - 1;
    load(A0, tile);
    for (;;) {
        int nxt = tile + stride;
        load(A1, nxt < t_last ? nxt : t_last);
        compute_store(A0, tile);
        tile = nxt;
        if (tile >= t_end) break;
        nxt = tile + stride;
        load(A0, nxt < t_last ? nxt : t_last);
        compute_store(A1, tile);
        tile = nxt;
        if (tile >= t_end) break;
    }
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
    L->off_wfrag16 = off; off += ubd_align_up((size_t)UBD_NUM_DIL * UBD_DIL16_FRAG_U32 * sizeof(unsigned), 256);
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
    const long tiles = (long)n * OH * ((OW + 15) / 16);
    const int grid = ubd_grid_for(tiles, h->num_cus, 4, 8);
    hipLaunchKernelGGL((sepconv16_kernel<CIN, STRIDE, IN_MODE, T>), dim3(grid), dim3(256), 0, st, x, y, frag, bias, n, H, W, OH, OW, pad_lo, sub, div);
}

template <typename T>
static int forward16_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n,
                          int H, int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st)
{
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    float *wfrag = (float *)(ws + L.off_wfrag32);
    unsigned *wfrag16 = (unsigned *)(ws + L.off_wfrag16);
    unsigned short *a1 = (unsigned short *)(ws + L.off_a1), *a2 = (unsigned short *)(ws + L.off_a2);
    ubd_launch_pack_direct(h, params, wfrag, st);                      // fp32 depthwise / pointwise fragments
    hipLaunchKernelGGL((pack16_kernel<T>), dim3(48), dim3(256), 0, st, params, wfrag16, h->off_dil_k[0], h->off_dil_k[1] - h->off_dil_k[0]);
    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const float *sf0 = wfrag, *sf1 = wfrag + per_sep, *sf2 = wfrag + 2 * per_sep;
    const int pad_s2 = h->cfg.fml_compatible ? 1 : 0;
    float sub = 0.f, div = 1.f;
    if (preprocessing == UBD_PRE_MOBILENET) { sub = 127.5f; div = 127.5f; }
    const bool u8 = in_dtype == UBD_IN_U8;
    if (h->cfg.c_in == 1) {
        if (u8) launch_sep16<1, 2, 1, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
        else launch_sep16<1, 2, 0, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
    } else {
        if (u8) launch_sep16<3, 2, 1, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
        else launch_sep16<3, 2, 0, T>(h, images, a1, sf0, params + h->off_sep_b[0], n, H, W, H2, W2, pad_s2, sub, div, st);
    }
    launch_sep16<UBD_C, 1, 2, T>(h, a1, a2, sf1, params + h->off_sep_b[1], n, H2, W2, H2, W2, 1, 0.f, 1.f, st);
    unsigned short *cur = (unsigned short *)(ws + L.off_acts[0]);
    launch_sep16<UBD_C, 2, 2, T>(h, a2, cur, sf2, params + h->off_sep_b[2], n, H2, W2, H4, W4, pad_s2, 0.f, 1.f, st);
    const unsigned in_bytes = (unsigned)((size_t)n * H4 * W4 * UBD_C * 2);
    const long tiles = (long)n * H4 * ((W4 + 15) / 16);
    for (int k = 0; k < UBD_NUM_DIL; ++k) {
        int grid = ubd_grid_for(tiles, h->num_cus, 4, 4);
        grid = (grid + 7) / 8 * 8;
        unsigned short *nxt = (unsigned short *)(ws + L.off_acts[k + 1]);
        hipLaunchKernelGGL((dilconv16_kernel<T>), dim3(grid), dim3(256), 0, st, cur, nxt,
                           (const u32x4 *)(wfrag16 + (size_t)k * UBD_DIL16_FRAG_U32), params + h->off_dil_b[k], n, H4, W4,
                           UBD_DILATIONS[k], in_bytes);
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
                         int W, float *logits, char *ws, const ubd_fwd16_layout &L, hipStream_t st)
{
    UBD_REQUIRE((size_t)n * (H / 4) * (W / 4) * UBD_C * 2 < 0xFFFFFFFFull, "ubd_forward: batch too large for 32-bit buffer offsets; split the batch");
    if (h->cfg.dtype == UBD_BF16) return forward16_impl<__bf16>(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st);
    return forward16_impl<_Float16>(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st);
}

int ubd_forward16(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H, int W,
                  float *logits, char *ws, size_t ws_bytes, hipStream_t st)
{
    UBD_REQUIRE(ws_bytes >= ubd_forward16_workspace_bytes(n, H, W), "ubd_forward: workspace too small for the 16-bit path");
    ubd_fwd16_layout L;
    ubd_fwd16_layout_compute(n, H, W, 0, &L);
    return ubd_forward16_layout(h, params, images, in_dtype, preprocessing, n, H, W, logits, ws, L, st);
}
