// Weight repacking shared by forward.hip, fwd16.hip and backward.hip: flat Keras-ordered parameters -> per-lane MFMA fragments.
// Every packer is a device function over a VIRTUAL thread grid (vtid of vthreads), so that the stand-alone kernels and the
// one-launch prologue of the bf16 train step (backward.hip train_prologue16_kernel: four packers + the zeroing of the gradient
// vector and of the loss scratch in ONE launch instead of six ~5-us launches) run the same code.
#pragma once
#include "common.h"

#define UBD_BWD_DGRAD_FLOATS (UBD_NUM_DIL * UBD_DIL_FRAG_FLOATS)
#define UBD_BWD_SEP_FLOATS (6 * 2 * 64)
#define UBD_BWD_DIRECT_FLOATS (UBD_BWD_DGRAD_FLOATS + 3 * UBD_BWD_SEP_FLOATS)
#define UBD_BWD_FRAG_FLOATS (UBD_BWD_DIRECT_FLOATS + UBD_NUM_DIL * UBD_WINO_FRAG_FLOATS)

template <typename T> __device__ __forceinline__ unsigned short ubd_to_bits16(float v) { return __builtin_bit_cast(unsigned short, (T)v); }

struct pack_args {
    size_t off_sep_dw[3], off_sep_pw[3];
    size_t off_dil_k[UBD_NUM_DIL];
    int c_in;
};

__device__ __forceinline__ void pack_weights_body(const float *__restrict__ params, float *__restrict__ wfrag, const pack_args &a, int vtid, int vthreads)
{
    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const int total = 3 * per_sep + UBD_NUM_DIL * UBD_DIL_FRAG_FLOATS;
    for (int idx = vtid; idx < total; idx += vthreads) {
        float v = 0.f;
        if (idx < 3 * per_sep) {
            int s = idx / per_sep, r = idx % per_sep;
            int cin = s == 0 ? a.c_in : UBD_C;
            if (r < UBD_SEP_FRAG_FLOATS) {
                int lane = r & 63, nt = (r >> 6) & 1, step = r >> 7;
                int q = lane >> 4, co = (lane & 15) + 16 * nt;
                int ch = cin == UBD_C ? 6 * q + step : ((step == 0 && q < cin) ? q : -1);
                if (ch >= 0 && co < UBD_C) v = params[a.off_sep_pw[s] + (size_t)ch * UBD_C + co];
            } else {
                r -= UBD_SEP_FRAG_FLOATS;
                int lane = r & 63, ts = r >> 6, step = ts % 6, tap = ts / 6;
                int q = lane >> 4;
                int ch = cin == UBD_C ? 6 * q + step : ((step == 0 && q < cin) ? q : -1);
                if (ch >= 0) v = params[a.off_sep_dw[s] + (size_t)tap * cin + ch];
            }
        } else {
            int r = idx - 3 * per_sep;
            int L = r / UBD_DIL_FRAG_FLOATS;
            r %= UBD_DIL_FRAG_FLOATS;
            int lane = r & 63, nt = (r >> 6) & 1, tj = r >> 7, j = tj % 6, t = tj / 6;
            int q = lane >> 4, co = (lane & 15) + 16 * nt;
            int ci = j < 4 ? 4 * q + j : 16 + 2 * q + (j - 4);
            if (co < UBD_C) v = params[a.off_dil_k[L] + ((size_t)t * UBD_C + ci) * UBD_C + co];
        }
        wfrag[idx] = v;
    }
}

// 16-bit B fragments of the dilated layers: lane (n = lane&15, q = lane>>4), chunk c, element j:
//   k = 32c + 8q + j (flat (tap, ci) index, zero for k >= 216), co = n + 16 nt (zero for co >= 24)
// transpose = 1: fragments of the data-gradient convolution, W'[t][ci'][co'] = W[8 - t][co'][ci'] (flipped taps,
// channels swapped), same lane layout.
template <typename T>
__device__ __forceinline__ void pack16_body(const float *__restrict__ params, unsigned *__restrict__ out, size_t off0, size_t layer_stride, int transpose, int vtid, int vthreads)
{
    const int total = UBD_NUM_DIL * UBD_DIL16_FRAG_U32;
    for (int idx = vtid; idx < total; idx += vthreads) {
        const int L = idx / UBD_DIL16_FRAG_U32;
        int r = idx % UBD_DIL16_FRAG_U32;
        const int dw = r & 3, lane = (r >> 2) & 63, cn = r >> 8, nt = cn & 1, c = cn >> 1;
        const int q = lane >> 4, co = (lane & 15) + 16 * nt;
        const float *wk = params + off0 + (size_t)L * layer_stride;
        unsigned short h[2];
        for (int e = 0; e < 2; ++e) {
            const int k = 32 * c + 8 * q + 2 * dw + e;
            float v = 0.f;
            if (k < 216 && co < UBD_C) {
                if (!transpose) v = wk[(size_t)k * UBD_C + co];
                else { const int t = k / UBD_C, ci = k - t * UBD_C; v = wk[((size_t)(8 - t) * UBD_C + co) * UBD_C + ci]; }
            }
            h[e] = ubd_to_bits16<T>(v);
        }
        out[idx] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    }
}

// Ready-to-use per-lane operands of the 24-channel separable layers L2 and L3 for the one-kernel 16-bit stem (sep123_16.h), which
// loads them per phase instead of keeping both layers' sets in registers: layer l (0: L2, 1: L3), slot s, lane (i = lane & 15,
// q = lane >> 4), 4 dwords.  Slots 0..4: tap-folded diagonal depthwise fragments of channels 0..15 (two taps x 16 channels per
// MFMA: k-slot 8q + e <-> tap 2j + (q >> 1), channel 8 (q & 1) + e == row i), slots 5..7: channels 16..23 (four taps x 8 channels:
// k-group q <-> tap 4j + q, row i = 4 qq + r <-> channel 16 + 2 qq + r, r < 2), slots 8, 9: the pointwise A operands of the two N
// tiles (k-slot 8q + e <-> this lane's channels 4q..4q+3, 16+2q, 17+2q; zeros for e = 6, 7) -- exactly what sepconv16_kernel<24, S>
// (fwd16.hip) builds in its prologue from the fp32 fragments.
struct pack_sep16_args { size_t off_dw[2], off_pw[2]; };
pack_sep16_args ubd_pack_sep16_args(const ubd_handle *h);      // fwd16.hip
template <typename T>
__device__ __forceinline__ void pack_sep16_ready_body(const float *__restrict__ params, unsigned *__restrict__ out, const pack_sep16_args &a, int vtid, int vthreads)
{
    for (int idx = vtid; idx < UBD_SEP16_READY_U32; idx += vthreads) {
        const int dw = idx & 3, lane = (idx >> 2) & 63, sl = (idx >> 8) % 10, l = (idx >> 8) / 10;
        const int i = lane & 15, q = lane >> 4;
        const float *kdw = params + a.off_dw[l], *kpw = params + a.off_pw[l];      // [tap][24] and [ch][24]
        unsigned short h[2] = {0, 0};
        if (sl < 5) {
            const int ts = 2 * sl + (q >> 1), e0 = i - 8 * (q & 1);
            if (ts < 9 && e0 >= 0 && e0 < 8 && (e0 >> 1) == dw) h[e0 & 1] = ubd_to_bits16<T>(kdw[ts * UBD_C + i]);
        } else if (sl < 8) {
            const int ts = 4 * (sl - 5) + q, r = i & 3, e = 2 * (i >> 2) + r;
            if (ts < 9 && r < 2 && (e >> 1) == dw) h[e & 1] = ubd_to_bits16<T>(kdw[ts * UBD_C + 16 + e]);
        } else if (dw < 3) {
            const int nt = sl - 8, co = i + 16 * nt;
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * dw + e;
                const int ch = s < 4 ? 4 * q + s : 16 + 2 * q + (s - 4);
                h[e] = ubd_to_bits16<T>(co < UBD_C ? kpw[(size_t)ch * UBD_C + co] : 0.f);
            }
        }
        out[idx] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    }
}

// Backward weight fragments:
//   dgrad[L][t'][j][nt][lane] = W_L[8-t'][co' ][ci']   with ci' = input channel of the dgrad conv
//        (= forward output channel) from (j, q) as in the forward packing, co' = (lane&15)+16nt
//   seppwT[s][step][tile][lane]: A operand of the dDW product, A[rho = lane&15][k = q]
//        = pw[ch(rho, tile)][co = 6q + step];   CIN==24: ch = 6*(rho>>2) + (rho&3) + 4*tile (tile 1: rho&3 < 2)
//                                               CIN< 24: tile 0 only, ch = rho>>2 if (rho&3)==0 and ch < CIN
struct pack_bwd_args {
    size_t off_sep_pw[3];
    size_t off_dil_k[UBD_NUM_DIL];
    int c_in;
};

__device__ __forceinline__ void pack_bwd_body(const float *__restrict__ params, float *__restrict__ out, const pack_bwd_args &a, int vtid, int vthreads)
{
    for (int idx = vtid; idx < UBD_BWD_DIRECT_FLOATS; idx += vthreads) {
        float v = 0.f;
        if (idx < UBD_BWD_DGRAD_FLOATS) {
            int L = idx / UBD_DIL_FRAG_FLOATS, r = idx % UBD_DIL_FRAG_FLOATS;
            int lane = r & 63, nt = (r >> 6) & 1, tj = r >> 7, j = tj % 6, t = tj / 6;
            int q = lane >> 4, cop = (lane & 15) + 16 * nt;
            int cip = j < 4 ? 4 * q + j : 16 + 2 * q + (j - 4);
            if (cop < UBD_C) v = params[a.off_dil_k[L] + ((size_t)(8 - t) * UBD_C + cop) * UBD_C + cip];
        } else {
            int r = idx - UBD_BWD_DGRAD_FLOATS;
            int s = r / UBD_BWD_SEP_FLOATS;
            r %= UBD_BWD_SEP_FLOATS;
            int cin = s == 0 ? a.c_in : UBD_C;
            int lane = r & 63, tile = (r >> 6) & 1, step = r >> 7;
            int rho = lane & 15, q = lane >> 4;
            int ch = -1;
            if (cin == UBD_C) {
                int sub = (rho & 3) + 4 * tile;
                if (sub < 6) ch = 6 * (rho >> 2) + sub;
            } else if (tile == 0 && (rho & 3) == 0 && (rho >> 2) < cin) {
                ch = rho >> 2;
            }
            if (ch >= 0) v = params[a.off_sep_pw[s] + (size_t)ch * UBD_C + 6 * q + step];
        }
        out[idx] = v;
    }
}

