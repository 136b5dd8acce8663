// Backward pass + train-step orchestration of the ubdvss hot path on gfx950.
//
// The reference gets its gradients from TensorFlow autodiff of the Keras graph
// (model.compile / fit_generator, train.py:110-112, :176-188); here every gradient kernel is
// written out.  Notation: layer output Y = relu(Z); G = dL/dZ ("masked" gradient); the kernels pass
// G tensors from layer to layer:
//   head_dx      G9 = (dlogits . hk^T) * (A9 > 0)
//   head_wgrad   dhk = A9^T dlogits, dhb = sum dlogits                       (MFMA, K-dim = pixels)
//   dil_wgrad    dW[t][ci][co] = sum_p X[p+off_t][ci] G[p][co], db = sum_p G (MFMA, K-dim = pixels,
//                M = 216 (+1 row of ones for the bias), N = 24)
//   dilconv<1>   G_below = conv(G, flipped/transposed W, same dilation) * (X > 0)   (forward kernel)
//   sep_bwd      separable layer: recomputes the depthwise output, dpw/db (MFMA), dDW = G pw^T (MFMA,
//                lands directly in the depthwise lane layout), ddw (VALU), writes dDW
//   sep_dx       G_below = depthwise-transpose(dDW) * (X > 0)
// Weight gradients: every block writes one row of a partial-sum matrix, reduce_partials_kernel adds the rows in a
// fixed order into the flat gradient vector (Keras get_weights() order, same as the parameters).
#include "common.h"
#include "pack.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Activation element access: the saved forward activations are fp32 (UBD_F32) or 16-bit (UBD_BF16 / UBD_F16);
// gradient tensors between layers are fp32, except in UBD_BF16 mode (bf16: bwd16.h, sepbwd16.h).
template <typename TX> __device__ __forceinline__ float ld_act(const void *base, size_t idx)
{
    return (float)((const TX *)base)[idx];
}
// 16-bit activation modes use every convolution kernel (and the depthwise intermediate) in the activation type
template <typename TX> __device__ __forceinline__ float rnd_act(float v)
{
    if constexpr (sizeof(TX) == 2) return (float)(TX)v;
    else return v;
}
// six consecutive channels starting at element index idx (idx % 2 == 0)
template <typename TX> __device__ __forceinline__ void ld_act6(const void *base, size_t idx, float (&v)[6])
{
    if constexpr (sizeof(TX) == 4) {
        const f32x2 *p = (const f32x2 *)((const float *)base + idx);
        const f32x2 a = p[0], c = p[1], d = p[2];
        v[0] = a[0]; v[1] = a[1]; v[2] = c[0]; v[3] = c[1]; v[4] = d[0]; v[5] = d[1];
    } else {
        const unsigned *p = (const unsigned *)((const unsigned short *)base + idx);
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        v[0] = (float)__builtin_bit_cast(TX, (unsigned short)(w0 & 0xFFFFu)); v[1] = (float)__builtin_bit_cast(TX, (unsigned short)(w0 >> 16));
        v[2] = (float)__builtin_bit_cast(TX, (unsigned short)(w1 & 0xFFFFu)); v[3] = (float)__builtin_bit_cast(TX, (unsigned short)(w1 >> 16));
        v[4] = (float)__builtin_bit_cast(TX, (unsigned short)(w2 & 0xFFFFu)); v[5] = (float)__builtin_bit_cast(TX, (unsigned short)(w2 >> 16));
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<__bf16>(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<_Float16>(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// gradient tensors between layers: bf16 for bf16 activations, fp32 otherwise (fp16 would underflow without loss scaling)
template <typename TX> struct UBD_G16 { static constexpr bool value = false; };
template <> struct UBD_G16<__bf16> { static constexpr bool value = true; };


// ------------------------------------------------------------------------------------ backward weight fragments (pack.h)
__global__ void pack_bwd_kernel(const float *__restrict__ params, float *__restrict__ out, pack_bwd_args a)
{
    pack_bwd_body(params, out, a, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

#define RP_COLS 16          // elements per block of reduce_partials_kernel
// ------------------------------------------------------------------------------------ partial sums
// Weight gradients are accumulated per block and written as one row of a [blocks][count] partial-sum matrix;
// reduce_partials_kernel adds the rows in a fixed order (deterministic, and no same-address atomics: a few
// thousand fp32 atomics per address cost ~0.4 ms per launch on MI355X).
// One launch reduces up to RP_MAX_JOBS partial-sum matrices (the head and the six dilated layers; the three separable
// layers): ten separate launches of ~6 us each, serialised behind their producers, cost 65 us of the 1.57 ms bf16 step.
#define RP_MAX_JOBS 8
struct rp_job {
    const float *part;
    float *out0, *out1, *out2;
    int nblocks, count, n0, n1, block0;                        // block0: first block of this job in the batched grid
};
struct rp_batch { rp_job job[RP_MAX_JOBS]; int njobs; };

// One block's share of a job: RP_COLS consecutive elements x (256 / RP_COLS) row groups.  Loads in flight per thread: 32 while the
// matrix has that many rows per group left, then 8, then 1 -- the additions happen in the order of the plain 8-wide loop either way
// (acc[u] takes rows ty + G u, + 8 G, + 16 G, ... one after the other), so the stand-alone kernel and the in-kernel tail below give
// the same bits.  s: 256 floats of LDS.
__device__ __forceinline__ void rp_reduce_group(const rp_job &J, int blk, float *s)
{
    const float *__restrict__ part = J.part;
    const int nblocks = J.nblocks, count = J.count, n0 = J.n0, n1 = J.n1;
    constexpr int G = 256 / RP_COLS;
    const int tx = threadIdx.x % RP_COLS, ty = threadIdx.x / RP_COLS;
    const int e = blk * RP_COLS + tx;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < count) {
        int b = ty;
        for (; b + 31 * G < nblocks; b += 32 * G) {
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[k][u] = part[(size_t)(b + G * (8 * k + u)) * count + e];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += v[k][u];
        }
        for (; b + 7 * G < nblocks; b += 8 * G) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(b + G * u) * count + e];
        }
        for (; b < nblocks; b += G) acc[0] += part[(size_t)b * count + e];
    }
    s[ty * RP_COLS + tx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (ty == 0 && e < count) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) v += s[g * RP_COLS + tx];           // fixed order: deterministic
        if (e < n0) J.out0[e] = v;
        else if (e < n0 + n1) J.out1[e - n0] = v;
        else J.out2[e - n0 - n1] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const rp_batch B)
{
    int j = 0;
#pragma unroll
    for (int k = 1; k < RP_MAX_JOBS; ++k)
        if (k < B.njobs && (int)blockIdx.x >= B.job[k].block0) j = k;            // block-uniform
    // block = RP_COLS consecutive elements x (256 / RP_COLS) row groups.
    // 16 columns per block: 5208-element rows give 326 blocks (one per CU and more) instead of 82.
    __shared__ float s[256];
    rp_reduce_group(B.job[j], (int)blockIdx.x - B.job[j].block0, s);
}

// bf16 train step: a weight-gradient kernel ends by totalling the partial rows of the producer IN FRONT of it (complete: stream
// order) -- the rows are microseconds old and still in the Infinity Cache, every block takes one or two column groups with
// 32 loads in flight, and the two stand-alone reduction launches of a pass (15.8 us each: 92 MB read back from memory) disappear.
// part == nullptr: nothing to do.  s: 256 floats of LDS that nobody else touches any more (the caller has passed a barrier).
__device__ __forceinline__ void rp_reduce_tail(const rp_job &J, float *s)
{
    if (J.part == nullptr) return;                                               // kernel-uniform
    const int ngroups = (J.count + RP_COLS - 1) / RP_COLS;
    for (int g = (int)blockIdx.x; g < ngroups; g += (int)gridDim.x) rp_reduce_group(J, g, s);
}

// Sum the four waves' 224 x 32 accumulator sets and write this block's row of the partial-sum matrix
// ([216*24 kernel gradient | 24 bias gradient]).  Waves take turns adding into one lane-linear LDS image
// (ds_read/write_b128, conflict-free): LDS float atomics cost ~3 cycles per LANE on gfx950 and made this epilogue
// the longest phase of the kernel.
// MT: accumulator tiles per wave, starting at M tile mt0 (a block of 4 x (14 / MT) waves: `ks` = the wave's k-step class, the order of the sum)
template <int MT = 14>
__device__ __forceinline__ void wgrad_block_reduce(const f32x4 (&acc)[MT][2], float *__restrict__ red /* 28 KiB */,
                                                   float *__restrict__ prow, int lane, int ks, int mt0 = 0)
{
    f32x4 *img = (f32x4 *)red;                         // [(mt, nt)][lane] x 4 floats (r)
    __syncthreads();                                   // tile buffers are free now
    for (int ph = 0; ph < 4; ++ph) {
        if (ks == ph) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 v = acc[mt][nt];
                    if (ph > 0) v += img[((mt0 + mt) * 2 + nt) * 64 + lane];
                    img[((mt0 + mt) * 2 + nt) * 64 + lane] = v;
                }
        }
        __syncthreads();
    }
    // D layout: col = lane & 15 (co), row = 4 * (lane >> 4) + r (rho within the M tile)
    for (int t = threadIdx.x; t < 217 * UBD_C; t += blockDim.x) {
        const int row = t / UBD_C, col = t - row * UBD_C;
        const int mt = row >> 4, rr = row & 15, nt = col >> 4;
        const int ln = 16 * (rr >> 2) + (col & 15);
        prow[t] = red[(((mt * 2 + nt) * 64 + ln) << 2) + (rr & 3)];
    }
}

// ------------------------------------------------------------------------------------ head
template <typename TX>
__global__ __launch_bounds__(256) void head_dx_kernel(const float *__restrict__ dlogits, const void *__restrict__ a9,
                                                      const float *__restrict__ hk, float *__restrict__ g, long npix, int k_out)
{
    __shared__ __attribute__((aligned(16))) float s_kT[(UBD_MAX_CLASSES + 1) * UBD_C];     // head kernel transposed: [k][c]
    for (int t = threadIdx.x; t < UBD_C * k_out; t += blockDim.x) { const int c = t / k_out, k = t - c * k_out; s_kT[k * UBD_C + c] = hk[t]; }
    __syncthreads();
    // one 4-channel chunk per thread: the 16-byte stores (and the activation loads) of a wave are contiguous
    f32x4 *pg = (f32x4 *)g;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < npix * 6; t += (long)gridDim.x * blockDim.x) {
        const long p = t / 6;
        const int c4 = (int)(t - p * 6);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < k_out; ++k) {
            const float dl = dlogits[p * k_out + k];
            const f32x4 wv = *(const f32x4 *)&s_kT[k * UBD_C + c4 * 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(dl, wv[e], acc[e]);
        }
        float av[4];
        if constexpr (sizeof(TX) == 4) {
            const f32x4 a = ((const f32x4 *)a9)[t];
            av[0] = a[0]; av[1] = a[1]; av[2] = a[2]; av[3] = a[3];
        } else {
            const TX *a = (const TX *)a9 + t * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = (float)a[e];
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = av[e] > 0.f ? acc[e] : 0.f;
        pg[t] = o;
    }
}

// dhk[c][k] = sum_p a9[p][c] dl[p][k]; dhb[k] = sum_p dl[p][k]  (row 24 of the A operand is all ones); k_out > 1.
// Tiles of 64 pixels are staged through LDS with contiguous 16-byte loads (activations widened to fp32, a ones column
// appended) and the MFMA operands are read back in their layout: lane (m, kq) takes A[c = m (+16)][px = 4 s + kq] and
// B[px][k = m (+16)].  Row stride 48 floats: the four pixel rows of a k-step start 16 banks apart (conflict-free).
// Wave w multiplies pixels 16 w .. 16 w + 15 of every tile.
#define HW_TILE 64
#define HW_STRIDE 48
// DXOUT (bf16 train step with classes, round 4): the head's DATA gradient G9 = (dlogits . hk^T) * (A9 > 0), rounded to TX, leaves from the same staged
// tile -- head_dx16_kernel read the 48 bytes per pixel of A9 a second time (22.5 + 23.7 us at 64 images and 8 classes).  Same expression in the
// same order as head_dx16_kernel (fmaf chain over the output channels k = 0, 1, ..): bit-identical G9.
template <typename TX, bool DXOUT = false>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const void *__restrict__ a9, const float *__restrict__ dlogits,
                                                         float *__restrict__ partials, long npix, int k_out,
                                                         const float *__restrict__ hk = nullptr, unsigned short *__restrict__ gout = nullptr)
{
    static_assert(!DXOUT || sizeof(TX) == 2, "the 16-bit gradient tensor");
    __shared__ __attribute__((aligned(16))) float s_kT[DXOUT ? (UBD_MAX_CLASSES + 1) * UBD_C : 4];   // head kernel transposed: [k][c]
    if constexpr (DXOUT)
        for (int t = threadIdx.x; t < UBD_C * k_out; t += 256) { const int c = t / k_out, k = t - c * k_out; s_kT[k * UBD_C + c] = hk[t]; }   // visible after the first barrier of the tile loop
    __shared__ __attribute__((aligned(16))) float sA[HW_TILE * HW_STRIDE];      // [px][c (24) | 1 | zeros]
    __shared__ __attribute__((aligned(16))) float sB[HW_TILE * (UBD_MAX_CLASSES + 1) + 32];   // flat copy of the tile's dlogits: [px][k_out]; columns k >= k_out
                                                                                            // of the B operand read the next pixel's values and are dropped at the end
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
    for (int t = threadIdx.x; t < HW_TILE * HW_STRIDE; t += 256) {                // ones column, zero padding (written once)
        const int c = t % HW_STRIDE;
        sA[t] = c == UBD_C ? 1.f : 0.f;
    }
    for (int t = threadIdx.x; t < HW_TILE * (UBD_MAX_CLASSES + 1) + 32; t += 256) sB[t] = 0.f;
    f32x4 acc[2][2] = {};
    const long ntiles = (npix + HW_TILE - 1) / HW_TILE;
    const bool wide = k_out > 16;
    // register staging: the next tile's global loads are in flight while the current tile is multiplied
    constexpr int CPP = sizeof(TX) == 4 ? 6 : 3, EPC = 16 / sizeof(TX);          // 16-byte chunks per pixel, elements per chunk
    constexpr int AREGS = (HW_TILE * CPP + 255) / 256;                            // 2 (fp32) / 1 (16-bit)
    constexpr int BREGS = (HW_TILE * (UBD_MAX_CLASSES + 1) + 255) / 256;          // 8
    u32x4 ra[AREGS];
    float rb[BREGS];
    auto fetch = [&](long tile) {
        const long p0 = tile * HW_TILE;
#pragma unroll
        for (int r = 0; r < AREGS; ++r) {
            const int ch = r * 256 + threadIdx.x;
            const int px = ch / CPP;
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[r] = (ch < HW_TILE * CPP && p0 + px < npix) ? ((const u32x4 *)a9)[p0 * CPP + ch] : z;
        }
#pragma unroll
        for (int r = 0; r < BREGS; ++r) {
            const int e = r * 256 + threadIdx.x;
            rb[r] = (e < HW_TILE * k_out && p0 * k_out + e < npix * k_out) ? dlogits[p0 * k_out + e] : 0.f;
        }
    };
    long tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long p0 = tile * HW_TILE;
        __syncthreads();                                                          // previous tile's operands consumed
#pragma unroll
        for (int r = 0; r < AREGS; ++r) {
            const int ch = r * 256 + threadIdx.x;
            if (ch < HW_TILE * CPP) {
                const int px = ch / CPP, part = ch - px * CPP;
                float *dst = sA + px * HW_STRIDE + part * EPC;
                if constexpr (sizeof(TX) == 4) {
                    *(f32x4 *)dst = __builtin_bit_cast(f32x4, ra[r]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dst[2 * e] = (float)__builtin_bit_cast(TX, (unsigned short)(ra[r][e] & 0xFFFFu));
                        dst[2 * e + 1] = (float)__builtin_bit_cast(TX, (unsigned short)(ra[r][e] >> 16));
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < BREGS; ++r) {
            const int e = r * 256 + threadIdx.x;
            if (e < HW_TILE * k_out) sB[e] = rb[r];
        }
        if (p0 + HW_TILE > npix)                                                  // ragged last tile: no ones beyond the data
            for (int px = threadIdx.x; px < HW_TILE; px += 256) sA[px * HW_STRIDE + UBD_C] = (p0 + px < npix) ? 1.f : 0.f;
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        if constexpr (DXOUT) {
            // one 16-byte chunk (8 channels of a pixel) per thread: 192 of the 256 threads; the pixel's d logits come from sB (broadcast reads),
            // the weights of the eight channels as two ds_read_b128 per output channel, the ReLU mask from the staged activation (fp32 copy: > 0)
            if (threadIdx.x < HW_TILE * 3) {
                const int px = (int)threadIdx.x / 3, c8 = (int)threadIdx.x - 3 * px;
                float g8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < k_out; ++k) {
                    const float dl = sB[px * k_out + k];
                    const f32x4 w0 = *(const f32x4 *)&s_kT[k * UBD_C + c8 * 8], w1 = *(const f32x4 *)&s_kT[k * UBD_C + c8 * 8 + 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { g8[e] = fmaf(dl, w0[e], g8[e]); g8[4 + e] = fmaf(dl, w1[e], g8[4 + e]); }
                }
                const f32x4 a0 = *(const f32x4 *)&sA[px * HW_STRIDE + c8 * 8], a1 = *(const f32x4 *)&sA[px * HW_STRIDE + c8 * 8 + 4];
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float alo = e < 2 ? a0[2 * e] : a1[2 * e - 4], ahi = e < 2 ? a0[2 * e + 1] : a1[2 * e - 3];
                    const unsigned lo = alo > 0.f ? (unsigned)__builtin_bit_cast(unsigned short, (TX)g8[2 * e]) : 0u;
                    const unsigned hi = ahi > 0.f ? (unsigned)__builtin_bit_cast(unsigned short, (TX)g8[2 * e + 1]) : 0u;
                    o[e] = lo | (hi << 16);
                }
                if (p0 + px < npix) ((u32x4 *)gout)[(p0 + px) * 3 + c8] = o;
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int px = 16 * wid + 4 * s4 + kq;
            const float a0 = sA[px * HW_STRIDE + m], a1 = sA[px * HW_STRIDE + 16 + m];
            const float b0 = sB[px * k_out + m];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            if (wide) {
                const float b1 = sB[px * k_out + 16 + m];
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    // D: col = lane&15 (k index), row = 4*(lane>>4) + r (channel index / ones row).  Block reduction in LDS first.
    __syncthreads();
    float *red = sA;                                                              // 32 x 32 floats
    for (int t = threadIdx.x; t < 32 * 32; t += blockDim.x) red[t] = 0.f;
    __syncthreads();
    // the four waves add their tiles one after the other (every (row, column) belongs to one lane per wave): a fixed order, so the
    // head gradients of a multi-class model repeat bit for bit like everything else (round 4; LDS float atomics added them in
    // whatever order the waves arrived)
    for (int ph = 0; ph < 4; ++ph) {
        if (wid == ph) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * mt + 4 * kq + r) * 32 + m + 16 * nt] += acc[mt][nt][r];
        }
        __syncthreads();
    }
    // one partial row [c (24) | ones][k_out] per block, summed by reduce_partials_kernel (a thousand blocks adding into the
    // same 25 k_out addresses would serialise for tens of microseconds)
    for (int t = threadIdx.x; t < 25 * 32; t += blockDim.x) {
        const int row = t >> 5, col = t & 31;
        if (col < k_out) partials[(size_t)blockIdx.x * (25 * k_out) + row * k_out + col] = red[t];
    }
}

// k_out == 1 (detection only, the benchmark configuration): dhk[c] = sum_p a9[p][c] dl[p], dhb = sum_p dl[p] is a
// plain streaming reduction: one pixel per lane per step (the 24 channels are 96 / 48 contiguous bytes), 25 fp32
// accumulators per lane, butterfly reduction per wave, one partial row per block (summed by reduce_partials_kernel).
template <typename TX>
__global__ __launch_bounds__(256) void head_wgrad1_kernel(const void *__restrict__ a9, const float *__restrict__ dlogits,
                                                          float *__restrict__ partials, long npix)
{
    float acc[UBD_C + 1];
#pragma unroll
    for (int c = 0; c <= UBD_C; ++c) acc[c] = 0.f;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const float dl = dlogits[p];
        float av[UBD_C];
#pragma unroll
        for (int c6 = 0; c6 < 4; ++c6) {
            float t6[6];
            ld_act6<TX>(a9, (size_t)p * UBD_C + 6 * c6, t6);
#pragma unroll
            for (int e = 0; e < 6; ++e) av[6 * c6 + e] = t6[e];
        }
#pragma unroll
        for (int c = 0; c < UBD_C; ++c) acc[c] = fmaf(av[c], dl, acc[c]);
        acc[UBD_C] += dl;
    }
    __shared__ float s_red[4][UBD_C + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c <= UBD_C; ++c) {
        float v = acc[c];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) s_red[wid][c] = v;
    }
    __syncthreads();
    if (threadIdx.x <= UBD_C)
        partials[(size_t)blockIdx.x * (UBD_C + 1) + threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

#include "bwd16.h"

// ------------------------------------------------------------------------------------ dilated wgrad
// dW[t][ci][co] = sum_p X[p + off_t][ci] G[p][co],  db[co] = sum_p G[p][co].
// GEMM view: M = 216 (+ one all-ones row -> bias gradient, padded to 14 tiles of 16), N = 24 (2 tiles), K = pixels.
// A convolution with dilation d is dense on each of the d x d phase sub-grids, so the work is cut into items
// (image, phase (ry, rx), 8 x 16 tile of the sub-grid): the block copies the X tile with a one-sub-pixel halo
// (10 x 18 pixels spaced d in the image) and the G tile into LDS by LDS-DMA (per-lane gather addresses,
// double-buffered against the MFMA phase), and all 9 taps x 24 channels of the A operand are then read from LDS
// (ds_read_b32) instead of 14 global gathers per k-step.  Each wave owns every 4th k-step (4 consecutive
// sub-pixels), accumulates the whole 224 x 32 result in 112 VGPRs across all its items, and the block reduces
// through LDS to one fp32 atomic per output at the very end.
#define WG_TH 8
#define WG_TW 16
#define WG_XW (WG_TW + 2)
#define WG_XPIX ((WG_TH + 2) * WG_XW)          // 180
#define WG_GPIX (WG_TH * WG_TW)                // 128
#define WG_ROUNDS 8                            // (180 * 6 + 128 * 6 + 255) / 256 with fp32 activations (7 with 16-bit)
#define WG_BUF_FLOATS (WG_ROUNDS * 256 * 4)    // 8192 floats = 32 KiB

// MS = 2 (round 5, fp32 activations): the block has EIGHT waves -- wave (ks = wid & 3, mh = wid >> 2) takes the k-steps ks, ks + 4, ... like
// before but only the seven M tiles 7 mh .. 7 mh + 6 of them (56 accumulator registers instead of 112, < 128 in all): two blocks per CU are
// then four waves per SIMD, where the fp32 MFMA issues every ~20 cycles per SIMD instead of every ~25-32 at two (tools/ubench/mfma_fill.hip).
// The sums are the same numbers added in the same order (k-step classes 0..3 per accumulator tile): bit-identical to MS = 1.
template <typename TX, int MS = 1>
__global__ __launch_bounds__(256 * MS, 2 * MS) void dil_wgrad_kernel(const void *__restrict__ x, const float *__restrict__ gz,
                                                           float *__restrict__ partials, int n, int h,
                                                           int w, int d)
{
    constexpr int NTH = 256 * MS, MT = 14 / MS;                    // threads per block, M tiles per wave
    __shared__ __attribute__((aligned(16))) float smem[2 * WG_BUF_FLOATS];     // 64 KiB: two tile buffers / final reduction
    __shared__ TX s_one[4];                                                     // 1.0: the A value of the ones row (bias gradient)
    if (threadIdx.x < 4) s_one[threadIdx.x] = (TX)1.f;                          // visible after the first item's barrier
    const int lane = threadIdx.x & 63, m = lane & 15, k = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ks = wid & 3, mh = wid >> 2;                         // k-step class, M half (MS = 1: mh = 0)
    constexpr int XCH = (int)sizeof(TX) * UBD_C / 16;              // 16-byte chunks per X pixel: 6 (fp32) or 3 (16-bit)
    constexpr int WG_CHUNKS = WG_XPIX * XCH + WG_GPIX * 6;
    constexpr int XBYTES = WG_XPIX * UBD_C * (int)sizeof(TX);      // G tile starts here (multiple of 16)

    // A-operand rows of the 14 M-tiles: dword offset of (tap, ci) relative to the X-tile pixel of the output position
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rho = 16 * (mt + mh * MT) + m;
        const int t = rho / UBD_C, ci = rho % UBD_C;
        aoff[mt] = ((t / 3) * WG_XW + (t % 3)) * UBD_C + ci;          // rows >= 216 are never read (see below)
    }
    // the last M tile (rows 208..223) holds 8 real rows, the ones row (bias gradient) and zeros: only the wave that owns it treats it specially
    const bool last_tile = mh == MS - 1;                                        // wave-uniform
    const bool row13_real = !last_tile || m < 8, row13_ones = last_tile && m == 8;
    const int aoff13c = row13_real ? aoff[MT - 1] : aoff[MT - 2];               // a valid offset for every lane (rows >= 216 read a neighbour's, unused)

    // work items
    const int sh = (h + d - 1) / d, sw = (w + d - 1) / d;              // largest sub-grid
    const int tiles_y = (sh + WG_TH - 1) / WG_TH, tiles_x = (sw + WG_TW - 1) / WG_TW;
    const int items = n * d * d * tiles_y * tiles_x;

    struct item_t { int img, ry, rx, sy0, sx0; };
    auto decode = [&](int it) {
        item_t r;
        const int tx = (int)((unsigned)it % (unsigned)tiles_x); it = (int)((unsigned)it / (unsigned)tiles_x);
        const int ty = (int)((unsigned)it % (unsigned)tiles_y); it = (int)((unsigned)it / (unsigned)tiles_y);
        r.rx = (int)((unsigned)it % (unsigned)d); it = (int)((unsigned)it / (unsigned)d);
        r.ry = (int)((unsigned)it % (unsigned)d);
        r.img = (int)((unsigned)it / (unsigned)d);
        r.sy0 = ty * WG_TH; r.sx0 = tx * WG_TW;
        return r;
    };
    // chunk c of the combined tile: c < XPIX*6 -> X pixel (with halo), else G pixel; returns image coords
    // chunk c of the combined tile: X pixels (with halo, XCH chunks each) first, then G pixels (6 chunks each)
    auto chunk_src = [&](const item_t &I, int c, bool &is_x, int &gy, int &gx, int &part) {
        c = c < WG_CHUNKS ? c : WG_CHUNKS - 1;
        is_x = c < WG_XPIX * XCH;
        int sy, sx;
        if (is_x) { const int pix = c / XCH; part = c - pix * XCH; sy = pix / WG_XW - 1; sx = pix % WG_XW - 1; }
        else { const int cg = c - WG_XPIX * XCH; const int gp = cg / 6; part = cg - gp * 6; sy = gp / WG_TW; sx = gp % WG_TW; }
        gy = I.ry + (I.sy0 + sy) * d;
        gx = I.rx + (I.sx0 + sx) * d;
    };
    // which pixel / 16-byte part a thread moves in round rd does not depend on the item: decoded ONCE (the divisions by 6, 18, 16 per round
    // and item were ~100 of the ~500 non-MFMA instructions a wave spends per item; round 5)
    constexpr int ROUNDS = (WG_CHUNKS + NTH - 1) / NTH;
    int cpk[ROUNDS];                                                   // (sy + 1) | (sx + 1) << 8 | part << 16 | is_x << 24
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        int c = rd * NTH + (int)threadIdx.x;
        c = c < WG_CHUNKS ? c : WG_CHUNKS - 1;
        const bool is_x = c < WG_XPIX * XCH;
        int sy, sx, part;
        if (is_x) { const int pix = c / XCH; part = c - pix * XCH; sy = pix / WG_XW - 1; sx = pix % WG_XW - 1; }
        else { const int cg = c - WG_XPIX * XCH; const int gp = cg / 6; part = cg - gp * 6; sy = gp / WG_TW; sx = gp % WG_TW; }
        cpk[rd] = (sy + 1) | ((sx + 1) << 8) | (part << 16) | ((is_x ? 1 : 0) << 24);
    }
    const unsigned lds_smem = ubd_lds_addr(smem);
    auto dma_item = [&](int it, int buf_floats) {
        const item_t I = decode(it);
        const char *xim = (const char *)x + (size_t)I.img * h * w * (UBD_C * sizeof(TX));
        const char *gim = (const char *)gz + (size_t)I.img * h * w * (UBD_C * sizeof(float));
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int cbase = rd * NTH + wid * 64;
            const int pk = cpk[rd];
            const int sy = (pk & 255) - 1, sx = ((pk >> 8) & 255) - 1, part = (pk >> 16) & 255;
            const bool is_x = (pk >> 24) != 0;
            int gy = I.ry + (I.sy0 + sy) * d, gx = I.rx + (I.sx0 + sx) * d;
            gy = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);                 // clamped; out-of-image pixels are zeroed later
            gx = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
            const unsigned pixel = (unsigned)(gy * w + gx);
            const char *src = is_x ? xim + (size_t)pixel * (UBD_C * sizeof(TX)) + part * 16
                                   : gim + (size_t)pixel * (UBD_C * sizeof(float)) + part * 16;
            // the asm form (common.h): hipcc orders every LDS read behind a builtin LDS-DMA in flight (s_waitcnt vmcnt(0) in front of the
            // first operand load), which put the next item's whole fetch in front of this item's MFMAs (round 5: 161 -> see DESIGN 5.4)
            ubd_glds16_at(src, lds_smem + (unsigned)(buf_floats + cbase * 4) * 4u);
        }
    };

    // XCD-aware item ranges (see dil_wgrad16_kernel): the phases of one image go through ONE L2
    const int xcd = blockIdx.x & 7;
    const int nblk_x = ((int)gridDim.x + 7 - xcd) >> 3;
    const int chunk = (items + 7) >> 3;
    const int it_begin = xcd * chunk;
    const int it_end = it_begin + chunk < items ? it_begin + chunk : items;
    f32x4 acc[MT][2] = {};
    int it = it_begin + (int)(blockIdx.x >> 3);
    if (it < it_end) dma_item(it, 0);
    for (int iter = 0; it < it_end; ++iter, it += nblk_x) {
        float *buf = smem + (iter & 1) * WG_BUF_FLOATS;
        const item_t I = decode(it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this item's DMA (issued one item ago, as asm) has landed
        __syncthreads();                              // ... for every wave; everyone left the other buffer
        if (it + nblk_x < it_end) dma_item(it + nblk_x, ((iter + 1) & 1) * WG_BUF_FLOATS);
        // zero the pixels that lie outside the image (halo / ragged sub-grid edge)
        const bool ragged = (I.ry + (I.sy0 - 1) * d < 0) || (I.rx + (I.sx0 - 1) * d < 0) ||
                            (I.ry + (I.sy0 + WG_TH) * d >= h) || (I.rx + (I.sx0 + WG_TW) * d >= w);   // block-uniform
        if (ragged) {
            for (int pix = threadIdx.x; pix < WG_XPIX + WG_GPIX; pix += NTH) {
                bool is_x; int gy, gx, part;
                const bool xp_ = pix < WG_XPIX;
                chunk_src(I, xp_ ? pix * XCH : WG_XPIX * XCH + (pix - WG_XPIX) * 6, is_x, gy, gx, part);
                if (gy < 0 || gy >= h || gx < 0 || gx >= w) {
                    f32x4 *z = (f32x4 *)((char *)buf + (xp_ ? pix * UBD_C * (int)sizeof(TX) : XBYTES + (pix - WG_XPIX) * UBD_C * 4));
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    const int nq = xp_ ? XCH : 6;
                    for (int q6 = 0; q6 < nq; ++q6) z[q6] = zero;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();             // raw barrier: the next item's DMA stays in flight
        }
        // k-steps: (row py, group of 4 consecutive sub-pixels); this wave takes every 4th
        const int rows_eff = min(WG_TH, sh - I.sy0), grp_eff = (min(WG_TW, sw - I.sx0) + 3) >> 2;
        const int nsteps = rows_eff * grp_eff;
        const TX *xt = (const TX *)buf;
        const float *gt = (const float *)((const char *)buf + XBYTES);
        // operands of k-step s (14 A values: (tap, ci) rows of the lane's sub-pixel; 2 B values: its gradient channels); the loads of
        // step s + 4 are issued in front of the 28 MFMAs of step s (round 5: the wave used to wait for every step's LDS round trip)
        auto ld_step = [&](int s, float (&a)[MT], float &b0, float &b1) {
            const int py = (int)((unsigned)s / (unsigned)grp_eff), pg = s - py * grp_eff;
            const int px = pg * 4 + k;                                 // this lane's sub-pixel column
            const TX *xp = xt + (py * WG_XW + px) * UBD_C;             // X-tile pixel of tap (0,0)
            const float *gp = gt + (py * WG_TW + px) * UBD_C;
            // every load unconditional and NO select after a load: an exec-masked load or a v_cndmask on a loaded value sits in the block
            // of the loads, and hipcc then waits for all of them (lgkmcnt(0)) before the MFMAs of the step in front -- the prefetch
            // would be gone.  Lanes without a value read a finite neighbour's: columns >= 8 of the second N tile and rows >= 217 are never
            // stored (wgrad_block_reduce), and the ones row (bias gradient) reads a 1.0 kept in LDS (the select is on the ADDRESS).
            b0 = gp[m];
            b1 = gp[16 + (m & 7)];
#pragma unroll
            for (int mt = 0; mt < MT - 1; ++mt) a[mt] = (float)xp[aoff[mt]];
            const TX *p13 = row13_ones ? (const TX *)s_one : xp + aoff13c;
            a[MT - 1] = (float)*p13;
        };
        float a0[MT], a1[MT], b00 = 0.f, b01 = 0.f, b10 = 0.f, b11 = 0.f;
        int s = ks;
        if (s < nsteps) ld_step(s, a0, b00, b01);
        for (; s < nsteps; s += 8) {
            if (s + 4 < nsteps) ld_step(s + 4, a1, b10, b11);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], b00, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], b01, acc[mt][1], 0, 0, 0);
            }
            if (s + 4 >= nsteps) break;
            if (s + 8 < nsteps) ld_step(s + 8, a0, b00, b01);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b10, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b11, acc[mt][1], 0, 0, 0);
            }
        }
    }
    wgrad_block_reduce<MT>(acc, smem, partials + (size_t)blockIdx.x * (217 * UBD_C), lane, ks, mh * MT);
}

// ------------------------------------------------------------------------------------ separable backward
// Persistent blocks over output tiles of 16 columns x TH rows (as the forward sepconv_kernel): the input
// patch and the G tile are staged in LDS (24 channels: LDS-DMA, clamped + zero-fixed at the image border;
// 1/3 channels: converted on the way through registers), every tap and both G layouts are then read from LDS.
template <int CIN, int STRIDE, int XB, int UPS = 0> struct sepb_cfg {          // XB = bytes per element of a 24-channel input
    // 1/3-channel layer with the in-block G tile (fp32 gradient path): 8-row tiles, 40 KB of LDS -- FOUR blocks per CU instead of two (the
    // kernel waited for memory two thirds of its time, PMC round 5)
    static constexpr int TH = ((CIN == UBD_C && STRIDE == 2) || (CIN != UBD_C && UPS > 0)) ? 8 : 16;
    static constexpr int PH = (TH - 1) * STRIDE + 3;
    static constexpr int PW = 15 * STRIDE + 3;
    static constexpr int XPIX = PH * PW;
    static constexpr int GPIX = TH * 16;
    static constexpr int XCH = XB * UBD_C / 16;                                        // DMA chunks per X pixel
    // LDS layout of the fp32 tiles (round 5).  A lane (pixel i, channel group q) reads 8-byte pairs at pixel_position + 6 q: at 24 dwords
    // per pixel, pixels i and i + 8 (stride 1) or i and i + 4, 8, 12 (stride 2) start in the same bank, every read took 2 (4) passes and
    // the LDS pipe was busy 54 % of the 24-channel kernel's time, 59 % of that in bank conflicts (profiles/r05_pmc_sep_bwd32.txt).
    // The LDS-DMA lands 16-byte chunks at consecutive slots, WHICH chunk a slot fetches is free -- so a pad chunk is left out:
    //   G tile, X patch of a stride-1 layer: 7 slots (28 dwords) per pixel; X patch of a stride-2 layer: 13 slots per PAIR of pixels.
    // Then the 32 lanes of a half-wave read 64 different banks (tools/lds_bank_model.py).
    static constexpr bool XSW = (CIN == UBD_C) && XB == 4;
    static constexpr int XROW_CH = !XSW ? PW * XCH : (STRIDE == 1 ? PW * 7 : (PW / 2) * 13 + (PW % 2) * 6);   // slots per X patch row
    static constexpr int XROW_DW = XROW_CH * 4;
    __host__ __device__ static constexpr int xpos_dw(int pc)                           // dword offset of pixel column pc in its row
    {
        return !XSW ? pc * (UBD_C * XB / 4) : (STRIDE == 1 ? pc * 28 : pc * 24 + (pc >> 1) * 4);
    }
    static constexpr int GPIX_DW = 28;                                                 // G tile: dwords per pixel
    // 1/3-channel patch: rows of XROW_E floats = whole 16-byte chunks with room for the skew between a chunk boundary and the patch's first float
    static constexpr int XROWC = (PW * CIN + 3 + 3) / 4, XROW_E = XROWC * 4;
    static constexpr int XFLOATS = (CIN == UBD_C) ? PH * XROW_DW : PH * XROW_E;
    static constexpr int XCHUNKS = (CIN == UBD_C) ? PH * XROW_CH : 0;
    static constexpr int GCHUNKS = GPIX * 7;                                           // a multiple of 64: every wave's 64 slots lie in ONE region
    static constexpr int CHUNKS = GCHUNKS + XCHUNKS;                                   // DMA slots: the G tile, then (24 channels) the X patch
    static constexpr int ROUNDS = (CHUNKS + 255) / 256;
    static constexpr int GOFF = (CIN == UBD_C) ? 0 : XFLOATS;                          // float offset of the DMA region
    static constexpr int LDS_FLOATS = GOFF + CHUNKS * 4;                               // exact: slots past CHUNKS are not fetched
};

// TX: element type of a 24-channel input patch; TR: activation type of the model (16-bit: kernels and the depthwise
// output are used rounded to TR, as in the forward pass)
// UPS > 0 (round 5, fp32 gradient path): the G tile is COMPUTED in the block instead of being read -- `G` then points at this layer's own
// output activation (the ReLU mask source, same shape as G), `up_ddw` at the dDW tensor of the layer above (stride UPS, top/left padding
// up_pad, map up_oh x up_ow) and `up_dw` at that layer's depthwise kernel [9][24]:
//     G[p][c] = (A[p][c] > 0) * sum_t dDW_up[(p + up_pad - t) / UPS][c] * dw_up[t][c]        (taps in sep_dx_kernel's order: the same bits)
// i.e. sep_dx_kernel's arithmetic on the tile, so that kernel's launch -- 403 MB read + 403 MB mask + 403 MB written per separable
// layer at 64 images -- and the G tensor itself disappear.
#ifdef UBD_STAMPS   // diagnostic build: s_memtime of every wave at the phase boundaries of its first 8 tiles (tools/stamps_sepb32.py)
#define SB32_STAMP_PARAM , unsigned long long *stamps = nullptr
#define SB32STAMP(k) do { if (stamps && stamp_it < 8 && (threadIdx.x & 63) == 0) stamps[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + stamp_it) * 12 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SB32_STAMP_PARAM
#define SB32STAMP(k) do {} while (0)
#endif
template <int CIN, int STRIDE, int IN_U8, typename TX, typename TR, int UPS = 0>
__global__ __launch_bounds__(256, (CIN != UBD_C && UPS > 0) ? 4 : 2) void sep_bwd_kernel(const void *__restrict__ xin, const float *__restrict__ G,
                                                      float *__restrict__ dDW, const float *__restrict__ fwdfrag,
                                                      const float *__restrict__ bwdfrag, float *__restrict__ partials, int n, int H, int W,
                                                      int OH, int OW, int pad_lo, float pre_sub, float pre_div,
                                                      const float *__restrict__ up_ddw = nullptr, const float *__restrict__ up_dw = nullptr,
                                                      int up_oh = 0, int up_ow = 0, int up_pad = 0 SB32_STAMP_PARAM)
{
    using C = sepb_cfg<CIN, STRIDE, (int)sizeof(TX), UPS>;
    static_assert(UPS == 0 || sizeof(TX) == 4 || CIN != UBD_C, "the in-block G tile needs an fp32 mask tile of G's size");
    constexpr int UPH = UPS == 1 ? C::TH + 2 : C::TH / 2 + 2, UPW = UPS == 1 ? 18 : 10;      // patch of the upper layer's dDW
    __shared__ __attribute__((aligned(16))) float s_up[UPS > 0 ? UPH * UPW * UBD_C + 9 * UBD_C : 4];
    constexpr int CPL = (CIN == UBD_C) ? 6 : 1;
    constexpr int NT_A = (CIN == UBD_C) ? 2 : 1;           // tiles of the dDW product
    constexpr int MT_PW = (CIN == UBD_C) ? 2 : 1;          // M tiles of the dpw product (CIN rows + ones row)
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ float s_dw[4][16][CIN == UBD_C ? UBD_C : 4];           // depthwise output of a row, transposed for the dpw product (columns = channels)
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, q = lane >> 4;
    const float *dwlane = fwdfrag + UBD_SEP_FRAG_FLOATS;
    float *gtile = lds + C::GOFF;                                       // DMA region: G tile, then the 24-channel X patch
    float *xpatch = (CIN == UBD_C) ? gtile + C::GCHUNKS * 4 : lds;
    const unsigned lds_dma = ubd_lds_addr(lds) + C::GOFF * 4u;             // LDS byte address of DMA slot 0

    float dwk[9][CPL];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CPL; ++s) dwk[t][s] = rnd_act<TR>(dwlane[(t * 6 + s) * 64 + lane]);
    float apw[6][NT_A];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int tl = 0; tl < NT_A; ++tl) apw[s][tl] = rnd_act<TR>(bwdfrag[(s * 2 + tl) * 64 + lane]);
    const bool ch_ok = (CIN == UBD_C) || (q < CIN);
    const int cb = (CIN == UBD_C) ? 6 * q : (q < CIN ? q : 0);

    float ddw[9][CPL];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CPL; ++s) ddw[t][s] = 0.f;
    f32x4 accpw[MT_PW][2] = {};

    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + C::TH - 1) / C::TH;
    const int total = n * tiles_y * tiles_x;
    ubd_tile_decoder tdec;
    tdec.init(tiles_x, tiles_y, total);
    [[maybe_unused]] int stamp_it = -1;
    const unsigned lds_up = ubd_lds_addr(s_up);
    if constexpr (UPS > 0) {
        if (threadIdx.x < 9 * UBD_C) s_up[UPH * UPW * UBD_C + threadIdx.x] = rnd_act<TR>(up_dw[threadIdx.x]);     // the upper layer's depthwise kernel: visible after the first tile's barrier
    }
    // 1/3-channel fp32 input whose rows are whole 16-byte chunks: fetched 16 bytes per lane (see the staging); xsk = floats between a chunk
    // boundary and the patch's first float
    const bool xwide = (CIN != UBD_C) && !IN_U8 && ((W * CIN) & 3) == 0 && ((uintptr_t)xin & 15) == 0 && (unsigned long long)H * W * CIN * 4 < (1ull << 31);   // 16-byte loads: aligned base, rows of whole chunks
    const int xsk = xwide ? ((-(CIN * pad_lo)) & 3) : 0;
    for (int ltile = blockIdx.x; ltile < total; ltile += gridDim.x) {
        int tx, ty, img;
        tdec.decode(ltile, tx, ty, img);                                   // neighbouring tiles on one XCD (shared halo lines)
        const int oy0 = ty * C::TH, ox0 = tx * 16;
        const int ix0 = ox0 * STRIDE - pad_lo, iy0 = oy0 * STRIDE - pad_lo;
        ++stamp_it;
        SB32STAMP(0);
        __syncthreads();                                               // previous tile fully consumed
        SB32STAMP(1);
        // the staging below decodes slot / element numbers that depend on the thread only: opaque per tile, or hipcc computes the decode of every
        // round once, in front of the tile loop, and parks it in scratch (27 spilled registers at the 128 of the four-blocks-per-CU variants)
        int lane_o = lane, tid_o = (int)threadIdx.x;
        asm volatile("" : "+v"(lane_o), "+v"(tid_o));
        // ---- stage X patch (24 ch) and G tile by LDS-DMA; clamped addresses, zero-fix below
        // Every wave-round fetches 64 consecutive slots of ONE region; (row, slot in the row) of its first slot are scalar arithmetic, a lane adds
        // its number and wraps into the next row at most once (rows are >= 64 slots), small divisions are one multiply, the address is a scalar
        // image base + a 32-bit lane offset (the launcher checks that an image fits).  Round 5: with one flat slot number per lane decoded by
        // 32-bit divisions and a 64-bit address per lane a round cost ~500 cycles of (mostly quarter-rate) VALU -- 7500 of the 24-channel
        // tile's 40000 (profiles/r05_stamps_sepb32.txt).
        static_assert(C::GCHUNKS % 64 == 0 && 16 * 7 >= 64 && (CIN != UBD_C || C::XROW_CH >= 64 || !C::XSW), "one region, one wrap per wave-round");
        const char *xim = (const char *)xin + (size_t)img * H * W * (UBD_C * sizeof(TX));     // wave-uniform bases
        const char *gim = (const char *)G + (size_t)img * OH * OW * (UBD_C * sizeof(float));
#pragma unroll 2
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int cbase = rd * 256 + wid * 64;
            if (cbase >= C::CHUNKS) break;                             // wave-uniform
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_dma + (unsigned)cbase * 16u));
            if (cbase < C::GCHUNKS) {
                const int row0 = cbase / 112;                          // scalar
                int rc = cbase - row0 * 112 + lane_o;
                const int wrap = rc >= 112 ? 1 : 0;
                rc -= wrap * 112;
                const int px = (rc * 147) >> 10, part = rc - px * 7;   // rc / 7 for rc < 209
                int gy = oy0 + row0 + wrap, gx = ox0 + px;
                gy = gy >= OH ? OH - 1 : gy;
                gx = gx >= OW ? OW - 1 : gx;
                const unsigned off = (unsigned)(gy * OW + gx) * (unsigned)(UBD_C * 4) + (unsigned)(part * 16);
                if (part < 6) ubd_glds16_sbase(gim, off, dst);
            } else if constexpr (CIN == UBD_C) {
                const int cx = cbase - C::GCHUNKS;
                int pr, pc, part;
                bool fetch;
                if constexpr (C::XSW) {
                    const int row0 = cx / C::XROW_CH;                  // scalar
                    int rc = cx - row0 * C::XROW_CH + lane_o;
                    const int wrap = rc >= C::XROW_CH ? 1 : 0;
                    rc -= wrap * C::XROW_CH;
                    pr = row0 + wrap;
                    if constexpr (STRIDE == 1) { pc = (rc * 147) >> 10; part = rc - pc * 7; fetch = part < 6; }
                    else {
                        const int pp = (rc * 79) >> 10, r13 = rc - pp * 13;      // rc / 13 for rc < 350: pair of pixels, slot inside (12: the pad)
                        const int hi = r13 >= 6 ? 1 : 0;
                        pc = 2 * pp + hi; part = r13 - 6 * hi; fetch = r13 < 12;
                    }
                } else {
                    const int c = cx + lane_o;
                    pr = c / C::XROW_CH;
                    const int rc = c - pr * C::XROW_CH;
                    pc = rc / C::XCH; part = rc - pc * C::XCH; fetch = true;
                }
                fetch = fetch && pr < C::PH;
                int gy = iy0 + pr, gx = ix0 + pc;
                gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
                const unsigned off = (unsigned)(gy * W + gx) * (unsigned)(UBD_C * sizeof(TX)) + (unsigned)(part * 16);
                if (fetch) ubd_glds16_sbase(xim, off, dst);
            }
        }
        if constexpr (UPS > 0) {
            // ---- the dDW patch of the layer above, by LDS-DMA too (clamped addresses: the G tile below reads a patch pixel only where it lies
            //      inside the upper map).  Round 5: first loaded behind the DMA's barrier (a second memory round trip per tile), then through
            //      registers in flight with the DMA -- 20 registers the four-blocks-per-CU variants do not have.
            const int uy0 = UPS == 1 ? oy0 + up_pad - 2 : ((oy0 + up_pad - 2) >> 1), ux0 = UPS == 1 ? ox0 + up_pad - 2 : ((ox0 + up_pad - 2) >> 1);
            const char *uim = (const char *)up_ddw + (size_t)img * up_oh * up_ow * (UBD_C * 4);
            constexpr int UCH = UPH * UPW * 6;
#pragma unroll
            for (int rd = 0; rd < (UCH + 255) / 256; ++rd) {
                const int cbase = rd * 256 + wid * 64;
                if (cbase >= UCH) break;                               // wave-uniform
                const unsigned c = (unsigned)(cbase + lane_o);
                const unsigned cc = c < (unsigned)UCH ? c : (unsigned)(UCH - 1);
                const unsigned pix = cc / 6u, part = cc - pix * 6u;
                const unsigned pr = pix / (unsigned)UPW, pc = pix - pr * UPW;
                int gy = uy0 + (int)pr, gx = ux0 + (int)pc;
                gy = gy < 0 ? 0 : (gy >= up_oh ? up_oh - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= up_ow ? up_ow - 1 : gx);
                const unsigned off = (unsigned)(gy * up_ow + gx) * (unsigned)(UBD_C * 4) + part * 16u;
                if (c < (unsigned)UCH) ubd_glds16_sbase(uim, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_up + (unsigned)cbase * 16u)));
            }
        }
        // small-channel input: through registers (converted on the way).  All of a thread's loads first (unconditional), the LDS stores behind
        // the upper patch's below: as a load -> store loop of seven trips this was 7 memory round trips per tile.  fp32 rows that are a whole
        // number of 16-byte chunks (`xwide`): 16 bytes per lane -- chunks left / right of the row or above / below the image are outside as a
        // whole and become zeros -- 7 wave-loads per tile instead of 27 (the vector memory pipe takes
        // ~170-200 cycles per wave-load whatever its width, tools/ubench/dma_issue.hip); patch rows then start `xsk` floats into their LDS row.
        constexpr int NXR = (CIN != UBD_C) ? (C::XPIX * CIN + 255) / 256 : 1;
        constexpr int NXW = (CIN != UBD_C) ? (C::PH * C::XROWC + 255) / 256 : 1;
        [[maybe_unused]] float xv[IN_U8 ? NXR : 1];
        [[maybe_unused]] f32x4 xw[IN_U8 ? 1 : NXW];
        if constexpr (CIN != UBD_C) {
            if (xwide) {
                // loaded, converted and stored in one piece below
            } else if constexpr (IN_U8) {
                constexpr unsigned ROWE = C::PW * CIN;                 // a patch row is ROWE consecutive elements of the image row
                const size_t ibase = (size_t)img * H * W * CIN;
#pragma unroll
                for (int k = 0; k < NXR; ++k) {
                    const unsigned e = (unsigned)(k * 256 + tid_o);
                    const unsigned ec = e < (unsigned)(C::XPIX * CIN) ? e : (unsigned)(C::XPIX * CIN - 1);
                    const unsigned pr = ec / ROWE, col = ec - pr * ROWE;
                    const unsigned pc = col / (unsigned)CIN, ch = col - pc * CIN;
                    const int gy = iy0 + (int)pr, gx = ix0 + (int)pc;
                    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
                    const float raw = (float)((const unsigned char *)xin)[ibase + ((size_t)gyc * W + gxc) * CIN + ch];
                    xv[k] = in ? (raw - pre_sub) / pre_div : 0.f;
                }
            } else {
                // fp32 rows that are no whole number of chunks (W * CIN % 4 != 0): element by element, stored at once (the slow path)
                for (int e = tid_o; e < C::XPIX * CIN; e += 256) {
                    const int pr = e / (C::PW * CIN), col = e - pr * (C::PW * CIN);
                    const int pc = col / CIN, ch = col - pc * CIN;
                    const int gy = iy0 + pr, gx = ix0 + pc;
                    float v = 0.f;
                    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = (((const float *)xin)[(((size_t)img * H + gy) * W + gx) * CIN + ch] - pre_sub) / pre_div;
                    xpatch[pr * C::XROW_E + col] = v;
                }
            }
        }
        if constexpr (CIN != UBD_C) {
            if (xwide) {
                const int a0f = ix0 * CIN - xsk, WC = W * CIN;                 // a0f: a multiple of 4 (tile origins are multiples of 32 pixels)
                const float *xim = (const float *)xin + (size_t)img * H * WC;
                bool okk[NXW];
#pragma unroll
                for (int k = 0; k < NXW; ++k) {
                    const unsigned c = (unsigned)(k * 256 + tid_o);
                    const unsigned pr = c / (unsigned)C::XROWC, pc = c - pr * C::XROWC;
                    const int gy = iy0 + (int)pr, f0 = a0f + 4 * (int)pc;
                    // a chunk is inside the image row or outside as a whole (rows are whole chunks): outside -> any valid address, zeroed below
                    okk[k] = c < (unsigned)(C::PH * C::XROWC) && (unsigned)gy < (unsigned)H && (unsigned)f0 < (unsigned)WC;
                    xw[k] = *(const f32x4 *)(xim + (okk[k] ? gy * WC + f0 : 0));
                }
#pragma unroll
                for (int k = 0; k < NXW; ++k) {
                    const unsigned c = (unsigned)(k * 256 + tid_o);
                    const bool ok = okk[k];                            // else: padding, exactly 0 after the preprocessing too
                    f32x4 v;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) v[e4] = ok ? (xw[k][e4] - pre_sub) / pre_div : 0.f;
                    if (c < (unsigned)(C::PH * C::XROWC)) *(f32x4 *)(xpatch + c * 4) = v;
                }
            } else if constexpr (IN_U8) {
#pragma unroll
                for (int k = 0; k < NXR; ++k) {
                    const unsigned e = (unsigned)(k * 256 + tid_o);
                    const unsigned pr = e / (unsigned)(C::PW * CIN);
                    if (e < (unsigned)(C::XPIX * CIN)) xpatch[e + pr * (unsigned)(C::XROW_E - C::PW * CIN)] = xv[k];     // rows of XROW_E floats
                }
            }
        }
        SB32STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's LDS-DMA has landed (asm form: hipcc keeps no count of it)
        __syncthreads();                                               // ... every wave's; LDS writes visible
        SB32STAMP(3);
        {
            const bool xborder = (CIN == UBD_C) && ((iy0 < 0) || (ix0 < 0) || (iy0 + C::PH > H) || (ix0 + C::PW > W));
            const bool gborder = (oy0 + C::TH > OH) || (ox0 + 16 > OW);
            if (xborder || gborder) {                                  // block-uniform
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                if (xborder)
                    for (int pix = threadIdx.x; pix < C::XPIX; pix += 256) {
                        const int pr = pix / C::PW, pc = pix - pr * C::PW;
                        const int gy = iy0 + pr, gx = ix0 + pc;
                        if (gy < 0 || gy >= H || gx < 0 || gx >= W) {
                            f32x4 *z = (f32x4 *)(xpatch + pr * C::XROW_DW + C::xpos_dw(pc));
#pragma unroll
                            for (int k6 = 0; k6 < C::XCH; ++k6) z[k6] = zero;
                        }
                    }
                if (gborder)
                    for (int pix = threadIdx.x; pix < C::GPIX; pix += 256)
                        if (oy0 + (pix >> 4) >= OH || ox0 + (pix & 15) >= OW) {
                            f32x4 *z = (f32x4 *)(gtile + pix * C::GPIX_DW);
#pragma unroll
                            for (int k6 = 0; k6 < 6; ++k6) z[k6] = zero;
                        }
                __syncthreads();
            }
        }

        SB32STAMP(4);
        if constexpr (UPS > 0) {
            // ---- the G tile from the layer above: its dDW patch (zeros outside its map) and its depthwise kernel into LDS, then every
            //      thread turns six 4-channel chunks of the mask tile into G in place
            float *upk = s_up + UPH * UPW * UBD_C;
            const int uy0 = UPS == 1 ? oy0 + up_pad - 2 : ((oy0 + up_pad - 2) >> 1), ux0 = UPS == 1 ? ox0 + up_pad - 2 : ((ox0 + up_pad - 2) >> 1);
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            // Tiles all of whose taps lie inside the upper map (all but the map's border tiles; block-uniform): no bounds, no clamps -- a chunk's
            // patch and weight addresses are ONE base each plus compile-time offsets (stride 2: the parity of the pixel is in the bases, a tap with
            // ky or kx = 3 is switched off by a select).  Same taps, same order, same sums as the general loop below.
            const bool interior = oy0 + up_pad - 2 >= 0 && ox0 + up_pad - 2 >= 0 && (oy0 + up_pad + C::TH - 1) / UPS < up_oh && (ox0 + up_pad + 15) / UPS < up_ow;
            if (interior) {
                for (int e = threadIdx.x; e < C::GPIX * 6; e += 256) {
                    const int pix = e / 6, part = e - pix * 6;
                    const int r = pix >> 4, c = pix & 15;
                    f32x4 acc = zero4;
                    if constexpr (UPS == 1) {
                        const float *pb = s_up + ((r + 2) * UPW + (c + 2)) * UBD_C + 4 * part;      // patch pixel of tap (0, 0): row oy0 + up_pad + r - uy0 = r + 2
                        const float *wb = upk + 4 * part;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const f32x4 v = *(const f32x4 *)(pb - (ky * UPW + kx) * UBD_C);
                                const f32x4 wv = *(const f32x4 *)(wb + (ky * 3 + kx) * UBD_C);
                                acc[0] = fmaf(v[0], wv[0], acc[0]); acc[1] = fmaf(v[1], wv[1], acc[1]);
                                acc[2] = fmaf(v[2], wv[2], acc[2]); acc[3] = fmaf(v[3], wv[3], acc[3]);
                            }
                    } else {
                        const int py = oy0 + r + up_pad, px = ox0 + c + up_pad;
                        const int qy = py & 1, qx = px & 1;
                        const float *pb = s_up + (((py >> 1) - uy0) * UPW + ((px >> 1) - ux0)) * UBD_C + 4 * part;   // tap (qy, qx)
                        const int t0 = qy * 3 + qx;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const bool ok = (a == 0 || qy == 0) && (b == 0 || qx == 0);          // ky = qy + 2 a <= 2, kx = qx + 2 b <= 2
                                int t = t0 + 6 * a + 2 * b;
                                t = t > 8 ? 8 : t;
                                const f32x4 vl = *(const f32x4 *)(pb - (a * UPW + b) * UBD_C);
                                const f32x4 wv = *(const f32x4 *)(upk + t * UBD_C + 4 * part);
                                const f32x4 v = ok ? vl : zero4;
                                acc[0] = fmaf(v[0], wv[0], acc[0]); acc[1] = fmaf(v[1], wv[1], acc[1]);
                                acc[2] = fmaf(v[2], wv[2], acc[2]); acc[3] = fmaf(v[3], wv[3], acc[3]);
                            }
                    }
                    f32x4 *pg = (f32x4 *)(gtile + pix * C::GPIX_DW + 4 * part);
                    const f32x4 mk = *pg;
                    *pg = (f32x4){mk[0] > 0.f ? acc[0] : 0.f, mk[1] > 0.f ? acc[1] : 0.f, mk[2] > 0.f ? acc[2] : 0.f, mk[3] > 0.f ? acc[3] : 0.f};
                }
            } else
            for (int e = threadIdx.x; e < C::GPIX * 6; e += 256) {
                const int pix = e / 6, part = e - pix * 6;
                const int py = oy0 + (pix >> 4) + up_pad, px = ox0 + (pix & 15) + up_pad;
                f32x4 acc = zero4;
                // No branch around the two LDS reads of a tap (round 5: as `if (ok) { read; read; fma }` every tap was an LDS round trip of its
                // own): the patch index is clamped into the patch, a tap that does not exist contributes 0 * w (the same sum: x + 0 = x).
                // Stride 2: only taps of the pixel's parity can exist ((p - k) even), 2 x 2 of the 9 -- visited in the same ky, kx order.
                constexpr int NTAP = UPS == 1 ? 3 : 2;
#pragma unroll
                for (int a = 0; a < NTAP; ++a) {
                    const int ky = UPS == 1 ? a : (py & 1) + 2 * a;
                    const int ty = py - ky;
                    const bool yok = ky <= 2 && ty >= 0 && (ty / UPS) < up_oh;
                    int ur = (UPS == 1 ? ty : (ty >> 1)) - uy0;
                    ur = ur < 0 ? 0 : (ur > UPH - 1 ? UPH - 1 : ur);
#pragma unroll
                    for (int b = 0; b < NTAP; ++b) {
                        const int kx = UPS == 1 ? b : (px & 1) + 2 * b;
                        const int tx = px - kx;
                        const bool ok = yok && kx <= 2 && tx >= 0 && (tx / UPS) < up_ow;
                        int uc = (UPS == 1 ? tx : (tx >> 1)) - ux0;
                        uc = uc < 0 ? 0 : (uc > UPW - 1 ? UPW - 1 : uc);
                        int t = ky * 3 + kx;
                        t = t > 8 ? 8 : t;
                        const f32x4 vl = *(const f32x4 *)(s_up + (ur * UPW + uc) * UBD_C + 4 * part);
                        const f32x4 wv = *(const f32x4 *)(upk + t * UBD_C + 4 * part);
                        const f32x4 v = ok ? vl : zero4;
                        acc[0] = fmaf(v[0], wv[0], acc[0]); acc[1] = fmaf(v[1], wv[1], acc[1]);
                        acc[2] = fmaf(v[2], wv[2], acc[2]); acc[3] = fmaf(v[3], wv[3], acc[3]);
                    }
                }
                f32x4 *pg = (f32x4 *)(gtile + pix * C::GPIX_DW + 4 * part);
                const f32x4 mk = *pg;
                *pg = (f32x4){mk[0] > 0.f ? acc[0] : 0.f, mk[1] > 0.f ? acc[1] : 0.f, mk[2] > 0.f ? acc[2] : 0.f, mk[3] > 0.f ? acc[3] : 0.f};
            }
            SB32STAMP(7);
            __syncthreads();
        }
        SB32STAMP(8);
#pragma unroll 1
        for (int r = wid; r < C::TH; r += 4) {
            const int oy = oy0 + r;
            if (oy >= OH) break;
            const int ox = ox0 + i;
            const bool pvalid = ox < OW;
            // ---- 1. G of this pixel, channels 6q..6q+5 (zero outside the map: zero-fixed tile)
            float g6[6];
            {
                const f32x2 *pg = (const f32x2 *)(gtile + (r * 16 + i) * C::GPIX_DW + 6 * q);
                const f32x2 v0 = pg[0], v1 = pg[1], v2 = pg[2];
                g6[0] = v0[0]; g6[1] = v0[1]; g6[2] = v1[0]; g6[3] = v1[1]; g6[4] = v2[0]; g6[5] = v2[1];
            }
            // ---- 2. dDW[i][ch] = sum_co G[i][co] pw[ch][co]  (rows = channels in lane layout, cols = pixels)
            f32x4 dA = {0.f, 0.f, 0.f, 0.f}, dB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                dA = __builtin_amdgcn_mfma_f32_16x16x4f32(apw[s][0], g6[s], dA, 0, 0, 0);
                if constexpr (NT_A == 2) dB = __builtin_amdgcn_mfma_f32_16x16x4f32(apw[s][1], g6[s], dB, 0, 0, 0);
            }
            float ddwv[CPL];
            if constexpr (CIN == UBD_C) {
                ddwv[0] = dA[0]; ddwv[1] = dA[1]; ddwv[2] = dA[2]; ddwv[3] = dA[3]; ddwv[4] = dB[0]; ddwv[5] = dB[1];
            } else {
                ddwv[0] = dA[0];
            }
            // ---- 3. one pass over the taps: depthwise output (for dpw) and depthwise kernel gradient
            float dwv[CPL];
#pragma unroll
            for (int s = 0; s < CPL; ++s) dwv[s] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int t = ky * 3 + kx;
                    const int pe = (CIN == UBD_C) ? (r * STRIDE + ky) * (C::XROW_DW * 4 / (int)sizeof(TX)) + C::xpos_dw(i * STRIDE + kx) * 4 / (int)sizeof(TX) + cb
                                                  : (r * STRIDE + ky) * C::XROW_E + xsk + (i * STRIDE + kx) * CIN + cb;     // element index (TX units)
                    const float *p = xpatch + pe;
                    if constexpr (CIN == UBD_C) {
                        float v[6];
                        ld_act6<TX>(xpatch, pe, v);
#pragma unroll
                        for (int s = 0; s < 6; ++s) { dwv[s] = fmaf(v[s], dwk[t][s], dwv[s]); ddw[t][s] = fmaf(v[s], ddwv[s], ddw[t][s]); }
                    } else {
                        dwv[0] = fmaf(p[0], dwk[t][0], dwv[0]);        // dwk is zero for lanes without a channel
                        ddw[t][0] = fmaf(ch_ok ? p[0] : 0.f, ddwv[0], ddw[t][0]);
                    }
                }
            // ---- 5. pointwise kernel / bias gradient: DW transposed through this wave's LDS tile, G read in place
#pragma unroll
            for (int s = 0; s < CPL; ++s)
                if (ch_ok) s_dw[wid][i][cb + s] = rnd_act<TR>(dwv[s]);
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int pr = 4 * g4 + q;               // lane (m = i, k = q): pixel pr of this row tile
                const float *gp = gtile + (r * 16 + pr) * C::GPIX_DW;
                const float b0 = gp[i];
                const float b1 = i < 8 ? gp[16 + i] : 0.f;
                float a0, a1 = 0.f;
                if constexpr (CIN == UBD_C) {
                    a0 = s_dw[wid][pr][i];
                    a1 = i < 8 ? s_dw[wid][pr][16 + i] : (i == 8 ? 1.f : 0.f);
                } else {
                    a0 = i < CIN ? s_dw[wid][pr][i] : (i == CIN ? 1.f : 0.f);
                }
                accpw[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, accpw[0][0], 0, 0, 0);
                accpw[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, accpw[0][1], 0, 0, 0);
                if constexpr (MT_PW == 2) {
                    accpw[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, accpw[1][0], 0, 0, 0);
                    accpw[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accpw[1][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 6. dDW for the data-gradient kernel
            if (dDW != nullptr && pvalid && ch_ok) {
                float *pd = dDW + (((size_t)img * OH + oy) * OW + ox) * CIN + cb;
                if constexpr (CIN == UBD_C) {
                    f32x2 *p2 = (f32x2 *)pd;
                    p2[0] = (f32x2){ddwv[0], ddwv[1]}; p2[1] = (f32x2){ddwv[2], ddwv[3]}; p2[2] = (f32x2){ddwv[4], ddwv[5]};
                } else {
                    pd[0] = ddwv[0];
                }
            }
        }
        SB32STAMP(9);
    }
    // ---- flush: block-level reduction in LDS, then this block's row of the partial-sum matrix
    //      row layout: [9*CIN depthwise | CIN*24 pointwise | 24 bias]
    constexpr int PART = 9 * CIN + CIN * UBD_C + UBD_C;
    __syncthreads();
    float *red = lds;                                   // tile buffers are free now
    for (int t = threadIdx.x; t < PART; t += blockDim.x) red[t] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            float v = ddw[t][s];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            ddw[t][s] = v;
        }
    // the four waves add their sums one after the other (inside a wave every address has one writer): a fixed order -- the fp32 /
    // fp16 train step repeats bit for bit too since round 4 (LDS float atomics added the waves in whatever order they arrived)
    for (int ph = 0; ph < 4; ++ph) {
        if ((int)(threadIdx.x >> 6) == ph) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < CPL; ++s)
                    if (i == 0 && ch_ok) red[t * CIN + cb + s] += ddw[t][s];
            // pointwise kernel / bias gradient: D col = i (co), row = 4q + r (+16 mt) (ci or the ones row)
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
    float *prow = partials + (size_t)blockIdx.x * PART;
    for (int t = threadIdx.x; t < PART; t += blockDim.x) prow[t] = red[t];
}

// G_below[q][c] = (sum_t dDW[(q + pad - t)/s][c] dw[t][c]) * (X[q][c] > 0)   (24-channel layers only)
template <int STRIDE, typename TX>
__global__ __launch_bounds__(256) void sep_dx_kernel(const float *__restrict__ dDW, const void *__restrict__ xmask,
                                                     float *__restrict__ gout, const float *__restrict__ fwdfrag, int n, int H,
                                                     int W, int OH, int OW, int pad_lo)
{
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const float *dwlane = fwdfrag + UBD_SEP_FRAG_FLOATS;
    float dwk[9][6];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < 6; ++s) dwk[t][s] = rnd_act<TX>(dwlane[(t * 6 + s) * 64 + lane]);
    const int tiles_x = (W + 15) >> 4;
    const int total = n * H * tiles_x;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < total; tile += nwaves) {
        const int xt = (int)((unsigned)tile % (unsigned)tiles_x);
        const int rowid = (int)((unsigned)tile / (unsigned)tiles_x);
        const int iy = (int)((unsigned)rowid % (unsigned)H);
        const int img = (int)((unsigned)rowid / (unsigned)H);
        const int ix = xt * 16 + i;
        if (ix >= W) continue;
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy - ky + pad_lo;
            const bool yok = ty >= 0 && (STRIDE == 1 || (ty & 1) == 0) && (ty / STRIDE) < OH;
            const int oy = ty / STRIDE;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix - kx + pad_lo;
                const bool ok = yok && tx >= 0 && (STRIDE == 1 || (tx & 1) == 0) && (tx / STRIDE) < OW;
                if (ok) {
                    const int ox = tx / STRIDE;
                    const f32x2 *p = (const f32x2 *)(dDW + (((size_t)img * OH + oy) * OW + ox) * UBD_C + 6 * q);
                    const f32x2 v0 = p[0], v1 = p[1], v2 = p[2];
                    const int t = ky * 3 + kx;
                    acc[0] = fmaf(v0[0], dwk[t][0], acc[0]); acc[1] = fmaf(v0[1], dwk[t][1], acc[1]);
                    acc[2] = fmaf(v1[0], dwk[t][2], acc[2]); acc[3] = fmaf(v1[1], dwk[t][3], acc[3]);
                    acc[4] = fmaf(v2[0], dwk[t][4], acc[4]); acc[5] = fmaf(v2[1], dwk[t][5], acc[5]);
                }
            }
        }
        const size_t e = (((size_t)img * H + iy) * W + ix) * UBD_C + 6 * q;
        float mk[6];
        ld_act6<TX>(xmask, e, mk);
        f32x2 *po = (f32x2 *)(gout + e);
        po[0] = (f32x2){mk[0] > 0.f ? acc[0] : 0.f, mk[1] > 0.f ? acc[1] : 0.f};
        po[1] = (f32x2){mk[2] > 0.f ? acc[2] : 0.f, mk[3] > 0.f ? acc[3] : 0.f};
        po[2] = (f32x2){mk[4] > 0.f ? acc[4] : 0.f, mk[5] > 0.f ? acc[5] : 0.f};
    }
}

#ifdef UBD_STAMPS   // diagnostic build only: one stamp buffer per (CIN, STRIDE) variant of sepb16_kernel, selected by the caller
static unsigned long long *g_sepb_stamps = nullptr;
static int g_sepb_stamps_cin = 0, g_sepb_stamps_stride = 0;
extern "C" void ubd_debug_set_stamps_sepb(void *p, int cin, int stride) { g_sepb_stamps = (unsigned long long *)p; g_sepb_stamps_cin = cin; g_sepb_stamps_stride = stride; }
#define SB_STAMP_ARG , ((CIN == g_sepb_stamps_cin && STRIDE == g_sepb_stamps_stride) ? g_sepb_stamps : nullptr)
#define WG_STAMP_ARG , ((g_sepb_stamps_cin == -1 && g_sepb_stamps_stride == dd) ? g_sepb_stamps : nullptr)      // cin = -1: dil_wgrad16 of dilation `stride`
#else
#define SB_STAMP_ARG
#define WG_STAMP_ARG
#endif
#include "sepbwd16.h"

// Host side of the batched reduction: every producer takes its own partial-sum matrix out of the workspace region and
// queues a job; rp_flush launches the jobs queued so far (after the dilated loop: those gradients feed the overlapped
// all-reduce; at the end of the pass).
struct rp_queue {
    rp_batch b;
    int nblocks;
    float *base;
    size_t used, cap;                                          // floats
};
static void rp_init(rp_queue *q, float *base, size_t cap_floats) { q->b.njobs = 0; q->nblocks = 0; q->base = base; q->used = 0; q->cap = cap_floats; }
// bf16 pass: the job queued last is handed to the NEXT producer kernel, which totals it at its end (rp_reduce_tail) -- rp_take_prev
// removes it from the batch (or returns an empty job); whatever is still queued when rp_flush is called goes to the stand-alone kernel
static rp_job rp_take_prev(rp_queue *q)
{
    rp_job j;
    memset(&j, 0, sizeof(j));
    if (q->b.njobs > 0) {
        j = q->b.job[--q->b.njobs];
        q->nblocks = j.block0;
        j.block0 = 0;
    }
    return j;
}
static int rp_flush(rp_queue *q, hipStream_t st)
{
    if (q->b.njobs > 0) hipLaunchKernelGGL(reduce_partials_kernel, dim3(q->nblocks), dim3(256), 0, st, q->b);
    q->b.njobs = 0; q->nblocks = 0;                            // the matrices stay allocated until the pass ends (the kernel is in flight)
    return 0;
}
// a [rows][count] matrix for the next producer, reduced into out0[n0] | out1[n1] | out2[rest]
static float *rp_add(rp_queue *q, int rows, int count, float *out0, int n0, float *out1, int n1, float *out2, hipStream_t st)
{
    if (q->b.njobs == RP_MAX_JOBS) rp_flush(q, st);
    const size_t need = ((size_t)rows * count + 63) & ~(size_t)63;
    if (q->used + need > q->cap) { ubd_set_error("backward: partial-sum region too small (%zu + %zu > %zu floats)", q->used, need, q->cap); return nullptr; }
    float *part = q->base + q->used;
    q->used += need;
    rp_job &j = q->b.job[q->b.njobs++];
    j.part = part; j.out0 = out0; j.out1 = out1; j.out2 = out2; j.nblocks = rows; j.count = count; j.n0 = n0; j.n1 = n1; j.block0 = q->nblocks;
    q->nblocks += (count + RP_COLS - 1) / RP_COLS;
    return part;
}

// ------------------------------------------------------------------------------------ host
// Workspace of a train step: forward layout (all activations kept; fp32 or 16-bit), then backward fragments,
// logits, dlogits, fp32 gradient ping-pong buffers, loss scratch, partial-sum matrix.
struct train_layout {
    ubd_fwd_layout fwd;            // dtype == UBD_F32
    ubd_fwd16_layout fwd16;        // 16-bit activations
    size_t off_bfrag, off_logits, off_dlogits, off_gq[2], off_ddw3, off_gb[2], off_loss, off_partials, partials_floats, total;
};

static void train_layout_compute(const ubd_handle *h, int n, int H, int W, train_layout *T)
{
    size_t off;
    if (h->cfg.dtype == UBD_F32) { ubd_fwd_layout_compute(h, n, H, W, 1, &T->fwd); off = T->fwd.total; }
    else { ubd_fwd16_layout_compute(n, H, W, 1, &T->fwd16); off = T->fwd16.total; }
    const size_t small = ubd_align_up((size_t)n * (H / 4) * (W / 4) * UBD_C * sizeof(float), 256);
    const size_t big = ubd_align_up((size_t)n * (H / 2) * (W / 2) * UBD_C * sizeof(float), 256);
    const size_t lg = ubd_align_up((size_t)n * (H / 4) * (W / 4) * h->k_out * sizeof(float), 256);
    T->off_bfrag = off;   off += ubd_align_up(UBD_BWD_FRAG_FLOATS * sizeof(float), 256);
    T->off_logits = off;  off += lg;
    T->off_dlogits = off; off += lg;
    T->off_gq[0] = off;   off += small;
    T->off_gq[1] = off;   off += small;
    T->off_ddw3 = off;    off += small;
    T->off_gb[0] = off;   off += big;
    T->off_gb[1] = off;   off += big;
    T->off_loss = off;    off += ubd_align_up(ubd_loss_workspace_bytes(h, n, H / 4, W / 4), 256);
    T->partials_floats = (size_t)8 * ((size_t)4 * h->num_cus + 8) * (217 * UBD_C);      // one matrix per producer (ten per pass), freed at the end of the pass
    T->off_partials = off; off += ubd_align_up(T->partials_floats * sizeof(float), 256);
    T->total = off;
}

extern "C" size_t ubd_train_workspace_bytes(const ubd_handle *h, int n, int height, int width)
{
    if (!h) return 0;
    train_layout T;
    train_layout_compute(h, n, height, width, &T);
    return T.total;
}

// UPS > 0: G is computed in the kernel from the layer above (G = this layer's output activation, the mask source; up_*: see the kernel)
template <int CIN, int STRIDE, typename TX, typename TR = TX, int UPS = 0>
static int launch_sep_bwd(const ubd_handle *h, const void *x, int in_u8, const float *G, float *dDW, const float *ffrag,
                           const float *bfrag, float *g_dw, float *g_pw, float *g_b, rp_queue *rq, int n, int H, int W,
                           int OH, int OW, int pad_lo, float sub, float div, hipStream_t st,
                           const float *up_ddw = nullptr, const float *up_dw = nullptr, int up_oh = 0, int up_ow = 0, int up_pad = 0)
{
    using C = sepb_cfg<CIN, STRIDE, (int)sizeof(TX), UPS>;
    // the kernel addresses a tile's pieces as image base + 32-bit byte offset
    if ((CIN == UBD_C && (unsigned long long)H * W * UBD_C * sizeof(TX) > 0xFFFFFFFFull) || (unsigned long long)OH * OW * UBD_C * 4 > 0xFFFFFFFFull) {
        ubd_set_error("separable backward: one image's activation map exceeds 4 GiB (%d x %d)", H, W);
        return -1;
    }
    if (UPS > 0 && (unsigned long long)up_oh * up_ow * UBD_C * 4 > 0xFFFFFFFFull) { ubd_set_error("separable backward: upper map exceeds 4 GiB"); return -1; }
    const int th = C::TH;
    const long tiles = (long)n * ((OH + th - 1) / th) * ((OW + 15) / 16);
    const size_t up_bytes = UPS == 0 ? 16 : ((UPS == 1 ? (size_t)(th + 2) * 18 : (size_t)(th / 2 + 2) * 10) * UBD_C + 9 * UBD_C) * sizeof(float);
    const size_t lds_bytes = C::LDS_FLOATS * sizeof(float) + 4 * 16 * (CIN == UBD_C ? UBD_C : 4) * sizeof(float) + up_bytes;
    int grid = h->num_cus * (lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 53 * 1024 ? 2 : (lds_bytes > 40 * 1024 || !(CIN != UBD_C && UPS > 0) ? 3 : 4)));   // 160 KB of LDS per CU
    if (grid > tiles) grid = (int)tiles;
    const int part = 9 * CIN + CIN * UBD_C + UBD_C;
    float *partials = rp_add(rq, grid, part, g_dw, 9 * CIN, g_pw, CIN * UBD_C, g_b, st);
    if (!partials) return -1;
    if (in_u8)
        hipLaunchKernelGGL((sep_bwd_kernel<CIN, STRIDE, 1, TX, TR, UPS>), dim3(grid), dim3(256), 0, st, x, G, dDW, ffrag, bfrag, partials, n, H, W, OH, OW, pad_lo, sub, div, up_ddw, up_dw, up_oh, up_ow, up_pad SB_STAMP_ARG);
    else
        hipLaunchKernelGGL((sep_bwd_kernel<CIN, STRIDE, 0, TX, TR, UPS>), dim3(grid), dim3(256), 0, st, x, G, dDW, ffrag, bfrag, partials, n, H, W, OH, OW, pad_lo, sub, div, up_ddw, up_dw, up_oh, up_ow, up_pad SB_STAMP_ARG);
    return 0;
}

template <int CIN, int STRIDE, int GSRC, typename T>
static int launch_sepb16(const ubd_handle *h, const void *x, int in_u8, const unsigned short *D, const unsigned *mbits, unsigned *xbits,
                          unsigned short *dDW, const float *dw_own, const float *pw_own, const float *dw_up, float *g_dw, float *g_pw,
                          float *g_b, rp_queue *rq, int n, int H, int W, int OH, int OW, int pad_lo, int DH, int DWd, int pad_up,
                          float sub, float div, hipStream_t st)
{
    using C = sepb16_cfg<CIN, STRIDE, GSRC>;
    // 1/3-channel fp32 input that is already preprocessed, rows a whole number of 16-byte chunks: the patch is fetched by LDS-DMA
    // (UBD_SEPB16_X=regs keeps the register path, which also serves every other input)
    bool xdma = false;
    if constexpr (CIN != UBD_C)
        xdma = !in_u8 && sub == 0.f && div == 1.f && (W * CIN) % 4 == 0 && ((uintptr_t)x & 15) == 0 && (unsigned)pad_lo <= 1u && !h->sepb_x_regs &&
               (size_t)H * W * CIN * 4 < (1ull << 31);      // the kernel's 'outside the image' offset 0x80000000 must lie beyond one image's bytes (the register path serves larger images)
    const long tiles = (long)n * ((OH + C::TH - 1) / C::TH) * ((OW + 15) / 16);
    int grid = h->num_cus * (xdma ? sepb16_cfg<CIN, STRIDE, GSRC, 1>::BLOCKS_PER_CU : C::BLOCKS_PER_CU);
    if (grid > tiles) grid = (int)tiles;
    const rp_job prev = h->chain_reduce ? rp_take_prev(rq) : rp_job{};
    float *partials = rp_add(rq, grid, C::PART, g_dw, 9 * CIN, g_pw, CIN * UBD_C, g_b, st);
    if (!partials) return -1;
    if constexpr (CIN != UBD_C) {
        if (xdma) {
            hipLaunchKernelGGL((sepb16_kernel<CIN, STRIDE, 2, GSRC, T>), dim3(grid), dim3(C::NT), 0, st, x, D, mbits, xbits, dDW, dw_own, pw_own, dw_up, partials, n, H, W, OH, OW, pad_lo, DH, DWd, pad_up, sub, div, prev SB_STAMP_ARG);
            return 0;
        }
    }
    if (in_u8)
        hipLaunchKernelGGL((sepb16_kernel<CIN, STRIDE, 1, GSRC, T>), dim3(grid), dim3(C::NT), 0, st, x, D, mbits, xbits, dDW, dw_own, pw_own, dw_up, partials, n, H, W, OH, OW, pad_lo, DH, DWd, pad_up, sub, div, prev SB_STAMP_ARG);
    else
        hipLaunchKernelGGL((sepb16_kernel<CIN, STRIDE, 0, GSRC, T>), dim3(grid), dim3(C::NT), 0, st, x, D, mbits, xbits, dDW, dw_own, pw_own, dw_up, partials, n, H, W, OH, OW, pad_lo, DH, DWd, pad_up, sub, div, prev SB_STAMP_ARG);
    return 0;
}

template <typename TX>
static int launch_head_wgrad(const ubd_handle *h, const void *a9, const float *dlogits, float *grads, rp_queue *rq, long npix,
                              hipStream_t st)
{
    if (h->k_out == 1) {
        long g1 = (npix + 255) / 256;
        if (g1 > h->num_cus * 4) g1 = h->num_cus * 4;
        float *partials = rp_add(rq, (int)g1, UBD_C + 1, grads + h->off_head_k, UBD_C, grads + h->off_head_b, 1, nullptr, st);
        if (!partials) return -1;
        hipLaunchKernelGGL((head_wgrad1_kernel<TX>), dim3((int)g1), dim3(256), 0, st, a9, dlogits, partials, npix);
        return 0;
    }
    long g2l = (npix + HW_TILE - 1) / HW_TILE;
    if (g2l > h->num_cus * 4) g2l = h->num_cus * 4;
    const int g2 = (int)g2l;
    const int cols = (UBD_C + 1) * h->k_out;
    float *partials = rp_add(rq, g2, cols, grads + h->off_head_k, UBD_C * h->k_out, grads + h->off_head_b, h->k_out, nullptr, st);
    if (!partials) return -1;
    hipLaunchKernelGGL((head_wgrad_kernel<TX>), dim3(g2), dim3(256), 0, st, a9, dlogits, partials, npix, h->k_out, (const float *)nullptr, (unsigned short *)nullptr);
    return 0;
}
// bf16 train step with classes: weight gradient AND data gradient of the head in one pass over A9 (head_wgrad_kernel<TX, true>)
template <typename TX>
static int launch_head_bwd16(const ubd_handle *h, const void *a9, const float *dlogits, const float *hk, unsigned short *g, float *grads, rp_queue *rq,
                             long npix, hipStream_t st)
{
    long g2l = (npix + HW_TILE - 1) / HW_TILE;
    if (g2l > h->num_cus * 4) g2l = h->num_cus * 4;
    const int g2 = (int)g2l;
    const int cols = (UBD_C + 1) * h->k_out;
    float *partials = rp_add(rq, g2, cols, grads + h->off_head_k, UBD_C * h->k_out, grads + h->off_head_b, h->k_out, nullptr, st);
    if (!partials) return -1;
    hipLaunchKernelGGL((head_wgrad_kernel<TX, true>), dim3(g2), dim3(256), 0, st, a9, dlogits, partials, npix, h->k_out, hk, g);
    return 0;
}

// Backward pass given the saved activations (element type TX): a1, a2 at half resolution, acts[0..6] = L3, L4..L9
// outputs at quarter resolution; wfrag = forward fp32 fragments (depthwise / pointwise per-lane weights).
// One launch in front of the bf16 train step: the four weight packers (fp32 stem fragments, 16-bit dilated fragments, backward
// fragments, transposed 16-bit fragments: 48 blocks each) + zero gradient vector + zero loss scratch -- they were four ~5-us
// kernels and two memset kernels scattered over the step, each serialised behind its predecessor (rocprofv3: 29 us of a 1.22-ms step).
struct train_prologue_args {
    pack_args pa;
    pack_bwd_args pb;
    pack_sep16_args ps;
    size_t off0, layer_stride, n_params, loss_zero_words;
    float *wfrag32, *bfrag, *grads;
    unsigned *wfrag16, *frag16t, *loss_zero;
};
template <typename T>
__global__ __launch_bounds__(256) void train_prologue16_kernel(const float *__restrict__ params, train_prologue_args a)
{
    const int part = (int)blockIdx.x / 48, vtid = ((int)blockIdx.x % 48) * 256 + (int)threadIdx.x, vthreads = 48 * 256;
    if (part == 0) pack_weights_body(params, a.wfrag32, a.pa, vtid, vthreads);
    else if (part == 1) {
        pack16_body<T>(params, a.wfrag16, a.off0, a.layer_stride, 0, vtid, vthreads);
        pack_sep16_ready_body<T>(params, a.wfrag16 + UBD_NUM_DIL * UBD_DIL16_FRAG_U32, a.ps, vtid, vthreads);
    }
    else if (part == 2) pack_bwd_body(params, a.bfrag, a.pb, vtid, vthreads);
    else pack16_body<T>(params, a.frag16t, a.off0, a.layer_stride, 1, vtid, vthreads);
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gthreads = (size_t)gridDim.x * 256;
    for (size_t i = gtid; i < a.n_params; i += gthreads) a.grads[i] = 0.f;
    for (size_t i = gtid; i < a.loss_zero_words; i += gthreads) a.loss_zero[i] = 0u;
}

template <typename TX>
static int backward_impl(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing, int n, int H,
                         int W, const void *a1, const void *a2, const void *const *acts, const float *wfrag, float *dlogits,
                         float *grads, char *ws, const train_layout &T, hipStream_t st, bool prepacked = false)
{
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const int act_dtype = h->cfg.dtype;
    float *bfrag = (float *)(ws + T.off_bfrag);
    const int per_sep = UBD_SEP_FRAG_FLOATS + UBD_SEP_DW_FLOATS;
    const long npix = (long)n * H4 * W4;
    if (!prepacked) {                                          // (the bf16 step's prologue kernel has zeroed and packed everything)
        UBD_CHECK_HIP(hipMemsetAsync(grads, 0, h->n_params * sizeof(float), st));
        pack_bwd_args pa;
        for (int s = 0; s < 3; ++s) pa.off_sep_pw[s] = h->off_sep_pw[s];
        for (int k = 0; k < UBD_NUM_DIL; ++k) pa.off_dil_k[k] = h->off_dil_k[k];
        pa.c_in = h->cfg.c_in;
        hipLaunchKernelGGL(pack_bwd_kernel, dim3(64), dim3(256), 0, st, params, bfrag, pa);
        if (h->use_wino && !(sizeof(TX) == 2 && UBD_G16<TX>::value)) ubd_launch_pack_wino(h, params, bfrag + UBD_BWD_DIRECT_FLOATS, 1, st);
    }

    float *gq[2] = {(float *)(ws + T.off_gq[0]), (float *)(ws + T.off_gq[1])};
    float *ddw3 = (float *)(ws + T.off_ddw3);
    float *gb[2] = {(float *)(ws + T.off_gb[0]), (float *)(ws + T.off_gb[1])};
    rp_queue rq;
    rp_init(&rq, (float *)(ws + T.off_partials), T.partials_floats);

    int grid = (int)((npix + 255) / 256);
    if (grid > h->num_cus * 8) grid = h->num_cus * 8;
    int cur = 0;
    if constexpr (sizeof(TX) == 2 && UBD_G16<TX>::value) {
        // ---- bf16 gradient tensors (bwd16.h): G9..G3 live in the two halves of gq[0]; G3 is widened into gq[1] for the
        //      separable backward kernels; the transposed 16-bit fragments reuse the Winograd part of bfrag
        unsigned short *g16[2] = {(unsigned short *)gq[0], (unsigned short *)gq[0] + (size_t)npix * UBD_C};
        unsigned *frag16t = (unsigned *)(bfrag + UBD_BWD_DIRECT_FLOATS);
        if (!prepacked) ubd_launch_pack16(h, params, frag16t, 1, st);
        if (h->k_out == 1) {                                   // one pass over A9 for both head gradients
            int g1 = (int)((npix * 3 + 255) / 256);
            if (g1 > h->num_cus * 6) g1 = h->num_cus * 6;
            g1 = (g1 + 2) / 3 * 3;                             // 256 g1 = 0 (mod 3): a thread keeps its channel group
            float *partials = rp_add(&rq, g1, UBD_C + 1, grads + h->off_head_k, UBD_C, grads + h->off_head_b, 1, nullptr, st);
            if (!partials) return -1;
            hipLaunchKernelGGL((head_bwd1_16_kernel<TX>), dim3(g1), dim3(256), 0, st, dlogits, (const unsigned short *)acts[6], params + h->off_head_k, g16[0], partials, npix);
        } else if (h->split_headbwd) {                        // UBD_HEADBWD=split: the two kernels (diagnostics / tests)
            hipLaunchKernelGGL((head_dx16_kernel<TX>), dim3(grid), dim3(256), 0, st, dlogits, (const unsigned short *)acts[6], params + h->off_head_k, g16[0], npix, h->k_out);
            if (launch_head_wgrad<TX>(h, acts[6], dlogits, grads, &rq, npix, st)) return -1;
        } else {
            if (launch_head_bwd16<TX>(h, acts[6], dlogits, params + h->off_head_k, g16[0], grads, &rq, npix, st)) return -1;
        }
        for (int k = UBD_NUM_DIL - 1; k >= 0; --k) {
            const void *X = acts[k];
            const int dd = UBD_DILATIONS[k];
            const int sw = (W4 + dd - 1) / dd, tw = sw <= 8 ? 8 : 16;              // narrow sub-grids: 8-wide tiles
            // sub-grids exactly 8 columns wide and at most 8 rows high (dilation 16 on 128 x 128 maps): two of them side by side in one 16-wide tile
            // (bwd16.h PAIR; UBD_DILBWD=pair8 keeps the 8-wide form)
            const bool pair = !h->split_dilbwd && !h->no_pair_dilbwd && tw == 8 && sw == 8 && (dd & 1) == 0 && W4 % dd == 0 && (H4 + dd - 1) / dd <= 8;
            const long items = pair ? (long)n * dd * (dd / 2)
                                    : (long)n * dd * dd * (((H4 + dd - 1) / dd + W16_TH(tw) - 1) / W16_TH(tw)) * ((sw + tw - 1) / tw);
            int gw = h->num_cus * ((tw == 8 && !pair) ? 3 : 2);     // 16 x 16 tiles: two blocks per CU (LDS, registers); 8-wide tiles: three; the same grid in the split mode (same order of the partial sums)
            if (gw > items) gw = (int)items;
            gw = (gw + 7) / 8 * 8;                                  // the item ranges are cut per XCD: all eight need a block
            const rp_job prev = h->chain_reduce ? rp_take_prev(&rq) : rp_job{};   // the head's / the layer above's partial rows: totalled at the end of this kernel
            float *partials = rp_add(&rq, gw, 217 * UBD_C, grads + h->off_dil_k[k], 216 * UBD_C, grads + h->off_dil_b[k], UBD_C, nullptr, st);
            if (!partials) return -1;
            // weight gradient AND data gradient of the layer from the same staged tiles (bwd16.h); UBD_DILBWD=split (a diagnostic / test
            // switch) keeps the separate data-gradient kernel
            const unsigned *wt = frag16t + (size_t)k * UBD_DIL16_FRAG_U32;
            const bool fuse_dx = !h->split_dilbwd;                 // 8-wide tiles (dilation 16 on 128-wide maps) too since round 4: the fused form with M-split accumulators     // 8-wide tiles (dilation 16 on 128-wide maps): fused 75 us vs 38 + 33 us apart (two blocks per CU instead of three)
            if (pair)
                hipLaunchKernelGGL((dil_wgrad16_kernel<TX, 16, true, false, true>), dim3(gw), dim3(256), 0, st, (const unsigned short *)X, g16[cur], partials, n, H4, W4, dd, (const u32x4 *)wt, g16[cur ^ 1], prev, w16_geometry<16, true>(n, H4, W4, dd) WG_STAMP_ARG);
            else if (tw == 8 && fuse_dx)
                hipLaunchKernelGGL((dil_wgrad16_kernel<TX, 8, true>), dim3(gw), dim3(256), 0, st, (const unsigned short *)X, g16[cur], partials, n, H4, W4, dd, (const u32x4 *)wt, g16[cur ^ 1], prev, w16_geometry<8, false>(n, H4, W4, dd) WG_STAMP_ARG);
            else if (tw == 8)
                hipLaunchKernelGGL((dil_wgrad16_kernel<TX, 8, false>), dim3(gw), dim3(256), 0, st, (const unsigned short *)X, g16[cur], partials, n, H4, W4, dd, (const u32x4 *)nullptr, (unsigned short *)nullptr, prev, w16_geometry<8, false>(n, H4, W4, dd) WG_STAMP_ARG);
            else if (fuse_dx)
                hipLaunchKernelGGL((dil_wgrad16_kernel<TX, 16, true>), dim3(gw), dim3(256), 0, st, (const unsigned short *)X, g16[cur], partials, n, H4, W4, dd, (const u32x4 *)wt, g16[cur ^ 1], prev, w16_geometry<16, false>(n, H4, W4, dd) WG_STAMP_ARG);
            else
                hipLaunchKernelGGL((dil_wgrad16_kernel<TX, 16, false>), dim3(gw), dim3(256), 0, st, (const unsigned short *)X, g16[cur], partials, n, H4, W4, dd, (const u32x4 *)nullptr, (unsigned short *)nullptr, prev, w16_geometry<16, false>(n, H4, W4, dd) WG_STAMP_ARG);
            if (!fuse_dx) ubd_launch_dilconv16(h, 1, wt, nullptr, X, dd, g16[cur], g16[cur ^ 1], n, H4, W4, st);
            cur ^= 1;
        }
        // chained reduction: the first dilated layer's rows are totalled at the end of L3's kernel below, so the dilated + head segment
        // is final (and its all-reduce may start) one kernel later than with the stand-alone reduction
        if (!h->chain_reduce) {
            rp_flush(&rq, st);
            if (ubd_comm_fused(h)) { const int rc = ubd_comm_begin_tail(h, grads, st); if (rc) return rc; }   // dilated + head gradients are final
        }
        // separable layers: G1 / G2 are built tile-wise in LDS from the bf16 dDW tensor of the layer above (sepbwd16.h)
        const int pad2 = h->cfg.fml_compatible ? 1 : 0;
        const float *dw0 = params + h->off_sep_dw[0], *dw1 = params + h->off_sep_dw[1], *dw2 = params + h->off_sep_dw[2];
        const float *pw0 = params + h->off_sep_pw[0], *pw1 = params + h->off_sep_pw[1], *pw2 = params + h->off_sep_pw[2];
        unsigned short *ddw3 = (unsigned short *)(ws + T.off_ddw3), *ddw2 = (unsigned short *)(ws + T.off_gb[0]);
        // ReLU bits of a2 and a1 (one word per pixel, sepbwd16.h): written by the kernel that reads the activation as its input, read by the next one
        unsigned *bits2 = (unsigned *)(ws + T.off_gb[1]), *bits1 = bits2 + ubd_align_up((size_t)n * H2 * W2, 64);
        if (launch_sepb16<UBD_C, 2, 0, TX>(h, a2, 0, g16[cur], nullptr, bits2, ddw3, dw2, pw2, dw2, grads + h->off_sep_dw[2], grads + h->off_sep_pw[2],
                                       grads + h->off_sep_b[2], &rq, n, H2, W2, H4, W4, pad2, H4, W4, 0, 0.f, 1.f, st)) return -1;
        if (h->chain_reduce && ubd_comm_fused(h)) { const int rc = ubd_comm_begin_tail(h, grads, st); if (rc) return rc; }   // dilated + head gradients are final
        if (launch_sepb16<UBD_C, 1, 2, TX>(h, a1, 0, ddw3, bits2, bits1, ddw2, dw1, pw1, dw2, grads + h->off_sep_dw[1], grads + h->off_sep_pw[1],
                                       grads + h->off_sep_b[1], &rq, n, H2, W2, H2, W2, 1, H4, W4, pad2, 0.f, 1.f, st)) return -1;
        float sub = 0.f, div = 1.f;
        if (preprocessing == UBD_PRE_MOBILENET) { sub = 127.5f; div = 127.5f; }
        const int u8 = in_dtype == UBD_IN_U8;
        int rc1;
        if (h->cfg.c_in == 1)
            rc1 = launch_sepb16<1, 2, 1, TX>(h, images, u8, ddw2, bits1, nullptr, nullptr, dw0, pw0, dw1, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0],
                                             grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad2, H2, W2, 1, sub, div, st);
        else
            rc1 = launch_sepb16<3, 2, 1, TX>(h, images, u8, ddw2, bits1, nullptr, nullptr, dw0, pw0, dw1, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0],
                                             grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad2, H2, W2, 1, sub, div, st);
        if (rc1) return -1;
        rp_flush(&rq, st);
        UBD_CHECK_HIP(hipGetLastError());
        if (ubd_comm_fused(h)) return ubd_comm_finish(h, grads, st);
        return 0;
    } else {
    // head
    hipLaunchKernelGGL((head_dx_kernel<TX>), dim3(grid), dim3(256), 0, st, dlogits, acts[6], params + h->off_head_k, gq[0], npix, h->k_out);
    if (launch_head_wgrad<TX>(h, acts[6], dlogits, grads, &rq, npix, st)) return -1;
    // dilated layers, top to bottom
    for (int k = UBD_NUM_DIL - 1; k >= 0; --k) {
        const void *X = acts[k];                                // input of dilated layer k (= output of the layer below)
        {
            const int dd = UBD_DILATIONS[k];
            const long items = (long)n * dd * dd * (((H4 + dd - 1) / dd + WG_TH - 1) / WG_TH) * (((W4 + dd - 1) / dd + WG_TW - 1) / WG_TW);
            int gw = h->num_cus * 2;
            if (gw > items) gw = (int)items;
            gw = (gw + 7) / 8 * 8;                                  // the item ranges are cut per XCD: all eight need a block
            float *partials = rp_add(&rq, gw, 217 * UBD_C, grads + h->off_dil_k[k], 216 * UBD_C, grads + h->off_dil_b[k], UBD_C, nullptr, st);
            if (!partials) return -1;
            if constexpr (sizeof(TX) == 4) {
                if (h->split_sepbwd32) hipLaunchKernelGGL((dil_wgrad_kernel<TX>), dim3(gw), dim3(256), 0, st, X, gq[cur], partials, n, H4, W4, dd);     // diagnostics: the four-wave form
                else hipLaunchKernelGGL((dil_wgrad_kernel<TX, 2>), dim3(gw), dim3(512), 0, st, X, gq[cur], partials, n, H4, W4, dd);
            } else
                hipLaunchKernelGGL((dil_wgrad_kernel<TX>), dim3(gw), dim3(256), 0, st, X, gq[cur], partials, n, H4, W4, dd);
        }
        if (h->use_wino)
            ubd_launch_dilconv_wino(h, 1, bfrag + UBD_BWD_DIRECT_FLOATS + (size_t)k * UBD_WINO_FRAG_FLOATS, X, act_dtype, UBD_DILATIONS[k], gq[cur], gq[cur ^ 1], n, H4, W4, st);
        else
            ubd_launch_dilconv(h, 1, bfrag + (size_t)k * UBD_DIL_FRAG_FLOATS, (const float *)X, UBD_DILATIONS[k], gq[cur], gq[cur ^ 1], n, H4, W4, st);
        cur ^= 1;
    }
    }
    rp_flush(&rq, st);
    if (ubd_comm_fused(h)) { const int rc = ubd_comm_begin_tail(h, grads, st); if (rc) return rc; }       // dilated + head gradients are final
    // separable layers
    const int pad_s2 = h->cfg.fml_compatible ? 1 : 0;
    const float *sf0 = wfrag, *sf1 = wfrag + per_sep, *sf2 = wfrag + 2 * per_sep;
    const float *bs0 = bfrag + UBD_BWD_DGRAD_FLOATS, *bs1 = bs0 + UBD_BWD_SEP_FLOATS, *bs2 = bs1 + UBD_BWD_SEP_FLOATS;
    // L3: input a2 (H2 x W2), output H4 x W4, G = gq[cur]
    if (launch_sep_bwd<UBD_C, 2, TX>(h, a2, 0, gq[cur], ddw3, sf2, bs2, grads + h->off_sep_dw[2], grads + h->off_sep_pw[2], grads + h->off_sep_b[2], &rq, n, H2, W2, H4, W4, pad_s2, 0.f, 1.f, st)) return -1;
    float sub = 0.f, div = 1.f;
    if (preprocessing == UBD_PRE_MOBILENET) { sub = 127.5f; div = 127.5f; }
    const int u8 = in_dtype == UBD_IN_U8;
    int rc1;
    if constexpr (sizeof(TX) == 4) {
        // fp32 activations (round 5): L2's and L1's kernels build their G tiles themselves from the dDW tensor of the layer above and
        // their own output activation (the ReLU mask) -- no sep_dx launches, no G tensors (UBD_SEPBWD=split keeps the two-kernel form)
        if (!h->split_sepbwd32) {
            const float *dwk2 = params + h->off_sep_dw[2], *dwk1 = params + h->off_sep_dw[1];
            if (launch_sep_bwd<UBD_C, 1, TX, TX, 2>(h, a1, 0, (const float *)a2, gb[1], sf1, bs1, grads + h->off_sep_dw[1], grads + h->off_sep_pw[1], grads + h->off_sep_b[1], &rq, n, H2, W2, H2, W2, 1, 0.f, 1.f, st,
                                                    ddw3, dwk2, H4, W4, pad_s2)) return -1;
            if (h->cfg.c_in == 1)
                rc1 = launch_sep_bwd<1, 2, float, TX, 1>(h, images, u8, (const float *)a1, nullptr, sf0, bs0, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0], grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad_s2, sub, div, st,
                                                         gb[1], dwk1, H2, W2, 1);
            else
                rc1 = launch_sep_bwd<3, 2, float, TX, 1>(h, images, u8, (const float *)a1, nullptr, sf0, bs0, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0], grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad_s2, sub, div, st,
                                                         gb[1], dwk1, H2, W2, 1);
            if (rc1) return -1;
            rp_flush(&rq, st);
            UBD_CHECK_HIP(hipGetLastError());
            if (ubd_comm_fused(h)) return ubd_comm_finish(h, grads, st);
            return 0;
        }
    }
    {
        const long tiles = (long)n * H2 * ((W2 + 15) / 16);
        const int g3 = ubd_grid_for(tiles, h->num_cus, 4, 8);
        hipLaunchKernelGGL((sep_dx_kernel<2, TX>), dim3(g3), dim3(256), 0, st, ddw3, a2, gb[0], sf2, n, H2, W2, H4, W4, pad_s2);
        // L2: input a1, output H2 x W2, G = gb[0]
        if (launch_sep_bwd<UBD_C, 1, TX>(h, a1, 0, gb[0], gb[1], sf1, bs1, grads + h->off_sep_dw[1], grads + h->off_sep_pw[1], grads + h->off_sep_b[1], &rq, n, H2, W2, H2, W2, 1, 0.f, 1.f, st)) return -1;
        hipLaunchKernelGGL((sep_dx_kernel<1, TX>), dim3(g3), dim3(256), 0, st, gb[1], a1, gb[0], sf1, n, H2, W2, H2, W2, 1);
    }
    // L1: input = images (fp32 / uint8), no data gradient
    if (h->cfg.c_in == 1)
        rc1 = launch_sep_bwd<1, 2, float, TX>(h, images, u8, gb[0], nullptr, sf0, bs0, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0], grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad_s2, sub, div, st);
    else
        rc1 = launch_sep_bwd<3, 2, float, TX>(h, images, u8, gb[0], nullptr, sf0, bs0, grads + h->off_sep_dw[0], grads + h->off_sep_pw[0], grads + h->off_sep_b[0], &rq, n, H, W, H2, W2, pad_s2, sub, div, st);
    if (rc1) return -1;
    rp_flush(&rq, st);
    UBD_CHECK_HIP(hipGetLastError());
    if (ubd_comm_fused(h)) return ubd_comm_finish(h, grads, st);
    return 0;
}

extern "C" int ubd_train_step(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                              const int32_t *y_true, int n, int height, int width, float *grads, float *loss,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    UBD_REQUIRE(h && params && images && y_true && grads && loss && workspace, "ubd_train_step: null argument");
    UBD_REQUIRE(n > 0 && height > 0 && width > 0 && (height % 4) == 0 && (width % 4) == 0, "ubd_train_step: height and width must be positive multiples of 4");
    UBD_REQUIRE(h->cfg.dtype == UBD_F32 || h->use_wino, "ubd_train_step: 16-bit activations need the Winograd data-gradient kernel (unset UBD_DILCONV=direct)");
    train_layout T;
    train_layout_compute(h, n, height, width, &T);
    UBD_REQUIRE(workspace_bytes >= T.total, "ubd_train_step: workspace too small (%zu < %zu)", workspace_bytes, T.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *logits = (float *)(ws + T.off_logits), *dlogits = (float *)(ws + T.off_dlogits);
    const long npix = (long)n * (height / 4) * (width / 4);
    const void *acts[7];
    int rc;
    if (h->cfg.dtype == UBD_F32) {
        rc = ubd_forward_impl(h, params, images, in_dtype, preprocessing, n, height, width, logits, ws, T.fwd, st);
        if (rc) return rc;
        rc = ubd_loss_impl(logits, h->k_out, y_true, npix, loss, dlogits, ws + T.off_loss, st, h);
        if (rc) return rc;
        for (int k = 0; k < 7; ++k) acts[k] = ws + T.fwd.off_acts[k];
        return backward_impl<float>(h, params, images, in_dtype, preprocessing, n, height, width, ws + T.fwd.off_a1, ws + T.fwd.off_a2,
                                    acts, (const float *)(ws + T.fwd.off_wfrag), dlogits, grads, ws, T, st);
    }
    UBD_REQUIRE(in_dtype == UBD_IN_F32 || in_dtype == UBD_IN_U8, "ubd_train_step: bad in_dtype %d", in_dtype);
    const bool one_prologue = h->cfg.dtype == UBD_BF16;                          // configs[2] / configs[3]
    if (one_prologue) {
        train_prologue_args a;
        for (int s = 0; s < 3; ++s) { a.pa.off_sep_dw[s] = h->off_sep_dw[s]; a.pa.off_sep_pw[s] = h->off_sep_pw[s]; a.pb.off_sep_pw[s] = h->off_sep_pw[s]; }
        for (int k = 0; k < UBD_NUM_DIL; ++k) { a.pa.off_dil_k[k] = h->off_dil_k[k]; a.pb.off_dil_k[k] = h->off_dil_k[k]; }
        a.pa.c_in = a.pb.c_in = h->cfg.c_in;
        a.ps = ubd_pack_sep16_args(h);
        a.off0 = h->off_dil_k[0]; a.layer_stride = h->off_dil_k[1] - h->off_dil_k[0];
        a.n_params = h->n_params; a.loss_zero_words = ubd_loss_zero_bytes() / 4;
        a.wfrag32 = (float *)(ws + T.fwd16.off_wfrag32); a.wfrag16 = (unsigned *)(ws + T.fwd16.off_wfrag16);
        a.bfrag = (float *)(ws + T.off_bfrag); a.frag16t = (unsigned *)(a.bfrag + UBD_BWD_DIRECT_FLOATS);
        a.grads = grads; a.loss_zero = (unsigned *)(ws + T.off_loss);
        hipLaunchKernelGGL((train_prologue16_kernel<__bf16>), dim3(4 * 48), dim3(256), 0, st, params, a);
    }
    rc = ubd_forward16_layout(h, params, images, one_prologue ? (in_dtype | UBD_IN_PREPACKED) : in_dtype, preprocessing, n, height, width, logits, ws, T.fwd16, st);
    if (rc) return rc;
    rc = ubd_loss_impl(logits, h->k_out, y_true, npix, loss, dlogits, ws + T.off_loss, st, h, one_prologue);
    if (rc) return rc;
    for (int k = 0; k < 7; ++k) acts[k] = ws + T.fwd16.off_acts[k];
    if (h->cfg.dtype == UBD_BF16)
        return backward_impl<__bf16>(h, params, images, in_dtype, preprocessing, n, height, width, ws + T.fwd16.off_a1, ws + T.fwd16.off_a2,
                                     acts, (const float *)(ws + T.fwd16.off_wfrag32), dlogits, grads, ws, T, st, true);
    return backward_impl<_Float16>(h, params, images, in_dtype, preprocessing, n, height, width, ws + T.fwd16.off_a1, ws + T.fwd16.off_a2,
                                   acts, (const float *)(ws + T.fwd16.off_wfrag32), dlogits, grads, ws, T, st);
}
