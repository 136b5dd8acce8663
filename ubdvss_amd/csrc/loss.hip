// Fused training loss (forward value + gradient w.r.t. the logits) on gfx950.
//
// Reference: semantic_segmentation/losses.py
//   :13-17   weights 15 (positive) / 1 (negative) / 5 (hard negative), detection 1, classification 1
//   :27-30   z = y_true > 0, p = sigmoid(y_pred[...,0])
//   :86-126  binary_classification_loss: K.binary_crossentropy (Keras clips p to [1e-7, 1-1e-7] in
//            fp32 and maps back to logits => x' = clamp(x, -16.118095, +15.942385), zero gradient
//            where the clamp is active), mean over positives, mean over negatives, mean of
//            tf.nn.top_k(ce * (1-z), k = min(max(n_pos,1), max(n_neg,1))) over the flattened batch
//   :65-83   classification_loss: masked sparse softmax CE over channels 1.., / max(n_pos,1)
//   :47-62   total = 1*detection + 1*classification
// The batch-global top-k is an exact 3-pass radix select (11/11/10 bits) on the fp32 bit pattern of
// ce*(1-z) (non-negative => monotone), ties at the k-th value resolved toward the lower flat index
// like tf.nn.top_k.
#include "common.h"
#include <mutex>

#define LOSS_BLOCK 256
#define LOSS_MAX_BLOCKS 256      // one block per CU: every block ends with ~50 same-address global atomics (header sums, non-empty bins)

struct loss_hdr {
    double sum_pos, sum_neg, sum_hard, sum_cls;   // [0..3]: (sum_pos, sum_neg) and (sum_hard, sum_cls) are all-reduced pairwise
    // counters, contiguous (all-reduced as 6 ints in the batch-global mode): positives, the detection confusion matrix with
    // pred = logit0 > 0 (keras_metrics.py:110-172; fn = n_pos - tp), positive pixels whose class argmax equals the label
    int n_pos, tp, tn, fp, spare, cls_correct;
    unsigned k;                                   // top-k size
    unsigned prefix_l[2];                         // radix-select prefix after level 0 / level 1
    unsigned k_rem_l[2];                          // rank still to resolve inside that prefix bin
    unsigned T;                                   // final threshold bits (k-th largest value)
    unsigned need_eq;                             // how many elements == T are selected
    unsigned pad[3];
};
#define LOSS_HDR_BYTES 256
#define LOSS1_COPIES 8            // one-launch form: blocks b, b + 8, ... share a copy of the histograms (same-address atomics of 256 blocks are carried out one after the other)

// Per-block partial sums of the statistics kernel.  Every block used to end with six atomic adds on the header: 256 blocks x 6
// read-modify-writes on ONE cache line are carried out one after the other at the memory side (~10 ns each) -- 15 of the statistics
// kernel's 26 us.  Now a block stores its sums in its own record (write-through stores, no contention), draws a ticket, and the last
// block out adds the records up in a fixed order (the loss values are bit-reproducible from run to run as a side effect).
struct loss_part { double a, b; int i0, i1, i2, i3; };          // sum_pos, sum_neg, n_pos, tp, tn, fp

struct loss_layout {
    size_t off_hdr, off_hist, off_histk, off_rec, off_blockties, off_rankties, off_part, off_ce, total;
};

static void loss_layout_compute(long npix, loss_layout *L)
{
    size_t off = 0;
    L->off_hdr = off;       off += LOSS_HDR_BYTES;
    L->off_hist = off;      off += 3 * 2048 * sizeof(unsigned);
    L->off_histk = off;     off += LOSS1_COPIES * 4096 * sizeof(unsigned);  // the one-launch form's histograms: LOSS1_COPIES copies of (2048 + 1024 + 1024) bins, zeroed with the header
    L->off_rec = off;       off += LOSS_MAX_BLOCKS * 64;                        // the one-launch form's per-block records + barrier flags (zeroed with the header)
    L->off_blockties = off; off += ubd_align_up((LOSS_MAX_BLOCKS + 1) * sizeof(unsigned), 256);
    L->off_rankties = off;  off += 1024;                       // batch-global mode: tie counts of every rank (<= 256 ranks)
    L->off_part = off;      off += ubd_align_up(LOSS_MAX_BLOCKS * sizeof(loss_part), 256);
    L->off_ce = off;        off += ubd_align_up((size_t)npix * sizeof(float), 256);
    L->total = off;
}

extern "C" size_t ubd_loss_workspace_bytes(const ubd_handle *, int n, int map_h, int map_w)
{
    loss_layout L;
    loss_layout_compute((long)n * map_h * map_w, &L);
    return L.total;
}

// Keras clip points evaluated in fp32
#define LOGIT_LO (-16.11809539794922f)
#define LOGIT_HI (15.942384719848633f)

__device__ __forceinline__ float bce_from_logit(float x, float z, float &xc)
{
    xc = fminf(fmaxf(x, LOGIT_LO), LOGIT_HI);
    return fmaxf(xc, 0.f) - xc * z + log1pf(expf(-fabsf(xc)));
}

__device__ __forceinline__ double block_reduce_sum(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;       // valid in thread 0
}

// NV sums at once: one pair of barriers instead of NV (results valid in thread 0); sh: (blockDim.x / 64) * NV doubles
template <int NV>
__device__ __forceinline__ void block_reduce_n(double (&v)[NV], double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += __shfl_down(v[k], o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[wid * NV + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] += sh[w * NV + k];
    }
}
__device__ __forceinline__ void block_reduce6(double (&v)[6], double *sh) { block_reduce_n<6>(v, sh); }

// The block's six sums (valid in thread 0) go to its record; returns (block-uniformly) whether this block was the last one out --
// then v holds, in thread 0, the totals over all blocks, added in block order.  Stores and loads of the records go past the caches
// (agent-scope accesses); the ticket is drawn after the stores have been performed.
__device__ __forceinline__ bool loss_publish_and_total(loss_part *__restrict__ part, unsigned *ctr, double (&v)[6], double *sh, int *s_flag)
{
    if (threadIdx.x == 0) {
        loss_part *mine = part + blockIdx.x;
        __hip_atomic_store(&mine->a, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->b, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->i0, (int)v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->i1, (int)v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->i2, (int)v[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->i3, (int)v[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned done = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = done == gridDim.x - 1;
    }
    __syncthreads();
    if (!*s_flag) return false;
    double t[6] = {0, 0, 0, 0, 0, 0};
    if (threadIdx.x < gridDim.x) {                          // grid <= LOSS_MAX_BLOCKS <= block size
        const loss_part *r = part + threadIdx.x;
        t[0] = __hip_atomic_load(&r->a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[1] = __hip_atomic_load(&r->b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[2] = (double)__hip_atomic_load(&r->i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[3] = (double)__hip_atomic_load(&r->i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[4] = (double)__hip_atomic_load(&r->i2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[5] = (double)__hip_atomic_load(&r->i3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_reduce6(t, sh);
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = t[k];
    return true;
}

// ---- pass 1: per-pixel BCE, batch sums, first-level histogram ------------------------------
// 1024 threads per block: with one block per CU (LOSS_MAX_BLOCKS) that is four waves per SIMD to hide the load -> exp/log ->
// LDS-atomic chain of a pixel, at the same number of block-level global atomics
#define LOSS_STATS_BLOCK 1024
__global__ __launch_bounds__(LOSS_STATS_BLOCK) void loss_stats_kernel(const float *__restrict__ logits, int k_out,
                                                                const int *__restrict__ y_true, long npix,
                                                                loss_hdr *hdr, unsigned *__restrict__ hist,
                                                                float *__restrict__ ce_buf, loss_part *__restrict__ part)
{
    __shared__ unsigned s_hist[2048];
    __shared__ double s_red[LOSS_STATS_BLOCK / 64 * 6];
    __shared__ int s_last;
    for (int t = threadIdx.x; t < 2048; t += blockDim.x) s_hist[t] = 0;
    __syncthreads();
    double sp = 0, sn = 0;
    int np = 0, c_tp = 0, c_tn = 0, c_fp = 0;
    // four pixels per trip with all their loads up front: one pixel per trip waited out a full memory round trip each time (a block
    // per CU, four waves per SIMD: nothing else to run meanwhile)
    const long stride = (long)gridDim.x * blockDim.x;
    auto pixel = [&](long p, float x, int yt) {
        const float z = yt > 0 ? 1.f : 0.f;
        const bool pred = x > 0.f;                            // keras_metrics.py:112
        c_tp += (pred && z > 0.f); c_tn += (!pred && z == 0.f); c_fp += (pred && z == 0.f);
        float xc;
        const float ce = bce_from_logit(x, z, xc);
        const float cn = ce * (1.f - z);
        ce_buf[p] = cn;
        sp += (double)(ce * z);
        sn += (double)cn;
        np += (z > 0.f);
        atomicAdd(&s_hist[__float_as_uint(cn) >> 21], 1u);
    };
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + 3 * stride < npix; p += 4 * stride) {
        float x[4];
        int yt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = logits[(p + u * stride) * k_out]; yt[u] = y_true[p + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) pixel(p + u * stride, x[u], yt[u]);
    }
    for (; p < npix; p += stride) pixel(p, logits[p * k_out], y_true[p]);
    __syncthreads();
    for (int t = threadIdx.x; t < 2048; t += blockDim.x)
        if (s_hist[t]) atomicAdd(&hist[t], s_hist[t]);
    double v[6] = {sp, sn, (double)np, (double)c_tp, (double)c_tn, (double)c_fp};
    block_reduce6(v, s_red);
    if (loss_publish_and_total(part, &hdr->pad[1], v, s_red, &s_last) && threadIdx.x == 0) {
        // the header is zero when the kernel starts (and the batch-global mode all-reduces these fields in place afterwards)
        hdr->sum_pos = v[0]; hdr->sum_neg = v[1];
        hdr->n_pos = (int)v[2]; hdr->tp = (int)v[3]; hdr->tn = (int)v[4]; hdr->fp = (int)v[5];
    }
}

// ---- radix-select scan: pick the bin that holds the k_rem-th largest element -----------------
// level 0: bins = bits >> 21 (2048), level 1: (bits >> 10) & 2047, level 2: bits & 1023.
// Runs at the head of the kernel that consumes its result (three single-block launches of ~10 us each saved): every
// block recomputes the same selection from the finished histogram of the previous kernel; block 0 records it in the
// header for the later kernels (each level has its own header fields, so late blocks never read a value block 0 has
// already replaced).
struct loss_sel { unsigned prefix, k_rem, k; };

__device__ __forceinline__ loss_sel loss_select_block(loss_hdr *hdr, const unsigned *__restrict__ hist, int level, long npix,
                                                      unsigned *s_part /* 256 */, loss_sel *s_out)
{
    const int nbins = level == 2 ? 1024 : 2048;
    const int per = nbins / 256;
    // thread t sums bins [nbins - (t+1)*per, nbins - t*per)  (descending order); inclusive prefix over the 256 segment
    // sums; the one thread whose segment holds the k_rem-th largest element finishes the scan
    unsigned mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[nbins - 1 - (threadIdx.x * per + j)];
    // inclusive prefix over the 256 threads: shuffles inside a wave, the four wave totals through LDS -- ONE barrier (the Hillis-Steele form in LDS
    // took sixteen: ~2 us at the head of each of the three kernels that start with this scan)
    unsigned incl_w = mine;
    {
        const int ln = threadIdx.x & 63;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl_w, o, 64);
            if (ln >= o) incl_w += v;
        }
        if (ln == 63) s_part[threadIdx.x >> 6] = incl_w;
        __syncthreads();
        for (int w2 = 0; w2 < (int)(threadIdx.x >> 6); ++w2) incl_w += s_part[w2];
    }
    unsigned k_rem, k, prev_prefix = 0;
    if (level == 0) {
        const long n_pos = hdr->n_pos;
        const long n_neg = npix - n_pos;
        const long a = n_pos > 1 ? n_pos : 1, b = n_neg > 1 ? n_neg : 1;
        k = (unsigned)(a < b ? a : b);       // losses.py:110
        k_rem = k;
    } else {
        k = hdr->k;
        k_rem = hdr->k_rem_l[level - 1];
        prev_prefix = hdr->prefix_l[level - 1];
    }
    const unsigned incl = incl_w, excl = incl - mine;
    // k_rem <= total count holds by construction (k <= number of masked negatives incl. zeros); the last segment takes
    // any remainder like the serial scan did
    const bool winner = (excl < k_rem && k_rem <= incl) || (threadIdx.x == 255 && incl < k_rem);
    if (winner) {
        unsigned acc = excl;
        int bin = nbins - 1 - (int)threadIdx.x * per;
        for (int j = 0; j < per - 1; ++j, --bin) {
            const unsigned c = hist[bin];
            if (acc + c >= k_rem) break;
            acc += c;
        }
        // `bin` holds the k_rem-th largest; `acc` elements are strictly above it
        const unsigned rem = k_rem - acc;
        loss_sel r;
        r.k = k; r.k_rem = rem;
        r.prefix = level == 0 ? (unsigned)bin : (level == 1 ? ((prev_prefix << 11) | (unsigned)bin) : ((prev_prefix << 10) | (unsigned)bin));
        *s_out = r;
        if (blockIdx.x == 0) {
            if (level == 0) hdr->k = k;
            if (level < 2) { hdr->prefix_l[level] = r.prefix; hdr->k_rem_l[level] = rem; }
            else { hdr->T = r.prefix; hdr->need_eq = rem; }
        }
    }
    __syncthreads();
    return *s_out;
}

// histogram of level `level` (1 or 2) over the elements inside the bin selected at level - 1 (prev_hist)
__global__ __launch_bounds__(LOSS_BLOCK) void loss_hist_kernel(const float *__restrict__ ce_buf, long npix, long npix_total,
                                                               loss_hdr *hdr, const unsigned *__restrict__ prev_hist,
                                                               unsigned *__restrict__ hist, int level)
{
    __shared__ unsigned s_hist[2048];
    __shared__ unsigned s_part[256];
    __shared__ loss_sel s_sel;
    for (int t = threadIdx.x; t < 2048; t += blockDim.x) s_hist[t] = 0;
    const unsigned prefix = loss_select_block(hdr, prev_hist, level - 1, npix_total, s_part, &s_sel).prefix;
    const long stride = (long)gridDim.x * blockDim.x;
    auto element = [&](unsigned b) {
        if (level == 1) {
            if ((b >> 21) == prefix) atomicAdd(&s_hist[(b >> 10) & 2047u], 1u);
        } else {
            if ((b >> 10) == prefix) atomicAdd(&s_hist[b & 1023u], 1u);
        }
    };
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + 7 * stride < npix; p += 8 * stride) {           // eight loads in flight per thread (see loss_stats_kernel)
        unsigned b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = __float_as_uint(ce_buf[p + u * stride]);
#pragma unroll
        for (int u = 0; u < 8; ++u) element(b[u]);
    }
    for (; p < npix; p += stride) element(__float_as_uint(ce_buf[p]));
    __syncthreads();
    for (int t = threadIdx.x; t < 2048; t += blockDim.x)
        if (s_hist[t]) atomicAdd(&hist[t], s_hist[t]);
}

// ---- ties: per-chunk count of elements == T (chunks are contiguous index ranges) --------------
__global__ __launch_bounds__(LOSS_BLOCK) void loss_tiecount_kernel(const float *__restrict__ ce_buf, long npix, long npix_total, long chunk,
                                                                   loss_hdr *hdr, const unsigned *__restrict__ hist2,
                                                                   unsigned *__restrict__ blockties)
{
    __shared__ double s_red[LOSS_BLOCK / 64];
    __shared__ unsigned s_part[256];
    __shared__ loss_sel s_sel;
    const loss_sel fin = loss_select_block(hdr, hist2, 2, npix_total, s_part, &s_sel);
    const unsigned T = fin.prefix;                                                              // final threshold bits
    // every element equal to the threshold is selected (the bin's count == the number still needed: always, unless the k-th value
    // repeats): nobody needs tie ranks -- the gradient kernel tests the same condition and does not read the counts
    if (hist2[T & 1023u] == fin.k_rem) return;
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
    int c = 0;
    long p = lo + threadIdx.x;
    for (; p + 7 * (long)blockDim.x < hi; p += 8 * (long)blockDim.x) {
        unsigned b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = __float_as_uint(ce_buf[p + u * (long)blockDim.x]);
#pragma unroll
        for (int u = 0; u < 8; ++u) c += (b[u] == T);
    }
    for (; p < hi; p += blockDim.x) c += (__float_as_uint(ce_buf[p]) == T);
    const double r = block_reduce_sum((double)c, s_red);
    if (threadIdx.x == 0) blockties[blockIdx.x] = (unsigned)r;
}

__global__ __launch_bounds__(1024) void loss_tiescan_kernel(unsigned *blockties, int nblocks)
{
    // exclusive scan of <= LOSS_MAX_BLOCKS (1024) counts by one 1024-thread block
    __shared__ unsigned s[1024];
    const int t = threadIdx.x;
    const unsigned mine = t < nblocks ? blockties[t] : 0u;
    s[t] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = t >= o ? s[t - o] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    if (t < nblocks) blockties[t] = s[t] - mine;
    if (t == nblocks - 1) blockties[nblocks] = s[t];
}

// sum_hard / sum_cls / cls_correct are passed in: inside loss_grad_kernel (last block out) they are read past the vector L1
__device__ __forceinline__ void loss_finalize_body(const loss_hdr *hdr, double sum_hard, double sum_cls, int cls_correct, long npix, int n_cls,
                                                   float *loss4)
{
    const double n_pos = hdr->n_pos > 1 ? (double)hdr->n_pos : 1.0;
    const long n_neg_l = npix - hdr->n_pos;
    const double n_neg = n_neg_l > 1 ? (double)n_neg_l : 1.0;
    double hard = sum_hard / (double)hdr->k;
    if (hard != hard) hard = 0.0;                                  // losses.py:117-121
    const double det = 15.0 * hdr->sum_pos / n_pos + 1.0 * hdr->sum_neg / n_neg + 5.0 * hard;
    const double cls = n_cls > 0 ? sum_cls / n_pos : 0.0;
    loss4[0] = (float)(1.0 * det + (n_cls > 0 ? 1.0 * cls : 0.0));
    loss4[1] = (float)det;
    loss4[2] = (float)cls;
    loss4[3] = (float)hdr->k;
    // monitoring values of the Keras train step (losses.py:138-191, keras_metrics.py:110-172): loss components
    // and the raw counters of the per-batch pixel metrics
    loss4[4] = (float)(hdr->sum_pos / n_pos);
    loss4[5] = (float)(hdr->sum_neg / n_neg);
    loss4[6] = (float)hard;
    loss4[7] = (float)hdr->n_pos;
    loss4[8] = (float)hdr->tp;
    loss4[9] = (float)hdr->tn;
    loss4[10] = (float)hdr->fp;
    loss4[11] = (float)(hdr->n_pos - hdr->tp);       // fn
    loss4[12] = (float)cls_correct;
    loss4[13] = (float)npix;
    loss4[14] = 0.f;
    loss4[15] = 0.f;
}

__global__ void loss_finalize_kernel(const loss_hdr *hdr, long npix, int n_cls, float *loss4)
{
    if (threadIdx.x || blockIdx.x) return;
    loss_finalize_body(hdr, hdr->sum_hard, hdr->sum_cls, hdr->cls_correct, npix, n_cls, loss4);
}

// one pixel of the gradient pass: hard-negative sum, gradient of the detection logit, classification part
struct loss_w { float w_pos, w_neg, w_hard, w_cls; int k_out, n_cls; };
__device__ __forceinline__ void loss_grad_pixel(const loss_w &W, const float *__restrict__ logits, float *__restrict__ dlogits, long p, unsigned bits,
                                                bool sel, float x, int yt, double &s_hard, double &s_cls, int &c_correct)
{
    const int k_out = W.k_out, n_cls = W.n_cls;
    const float z = yt > 0 ? 1.f : 0.f;
    if (sel) s_hard += (double)__uint_as_float(bits);
    const float xc = fminf(fmaxf(x, LOGIT_LO), LOGIT_HI);
    const bool inside = (x >= LOGIT_LO) && (x <= LOGIT_HI);
    if (dlogits) {
        const float sig = 1.f / (1.f + expf(-xc));
        const float coef = z * W.w_pos + (1.f - z) * (W.w_neg + (sel ? W.w_hard : 0.f));
        dlogits[p * k_out] = inside ? (sig - z) * coef : 0.f;
    }
    if (n_cls > 0) {
        const float *lg = logits + p * k_out + 1;
        if (yt > 0) {
            float mx = lg[0];
            int amax = 0;
            for (int c = 1; c < n_cls; ++c)
                if (lg[c] > mx) { mx = lg[c]; amax = c; }         // first maximum, like tf.argmax
            float sum = 0.f;
            for (int c = 0; c < n_cls; ++c) sum += expf(lg[c] - mx);
            const float lse = mx + logf(sum);
            const int lab = yt - 1 < n_cls ? yt - 1 : n_cls - 1;
            c_correct += (amax == lab);
            s_cls += (double)(lse - lg[lab]);
            if (dlogits)
                for (int c = 0; c < n_cls; ++c)
                    dlogits[p * k_out + 1 + c] = (expf(lg[c] - lse) - (c == lab ? 1.f : 0.f)) * W.w_cls;
        } else if (dlogits) {
            for (int c = 0; c < n_cls; ++c) dlogits[p * k_out + 1 + c] = 0.f;
        }
    }
}

// ---- gradient + hard-negative / classification sums --------------------------------------------
#define LOSS_GRAD_BLOCK 1024       // as loss_stats: four waves per SIMD on the one block a CU gets
__global__ __launch_bounds__(LOSS_GRAD_BLOCK) void loss_grad_kernel(const float *__restrict__ logits, int k_out,
                                                               const int *__restrict__ y_true, long npix, long chunk,
                                                               loss_hdr *hdr, const unsigned *__restrict__ blockties,
                                                               const float *__restrict__ ce_buf, float *__restrict__ dlogits,
                                                               long npix_total, const unsigned *__restrict__ rank_ties, int rank,
                                                               int raw_ties, float *__restrict__ loss4, const unsigned *__restrict__ hist2)
{
    // raw_ties: blockties holds the per-block COUNTS (no loss_tiescan launch): the block sums the counts in front of it itself.
    // loss4 != nullptr: the last block out also evaluates the loss values (no loss_finalize launch).  Both are set on one GPU /
    // with per-replica losses; the batch-global mode needs the host-enqueued collectives in between and keeps the two launches.
    __shared__ double s_red[LOSS_GRAD_BLOCK / 64];
    __shared__ unsigned s_wave_ties[LOSS_GRAD_BLOCK / 64];
    const unsigned T = hdr->T, need_eq = hdr->need_eq;
    const double n_pos = hdr->n_pos > 1 ? (double)hdr->n_pos : 1.0;
    const long n_neg_l = npix_total - hdr->n_pos;
    const double n_neg = n_neg_l > 1 ? (double)n_neg_l : 1.0;
    const float w_pos = (float)(15.0 / n_pos), w_neg = (float)(1.0 / n_neg), w_hard = (float)(5.0 / (double)hdr->k);
    const float w_cls = (float)(1.0 / n_pos);
    const int n_cls = k_out - 1;
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double s_hard = 0, s_cls = 0;
    int c_correct = 0;
    const loss_w Wt = {w_pos, w_neg, w_hard, w_cls, k_out, n_cls};
    auto pixel = [&](long p, unsigned bits, bool sel, float x, int yt) { loss_grad_pixel(Wt, logits, dlogits, p, bits, sel, x, yt, s_hard, s_cls, c_correct); };
    // Every element equal to the threshold is selected when the threshold bin of the last histogram holds exactly the number still
    // needed (always, unless the k-th value repeats, e.g. an exact 0): then no tie needs a rank, and the pass is a plain stream --
    // no ballots, no two block barriers per trip, no prefix over the blocks' tie counts (loss_tiecount_kernel has returned early on
    // the same test and left them unwritten).  In the batch-global mode histogram and need_eq are global, so is the test.
    if (hist2[T & 1023u] == need_eq) {
        long p = lo + threadIdx.x;
        for (; p + 3 * (long)blockDim.x < hi; p += 4 * (long)blockDim.x) {        // four pixels' loads in flight
            unsigned b[4];
            float x[4];
            int yt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long pu = p + u * (long)blockDim.x;
                b[u] = __float_as_uint(ce_buf[pu]); x[u] = logits[pu * k_out]; yt[u] = y_true[pu];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) pixel(p + u * (long)blockDim.x, b[u], b[u] >= T, x[u], yt[u]);
        }
        for (; p < hi; p += blockDim.x) {
            const unsigned b = __float_as_uint(ce_buf[p]);
            pixel(p, b, b >= T, logits[p * k_out], y_true[p]);
        }
    } else {
    unsigned tie_base;                                  // ties before this iteration of this block
    if (raw_ties) {
        const double before = block_reduce_sum(threadIdx.x < blockIdx.x ? (double)blockties[threadIdx.x] : 0.0, s_red);   // grid <= LOSS_MAX_BLOCKS <= block size
        if (threadIdx.x == 0) s_wave_ties[0] = (unsigned)before;          // the sum is valid in thread 0: hand it to everybody
        __syncthreads();
        tie_base = s_wave_ties[0];
        __syncthreads();
    } else {
        tie_base = blockties[blockIdx.x];
    }
    for (int j = 0; j < rank; ++j) tie_base += rank_ties[j];    // batch-global mode: the ranks before this one hold the lower flat indices
    for (long base = lo; base < hi; base += blockDim.x) {
        const long p = base + threadIdx.x;
        const bool active = p < hi;
        unsigned bits = 0;
        if (active) bits = __float_as_uint(ce_buf[p]);
        const bool is_tie = active && (bits == T);
        // rank of this tie among all ties in flat-index order
        const unsigned long long bal = __ballot(is_tie);
        const unsigned before_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) s_wave_ties[wid] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before_waves = 0, total_iter = 0;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) {
            const unsigned c = s_wave_ties[w2];
            if (w2 < wid) before_waves += c;
            total_iter += c;
        }
        const unsigned tie_rank = tie_base + before_waves + before_in_wave;
        tie_base += total_iter;
        if (!active) continue;
        pixel(p, bits, (bits > T) || (is_tie && tie_rank < need_eq), logits[p * k_out], y_true[p]);
    }
    }
    double r = block_reduce_sum(s_hard, s_red);
    if (threadIdx.x == 0 && r != 0) atomicAdd(&hdr->sum_hard, r);
    r = block_reduce_sum(s_cls, s_red);
    if (threadIdx.x == 0 && r != 0) atomicAdd(&hdr->sum_cls, r);
    r = block_reduce_sum((double)c_correct, s_red);
    if (threadIdx.x == 0 && r != 0) atomicAdd(&hdr->cls_correct, (int)r);
    if (loss4 && threadIdx.x == 0) {
        // last block out: every block's sums have been added at the L2 by then (the counter add follows them in program order
        // on the same lane; the loads below go past this CU's vector L1, which may still hold the header as the kernel found it).
        // (Per-block records + a total by the last block, as in loss_stats_kernel, were measured here too: 17.7 -> 19.9 us -- three
        // sums are not worth the extra round trip of the last block.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned done = __hip_atomic_fetch_add(&hdr->pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            const double sh = __hip_atomic_load(&hdr->sum_hard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double sc = __hip_atomic_load(&hdr->sum_cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int cc = __hip_atomic_load(&hdr->cls_correct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss_finalize_body(hdr, sh, sc, cc, npix_total, k_out - 1, loss4);
        }
    }
}

// ================================== the whole evaluation in ONE launch ==================================================
// One GPU / per-replica loss (the batch-global mode keeps the chain above: its collectives are enqueued by the host between the
// stages).  A block of 1024 threads owns a contiguous chunk of R * 1024 pixels and keeps them IN REGISTERS (logit, bit pattern of
// ce * (1 - z), label sign) from the first load to the gradient store: nothing is written and read back in between.  The radix
// select keeps the chain's three levels (11 / 10 / 10 bits; a 16 / 15-bit split was built first: two stages, but ~3000 non-empty
// bins per block = 770 k global atomics = 60 us).  The stages are separated by grid-wide barriers (every block is resident: grid <=
// CUs, one block per CU): a block signals a stage by a plain store to its own record, block 0 watches the records, does the serial
// part of the stage (the selection scan; at the end the totals in block order) and releases the others through a word that carries
// what it decided.  A fourth barrier exists only when the k-th value repeats and not all of its copies are selected
// (tie ranks in flat-index order need the counts of the blocks in front).  Everything one block hands to another goes through
// agent-scope atomics / cache-bypassing loads and stores ordered by s_waitcnt -- no release / acquire fences: on this part each
// one is an L2 write-back + invalidate (measured: 19 us per barrier with them).
// losses.py:99-116; same tie rule, same gradient arithmetic as the chain (gradients bit-equal, test_gpu_loss.py).
#define LOSS1_BLOCK 1024
#define LOSS1_MAX_R 4              // 8 pixels per thread spill (49 registers at the 128-register bound of a 1024-thread block)
#define LOSS1_DIRECT_MAX 512       // a block with at most this many elements in the selected bin adds them to the next histogram one global atomic each
#define LOSS1_SPIN_LIMIT (1u << 21)

// lives at header + 128: the arrival counter on its own line; rel[j]: the word that releases barrier j AND carries what the stage decided
// (bit 63: valid) -- a block learns "released" and the selection in ONE memory round trip:
//   rel[0] = prefix0 [10:0] | k_rem0 [31:11] | n_pos [52:32]      rel[1] = prefix1 [20:0] | k_rem1 [41:21]
//   rel[2] = T [30:0] | need_eq [51:31] | ties [52]                rel[3] = the tie barrier          (counts <= 2^20: npix <= 256 * 4096)
struct loss_sync { unsigned arrive, pad0[15]; unsigned long long rel[4]; unsigned timeout, pad1[7]; };
static_assert(sizeof(loss_hdr) <= 128 && sizeof(loss_sync) == 128, "loss header layout");
#define LOSS1_VALID (1ull << 63)

#define LOSS1_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LOSS1_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// A block's record: its sums for the gathering block and its barrier flag, on a line of its own (64 bytes).  Arrival is a plain store of the
// stage number -- no read-modify-write on one counter: 256 same-address atomics are carried out one after the other (~2.5 us per barrier).
// counts: bit 63 = stage 0 finished (the flag of barrier 1: the counters decide k, one round trip); hard: sign bit = the block is done (its value is >= 0)
struct loss1_rec { double a, b; unsigned long long counts; unsigned flag, ties; unsigned long long hard; double cls; unsigned correct, pad[3]; };
static_assert(sizeof(loss1_rec) == 64, "loss1_rec");
// every thread of the block calls it after its last global atomic / store of the stage
__device__ __forceinline__ void loss1_signal(loss1_rec *rec, unsigned stage)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // this wave's atomics and record stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) LOSS1_ST(&rec[blockIdx.x].flag, stage);
}
// block 0: until every block has signalled `stage` (thread t watches block t: one memory round trip per look)
__device__ __forceinline__ void loss1_gather(const loss1_rec *rec, unsigned stage, unsigned grid, loss_sync *sy)
{
    unsigned spins = 0;
    for (;;) {
        const unsigned f = threadIdx.x < grid ? LOSS1_LD(&rec[threadIdx.x].flag) : stage;
        if (__syncthreads_and(f >= stage)) break;
        if (++spins > LOSS1_SPIN_LIMIT) {
            if (threadIdx.x == 0) LOSS1_ST(&sy->timeout, 1u);
            break;
        }
    }
}
// thread 0 of every block but block 0: polls the stage's word; the payload comes back in *s_word (valid after the barrier)
__device__ __forceinline__ void loss1_wait(loss_sync *sy, int j, unsigned long long *s_word)
{
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long w;
        while (!((w = LOSS1_LD(&sy->rel[j])) & LOSS1_VALID)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > LOSS1_SPIN_LIMIT) {                            // a block that never became resident: give up loudly (NaN loss), never hang the queue
                LOSS1_ST(&sy->timeout, 1u);
                break;
            }
        }
        *s_word = w;
    }
    __syncthreads();
}

// the bin that holds the k_rem-th largest element of a finished NB-bin histogram, by the 1024 threads of one block (descending scan);
// s_out[0] = bin, s_out[1] = rank inside the bin, s_out[2] = the bin's count
template <int NB>
__device__ __forceinline__ void loss1_select(const unsigned *hist, unsigned k_rem, unsigned *s_w, unsigned *s_out)
{
    constexpr int per = NB / LOSS1_BLOCK;          // 2 or 1
    const int top = NB - 1 - (int)threadIdx.x * per;                       // this thread's bins: top, top - 1
    unsigned c[per], mine = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) {
        c[j] = 0;
#pragma unroll
        for (int cp = 0; cp < LOSS1_COPIES; ++cp) c[j] += LOSS1_LD(hist + cp * 4096 + top - j);
        mine += c[j];
    }
    unsigned incl = mine;
    const int ln = threadIdx.x & 63, wd = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o, 64);
        if (ln >= o) incl += v;
    }
    __syncthreads();
    if (ln == 63) s_w[wd] = incl;
    __syncthreads();
    for (int w2 = 0; w2 < wd; ++w2) incl += s_w[w2];
    const unsigned excl = incl - mine;
    const bool winner = (excl < k_rem && k_rem <= incl) || (threadIdx.x == LOSS1_BLOCK - 1 && incl < k_rem);
    if (winner) {
        unsigned acc = excl;
        int j = 0;
#pragma unroll
        for (; j < per - 1; ++j) {
            if (acc + c[j] >= k_rem) break;
            acc += c[j];
        }
        s_out[0] = (unsigned)(top - j); s_out[1] = k_rem - acc; s_out[2] = c[j];
    }
    __syncthreads();
}

// histogram of the next 10 bits over this block's elements whose leading bits equal `prefix`: a handful of elements per block as a rule
// (one global atomic each), an LDS histogram when the selected bin is crowded (quantised / saturated logits)
template <int R>
__device__ __forceinline__ void loss1_refine(const unsigned (&bits)[R], int match_shift, unsigned prefix, int bin_shift, unsigned *hist,
                                             unsigned *s_h, unsigned *s_cnt)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned mine = 0;
#pragma unroll
    for (int u = 0; u < R; ++u) mine += (bits[u] != 0xffffffffu && ((bits[u] & 0x7fffffffu) >> match_shift) == prefix);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(s_cnt, mine);
    __syncthreads();
    const unsigned in_block = *s_cnt;
    __syncthreads();
    if (tid == 0) *s_cnt = 0;
    if (in_block && in_block <= LOSS1_DIRECT_MAX) {
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (bits[u] != 0xffffffffu && ((bits[u] & 0x7fffffffu) >> match_shift) == prefix) atomicAdd(&hist[(bits[u] >> bin_shift) & 1023u], 1u);
    } else if (in_block) {
        s_h[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (bits[u] != 0xffffffffu && ((bits[u] & 0x7fffffffu) >> match_shift) == prefix) {
                // one wave's elements of one bin (all of them, when the logits are constant) as ONE LDS atomic
                const unsigned bin = (bits[u] >> bin_shift) & 1023u;
                const unsigned first = __builtin_amdgcn_readfirstlane(bin);
                const unsigned long long same = __ballot(bin == first);
                if (bin != first) atomicAdd(&s_h[bin], 1u);
                else if (lane == __builtin_ctzll(same)) atomicAdd(&s_h[bin], (unsigned)__popcll(same));
            }
        __syncthreads();
        if (s_h[tid]) atomicAdd(&hist[tid], s_h[tid]);
    }
}

template <int R>
__global__ __launch_bounds__(LOSS1_BLOCK) void loss_one_kernel(const float *__restrict__ logits, int k_out, const int *__restrict__ y_true, long npix,
                                                               loss_hdr *hdr, unsigned *hist, unsigned *blockties,
                                                               loss1_rec *rec, float *__restrict__ dlogits, float *__restrict__ loss4, unsigned long long *dbg)
{
#ifdef LOSS1_STAMPS
#define STAMP(i) do { if (threadIdx.x == 0) dbg[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    STAMP(0);
    __shared__ unsigned s_h[2048];
    __shared__ double s_red[LOSS1_BLOCK / 64 * 6];
    __shared__ unsigned s_w[LOSS1_BLOCK / 64];
    __shared__ unsigned s_misc[8];
    __shared__ unsigned long long s_word, s_word_tie;
    __shared__ unsigned s_cnt;
    loss_sync *sy = (loss_sync *)((char *)hdr + 128);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned grid = gridDim.x;
    const long lo = (long)blockIdx.x * (R * LOSS1_BLOCK);
    const int n_cls = k_out - 1;
    unsigned *mycopy = hist + (blockIdx.x & (LOSS1_COPIES - 1)) * 4096;       // [0, 2048): level 0, [2048, 3072): level 1, [3072, 4096): level 2

    // ---- stage 0: per-pixel BCE, batch sums, level-0 histogram
    s_h[tid] = 0; s_h[tid + LOSS1_BLOCK] = 0;
    if (tid == 0) { s_cnt = 0; s_misc[1] = 0; }
    float x[R];
    unsigned bits[R];                 // bit 31: positive label; 0xffffffff: no pixel
    {
        int yt[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const long p = lo + u * LOSS1_BLOCK + tid;
            const bool valid = p < npix;
            x[u] = valid ? logits[p * k_out] : 0.f;
            yt[u] = valid ? y_true[p] : -1;
        }
        __syncthreads();
        STAMP(1);
        double sp = 0, sn = 0;
        unsigned long long cnt = 0;                                      // n_pos | tp << 16 | tn << 32 | fp << 48: a block has <= 4096 pixels
        unsigned zeros = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (lo + u * LOSS1_BLOCK + tid < npix) {
                const float z = yt[u] > 0 ? 1.f : 0.f;
                const bool pred = x[u] > 0.f;                            // keras_metrics.py:112
                cnt += (unsigned long long)(z > 0.f) | ((unsigned long long)(pred && z > 0.f) << 16) |
                       ((unsigned long long)(!pred && z == 0.f) << 32) | ((unsigned long long)(pred && z == 0.f) << 48);
                float xc;
                const float ce = bce_from_logit(x[u], z, xc);
                const float cn = ce * (1.f - z);
                sp += (double)(ce * z);
                sn += (double)cn;
                const unsigned b = __float_as_uint(cn);
                bits[u] = b | (z > 0.f ? 0x80000000u : 0u);
                if (b == 0u) ++zeros;                                    // every positive and every exact zero: counted, not one LDS atomic each on ONE address
                else atomicAdd(&s_h[b >> 20], 1u);
            } else bits[u] = 0xffffffffu;
        }
        // wave totals (fixed shuffle tree: the sums are reproducible), then the 16 wave records through LDS to the first 16 lanes
        for (int o = 32; o > 0; o >>= 1) {
            sp += __shfl_down(sp, o, 64); sn += __shfl_down(sn, o, 64);
            cnt += __shfl_down(cnt, o, 64); zeros += __shfl_down(zeros, o, 64);
        }
        if (lane == 0) { s_red[wid * 4] = sp; s_red[wid * 4 + 1] = sn; ((unsigned long long *)s_red)[wid * 4 + 2] = cnt; ((unsigned long long *)s_red)[wid * 4 + 3] = zeros; }
        __syncthreads();
        STAMP(2);
        for (int t = tid; t < 2048; t += LOSS1_BLOCK)
            if (s_h[t]) atomicAdd(&mycopy[t], s_h[t]);
        if (wid == 0) {
            const int w2 = lane & 15;
            sp = s_red[w2 * 4]; sn = s_red[w2 * 4 + 1]; cnt = ((unsigned long long *)s_red)[w2 * 4 + 2]; unsigned long long zz = ((unsigned long long *)s_red)[w2 * 4 + 3];
            for (int o = 8; o > 0; o >>= 1) {
                sp += __shfl_down(sp, o, 64); sn += __shfl_down(sn, o, 64);
                cnt += __shfl_down(cnt, o, 64); zz += __shfl_down(zz, o, 64);
            }
            if (lane == 0) {
                if (zz) atomicAdd(&mycopy[0], (unsigned)zz);
                loss1_rec *mine = rec + blockIdx.x;
                LOSS1_ST(&mine->a, sp); LOSS1_ST(&mine->b, sn);
                s_word = cnt;
            }
        }
    }
    STAMP(3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // this wave's histogram atomics (and the two sums) have been performed
    __syncthreads();
    if (tid == 0) LOSS1_ST(&rec[blockIdx.x].counts, s_word | LOSS1_VALID);
    if (blockIdx.x == 0) {
        // gather: thread t watches block t's counters; n_pos decides k -- the sums wait until the end of the kernel
        unsigned long long c = LOSS1_VALID;
        for (unsigned spins = 0;; ++spins) {
            if ((unsigned)tid < grid) c = LOSS1_LD(&rec[tid].counts);
            if (__syncthreads_and((c & LOSS1_VALID) != 0)) break;
            if (spins > LOSS1_SPIN_LIMIT) { if (tid == 0) LOSS1_ST(&sy->timeout, 1u); break; }
        }
        STAMP(12);
        unsigned np = (unsigned)tid < grid ? (unsigned)(c & 0xffffu) : 0u;
        for (int o = 32; o > 0; o >>= 1) np += __shfl_down(np, o, 64);
        if (lane == 0 && np) atomicAdd(&s_misc[1], np);
        __syncthreads();
        if (tid == 0) {
            const long n_pos = (long)s_misc[1], n_neg = npix - n_pos;
            const long a = n_pos > 1 ? n_pos : 1, b = n_neg > 1 ? n_neg : 1;
            s_misc[0] = (unsigned)(a < b ? a : b);                       // k, losses.py:110
        }
        __syncthreads();
        loss1_select<2048>(hist, s_misc[0], s_w, s_misc + 4);
        if (tid == 0) {
            s_word = LOSS1_VALID | (unsigned long long)s_misc[4] | ((unsigned long long)s_misc[5] << 11) | ((unsigned long long)s_misc[1] << 32);
            LOSS1_ST(&sy->rel[0], s_word);
            LOSS1_ST(&hdr->k, s_misc[0]); LOSS1_ST(&hdr->prefix_l[0], s_misc[4]); LOSS1_ST(&hdr->k_rem_l[0], s_misc[5]);
        }
        __syncthreads();
    } else {
        loss1_wait(sy, 0, &s_word);
    }
    const unsigned prefix0 = (unsigned)s_word & 2047u, k_rem0 = (unsigned)(s_word >> 11) & 0x1fffffu;
    const long n_pos_l = (long)((s_word >> 32) & 0x1fffffu);
    STAMP(4);

    // ---- stage 1: bits [19:10] of the elements inside the selected level-0 bin
    loss1_refine<R>(bits, 20, prefix0, 10, mycopy + 2048, s_h, &s_cnt);
    STAMP(5);
    loss1_signal(rec, 2u);
    if (blockIdx.x == 0) {
        loss1_gather(rec, 2u, grid, sy);
        STAMP(13);
        loss1_select<1024>(hist + 2048, k_rem0, s_w, s_misc + 4);
        if (tid == 0) {
            s_word = LOSS1_VALID | (unsigned long long)((prefix0 << 10) | s_misc[4]) | ((unsigned long long)s_misc[5] << 21);
            LOSS1_ST(&sy->rel[1], s_word);
            LOSS1_ST(&hdr->prefix_l[1], (prefix0 << 10) | s_misc[4]); LOSS1_ST(&hdr->k_rem_l[1], s_misc[5]);
        }
        __syncthreads();
    } else {
        loss1_wait(sy, 1, &s_word);
    }
    const unsigned prefix1 = (unsigned)s_word & 0x1fffffu, k_rem1 = (unsigned)(s_word >> 21) & 0x1fffffu;
    STAMP(6);

    // ---- stage 2: bits [9:0] inside the selected level-1 bin
    loss1_refine<R>(bits, 10, prefix1, 0, mycopy + 3072, s_h, &s_cnt);
    STAMP(7);
    loss1_signal(rec, 3u);
    if (blockIdx.x == 0) {
        loss1_gather(rec, 3u, grid, sy);
        STAMP(14);
        loss1_select<1024>(hist + 3072, k_rem1, s_w, s_misc + 4);
        if (tid == 0) {
            const unsigned Tb = (prefix1 << 10) | s_misc[4];
            s_word = LOSS1_VALID | (unsigned long long)Tb | ((unsigned long long)s_misc[5] << 31) | ((unsigned long long)(s_misc[6] != s_misc[5]) << 52);
            LOSS1_ST(&sy->rel[2], s_word);
            LOSS1_ST(&hdr->T, Tb); LOSS1_ST(&hdr->need_eq, s_misc[5]);
        }
        __syncthreads();
    } else {
        loss1_wait(sy, 2, &s_word);
    }
    const unsigned T = (unsigned)s_word & 0x7fffffffu, need_eq = (unsigned)(s_word >> 31) & 0x1fffffu;
    const bool ties = ((s_word >> 52) & 1ull) != 0;
    const unsigned k = (unsigned)((n_pos_l > 1 ? n_pos_l : 1) < (npix - n_pos_l > 1 ? npix - n_pos_l : 1) ? (n_pos_l > 1 ? n_pos_l : 1) : (npix - n_pos_l > 1 ? npix - n_pos_l : 1));
    __syncthreads();
    STAMP(8);

    // ---- stage 3 (only when the k-th value repeats and not every copy is selected): ranks of the ties in flat-index order
    unsigned tie_rank[R];
#pragma unroll
    for (int u = 0; u < R; ++u) tie_rank[u] = 0;
    if (ties) {
        unsigned c = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) c += (bits[u] != 0xffffffffu && (bits[u] & 0x7fffffffu) == T);
        const double tot = block_reduce_sum((double)c, s_red);
        if (tid == 0) LOSS1_ST(&blockties[blockIdx.x], (unsigned)tot);
        loss1_signal(rec, 4u);
        if (blockIdx.x == 0) { loss1_gather(rec, 4u, grid, sy); if (tid == 0) LOSS1_ST(&sy->rel[3], LOSS1_VALID); }
        else loss1_wait(sy, 3, &s_word_tie);
        const double before = block_reduce_sum((unsigned)tid < blockIdx.x ? (double)LOSS1_LD(&blockties[tid]) : 0.0, s_red);
        if (tid == 0) s_misc[5] = (unsigned)before;
        __syncthreads();
        unsigned tie_base = s_misc[5];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const bool is_tie = bits[u] != 0xffffffffu && (bits[u] & 0x7fffffffu) == T;
            const unsigned long long bal = __ballot(is_tie);
            __syncthreads();
            if (lane == 0) s_w[wid] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before_waves = 0, total_iter = 0;
            for (int w2 = 0; w2 < LOSS1_BLOCK / 64; ++w2) {
                const unsigned cw = s_w[w2];
                if (w2 < wid) before_waves += cw;
                total_iter += cw;
            }
            tie_rank[u] = tie_base + before_waves + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            tie_base += total_iter;
        }
    }

    // ---- stage 4: gradient, hard-negative / classification sums, loss values by the last block out
    {
        const double n_pos = n_pos_l > 1 ? (double)n_pos_l : 1.0;
        const long n_neg_l = npix - n_pos_l;
        const double n_neg = n_neg_l > 1 ? (double)n_neg_l : 1.0;
        const loss_w Wt = {(float)(15.0 / n_pos), (float)(1.0 / n_neg), (float)(5.0 / (double)k), (float)(1.0 / n_pos), k_out, n_cls};
        double s_hard = 0, s_cls = 0;
        int c_correct = 0;
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (bits[u] != 0xffffffffu) {
                const long p = lo + u * LOSS1_BLOCK + tid;
                const unsigned b = bits[u] & 0x7fffffffu;
                const bool sel = ties ? ((b > T) || (b == T && tie_rank[u] < need_eq)) : (b >= T);
                const int yt = n_cls > 0 ? y_true[p] : (int)(bits[u] >> 31);
                loss_grad_pixel(Wt, logits, dlogits, p, b, sel, x[u], yt, s_hard, s_cls, c_correct);
            }
        STAMP(9);
        double v[3] = {s_hard, s_cls, (double)c_correct};
        block_reduce_n<3>(v, s_red);
        if (tid == 0) {
            loss1_rec *mine = rec + blockIdx.x;
            if (n_cls > 0) {
                LOSS1_ST(&mine->cls, v[1]); LOSS1_ST(&mine->correct, (unsigned)v[2]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            LOSS1_ST(&mine->hard, (unsigned long long)__double_as_longlong(v[0]) | LOSS1_VALID);       // a sum of non-negative values: the sign bit says "done"
        }
        if (blockIdx.x == 0) {
            // block 0 stays for the loss values: every block's sums added in block order (the loss is bit-reproducible from run to run).
            // The first look also fetches the stage-0 sums, final since barrier 1.
            double t[6] = {0, 0, 0, 0, 0, 0};
            if ((unsigned)tid < grid) {
                const loss1_rec *r = rec + tid;
                t[0] = LOSS1_LD(&r->a); t[1] = LOSS1_LD(&r->b);
                const unsigned long long c = LOSS1_LD(&r->counts);
                t[2] = (double)(c & 0xffffu); t[3] = (double)((c >> 16) & 0xffffu); t[4] = (double)((c >> 32) & 0xffffu); t[5] = (double)((c >> 48) & 0x7fffu);
            }
            unsigned long long hw = LOSS1_VALID;
            for (unsigned spins = 0;; ++spins) {
                if ((unsigned)tid < grid) hw = LOSS1_LD(&rec[tid].hard);
                if (__syncthreads_and((hw & LOSS1_VALID) != 0)) break;
                if (spins > LOSS1_SPIN_LIMIT) { if (tid == 0) LOSS1_ST(&sy->timeout, 1u); break; }
            }
            // nine totals in one pass; only the first four waves hold records (grid <= 256)
            double e[9] = {t[0], t[1], t[2], t[3], t[4], t[5], (unsigned)tid < grid ? __longlong_as_double((long long)(hw & ~LOSS1_VALID)) : 0.0, 0, 0};
            if (n_cls > 0 && (unsigned)tid < grid) { e[7] = LOSS1_LD(&rec[tid].cls); e[8] = (double)LOSS1_LD(&rec[tid].correct); }
            if (wid < 4) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int q = 0; q < 9; ++q) e[q] += __shfl_down(e[q], o, 64);
                if (lane == 0)
#pragma unroll
                    for (int q = 0; q < 9; ++q) s_red[wid * 9 + q] = e[q];
            }
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 9; ++q) e[q] = ((s_red[q] + s_red[9 + q]) + s_red[18 + q]) + s_red[27 + q];
                loss_hdr snap;
                snap.sum_pos = e[0]; snap.sum_neg = e[1]; snap.n_pos = (int)e[2]; snap.tp = (int)e[3]; snap.tn = (int)e[4]; snap.fp = (int)e[5]; snap.k = k;
                hdr->sum_pos = e[0]; hdr->sum_neg = e[1]; hdr->n_pos = snap.n_pos; hdr->tp = snap.tp; hdr->tn = snap.tn; hdr->fp = snap.fp;
                hdr->sum_hard = e[6]; hdr->sum_cls = e[7]; hdr->cls_correct = (int)e[8];
                loss_finalize_body(&snap, e[6], e[7], (int)e[8], npix, n_cls, loss4);
                if (LOSS1_LD(&sy->timeout))
                    for (int j = 0; j < 7; ++j) loss4[j] = __uint_as_float(0x7fc00000u);
            }
        }
    }
    STAMP(10);
#undef STAMP
}
#undef LOSS1_LD
#undef LOSS1_ST

struct loss1_dev_state { std::mutex mu; hipStream_t last = nullptr; bool any = false, multi = false; hipEvent_t ev = nullptr; };
static loss1_dev_state g_loss1_dev[16];          // by device ordinal

// the one-launch form applies when every block can be resident (grid <= CUs) with its chunk in registers (<= 4 pixels per thread: 1 M pixels on 256 CUs)
static bool loss_one_launch(const float *logits, int k_out, const int32_t *y_true, long npix, float *loss, float *dlogits, loss_hdr *hdr,
                            unsigned *hist, unsigned *blockties, loss1_rec *rec, int max_grid, hipStream_t st, void *dbg)
{
    if (max_grid > LOSS_MAX_BLOCKS) max_grid = LOSS_MAX_BLOCKS;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }   // a graph replays anywhere: the chain has no barrier
    for (int r = 1; r <= LOSS1_MAX_R; r *= 2) {
        const long g = (npix + (long)r * LOSS1_BLOCK - 1) / ((long)r * LOSS1_BLOCK);
        if (g > max_grid) continue;
#define LOSS1_LAUNCH(RR) hipLaunchKernelGGL((loss_one_kernel<RR>), dim3((int)g), dim3(LOSS1_BLOCK), 0, st, logits, k_out, y_true, npix, hdr, hist, blockties, rec, dlogits, loss, (unsigned long long *)dbg)
        // Two of these kernels on two streams could each hold a part of the CUs and wait for the rest for ever (every block spins at the
        // barriers until all of its grid is resident; the spin limit would end it with a NaN loss after seconds).  Launches on ONE stream are
        // ordered anyway (the train step, bench.py: no cost); the first launch from a second stream of this process drains the device once,
        // and from then on every launch waits for the previous one's event.  Processes sharing a GPU: UBD_LOSS=chain.
        int dev = 0;
        (void)hipGetDevice(&dev);
        loss1_dev_state &S = g_loss1_dev[dev & 15];
        std::lock_guard<std::mutex> lk(S.mu);
#ifdef LOSS1_NO_STREAM_ORDER          // experiment build: shows that tests/test_gpu_loss.py::test_one_launch_form_from_two_streams_at_once has power
        if (false) {
#else
        if (S.any && S.last != st) {
#endif
            if (!S.multi) {
                if (hipDeviceSynchronize() != hipSuccess || hipEventCreateWithFlags(&S.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
                S.multi = true;
            } else if (hipStreamWaitEvent(st, S.ev, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
        if (r == 1) LOSS1_LAUNCH(1); else if (r == 2) LOSS1_LAUNCH(2); else LOSS1_LAUNCH(4);
#undef LOSS1_LAUNCH
        if (S.multi) (void)hipEventRecord(S.ev, st);
        S.last = st; S.any = true;
        return true;
    }
    return false;
}

// h != nullptr with a UBD_COMM_GLOBAL_LOSS communicator: the reductions of losses.py:86-126 (n_pos, n_neg, the two means, the
// top-k over the flattened batch, the classification mean) run over the images of ALL ranks -- the reference's semantics at
// the global batch (option 2).  Rank r holds the flat indices [r * npix, (r + 1) * npix) (equal shards).
// Eight small collectives on the caller's stream: (sums, counters, level-0 histogram), the two refined histograms, the tie
// counts (all-gather), (hard-negative / classification sums, class hits).  Integer histograms make every rank select the
// same threshold bit pattern; ties at the threshold go to the lower GLOBAL flat index like tf.nn.top_k.
size_t ubd_loss_zero_bytes(void) { return LOSS_HDR_BYTES + (3 * 2048 + LOSS1_COPIES * 4096) * sizeof(unsigned) + LOSS_MAX_BLOCKS * 64; }   // header + the three histograms: zero before every evaluation

int ubd_loss_impl(const float *logits, int k_out, const int32_t *y_true, long npix, float *loss, float *dlogits,
                  char *ws, hipStream_t st, ubd_handle *h, bool prezeroed)
{
    loss_layout L;
    loss_layout_compute(npix, &L);
    loss_hdr *hdr = (loss_hdr *)(ws + L.off_hdr);
    unsigned *hist = (unsigned *)(ws + L.off_hist);
    unsigned *blockties = (unsigned *)(ws + L.off_blockties);
    unsigned *rankties = (unsigned *)(ws + L.off_rankties);
    float *ce = (float *)(ws + L.off_ce);
    loss_part *part = (loss_part *)(ws + L.off_part);
    const bool glob = h && ubd_comm_global_loss(h);
    const int world = glob ? ubd_comm_world(h) : 1, rank = glob ? ubd_comm_rank(h) : 0;
    UBD_REQUIRE(world <= 256, "ubd_loss: batch-global loss supports at most 256 ranks");
    const long npix_total = npix * world;
    UBD_REQUIRE(npix_total < (1L << 31), "ubd_loss: too many pixels in the global batch");
    if (!prezeroed) UBD_CHECK_HIP(hipMemsetAsync(ws, 0, L.off_blockties, st));     // header + 3 histograms (the bf16 train step's prologue kernel has done it)
    int grid = (int)((npix + LOSS_BLOCK - 1) / LOSS_BLOCK);
    if (grid > LOSS_MAX_BLOCKS) grid = LOSS_MAX_BLOCKS;
    long chunk = (npix + grid - 1) / grid;
    chunk = (chunk + LOSS_BLOCK - 1) / LOSS_BLOCK * LOSS_BLOCK;
    const int cgrid = (int)((npix + chunk - 1) / chunk);
    int rc;
    if (!glob && !(h && h->loss_chain) &&
        loss_one_launch(logits, k_out, y_true, npix, loss, dlogits, hdr, (unsigned *)(ws + L.off_histk), blockties, (loss1_rec *)(ws + L.off_rec), h ? h->num_cus : LOSS_MAX_BLOCKS, st, ce)) {    // ce: unused by this form (stamps of a -DLOSS1_STAMPS build)
        UBD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(loss_stats_kernel, dim3(grid), dim3(LOSS_STATS_BLOCK), 0, st, logits, k_out, y_true, npix, hdr, hist, ce, part);
    if (glob) {
        if ((rc = ubd_comm_allreduce_raw(h, &hdr->sum_pos, 2, UBD_RED_F64, st))) return rc;
        if ((rc = ubd_comm_allreduce_raw(h, &hdr->n_pos, 6, UBD_RED_I32, st))) return rc;
        if ((rc = ubd_comm_allreduce_raw(h, hist, 2048, UBD_RED_U32, st))) return rc;
    }
    hipLaunchKernelGGL(loss_hist_kernel, dim3(grid), dim3(LOSS_BLOCK), 0, st, ce, npix, npix_total, hdr, hist, hist + 2048, 1);
    if (glob && (rc = ubd_comm_allreduce_raw(h, hist + 2048, 2048, UBD_RED_U32, st))) return rc;
    hipLaunchKernelGGL(loss_hist_kernel, dim3(grid), dim3(LOSS_BLOCK), 0, st, ce, npix, npix_total, hdr, hist + 2048, hist + 4096, 2);
    if (glob && (rc = ubd_comm_allreduce_raw(h, hist + 4096, 2048, UBD_RED_U32, st))) return rc;
    hipLaunchKernelGGL(loss_tiecount_kernel, dim3(cgrid), dim3(LOSS_BLOCK), 0, st, ce, npix, npix_total, chunk, hdr, hist + 4096, blockties);
    if (glob) {
        hipLaunchKernelGGL(loss_tiescan_kernel, dim3(1), dim3(1024), 0, st, blockties, cgrid);
        if ((rc = ubd_comm_allgather_u32(h, blockties + cgrid, rankties, st))) return rc;     // this rank's tie count -> everyone
        hipLaunchKernelGGL(loss_grad_kernel, dim3(cgrid), dim3(LOSS_GRAD_BLOCK), 0, st, logits, k_out, y_true, npix, chunk, hdr, blockties, ce, dlogits,
                           npix_total, rankties, rank, 0, (float *)nullptr, hist + 4096);
        if ((rc = ubd_comm_allreduce_raw(h, &hdr->sum_hard, 2, UBD_RED_F64, st))) return rc;
        if ((rc = ubd_comm_allreduce_raw(h, &hdr->cls_correct, 1, UBD_RED_I32, st))) return rc;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, hdr, npix_total, k_out - 1, loss);
    } else {                                             // one launch: prefix of the tie counts, gradient, loss values
        hipLaunchKernelGGL(loss_grad_kernel, dim3(cgrid), dim3(LOSS_GRAD_BLOCK), 0, st, logits, k_out, y_true, npix, chunk, hdr, blockties, ce, dlogits,
                           npix_total, rankties, rank, 1, loss, hist + 4096);
    }
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int ubd_loss(ubd_handle *h, const float *logits, const int32_t *y_true, int n, int map_h, int map_w,
                        float *loss, float *dlogits, void *workspace, size_t workspace_bytes, void *stream)
{
    UBD_REQUIRE(h && logits && y_true && loss && workspace, "ubd_loss: null argument");
    UBD_REQUIRE(n > 0 && map_h > 0 && map_w > 0, "ubd_loss: bad sizes");
    const long npix = (long)n * map_h * map_w;
    UBD_REQUIRE(npix < (1L << 31), "ubd_loss: too many pixels");
    UBD_REQUIRE(workspace_bytes >= ubd_loss_workspace_bytes(h, n, map_h, map_w), "ubd_loss: workspace too small");
    return ubd_loss_impl(logits, h->k_out, y_true, npix, loss, dlogits, (char *)workspace, (hipStream_t)stream, h);
}
