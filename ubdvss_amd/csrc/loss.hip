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

// Per-block partial sums of the statistics kernel.  Every block used to end with six atomic adds on the header: 256 blocks x 6
// read-modify-writes on ONE cache line are carried out one after the other at the memory side (~10 ns each) -- 15 of the statistics
// kernel's 26 us.  Now a block stores its sums in its own record (write-through stores, no contention), draws a ticket, and the last
// block out adds the records up in a fixed order (the loss values are bit-reproducible from run to run as a side effect).
struct loss_part { double a, b; int i0, i1, i2, i3; };          // sum_pos, sum_neg, n_pos, tp, tn, fp

struct loss_layout {
    size_t off_hdr, off_hist, off_blockties, off_rankties, off_part, off_ce, total;
};

static void loss_layout_compute(long npix, loss_layout *L)
{
    size_t off = 0;
    L->off_hdr = off;       off += LOSS_HDR_BYTES;
    L->off_hist = off;      off += 3 * 2048 * sizeof(unsigned);
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

// six sums at once: one pair of barriers instead of six (results valid in thread 0); sh: (blockDim.x / 64) * 6 doubles
__device__ __forceinline__ void block_reduce6(double (&v)[6], double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] += __shfl_down(v[k], o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) sh[wid * 6 + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] += sh[w * 6 + k];
    }
}

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
    // one pixel: hard-negative sum, gradient of the detection logit, classification part
    auto pixel = [&](long p, unsigned bits, bool sel, float x, int yt) {
        const float z = yt > 0 ? 1.f : 0.f;
        if (sel) s_hard += (double)__uint_as_float(bits);
        const float xc = fminf(fmaxf(x, LOGIT_LO), LOGIT_HI);
        const bool inside = (x >= LOGIT_LO) && (x <= LOGIT_HI);
        if (dlogits) {
            const float sig = 1.f / (1.f + expf(-xc));
            const float coef = z * w_pos + (1.f - z) * (w_neg + (sel ? w_hard : 0.f));
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
                        dlogits[p * k_out + 1 + c] = (expf(lg[c] - lse) - (c == lab ? 1.f : 0.f)) * w_cls;
            } else if (dlogits) {
                for (int c = 0; c < n_cls; ++c) dlogits[p * k_out + 1 + c] = 0.f;
            }
        }
    };
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

// h != nullptr with a UBD_COMM_GLOBAL_LOSS communicator: the reductions of losses.py:86-126 (n_pos, n_neg, the two means, the
// top-k over the flattened batch, the classification mean) run over the images of ALL ranks -- the reference's semantics at
// the global batch (option 2).  Rank r holds the flat indices [r * npix, (r + 1) * npix) (equal shards).
// Eight small collectives on the caller's stream: (sums, counters, level-0 histogram), the two refined histograms, the tie
// counts (all-gather), (hard-negative / classification sums, class hits).  Integer histograms make every rank select the
// same threshold bit pattern; ties at the threshold go to the lower GLOBAL flat index like tf.nn.top_k.
size_t ubd_loss_zero_bytes(void) { return LOSS_HDR_BYTES + 3 * 2048 * sizeof(unsigned); }   // header + the three histograms: zero before every evaluation

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
