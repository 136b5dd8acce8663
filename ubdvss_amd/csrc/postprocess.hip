// Device postprocess of the ubdvss hot path on gfx950: logits -> binary map -> external
// 8-connected components -> contourArea filter -> minAreaRect -> boxPoints -> rounded quads
// (+ optional per-object class vote), all without leaving the GPU and without any serial
// border following.
//
// Reference call sites: semantic_segmentation/model_runner.py:121-134,
// segmap_manager.py:41-69, utils.py:51-60, :135-138 (cv2.findContours RETR_EXTERNAL /
// CHAIN_APPROX_SIMPLE, contourArea, minAreaRect, boxPoints, drawContours fill).
// OpenCV 3.4 semantics are restated from its published algorithms (oracle/cv_post.c holds the
// sequential restatement this file is tested against; parity vs cv2 itself is unpinned because
// OpenCV is not available in the build environment).
//
// Fully parallel formulation (all equivalences are checked against the sequential restatement in
// tests/test_oracle_post.py on thousands of random maps):
//   * RETR_EXTERNAL: a component is returned iff the pixel north of its raster-first pixel is the
//     image frame or OUTSIDE background (4-connected background region touching the frame).
//   * filled contour (drawContours thickness=-1) = every pixel enclosed by the component = pixels
//     whose nesting chain (region -> region north of its raster-first pixel -> ...) ends in it.
//   * contourArea of the traced outer border = Q4 + Q3/2 over the 2x2 pixel quads of the FILLED
//     region (Gray's bit-quad area): quads with 4 corners inside count 1, with 3 corners 1/2.
//   * minAreaRect needs only the convex hull, which is the hull of the per-row x extents.
// Pipeline (one launch each, grid over all pixels of the batch unless noted):
//   init      fg = logit0 > thr (strict); union-find node per pixel (+ node 0 = frame/outside),
//             initialised to the start of the pixel's horizontal run via wave ballots
//   merge     lock-free union-find (atomicMin, min-index roots), only the non-redundant links:
//             foreground 8-connected, background 4-connected, frame contact
//   flatten   label = root
//   roots     external roots get a slot; owner: per pixel, slot of the enclosing external component
//   area      bit-quad area per slot (wave-aggregated atomics); keep: contourArea > min_area
//   extents   per-row min/max x of every kept object; boxes (one lane per object): hull from the
//             row extents ordered like cv::convexHull(clockwise=true), then rotatingCalipers /
//             minAreaRect / boxPoints in OpenCV's float32/float64 operation order, np.round(x*scale)
//   vote      (n_classes > 0) mean softmax over the filled region, argmax
//   emit      objects ordered like cv2 returns them (last discovered first)
#include "common.h"

#pragma clang fp contract(off)

#define CV_PI 3.1415926535897932384626433832795
#define STAGE_INTS 10     // root, quad[8], spare

struct pp_layout {
    size_t off_nroots, off_nkept;   // int32 [n] each (zeroed every call, start of the workspace)
    size_t off_label;               // int32 [n][hw+1]
    size_t off_fg;                  // uint8 [n][hw]
    size_t off_owner;               // int32 [n][hw]    slot of the enclosing external component or -1
    size_t off_rootslot;            // int32 [n][hw]    valid at external root pixels
    size_t off_roots;               // int32 [n][root_cap]  root pixel of slot
    size_t off_area2;               // int32 [n][root_cap]  2 * contourArea
    size_t off_kept;                // int32 [n][root_cap]  kept index or -1
    size_t off_stage;               // int32 [n][cap][STAGE_INTS]
    size_t off_ymax;                // int32 [n][cap]
    size_t off_rows;                // int32 [n][cap][6*h]: row extents (2h) + hull points (2h points)
    size_t off_vote;                // float [n][cap][n_cls+1]
    size_t total;
    int root_cap;
};

static void pp_layout_compute(int n, int h, int w, int cap, int n_cls, pp_layout *L)
{
    const size_t hw = (size_t)h * w;
    size_t off = 0;
    // most external 8-connected components a map can hold: isolated pixels on every other row and column
    L->root_cap = ((h + 1) / 2) * ((w + 1) / 2) + 1;
    L->off_nroots = off;   off += ubd_align_up(sizeof(int) * n, 256);
    L->off_nkept = off;    off += ubd_align_up(sizeof(int) * n, 256);
    L->off_label = off;    off += ubd_align_up(sizeof(int) * n * (hw + 1), 256);
    L->off_fg = off;       off += ubd_align_up(n * hw, 256);
    L->off_owner = off;    off += ubd_align_up(sizeof(int) * n * hw, 256);
    L->off_rootslot = off; off += ubd_align_up(sizeof(int) * n * hw, 256);
    L->off_roots = off;    off += ubd_align_up(sizeof(int) * (size_t)n * L->root_cap, 256);
    L->off_area2 = off;    off += ubd_align_up(sizeof(int) * (size_t)n * L->root_cap, 256);
    L->off_kept = off;     off += ubd_align_up(sizeof(int) * (size_t)n * L->root_cap, 256);
    L->off_stage = off;    off += ubd_align_up(sizeof(int) * (size_t)n * cap * STAGE_INTS, 256);
    L->off_ymax = off;     off += ubd_align_up(sizeof(int) * (size_t)n * cap, 256);
    L->off_rows = off;     off += ubd_align_up(sizeof(int) * (size_t)n * cap * 6 * h, 256);
    L->off_vote = off;     off += ubd_align_up(sizeof(float) * (size_t)n * cap * (n_cls + 1), 256);
    L->total = off;
}

extern "C" size_t ubd_postprocess_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w, int cap)
{
    pp_layout L;
    pp_layout_compute(n, map_h, map_w, cap, h ? h->cfg.n_classes : 0, &L);
    return L.total;
}

// ------------------------------------------------------------------------------------ init
// blockDim must be a multiple of 64; lanes of a wave hold 64 consecutive flat pixels.
__global__ __launch_bounds__(256) void pp_init_kernel(const float *__restrict__ logits, int k_out, float thr, long npix,
                                                      int hw, int w, unsigned char *__restrict__ fg,
                                                      int *__restrict__ label, int *__restrict__ binary_map)
{
    const int lane = threadIdx.x & 63;
    const long nround = (npix + 63) / 64 * 64;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < nround; p += (long)gridDim.x * blockDim.x) {
        const bool valid = p < npix;
        int f = 0, img = 0, loc = 0, x = 0;
        if (valid) {
            img = (int)(p / hw); loc = (int)(p % hw); x = loc % w;
            f = logits[p * k_out] > thr ? 1 : 0;                 // strict >, model_runner.py:124
            fg[p] = (unsigned char)f;
            if (binary_map) binary_map[p] = f;
        }
        // same class as the pixel to the left (same row)?
        int fl = __shfl_up(f, 1, 64);
        if (lane == 0 && valid && x > 0) fl = logits[(p - 1) * k_out] > thr ? 1 : 0;
        const bool same_left = valid && x > 0 && fl == f;
        const unsigned long long breaks = __ballot(!same_left);  // bit l: lane l starts a run (or is invalid)
        if (valid) {
            const unsigned long long below = breaks & ((2ull << lane) - 1ull);   // lanes <= mine
            int start_off;                                        // distance back to the run start
            if (below) start_off = lane - (63 - __clzll(below));
            else start_off = lane + 1;                            // run continues into the previous wave: link there
            int *lab = label + (size_t)img * (hw + 1);
            lab[loc + 1] = loc + 1 - start_off;
            if (loc == 0) lab[0] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------ union-find
__device__ __forceinline__ int uf_find(int *lab, int a)
{
    int p = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != a) {
        a = p;
        p = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return a;
}

__device__ __forceinline__ void uf_union(int *lab, int a, int b)
{
    for (;;) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }          // a > b: hang a under b
        int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;                                         // a was no longer a root: retry with its parent
    }
}

// Only links that are not implied by the run initialisation or by a neighbour's links.
__global__ __launch_bounds__(256) void pp_merge_kernel(const unsigned char *__restrict__ fg, int *__restrict__ label,
                                                       long npix, int h, int w)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const int y = loc / w, x = loc % w;
        const unsigned char *m = fg + (size_t)img * hw;
        int *lab = label + (size_t)img * (hw + 1);
        const int me = loc + 1;
        const int c = m[loc];
        const bool W = x > 0 && m[loc - 1] == c;
        if (c) {
            if (y > 0) {
                const bool N = m[loc - w];
                const bool NW = x > 0 && m[loc - w - 1];
                if (N) {
                    if (!(W && NW)) uf_union(lab, me, me - w);
                } else {
                    if (NW && !W) uf_union(lab, me, me - w - 1);
                    const bool NE = x < w - 1 && m[loc - w + 1];
                    const bool E = x < w - 1 && m[loc + 1];
                    if (NE && !E) uf_union(lab, me, me - w + 1);
                }
            }
        } else {
            if (y > 0 && !m[loc - w]) {
                const bool NW = x > 0 && !m[loc - w - 1];
                if (!(W && NW)) uf_union(lab, me, me - w);
            }
            // frame contact: one link per run on the first / last row, the row ends elsewhere
            const bool row_edge = (y == 0 || y == h - 1) && !W;
            if (row_edge || x == 0 || x == w - 1) uf_union(lab, me, 0);
        }
    }
}

__global__ __launch_bounds__(256) void pp_flatten_kernel(int *__restrict__ label, int n, int hw)
{
    const long total = (long)n * (hw + 1);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / (hw + 1)), node = (int)(p % (hw + 1));
        int *lab = label + (size_t)img * (hw + 1);
        const int r = uf_find(lab, node);
        __hip_atomic_store(&lab[node], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------ roots / owner
__global__ __launch_bounds__(256) void pp_roots_kernel(const unsigned char *__restrict__ fg, const int *__restrict__ label,
                                                       long npix, int h, int w, int *__restrict__ nroots,
                                                       int *__restrict__ roots, int *__restrict__ rootslot,
                                                       int *__restrict__ area2, int root_cap)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        if (!fg[p]) continue;
        const int *lab = label + (size_t)img * (hw + 1);
        if (lab[loc + 1] != loc + 1) continue;             // not the raster-first pixel of its component
        const bool external = (loc < w) || (lab[loc + 1 - w] == 0);
        if (!external) continue;
        const int idx = atomicAdd(&nroots[img], 1);        // idx < root_cap always: at most ceil(h/2) * ceil(w/2) components
        roots[(size_t)img * root_cap + idx] = loc;
        area2[(size_t)img * root_cap + idx] = 0;
        rootslot[p] = idx;
    }
}

// owner[p] = slot of the external component that encloses pixel p, or -1.
__global__ __launch_bounds__(256) void pp_owner_kernel(const unsigned char *__restrict__ fg, const int *__restrict__ label,
                                                       const int *__restrict__ rootslot, long npix, int h, int w,
                                                       int *__restrict__ owner)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const unsigned char *m = fg + (size_t)img * hw;
        const int *lab = label + (size_t)img * (hw + 1);
        int node = lab[loc + 1];
        int own = -1;
        for (int guard = 0; guard < 4096; ++guard) {
            if (node == 0) break;                           // outside background
            const int r = node - 1;                         // raster-first pixel of this region
            if (r < w) { if (m[r]) own = rootslot[(size_t)img * hw + r]; break; }
            const int up = lab[r - w + 1];                  // region north of it
            if (m[r] && up == 0) { own = rootslot[(size_t)img * hw + r]; break; }
            node = up;
        }
        owner[p] = own;
    }
}

// ------------------------------------------------------------------------------------ area
// Adds `val` to acc[key] for all lanes with key >= 0, one atomic per distinct key in the wave.
__device__ __forceinline__ void wave_atomic_add_by_key(int *acc, int key, int val)
{
    unsigned long long todo = __ballot(key >= 0 && val != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, 64);
        const bool mine = (key == k) && (val != 0);
        int v = mine ? val : 0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == leader) atomicAdd(&acc[k], v);
        todo &= ~__ballot(mine);
    }
}

__global__ __launch_bounds__(256) void pp_area_kernel(const int *__restrict__ owner, long npix, int h, int w,
                                                      int *__restrict__ area2, int root_cap)
{
    const int hw = h * w;
    const long nround = (npix + 63) / 64 * 64;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < nround; p += (long)gridDim.x * blockDim.x) {
        int key = -1, val = 0;
        if (p < npix) {
            const int img = (int)(p / hw), loc = (int)(p % hw);
            const int y = loc / w, x = loc % w;
            if (x < w - 1 && y < h - 1) {
                const int o0 = owner[p], o1 = owner[p + 1], o2 = owner[p + w], o3 = owner[p + w + 1];
                const int o = max(max(o0, o1), max(o2, o3));          // all non-negative owners in a quad agree
                if (o >= 0) {
                    const int cnt = (o0 == o) + (o1 == o) + (o2 == o) + (o3 == o);
                    val = cnt == 4 ? 2 : (cnt == 3 ? 1 : 0);
                    key = img * root_cap + o;
                }
            }
        }
        wave_atomic_add_by_key(area2, key, val);
    }
}

__global__ __launch_bounds__(256) void pp_keep_kernel(int n, int h, const int *__restrict__ nroots, const int *__restrict__ roots,
                                                      const int *__restrict__ area2, int root_cap, float min_area,
                                                      int *__restrict__ nkept, int *__restrict__ kept,
                                                      int *__restrict__ stage, int *__restrict__ ymax,
                                                      int *__restrict__ rows, int cap, float *__restrict__ vote, int n_cls)
{
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        const int nr = nroots[img];
        for (int s = threadIdx.x; s < nr; s += blockDim.x) {
            const size_t gi = (size_t)img * root_cap + s;
            const double area = (double)area2[gi] * 0.5;
            int k = -1;
            if (area > (double)min_area) {                            // utils.py:55 (strict >)
                k = atomicAdd(&nkept[img], 1);
                if (k < cap) {
                    int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
                    st[0] = roots[gi];
                    ymax[(size_t)img * cap + k] = 0;
                    int *r = rows + ((size_t)img * cap + k) * (size_t)(6 * h);
                    for (int y = 0; y < h; ++y) { r[2 * y] = 0x7fffffff; r[2 * y + 1] = -1; }
                    if (n_cls > 0) {
                        float *v = vote + ((size_t)img * cap + k) * (n_cls + 1);
                        for (int c = 0; c <= n_cls; ++c) v[c] = 0.f;
                    }
                } else {
                    k = -1;                                           // overflow: reported through counts[] > cap
                }
            }
            kept[gi] = k;
        }
    }
}

// ------------------------------------------------------------------------------------ extents
__global__ __launch_bounds__(256) void pp_extents_kernel(const int *__restrict__ owner, const int *__restrict__ kept, long npix,
                                                         int h, int w, int root_cap, int cap, int *__restrict__ rows,
                                                         int *__restrict__ ymax)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int o = owner[p];
        if (o < 0) continue;
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const int y = loc / w, x = loc % w;
        const bool left_end = (x == 0) || owner[p - 1] != o;
        const bool right_end = (x == w - 1) || owner[p + 1] != o;
        const bool bottom = (y == h - 1) || owner[p + w] != o;
        if (!(left_end || right_end || bottom)) continue;
        const int k = kept[(size_t)img * root_cap + o];
        if (k < 0) continue;
        int *r = rows + ((size_t)img * cap + k) * (size_t)(6 * h);
        if (left_end) atomicMin(&r[2 * y], x);
        if (right_end) atomicMax(&r[2 * y + 1], x);
        if (bottom) atomicMax(&ymax[(size_t)img * cap + k], y);
    }
}

struct ipt { int x, y; };
__device__ __forceinline__ long long cross3(ipt o, ipt a, ipt b)
{
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

// cv::minAreaRect + cv::boxPoints on a strictly convex polygon `hp` (n >= 3) ordered like
// cv::convexHull(clockwise=true) -- same float32/float64 operation order as OpenCV 3.4
// rotatingCalipers (rotcalipers.cpp), minAreaRect (rotcalipers.cpp) and RotatedRect::points.
// `etab` (optional): per-edge table [3][n] = (vx, vy, 1/length) of edge i -> i+1 computed beforehand with exactly the
// arithmetic of `vec` below (the wave-cooperative kernel fills it one edge per lane, which takes the double-precision
// square roots and divisions out of the serial calipers loop).
__device__ void min_area_box(const ipt *hp, int n, float *box8, const float *etab = nullptr)
{
    float cxr = 0.f, cyr = 0.f, bw = 0.f, bh = 0.f, angle = 0.f;
    if (n > 2) {
        float minarea = 3.402823466e+38f;
        int buf_i0 = 0, buf_i5 = 0;
        float buf1 = 0.f, buf2 = 0.f, buf3 = 0.f, buf4 = 0.f;
        int left = 0, bottom = 0, right = 0, top = 0;
        float left_x, right_x, top_y, bottom_y;
        left_x = right_x = (float)hp[0].x;
        top_y = bottom_y = (float)hp[0].y;
        for (int i = 0; i < n; ++i) {
            const float px = (float)hp[i].x, py = (float)hp[i].y;
            if (px < left_x) left_x = px, left = i;
            if (px > right_x) right_x = px, right = i;
            if (py > top_y) top_y = py, top = i;
            if (py < bottom_y) bottom_y = py, bottom = i;
        }
        auto vec = [&](int i, float &vx, float &vy, float &inv) {
            if (etab) { vx = etab[i]; vy = etab[n + i]; inv = etab[2 * n + i]; return; }
            const int j = (i + 1 < n) ? i + 1 : 0;
            const double dx = (float)hp[j].x - (float)hp[i].x;
            const double dy = (float)hp[j].y - (float)hp[i].y;
            vx = (float)dx; vy = (float)dy;
            inv = (float)(1. / sqrt(dx * dx + dy * dy));
        };
        float orientation = 0.f;
        {
            float ax_, ay_, t_;
            vec(n - 1, ax_, ay_, t_);
            double ax = ax_, ay = ay_;
            for (int i = 0; i < n; ++i) {
                float bx_, by_;
                vec(i, bx_, by_, t_);
                const double bx = bx_, by = by_;
                const double convexity = ax * by - ay * bx;
                if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
                ax = bx; ay = by;
            }
        }
        float base_a = orientation, base_b = 0.f;
        int seq[4] = {bottom, right, top, left};
        for (int k = 0; k < n; ++k) {
            float vx[4], vy[4], inv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vec(seq[e], vx[e], vy[e], inv[e]);
            const float dp0 = +base_a * vx[0] + base_b * vy[0];
            const float dp1 = -base_b * vx[1] + base_a * vy[1];
            const float dp2 = -base_a * vx[2] - base_b * vy[2];
            const float dp3 = +base_b * vx[3] - base_a * vy[3];
            float maxcos = dp0 * inv[0];
            int main_element = 0;
            float c1 = dp1 * inv[1]; if (c1 > maxcos) { main_element = 1; maxcos = c1; }
            float c2 = dp2 * inv[2]; if (c2 > maxcos) { main_element = 2; maxcos = c2; }
            float c3 = dp3 * inv[3]; if (c3 > maxcos) { main_element = 3; maxcos = c3; }
            {
                const float lead_x = vx[main_element] * inv[main_element];
                const float lead_y = vy[main_element] * inv[main_element];
                switch (main_element) {
                case 0: base_a = lead_x;  base_b = lead_y;  break;
                case 1: base_a = lead_y;  base_b = -lead_x; break;
                case 2: base_a = -lead_x; base_b = -lead_y; break;
                default: base_a = -lead_y; base_b = lead_x; break;
                }
            }
            seq[main_element] += 1;
            seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
            {
                float dx = (float)hp[seq[1]].x - (float)hp[seq[3]].x;
                float dy = (float)hp[seq[1]].y - (float)hp[seq[3]].y;
                const float width = dx * base_a + dy * base_b;
                dx = (float)hp[seq[2]].x - (float)hp[seq[0]].x;
                dy = (float)hp[seq[2]].y - (float)hp[seq[0]].y;
                const float height = -dx * base_b + dy * base_a;
                const float area = width * height;
                if (area <= minarea) {
                    minarea = area;
                    buf_i0 = seq[3];
                    buf1 = base_a; buf2 = width; buf3 = base_b; buf4 = height;
                    buf_i5 = seq[0];
                }
            }
        }
        const float A1 = buf1, B1 = buf3, A2 = -buf3, B2 = buf1;
        const float C1 = A1 * (float)hp[buf_i0].x + (float)hp[buf_i0].y * B1;
        const float C2 = A2 * (float)hp[buf_i5].x + (float)hp[buf_i5].y * B2;
        const float idet = 1.f / (A1 * B2 - A2 * B1);
        const float ox = (C1 * B2 - C2 * B1) * idet;
        const float oy = (A1 * C2 - A2 * C1) * idet;
        const float o1x = A1 * buf2, o1y = B1 * buf2, o2x = A2 * buf4, o2y = B2 * buf4;
        cxr = ox + (o1x + o2x) * 0.5f;
        cyr = oy + (o1y + o2y) * 0.5f;
        bw = (float)sqrt((double)o1x * o1x + (double)o1y * o1y);
        bh = (float)sqrt((double)o2x * o2x + (double)o2y * o2y);
        angle = (float)atan2((double)o1y, (double)o1x);
    } else if (n == 2) {
        cxr = ((float)hp[0].x + (float)hp[1].x) * 0.5f;
        cyr = ((float)hp[0].y + (float)hp[1].y) * 0.5f;
        const double dx = (float)hp[1].x - (float)hp[0].x, dy = (float)hp[1].y - (float)hp[0].y;
        bw = (float)sqrt(dx * dx + dy * dy);
        bh = 0.f;
        angle = (float)atan2(dy, dx);
    } else if (n == 1) {
        cxr = (float)hp[0].x; cyr = (float)hp[0].y;
    }
    angle = (float)(angle * 180 / CV_PI);
    // RotatedRect::points
    const double _angle = angle * CV_PI / 180.;
    const float b = (float)cos(_angle) * 0.5f;
    const float a = (float)sin(_angle) * 0.5f;
    box8[0] = cxr - a * bh - b * bw;
    box8[1] = cyr + b * bh - a * bw;
    box8[2] = cxr + a * bh - b * bw;
    box8[3] = cyr - b * bh - a * bw;
    box8[4] = 2 * cxr - box8[0];
    box8[5] = 2 * cyr - box8[1];
    box8[6] = 2 * cxr - box8[2];
    box8[7] = 2 * cyr - box8[3];
}

// Builds the strictly convex hull from per-row extents rows[2*r], rows[2*r+1], r = 0..nrows-1
// (row y = y0 + r) IN PLACE (hull points overwrite `rows` storage viewed as ipt[]), ordered like
// cv::convexHull(points, clockwise=true): start at the min-x (then min-y) vertex, first toward +y.
// Returns the vertex count; `hp` receives the pointer.
__device__ int hull_from_rows(int *rows, int nrows, int y0, ipt *scratch_pts)
{
    // left chain (top -> bottom) into scratch_pts[0..), right chain (top -> bottom) after it
    // scratch_pts has room for 2*nrows points.
    ipt *lc = scratch_pts;
    int nl = 0;
    for (int r = 0; r < nrows; ++r) {
        ipt p = {rows[2 * r], y0 + r};
        while (nl >= 2 && cross3(lc[nl - 2], lc[nl - 1], p) >= 0) --nl;
        lc[nl++] = p;
    }
    ipt *rc = scratch_pts + nl;
    int nr = 0;
    for (int r = 0; r < nrows; ++r) {
        ipt p = {rows[2 * r + 1], y0 + r};
        while (nr >= 2 && cross3(rc[nr - 2], rc[nr - 1], p) <= 0) --nr;
        rc[nr++] = p;
    }
    // cyclic order: lc[0..nl-1] (downwards), then rc[nr-1..0] (upwards)
    // reverse rc in place
    for (int a = 0, b = nr - 1; a < b; ++a, --b) { ipt t = rc[a]; rc[a] = rc[b]; rc[b] = t; }
    int n = nl + nr;
    ipt *P = scratch_pts;
    // drop duplicated junction points
    if (nr > 0 && P[nl - 1].x == P[nl].x && P[nl - 1].y == P[nl].y) {           // bottom
        for (int k = nl; k < n - 1; ++k) P[k] = P[k + 1];
        --n;
    }
    if (n > 1 && P[n - 1].x == P[0].x && P[n - 1].y == P[0].y) --n;            // top
    // remove collinear / repeated vertices until stable (junctions may be collinear)
    bool changed = true;
    while (changed && n > 2) {
        changed = false;
        for (int k = 0; k < n && n > 2; ++k) {
            const ipt a = P[(k + n - 1) % n], b = P[k], c = P[(k + 1) % n];
            if (cross3(a, b, c) == 0) {
                for (int m = k; m < n - 1; ++m) P[m] = P[m + 1];
                --n; --k; changed = true;
            }
        }
    }
    if (n <= 2) {
        // degenerate (all collinear): cv2 returns the two extreme points, lexicographic min first
        if (n == 2) {
            const bool swap = (P[1].x < P[0].x) || (P[1].x == P[0].x && P[1].y < P[0].y);
            if (swap) { ipt t = P[0]; P[0] = P[1]; P[1] = t; }
        }
        return n;
    }
    // rotate so that the min-x (then min-y) vertex comes first
    int s = 0;
    for (int k = 1; k < n; ++k)
        if (P[k].x < P[s].x || (P[k].x == P[s].x && P[k].y < P[s].y)) s = k;
    if (s != 0) {   // rotate left by s in place (three reversals)
        auto rev = [&](int a, int b) { for (; a < b; ++a, --b) { ipt t = P[a]; P[a] = P[b]; P[b] = t; } };
        rev(0, s - 1); rev(s, n - 1); rev(0, n - 1);
    }
    return n;
}


// one lane per kept object: hull from row extents -> minAreaRect -> boxPoints -> rounded quad
__global__ __launch_bounds__(64) void pp_boxes_kernel(int n, int h, int w, const int *__restrict__ nkept, int *__restrict__ stage,
                                                      const int *__restrict__ ymax, int *__restrict__ rows_ws, int cap, int scale)
{
    const long total = (long)n * cap;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int img = (int)(t / cap), k = (int)(t % cap);
        if (k >= min(nkept[img], cap)) continue;
        int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
        const int y0 = st[0] / w;
        const int nrows = ymax[(size_t)img * cap + k] - y0 + 1;
        int *rows = rows_ws + ((size_t)img * cap + k) * (size_t)(6 * h);
        ipt *pts = (ipt *)(rows + 2 * h);                             // 4h ints = room for 2h points
        const int nh = hull_from_rows(rows + 2 * y0, nrows, y0, pts);
        float box[8];
        min_area_box(pts, nh, box);
#pragma unroll
        for (int j = 0; j < 8; ++j) st[1 + j] = (int)rintf(box[j] * (float)scale);   // np.round: half to even
        st[9] = 0;
    }
}

// Wave-cooperative variant (default): one wave per kept object, row extents and hull in LDS.
// The left / right hull chains are found by wave-parallel gift wrapping over the row extents (exact
// integer slope comparisons); lane 0 then orders the polygon like cv::convexHull and runs the calipers.
__device__ __forceinline__ void hull_finish(ipt *P, int nl, int nr, int &n_out)
{
    // P = lc[0..nl-1] (top -> bottom) followed by rc[0..nr-1] (top -> bottom): reverse rc
    ipt *rc = P + nl;
    for (int a = 0, b = nr - 1; a < b; ++a, --b) { ipt t = rc[a]; rc[a] = rc[b]; rc[b] = t; }
    int n = nl + nr;
    if (nr > 0 && P[nl - 1].x == P[nl].x && P[nl - 1].y == P[nl].y) {           // bottom junction
        for (int k = nl; k < n - 1; ++k) P[k] = P[k + 1];
        --n;
    }
    if (n > 1 && P[n - 1].x == P[0].x && P[n - 1].y == P[0].y) --n;            // top junction
    bool changed = true;
    while (changed && n > 2) {
        changed = false;
        for (int k = 0; k < n && n > 2; ++k) {
            const ipt a = P[(k + n - 1) % n], b = P[k], c = P[(k + 1) % n];
            if (cross3(a, b, c) == 0) {
                for (int m = k; m < n - 1; ++m) P[m] = P[m + 1];
                --n; --k; changed = true;
            }
        }
    }
    if (n == 2) {
        const bool swap = (P[1].x < P[0].x) || (P[1].x == P[0].x && P[1].y < P[0].y);
        if (swap) { ipt t = P[0]; P[0] = P[1]; P[1] = t; }
    } else if (n > 2) {
        int s = 0;
        for (int k = 1; k < n; ++k)
            if (P[k].x < P[s].x || (P[k].x == P[s].x && P[k].y < P[s].y)) s = k;
        if (s != 0) {
            auto rev = [&](int a, int b) { for (; a < b; ++a, --b) { ipt t = P[a]; P[a] = P[b]; P[b] = t; } };
            rev(0, s - 1); rev(s, n - 1); rev(0, n - 1);
        }
    }
    n_out = n;
}

// One wave, one kept object: row extents (global, 2 ints per row from row y0) -> hull -> minAreaRect -> boxPoints -> rounded quad
// into st[1..8].  rws: the wave's LDS scratch of 12 * h + 4 ints (row extents | hull points | edge table).  ATOMIC: the
// extents were accumulated by atomics of THIS launch (the fused one-launch front end): read them past the CU's vector L1.
template <bool ATOMIC>
__device__ __forceinline__ void pp_box_object(int *rws, const int *__restrict__ g, int nrows, int y0, int h, int lane, int scale,
                                              int *__restrict__ st)
{
    ipt *pts = (ipt *)(rws + 2 * h);
    float *etab = (float *)(rws + 6 * h);                   // edge table of the hull: 3 x (<= 2h) floats
    for (int r = lane; r < 2 * nrows; r += 64)
        rws[r] = ATOMIC ? __hip_atomic_load(&g[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int cnt[2] = {0, 0};
#pragma unroll
    for (int side = 0; side < 2; ++side) {            // 0: left chain (min x), 1: right chain (max x)
        // Gift wrapping down the chain: from vertex row c the next vertex is the later row with the
        // extreme slope dx/dy (min for the left chain, max for the right one; farthest on ties, which
        // drops collinear points).  Candidates are spread over the 64 lanes, fractions compared
        // exactly by int32 cross-multiplication (|dx|, dy < 2^15), then a xor-butterfly reduction.
        ipt *out = pts + (side ? cnt[0] : 0);
        int nout = 0, c = 0;
        for (;;) {
            const int xc = rws[2 * c + side];
            if (lane == 0) out[nout] = (ipt){xc, y0 + c};
            ++nout;
            if (c >= nrows - 1) break;
            int bn = 0, bd = 0, br = -1;
            for (int r = c + 1 + lane; r < nrows; r += 64) {
                const int nn = rws[2 * r + side] - xc, dd = r - c;
                bool better = true;
                if (bd != 0) {
                    const int lhs = nn * bd, rhs = bn * dd;
                    better = side ? (lhs >= rhs) : (lhs <= rhs);      // later row wins ties
                }
                if (better) { bn = nn; bd = dd; br = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int on = __shfl_xor(bn, o, 64), od = __shfl_xor(bd, o, 64), orr = __shfl_xor(br, o, 64);
                bool take;
                if (od == 0) take = false;
                else if (bd == 0) take = true;
                else {
                    const int lhs = on * bd, rhs = bn * od;
                    take = lhs == rhs ? (orr > br) : (side ? (lhs > rhs) : (lhs < rhs));
                }
                if (take) { bn = on; bd = od; br = orr; }
            }
            c = br;
        }
        cnt[side] = nout;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int nh = 0;
    if (lane == 0) hull_finish(pts, cnt[0], cnt[1], nh);
    nh = __builtin_amdgcn_readfirstlane(nh);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < nh; e += 64) {           // one hull edge per lane (same arithmetic as min_area_box::vec)
        const int j = (e + 1 < nh) ? e + 1 : 0;
        const double dx = (float)pts[j].x - (float)pts[e].x;
        const double dy = (float)pts[j].y - (float)pts[e].y;
        etab[e] = (float)dx; etab[nh + e] = (float)dy;
        etab[2 * nh + e] = (float)(1. / sqrt(dx * dx + dy * dy));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float box[8];
        min_area_box(pts, nh, box, etab);
#pragma unroll
        for (int j = 0; j < 8; ++j) st[1 + j] = (int)rintf(box[j] * (float)scale);   // np.round: half to even
        st[9] = 0;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(512) void pp_boxes_wave_kernel(int n, int h, int w, const int *__restrict__ nkept,
                                                            int *__restrict__ stage, const int *__restrict__ ymax,
                                                            const int *__restrict__ rows_ws, int cap, int scale)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int *rws = smem + wid * (12 * h + 4);
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        const int nk = min(nkept[img], cap);
        for (int k = wid; k < nk; k += nw) {
            int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
            const int y0 = st[0] / w;
            const int nrows = ymax[(size_t)img * cap + k] - y0 + 1;
            const int *g = rows_ws + ((size_t)img * cap + k) * (size_t)(6 * h) + 2 * y0;
            pp_box_object<false>(rws, g, nrows, y0, h, lane, scale, st);
        }
    }
}

// ------------------------------------------------------------------------------------ vote
// Class vote (segmap_manager.py:59-67): mean over the filled contour of softmax(class logits).
__global__ __launch_bounds__(256) void pp_vote_kernel(const float *__restrict__ logits, int k_out, const int *__restrict__ owner,
                                                      const int *__restrict__ kept, long npix, int hw, int root_cap, int cap,
                                                      float *__restrict__ vote)
{
    const int n_cls = k_out - 1;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int o = owner[p];
        if (o < 0) continue;
        const int img = (int)(p / hw);
        const int k = kept[(size_t)img * root_cap + o];
        if (k < 0) continue;
        const float *lg = logits + p * k_out + 1;
        float mx = lg[0];
        for (int c = 1; c < n_cls; ++c) mx = fmaxf(mx, lg[c]);
        float sum = 0.f;
        for (int c = 0; c < n_cls; ++c) sum += expf(lg[c] - mx);
        float *v = vote + ((size_t)img * cap + k) * (n_cls + 1);
        for (int c = 0; c < n_cls; ++c) atomicAdd(&v[c], expf(lg[c] - mx) / sum);
    }
}

// ------------------------------------------------------------------------------------ emit
__global__ __launch_bounds__(256) void pp_emit_kernel(int n, const int *__restrict__ nkept, const int *__restrict__ stage,
                                                      const float *__restrict__ vote, int n_cls, int cap, int *__restrict__ quads,
                                                      int *__restrict__ classes, int *__restrict__ counts)
{
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        const int total = nkept[img];
        const int nk = min(total, cap);
        if (threadIdx.x == 0) counts[img] = total;
        const int *st = stage + (size_t)img * cap * STAGE_INTS;
        for (int s = threadIdx.x; s < nk; s += blockDim.x) {
            const int root = st[s * STAGE_INTS];
            int rank = 0;                                        // cv2 order: last discovered first
            for (int t = 0; t < nk; ++t) rank += (st[t * STAGE_INTS] > root) ? 1 : 0;
            int *q = quads + ((size_t)img * cap + rank) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = st[s * STAGE_INTS + 1 + j];
            if (classes) {
                int best = 0;
                if (n_cls > 0) {
                    const float *v = vote + ((size_t)img * cap + s) * (n_cls + 1);
                    float bv = v[0];
                    for (int c = 1; c < n_cls; ++c)
                        if (v[c] > bv) { bv = v[c]; best = c; }
                }
                classes[(size_t)img * cap + rank] = best;
            }
        }
    }
}


// ------------------------------------------------------------------------------------ fused front end (LDS)
// Maps of at most PP_LDS_MAX_HW pixels (128 x 128: the 512 x 512 input of the headline configuration): init, merge,
// flatten, roots, owner, area, keep and extents of ONE image run in ONE block with the union-find forest, the
// foreground bits and the owner map in LDS -- one launch instead of nine, no global atomics on the forest.  Every phase
// is the corresponding kernel above restated on LDS arrays (same links, same external rule, same bit-quad area), so
// the results are identical; the phases are separated by block barriers instead of kernel boundaries.
#define PP_LDS_MAX_HW 16384
#define PP_LDS_THREADS 1024

// find with intermediate pointer jumping (as in ECL-CC): every node visited is re-pointed at its grandparent.  Parents
// only ever move to smaller ancestors of the same tree, so concurrent finds and hooks stay correct and the final roots
// (minimum node of each region) do not depend on the interleaving.
__device__ __forceinline__ int uf_find_wg(int *lab, int a)
{
    int curr = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (curr != a) {
        int prev = a, next;
        while (curr > (next = __hip_atomic_load(&lab[curr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
            __hip_atomic_store(&lab[prev], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = curr;
            curr = next;
        }
    }
    return curr;
}

// read-only find (no stores at all)
__device__ __forceinline__ int uf_find_ro_wg(const int *lab, int a)
{
    int p = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != a) {
        a = p;
        p = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return a;
}

__device__ __forceinline__ void uf_union_wg(int *lab, int a, int b)
{
    for (;;) {
        a = uf_find_wg(lab, a);
        b = uf_find_wg(lab, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        const int old = __hip_atomic_fetch_min(&lab[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == a) return;
        a = old;
    }
}

// LDS layout (bytes): label int32 [hw + 1] | owner int16 [hw] | rootslot int16 [hw] | fg uint8 [hw] | 2 counters.
// After the owner phase the label array is dead and is reused as area2 [root_cap] | kept [root_cap].
static size_t pp_front_lds_bytes(int hw, int root_cap)
{
    const size_t lab_ints = (size_t)hw + 1 > 2 * (size_t)root_cap ? (size_t)hw + 1 : 2 * (size_t)root_cap;
    return ubd_align_up(lab_ints * 4, 16) + (size_t)hw * 2 * 2 + ubd_align_up(hw, 16) + 16;
}

#ifdef UBD_STAMPS   // diagnostic build only (tools/build_diag.sh)
static unsigned long long *g_pp_stamps = nullptr;
extern "C" void ubd_debug_set_stamps_pp(void *p) { g_pp_stamps = (unsigned long long *)p; }
#endif
// TAIL: the block also fits its image's boxes (one wave per kept object, scratch in the dead parts of the forest / root-slot
// arrays), takes the class vote and emits the lists -- the whole postprocess of a batch is then ONE launch.  (Measured in
// round 3: in the two-stream pipeline every launch on the postprocess stream costs the forward stream ~5 us of dispatch
// time whatever the kernel does; tools/pipe_ablation2.py.)
template <bool TAIL>
__global__ __launch_bounds__(PP_LDS_THREADS) void pp_front_lds_kernel(
    const float *__restrict__ logits, int k_out, float thr, int h, int w, float min_area, int cap, int n_cls, int root_cap,
    int *__restrict__ binary_map, int *__restrict__ g_nroots, int *__restrict__ g_nkept, int *__restrict__ g_owner,
    int *__restrict__ g_roots, int *__restrict__ g_kept, int *__restrict__ stage, int *__restrict__ ymax,
    int *__restrict__ rows, float *__restrict__ vote, int poison, int scale, int *__restrict__ quads,
    int *__restrict__ classes, int *__restrict__ counts
#ifdef UBD_STAMPS
    , unsigned long long *__restrict__ stamps
#endif
    )
{
#ifdef UBD_STAMPS
    int stamp_k = 0;
#define PPSTAMP() do { if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 16 + stamp_k] = __builtin_amdgcn_s_memtime(); ++stamp_k; } while (0)
#else
#define PPSTAMP() do {} while (0)
#endif
    PPSTAMP();
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int hw = h * w;
    int *lab = smem;
    const int lab_ints = hw + 1 > 2 * root_cap ? hw + 1 : 2 * root_cap;
    short *own16 = (short *)((char *)smem + (((size_t)lab_ints * 4 + 15) & ~(size_t)15));
    short *rs16 = own16 + hw;
    unsigned char *m = (unsigned char *)(rs16 + hw);
    int *ctr = (int *)(m + ((hw + 15) & ~15));               // [0] roots, [1] kept, [2] integrity flag (UBD_PP_POISON)
    int *area2 = lab, *kept = lab + root_cap;                  // aliases, valid after the owner phase
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const size_t pbase = (size_t)img * hw;
    const int hw64 = (hw + 63) & ~63;
    // pixel -> (row, column) without an integer division by the run-time width (~25 vector instructions each, once per pixel
    // and phase on a kernel that is issue-bound at 16 waves per CU): floor((loc + 0.5) * (1 / w)) is exact for loc < hw <= 2^14
    // (the quotient's distance to the next integer is >= 0.5 / w, i.e. >= 2^-15 relative, against 2^-22 of rounding)
    const float rcp_w = 1.0f / (float)w;
    auto row_of = [&](int loc) { return (int)(((float)loc + 0.5f) * rcp_w); };

    // UBD_PP_POISON (tests): every LDS word the kernel uses starts as 0x7fff7fff instead of whatever the previous block left
    // there, so that a read of a never-written entry gives an impossible node / slot instead of a plausible small integer
    if (poison) {
        const int words = (int)(((char *)(ctr + 4) - (char *)smem) / 4);
        for (int i = tid; i < words; i += PP_LDS_THREADS) smem[i] = 0x7fff7fff;
        __syncthreads();
    }
    // ---- init (pp_init_kernel); the thread's logits are requested up front, PP_LDS_MAX_HW / PP_LDS_THREADS at most
    if (tid < 3) ctr[tid] = 0;                                 // [2]: integrity flag of the test mode
    constexpr int PER_THREAD = PP_LDS_MAX_HW / PP_LDS_THREADS;
    float lg[PER_THREAD], lgl[PER_THREAD];                     // lgl: logit left of the wave's first pixel (lane 0 only), requested with the
#pragma unroll                                               // rest: fetched inside the loop it was one dependent memory round trip per iteration
    for (int it = 0; it < PER_THREAD; ++it) {
        const int loc = tid + it * PP_LDS_THREADS;
        lg[it] = loc < hw ? logits[(pbase + loc) * k_out] : 0.f;
        lgl[it] = (lane == 0 && loc < hw && loc > 0) ? logits[(pbase + loc - 1) * k_out] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < PER_THREAD; ++it) {
        const int loc = tid + it * PP_LDS_THREADS;
        if (loc - lane >= hw64) break;                                              // wave-uniform
        const bool valid = loc < hw;
        int f = 0, x = 0;
        if (valid) {
            x = loc - row_of(loc) * w;
            f = lg[it] > thr ? 1 : 0;                                               // strict >, model_runner.py:124
            m[loc] = (unsigned char)f;
            if (binary_map) binary_map[pbase + loc] = f;
        }
        int fl = __shfl_up(f, 1, 64);
        if (lane == 0 && valid && x > 0) fl = lgl[it] > thr ? 1 : 0;
        const bool same_left = valid && x > 0 && fl == f;
        const unsigned long long breaks = __ballot(!same_left);
        if (valid) {
            const unsigned long long below = breaks & ((2ull << lane) - 1ull);
            const int start_off = below ? lane - (63 - __clzll(below)) : lane + 1;
            lab[loc + 1] = loc + 1 - start_off;
            if (loc == 0) lab[0] = 0;
        }
    }
    __syncthreads();
    PPSTAMP();

    // ---- merge (pp_merge_kernel).  Only a few lanes of a wave have a link to make at any pixel, and a union is a chain of
    // dependent LDS round trips: executed in place, every iteration of the pixel loop cost one union latency (~3 k cycles x 16
    // iterations = 50 k of the kernel's 122 k cycles).  Two passes instead: the wave first queues its links as 16-bit jobs
    // (pixel << 2 | direction; compacted with ballots into a wave-private slice of the still unused owner / root-slot arrays),
    // then runs them 64 at a time, one job per lane.  The forest and its roots do not depend on the order of the unions.
    {
        const int iters = (hw + PP_LDS_THREADS - 1) / PP_LDS_THREADS;
        const bool queued = (size_t)iters * 4096 <= (size_t)hw * 4;       // 2 jobs per pixel fit the slice (hw a multiple of 1024)
        unsigned short *queue = (unsigned short *)own16 + (size_t)(tid >> 6) * iters * 128;
        int njobs = 0;                                                    // wave-uniform
        for (int loc = tid; loc - lane < hw; loc += PP_LDS_THREADS) {     // wave-uniform trip count
            int ja = -1, jb = -1;                                         // direction of the first / second link: 0 N, 1 NW, 2 NE, 3 frame
            if (loc < hw) {
                const int y = row_of(loc), x = loc - y * w;
                const int c = m[loc];
                const bool W = x > 0 && m[loc - 1] == c;
                if (c) {
                    if (y > 0) {
                        const bool N = m[loc - w];
                        const bool NW = x > 0 && m[loc - w - 1];
                        if (N) {
                            if (!(W && NW)) ja = 0;
                        } else {
                            if (NW && !W) ja = 1;
                            const bool NE = x < w - 1 && m[loc - w + 1];
                            const bool E = x < w - 1 && m[loc + 1];
                            if (NE && !E) jb = 2;
                        }
                    }
                } else {
                    if (y > 0 && !m[loc - w]) {
                        const bool NW = x > 0 && !m[loc - w - 1];
                        if (!(W && NW)) ja = 0;
                    }
                    const bool row_edge = (y == 0 || y == h - 1) && !W;
                    if (row_edge || x == 0 || x == w - 1) jb = 3;
                }
            }
            if (queued) {
                const unsigned long long ba = __ballot(ja >= 0), bb = __ballot(jb >= 0);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (ja >= 0) queue[njobs + __popcll(ba & below)] = (unsigned short)((loc << 2) | ja);
                njobs += __popcll(ba);
                if (jb >= 0) queue[njobs + __popcll(bb & below)] = (unsigned short)((loc << 2) | jb);
                njobs += __popcll(bb);
            } else {
                const int me = loc + 1;
                if (ja >= 0) uf_union_wg(lab, me, ja == 0 ? me - w : me - w - 1);
                if (jb >= 0) uf_union_wg(lab, me, jb == 2 ? me - w + 1 : 0);
            }
        }
        if (queued)
            for (int j = lane; j < njobs; j += 64) {
                const int job = queue[j], me = (job >> 2) + 1, dir = job & 3;
                uf_union_wg(lab, me, dir == 0 ? me - w : (dir == 1 ? me - w - 1 : (dir == 2 ? me - w + 1 : 0)));
            }
    }
    __syncthreads();
    PPSTAMP();

    // ---- flatten (pp_flatten_kernel).  READ-ONLY find: the only stores of this phase are true roots into the thread's own
    // node.  (The compressing find of the merge phase must not be used here: its `lab[prev] = next` stores re-point OTHER
    // nodes at a grandparent read earlier and can land after the owner of that node has stored its final root, leaving a
    // stale non-root there -- the round-2 wrong-quad defect.)  Any value a concurrent reader sees in lab[x] is an ancestor
    // of x or its root, so the walks stay correct while the stores land.
    for (int node = tid; node <= hw; node += PP_LDS_THREADS) {
#ifdef UBD_PP_RACY_FLATTEN   // diagnostic build only (tools/prove_stress_power.sh): round 2's compressing find, to show that the stress test catches it
        const int r = uf_find_wg(lab, node);
#else
        const int r = uf_find_ro_wg(lab, node);
#endif
        __hip_atomic_store(&lab[node], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the owner / root-slot arrays were the union job queue until the barrier above: no entry is valid yet.  -1 = "no slot", so
    // a read of an entry this launch never wrote cannot return a plausible slot (stale queue jobs are small integers).
    {
        int *z = (int *)own16;                                   // own16 [hw] | rs16 [hw] = hw ints from a 16-byte-aligned base
        for (int i = tid; i < hw; i += PP_LDS_THREADS) z[i] = -1;
    }
    __syncthreads();
    PPSTAMP();

    // ---- roots (pp_roots_kernel)
    for (int loc = tid; loc < hw; loc += PP_LDS_THREADS) {
        if (!m[loc] || lab[loc + 1] != loc + 1) continue;
        const bool external = (loc < w) || (lab[loc + 1 - w] == 0);
        if (!external) continue;
        const int idx = atomicAdd(&ctr[0], 1);
        g_roots[(size_t)img * root_cap + idx] = loc;
        rs16[loc] = (short)idx;
    }
    __syncthreads();
    PPSTAMP();

    // ---- owner (pp_owner_kernel)
    for (int loc = tid; loc < hw; loc += PP_LDS_THREADS) {
        // the forest is flat (every entry is a root: checked by tests/test_gpu_postprocess.py under UBD_PP_POISON through the
        // slot of a non-root being -1), and a slot is read only at a root's own raster-first pixel
        int node = lab[loc + 1];
        int own = -1;
        for (int guard = 0; guard < 4096; ++guard) {
            if (poison && lab[node] != node) atomicOr(&ctr[2], 1);          // test mode: a non-root survived the flatten phase
            if (node == 0) break;
            const int r = node - 1;
            if (r < w) { if (m[r]) own = rs16[r]; break; }
            const int up = lab[r - w + 1];
            if (m[r] && up == 0) { own = rs16[r]; break; }
            node = up;
        }
        if (poison && (own < -1 || own >= ctr[0])) atomicOr(&ctr[2], 1);   // test mode: a slot that no root wrote
        own16[loc] = (short)own;
        if (!TAIL && g_owner) g_owner[pbase + loc] = own;
    }
    __syncthreads();
    PPSTAMP();
    const int nroots = ctr[0];
    for (int s = tid; s < nroots; s += PP_LDS_THREADS) area2[s] = 0;       // the forest is dead from here on
    __syncthreads();
    PPSTAMP();

    // ---- area (pp_area_kernel)
    for (int loc = tid; loc < hw; loc += PP_LDS_THREADS) {
        int key = -1, val = 0;
        if (loc < hw) {
            const int y = row_of(loc), x = loc - y * w;
            if (x < w - 1 && y < h - 1) {
                const int o0 = own16[loc], o1 = own16[loc + 1], o2 = own16[loc + w], o3 = own16[loc + w + 1];
                const int o = max(max(o0, o1), max(o2, o3));
                if (o >= 0) {
                    const int cnt = (o0 == o) + (o1 == o) + (o2 == o) + (o3 == o);
                    val = cnt == 4 ? 2 : (cnt == 3 ? 1 : 0);
                    key = o;
                }
            }
        }
        if (val != 0) atomicAdd(&area2[key], val);                                  // LDS atomic (a wave-level pre-reduction of equal keys measured slower: 14 k -> 20 k cycles)
    }
    __syncthreads();
    PPSTAMP();

    // ---- keep (pp_keep_kernel)
    for (int s = tid; s < nroots; s += PP_LDS_THREADS) {
        const double area = (double)area2[s] * 0.5;
        int k = -1;
        if (area > (double)min_area) {                                        // utils.py:55 (strict >)
            k = atomicAdd(&ctr[1], 1);
            if (k < cap) {
                int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
                st[0] = g_roots[(size_t)img * root_cap + s];
                ymax[(size_t)img * cap + k] = 0;
                if (n_cls > 0) {
                    float *v = vote + ((size_t)img * cap + k) * (n_cls + 1);
                    for (int c = 0; c <= n_cls; ++c) v[c] = 0.f;
                }
            } else {
                k = -1;
            }
        }
        kept[s] = k;
        if (!TAIL && g_kept) g_kept[(size_t)img * root_cap + s] = k;
    }
    __syncthreads();
    PPSTAMP();
    {
        const int nk = min(ctr[1], cap);                                          // row extents of the kept objects: (+inf, -1)
        int2 *r = (int2 *)(rows + (size_t)img * cap * (size_t)(6 * h));
        for (int e = tid; e < nk * h; e += PP_LDS_THREADS) {
            const int k = e / h, y = e - k * h;
            r[(size_t)k * (3 * h) + y] = make_int2(0x7fffffff, -1);
        }
    }
    __syncthreads();
    PPSTAMP();

    // ---- extents (pp_extents_kernel)
    for (int loc = tid; loc < hw; loc += PP_LDS_THREADS) {
        const int o = own16[loc];
        if (o < 0) continue;
        const int y = row_of(loc), x = loc - y * w;
        const bool left_end = (x == 0) || own16[loc - 1] != o;
        const bool right_end = (x == w - 1) || own16[loc + 1] != o;
        const bool bottom = (y == h - 1) || own16[loc + w] != o;
        if (!(left_end || right_end || bottom)) continue;
        const int k = kept[o];
        if (k < 0) continue;
        int *r = rows + ((size_t)img * cap + k) * (size_t)(6 * h);
        if (left_end) atomicMin(&r[2 * y], x);
        if (right_end) atomicMax(&r[2 * y + 1], x);
        if (bottom) atomicMax(&ymax[(size_t)img * cap + k], y);
    }
    if (tid == 0) { g_nroots[img] = nroots; g_nkept[img] = ctr[1] | (ctr[2] << 30); }   // test mode: an integrity failure shows as an impossible count
    PPSTAMP();
    if constexpr (TAIL) {
        // ---- boxes (pp_boxes_wave_kernel): the row extents were accumulated by this block's atomics at the L2 -- every wave
        // waits for its own (vmcnt) before the barrier, and the readers go past the vector L1 (pp_box_object<true>).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nk = min(ctr[1], cap);
        const int wid = tid >> 6;
        {
            // per-wave scratch of 12 h + 4 ints in what is dead by now: the forest behind area2 | kept, and the root-slot array
            const int S = 12 * h + 4;
            const int nA = (lab_ints - 2 * root_cap) / S, nB = (hw & 1) ? 0 : (hw / 2) / S;
            const int nwv = min(PP_LDS_THREADS / 64, nA + nB);                       // >= 1 (host)
            int *scratch = wid < nA ? lab + 2 * root_cap + wid * S : (int *)rs16 + (wid - nA) * S;
            if (wid < nwv)
                for (int k = wid; k < nk; k += nwv) {
                    int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
                    const int y0 = row_of(st[0]);
                    const int ym = __hip_atomic_load(&ymax[(size_t)img * cap + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int *g = rows + ((size_t)img * cap + k) * (size_t)(6 * h) + 2 * y0;
                    pp_box_object<true>(scratch, g, ym - y0 + 1, y0, h, lane, scale, st);
                }
        }
        // ---- vote (pp_vote_kernel): mean softmax over the filled region
        if (n_cls > 0) {
            for (int loc = tid; loc < hw; loc += PP_LDS_THREADS) {
                const int o = own16[loc];
                if (o < 0) continue;
                const int k = kept[o];
                if (k < 0) continue;
                const float *lg = logits + (pbase + loc) * k_out + 1;
                float mx = lg[0];
                for (int c = 1; c < n_cls; ++c) mx = fmaxf(mx, lg[c]);
                float sum = 0.f;
                for (int c = 0; c < n_cls; ++c) sum += expf(lg[c] - mx);
                float *v = vote + ((size_t)img * cap + k) * (n_cls + 1);
                for (int c = 0; c < n_cls; ++c) atomicAdd(&v[c], expf(lg[c] - mx) / sum);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        PPSTAMP();
        // ---- emit (pp_emit_kernel): objects ordered like cv2 returns them (last discovered first)
        if (tid == 0) counts[img] = ctr[1] | (ctr[2] << 30);
        const int *stg = stage + (size_t)img * cap * STAGE_INTS;
        for (int sidx = tid; sidx < nk; sidx += PP_LDS_THREADS) {
            const int root = stg[sidx * STAGE_INTS];
            int rank = 0;
            for (int t = 0; t < nk; ++t) rank += (stg[t * STAGE_INTS] > root) ? 1 : 0;
            int *qd = quads + ((size_t)img * cap + rank) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) qd[j] = stg[sidx * STAGE_INTS + 1 + j];
            if (classes) {
                int best = 0;
                if (n_cls > 0) {
                    const float *v = vote + ((size_t)img * cap + sidx) * (n_cls + 1);
                    float bv = __hip_atomic_load(&v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int c = 1; c < n_cls; ++c) {
                        const float vc = __hip_atomic_load(&v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (vc > bv) { bv = vc; best = c; }
                    }
                }
                classes[(size_t)img * cap + rank] = best;
            }
        }
        PPSTAMP();
    }
#undef PPSTAMP
}

// ------------------------------------------------------------------------------------ host
extern "C" int ubd_postprocess(ubd_handle *hd, const float *logits, int n, int map_h, int map_w,
                               float logit_threshold, int scale, float min_area, int32_t *binary_map,
                               int32_t *quads, int32_t *classes, int32_t *counts, int cap, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    UBD_REQUIRE(hd && logits && quads && counts && workspace, "ubd_postprocess: null argument");
    UBD_REQUIRE(n > 0 && map_h > 0 && map_w > 0 && cap > 0, "ubd_postprocess: bad sizes n=%d h=%d w=%d cap=%d", n, map_h, map_w, cap);
    UBD_REQUIRE(map_h < 32768 && map_w < 32768 && (long)n * map_h * map_w < (1L << 31), "ubd_postprocess: map too large");
    const int n_cls = hd->cfg.n_classes;
    UBD_REQUIRE(n_cls == 0 || classes, "ubd_postprocess: classes buffer required when n_classes > 0");
    pp_layout L;
    pp_layout_compute(n, map_h, map_w, cap, n_cls, &L);
    UBD_REQUIRE(workspace_bytes >= L.total, "ubd_postprocess: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *nroots = (int *)(ws + L.off_nroots), *nkept = (int *)(ws + L.off_nkept);
    int *label = (int *)(ws + L.off_label);
    unsigned char *fg = (unsigned char *)(ws + L.off_fg);
    int *owner = (int *)(ws + L.off_owner), *rootslot = (int *)(ws + L.off_rootslot);
    int *roots = (int *)(ws + L.off_roots), *area2 = (int *)(ws + L.off_area2), *kept = (int *)(ws + L.off_kept);
    int *stage = (int *)(ws + L.off_stage), *ymax = (int *)(ws + L.off_ymax), *rows = (int *)(ws + L.off_rows);
    float *vote = (float *)(ws + L.off_vote);
    const int hw = map_h * map_w;
    const long npix = (long)n * hw;
    int grid = (int)((npix + 255) / 256);
    const int gmax = hd->num_cus * 8;
    if (grid > gmax) grid = gmax;
    // test hooks, read per call (SegmapManager.postprocess keeps its handles): the multi-launch front end at any map size; LDS
    // poisoning + forest integrity check of the one-launch front end
    const bool force_global = getenv("UBD_PP_GLOBAL") != nullptr;
    const int pp_poison = getenv("UBD_PP_POISON") != nullptr;
    // one launch for the whole postprocess when a wave's box scratch (12 h + 4 ints) fits the dead part of the block's LDS
    bool fused_tail = false;
    if (hw <= PP_LDS_MAX_HW && !force_global) {
        if (!hd->pp_lds_attr_set) {                              // per handle = per device (the attribute belongs to the device's code object)
            const int lds_max = (int)pp_front_lds_bytes(PP_LDS_MAX_HW, PP_LDS_MAX_HW / 2 + 2);   // 1-pixel-high maps hold the most roots per pixel
            UBD_CHECK_HIP(hipFuncSetAttribute((const void *)pp_front_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
            UBD_CHECK_HIP(hipFuncSetAttribute((const void *)pp_front_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
            hd->pp_lds_attr_set = 1;
        }
        {
            const long lab_ints = (long)hw + 1 > 2L * L.root_cap ? (long)hw + 1 : 2L * L.root_cap, S = 12L * map_h + 4;
            fused_tail = (lab_ints - 2L * L.root_cap) / S + ((hw & 1) ? 0 : (hw / 2) / S) >= 1 && getenv("UBD_PP_SPLIT") == nullptr;   // UBD_PP_SPLIT: test hook, separate tail launches
        }
#ifdef UBD_STAMPS
#define PP_STAMP_ARG , g_pp_stamps
#else
#define PP_STAMP_ARG
#endif
        if (fused_tail)
            hipLaunchKernelGGL(pp_front_lds_kernel<true>, dim3(n), dim3(PP_LDS_THREADS), pp_front_lds_bytes(hw, L.root_cap), st, logits, hd->k_out, logit_threshold,
                               map_h, map_w, min_area, cap, n_cls, L.root_cap, binary_map, nroots, nkept, (int *)nullptr, roots,
                               (int *)nullptr, stage, ymax, rows, vote, pp_poison, scale, quads, classes, counts PP_STAMP_ARG);
        else
            hipLaunchKernelGGL(pp_front_lds_kernel<false>, dim3(n), dim3(PP_LDS_THREADS), pp_front_lds_bytes(hw, L.root_cap), st, logits, hd->k_out, logit_threshold,
                               map_h, map_w, min_area, cap, n_cls, L.root_cap, binary_map, nroots, nkept, n_cls > 0 ? owner : nullptr, roots,
                               n_cls > 0 ? kept : nullptr, stage, ymax, rows, vote, pp_poison, scale, quads, classes, counts PP_STAMP_ARG);
    } else {
    UBD_CHECK_HIP(hipMemsetAsync(ws, 0, L.off_label, st));          // the two per-image counters
    hipLaunchKernelGGL(pp_init_kernel, dim3(grid), dim3(256), 0, st, logits, hd->k_out, logit_threshold, npix, hw, map_w, fg, label, binary_map);
    hipLaunchKernelGGL(pp_merge_kernel, dim3(grid), dim3(256), 0, st, fg, label, npix, map_h, map_w);
    hipLaunchKernelGGL(pp_flatten_kernel, dim3(grid), dim3(256), 0, st, label, n, hw);
    hipLaunchKernelGGL(pp_roots_kernel, dim3(grid), dim3(256), 0, st, fg, label, npix, map_h, map_w, nroots, roots, rootslot, area2, L.root_cap);
    hipLaunchKernelGGL(pp_owner_kernel, dim3(grid), dim3(256), 0, st, fg, label, rootslot, npix, map_h, map_w, owner);
    hipLaunchKernelGGL(pp_area_kernel, dim3(grid), dim3(256), 0, st, owner, npix, map_h, map_w, area2, L.root_cap);
    hipLaunchKernelGGL(pp_keep_kernel, dim3(n), dim3(256), 0, st, n, map_h, nroots, roots, area2, L.root_cap, min_area, nkept, kept, stage, ymax, rows, cap, vote, n_cls);
    hipLaunchKernelGGL(pp_extents_kernel, dim3(grid), dim3(256), 0, st, owner, kept, npix, map_h, map_w, L.root_cap, cap, rows, ymax);
    }
    if (!fused_tail) {
        // one wave per kept object; 8 waves per image block, fewer while their LDS does not fit
        int waves = 8;
        size_t lds = (size_t)waves * (12 * map_h + 4) * sizeof(int);
        while (lds > 64 * 1024 && waves > 1) { waves /= 2; lds /= 2; }
        if (lds <= 64 * 1024) {
            hipLaunchKernelGGL(pp_boxes_wave_kernel, dim3(n), dim3(64 * waves), lds, st, n, map_h, map_w, nkept, stage, ymax, rows, cap, scale);
        } else {                                       // very tall maps: serial per-object fallback in global memory
            const long total = (long)n * cap;
            int bgrid = (int)((total + 63) / 64);
            if (bgrid > gmax) bgrid = gmax;
            hipLaunchKernelGGL(pp_boxes_kernel, dim3(bgrid), dim3(64), 0, st, n, map_h, map_w, nkept, stage, ymax, rows, cap, scale);
        }
    }
    if (!fused_tail) {
        if (n_cls > 0)
            hipLaunchKernelGGL(pp_vote_kernel, dim3(grid), dim3(256), 0, st, logits, hd->k_out, owner, kept, npix, hw, L.root_cap, cap, vote);
        hipLaunchKernelGGL(pp_emit_kernel, dim3(n), dim3(256), 0, st, n, nkept, stage, vote, n_cls, cap, quads, classes, counts);
    }
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}
