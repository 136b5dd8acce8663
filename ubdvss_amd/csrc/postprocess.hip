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
#include "pp_lds.h"

#pragma clang fp contract(off)


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

__global__ __launch_bounds__(512) void pp_boxes_wave_kernel(int n, int h, int w, const int *__restrict__ nkept,
                                                            int *__restrict__ stage, const int *__restrict__ ymax,
                                                            const int *__restrict__ rows_ws, int cap, int scale, int serial_tail)
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
            pp_box_object<false>(rws, g, nrows, y0, h, lane, scale, st, serial_tail != 0);
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


// ------------------------------------------------------------------------------------ one-launch front end (pp_lds.h)
#ifdef UBD_STAMPS   // diagnostic build only (tools/build_diag.sh)
static unsigned long long *g_pp_stamps = nullptr;
extern "C" void ubd_debug_set_stamps_pp(void *p) { g_pp_stamps = (unsigned long long *)p; }
#endif
template <bool TAIL, int NT = PP_LDS_THREADS>
__global__ __launch_bounds__(NT) void pp_front_lds_kernel(pp_lds_args a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    pp_image_lds<NT, TAIL>(smem, a, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------ host
// Does the in-block tail (boxes + vote + emit) fit?  A wave's box scratch is 12 h + 4 ints in what is dead of the block's LDS.
static bool pp_tail_fits(int map_h, int map_w, int root_cap)
{
    const long hw = (long)map_h * map_w;
    const long lab_ints = hw + 1 > 2L * root_cap ? hw + 1 : 2L * root_cap, S = 12L * map_h + 4;
    return (lab_ints - 2L * root_cap) / S + ((hw & 1) ? 0 : (hw / 2) / S) >= 1;
}

int ubd_pp_fill_job(ubd_handle *hd, const float *logits, int n, int map_h, int map_w, float logit_threshold, int scale, float min_area,
                    int32_t *binary_map, int32_t *quads, int32_t *classes, int32_t *counts, int cap, void *workspace,
                    size_t workspace_bytes, int threads, pp_lds_args *a)
{
    if (!(hd && logits && quads && counts && workspace)) { ubd_set_error("ubd_postprocess: null argument"); return -1; }
    if (!(n > 0 && map_h > 0 && map_w > 0 && cap > 0)) { ubd_set_error("ubd_postprocess: bad sizes n=%d h=%d w=%d cap=%d", n, map_h, map_w, cap); return -1; }
    if (!(map_h < 32768 && map_w < 32768 && (long)n * map_h * map_w < (1L << 31))) { ubd_set_error("ubd_postprocess: map too large"); return -1; }
    const int n_cls = hd->cfg.n_classes;
    if (!(n_cls == 0 || classes)) { ubd_set_error("ubd_postprocess: classes buffer required when n_classes > 0"); return -1; }
    pp_layout L;
    pp_layout_compute(n, map_h, map_w, cap, n_cls, &L);
    if (workspace_bytes < L.total) { ubd_set_error("ubd_postprocess: workspace too small (%zu < %zu)", workspace_bytes, L.total); return -1; }
    const long hw = (long)map_h * map_w;
    if (hw > PP_LDS_MAX_HW || (PP_LDS_MAX_HW % threads) != 0 || hd->pp_global || hd->pp_split || !pp_tail_fits(map_h, map_w, L.root_cap)) return 0;
    char *ws = (char *)workspace;
    a->logits = logits; a->n = n; a->k_out = hd->k_out; a->h = map_h; a->w = map_w; a->cap = cap; a->n_cls = n_cls; a->root_cap = L.root_cap;
    a->poison = hd->pp_poison; a->serial_tail = hd->pp_serial_tail; a->scale = scale; a->thr = logit_threshold; a->min_area = min_area;
    a->binary_map = binary_map; a->g_nroots = (int *)(ws + L.off_nroots); a->g_nkept = (int *)(ws + L.off_nkept);
    a->g_roots = (int *)(ws + L.off_roots); a->stage = (int *)(ws + L.off_stage); a->ymax = (int *)(ws + L.off_ymax); a->rows = (int *)(ws + L.off_rows);
    a->vote = (float *)(ws + L.off_vote); a->quads = quads; a->classes = classes; a->counts = counts;
    a->g_owner = nullptr; a->g_kept = nullptr; a->stamps = nullptr;
    return 1;
}

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
    // test hooks (read once, in ubd_create): the multi-launch front end at any map size; LDS poisoning + forest integrity check
    const bool force_global = hd->pp_global != 0;
    const int pp_poison = hd->pp_poison;
    // one launch for the whole postprocess when a wave's box scratch (12 h + 4 ints) fits the dead part of the block's LDS
    bool fused_tail = false;
    if (hw <= PP_LDS_MAX_HW && !force_global) {
        if (!hd->pp_lds_attr_set) {                              // per handle = per device (the attribute belongs to the device's code object)
            const int lds_max = (int)pp_front_lds_bytes(PP_LDS_MAX_HW, PP_LDS_MAX_HW / 2 + 2);   // 1-pixel-high maps hold the most roots per pixel
            UBD_CHECK_HIP(hipFuncSetAttribute((const void *)pp_front_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
            UBD_CHECK_HIP(hipFuncSetAttribute((const void *)pp_front_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
            hd->pp_lds_attr_set = 1;
        }
        fused_tail = pp_tail_fits(map_h, map_w, L.root_cap) && !hd->pp_split;   // UBD_PP_SPLIT: test hook, separate tail launches
        pp_lds_args a;
        a.logits = logits; a.n = n; a.k_out = hd->k_out; a.h = map_h; a.w = map_w; a.cap = cap; a.n_cls = n_cls; a.root_cap = L.root_cap;
        a.poison = pp_poison; a.serial_tail = hd->pp_serial_tail; a.scale = scale; a.thr = logit_threshold; a.min_area = min_area;
        a.binary_map = binary_map; a.g_nroots = nroots; a.g_nkept = nkept; a.g_roots = roots; a.stage = stage; a.ymax = ymax; a.rows = rows;
        a.vote = vote; a.quads = quads; a.classes = classes; a.counts = counts;
        a.g_owner = (!fused_tail && n_cls > 0) ? owner : nullptr;
        a.g_kept = (!fused_tail && n_cls > 0) ? kept : nullptr;
#ifdef UBD_STAMPS
        a.stamps = g_pp_stamps;
#else
        a.stamps = nullptr;
#endif
        if (fused_tail && hd->pp_threads_512) {        // diagnostics: the block shape the job has inside the stem kernel (512 threads)
            static bool attr512 = false;
            if (!attr512) { UBD_CHECK_HIP(hipFuncSetAttribute((const void *)pp_front_lds_kernel<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS_MAX_BYTES)); attr512 = true; }
            hipLaunchKernelGGL((pp_front_lds_kernel<true, 512>), dim3(n), dim3(512), pp_front_lds_bytes(hw, L.root_cap), st, a);
        } else if (fused_tail)
            hipLaunchKernelGGL(pp_front_lds_kernel<true>, dim3(n), dim3(PP_LDS_THREADS), pp_front_lds_bytes(hw, L.root_cap), st, a);
        else
            hipLaunchKernelGGL(pp_front_lds_kernel<false>, dim3(n), dim3(PP_LDS_THREADS), pp_front_lds_bytes(hw, L.root_cap), st, a);
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
            hipLaunchKernelGGL(pp_boxes_wave_kernel, dim3(n), dim3(64 * waves), lds, st, n, map_h, map_w, nkept, stage, ymax, rows, cap, scale, hd->pp_serial_tail != 0);
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
