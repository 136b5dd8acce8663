// Device postprocess of the ubdvss hot path on gfx950: logits -> binary map -> external
// 8-connected components -> contourArea filter -> minAreaRect -> boxPoints -> rounded quads
// (+ optional per-object class vote), all without leaving the GPU.
//
// Reference call sites: semantic_segmentation/model_runner.py:121-134,
// segmap_manager.py:41-69, utils.py:51-60, :135-138 (cv2.findContours RETR_EXTERNAL /
// CHAIN_APPROX_SIMPLE, contourArea, minAreaRect, boxPoints, drawContours fill).
// OpenCV 3.4 semantics are restated from its published algorithms (see oracle/cv_post.c
// for the sequential restatement this file is tested against; parity vs cv2 itself is
// unpinned because OpenCV is not available in the build environment).
//
// Parallel formulation:
//   1. threshold_init : fg = logit0 > thr; one union-find node per pixel (+ node 0 = the
//                       image frame / outside).
//   2. ccl_merge      : lock-free union-find (atomicMin, min-index roots): foreground
//                       8-connected, background 4-connected, border background pixels are
//                       merged with node 0.  ccl_flatten: label = root.
//   3. find_roots     : a foreground root (raster-first pixel of its component) is an
//                       *external* contour start iff the pixel north of it is outside the
//                       image or belongs to the outside background (root 0).  This is
//                       exactly cv2's RETR_EXTERNAL rule (components nested in holes are
//                       dropped); verified against the sequential scan in tests.
//   4. trace_boxes    : one lane per external component follows its outer border with the
//                       Suzuki-Abe / icvFetchContour stepping rule, accumulating the
//                       shoelace sum (contourArea) exactly in int64 and the per-row x
//                       extents; kept components get their convex hull from the row extents
//                       (monotone chains), ordered like cv::convexHull(clockwise=true), then
//                       rotatingCalipers / minAreaRect / boxPoints in the same float32 /
//                       float64 operation order as OpenCV, np.round (half-to-even) * scale.
//   5. class_vote     : (n_classes > 0) per pixel, the enclosing external component is found
//                       through the hole/nesting parent chain; softmax probabilities are
//                       accumulated per object; argmax of the mean.
//   6. emit           : objects ordered like cv2 returns them (last discovered first).
#include "common.h"

#pragma clang fp contract(off)

#define CV_PI 3.1415926535897932384626433832795

struct pp_layout {
    size_t off_label;    // int32 [n][hw+1]
    size_t off_fg;       // uint8 [n][hw]
    size_t off_nroots;   // int32 [n]
    size_t off_nkept;    // int32 [n]
    size_t off_roots;    // int32 [n][root_cap]
    size_t off_stage;    // int32 [n][cap][STAGE_INTS]
    size_t off_rows;     // int32 [n][cap][6*h]: row extents (2h) + hull points (2h points)
    size_t off_vote;     // float [n][cap][n_cls+1]
    size_t total;
    int root_cap;
};
#define STAGE_INTS 10     // root, quad[8], class

static void pp_layout_compute(int n, int h, int w, int cap, int n_cls, pp_layout *L)
{
    const size_t hw = (size_t)h * w;
    size_t off = 0;
    L->root_cap = (int)(hw / 4 + 2);
    L->off_nroots = off; off += ubd_align_up(sizeof(int) * n, 256);
    L->off_nkept = off;  off += ubd_align_up(sizeof(int) * n, 256);
    L->off_label = off;  off += ubd_align_up(sizeof(int) * n * (hw + 1), 256);
    L->off_fg = off;     off += ubd_align_up(n * hw, 256);
    L->off_roots = off;  off += ubd_align_up(sizeof(int) * (size_t)n * L->root_cap, 256);
    L->off_stage = off;  off += ubd_align_up(sizeof(int) * (size_t)n * cap * STAGE_INTS, 256);
    L->off_rows = off;   off += ubd_align_up(sizeof(int) * (size_t)n * cap * 6 * h, 256);
    L->off_vote = off;   off += ubd_align_up(sizeof(float) * (size_t)n * cap * (n_cls + 1), 256);
    L->total = off;
}

extern "C" size_t ubd_postprocess_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w, int cap)
{
    pp_layout L;
    pp_layout_compute(n, map_h, map_w, cap, h ? h->cfg.n_classes : 0, &L);
    return L.total;
}

// ------------------------------------------------------------------------------------ 1
__global__ void threshold_init_kernel(const float *__restrict__ logits, int k_out, float thr, long npix, int hw,
                                      unsigned char *__restrict__ fg, int *__restrict__ label,
                                      int *__restrict__ binary_map)
{
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const int f = logits[p * k_out] > thr ? 1 : 0;    // strict >, model_runner.py:124
        fg[p] = (unsigned char)f;
        if (binary_map) binary_map[p] = f;
        int *lab = label + (size_t)img * (hw + 1);
        lab[loc + 1] = loc + 1;
        if (loc == 0) lab[0] = 0;
    }
}

// ------------------------------------------------------------------------------------ 2
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

__global__ void ccl_merge_kernel(const unsigned char *__restrict__ fg, int *__restrict__ label, long npix, int h, int w)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const int y = loc / w, x = loc % w;
        const unsigned char *m = fg + (size_t)img * hw;
        int *lab = label + (size_t)img * (hw + 1);
        const int me = loc + 1;
        if (m[loc]) {
            if (x > 0 && m[loc - 1]) uf_union(lab, me, me - 1);
            if (y > 0) {
                if (m[loc - w]) uf_union(lab, me, me - w);
                if (x > 0 && m[loc - w - 1]) uf_union(lab, me, me - w - 1);
                if (x < w - 1 && m[loc - w + 1]) uf_union(lab, me, me - w + 1);
            }
        } else {
            if (x == 0 || y == 0 || x == w - 1 || y == h - 1) uf_union(lab, me, 0);
            if (x > 0 && !m[loc - 1]) uf_union(lab, me, me - 1);
            if (y > 0 && !m[loc - w]) uf_union(lab, me, me - w);
        }
    }
}

__global__ void ccl_flatten_img_kernel(int *__restrict__ label, int n, int hw)
{
    const long total = (long)n * (hw + 1);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / (hw + 1)), node = (int)(p % (hw + 1));
        int *lab = label + (size_t)img * (hw + 1);
        const int r = uf_find(lab, node);
        __hip_atomic_store(&lab[node], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------ 3
__global__ void find_roots_kernel(const unsigned char *__restrict__ fg, const int *__restrict__ label, long npix,
                                  int h, int w, int *__restrict__ nroots, int *__restrict__ roots, int root_cap)
{
    const int hw = h * w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        if (!fg[p]) continue;
        const int *lab = label + (size_t)img * (hw + 1);
        if (lab[loc + 1] != loc + 1) continue;             // not the raster-first pixel
        const int y = loc / w;
        const bool external = (y == 0) || (lab[loc + 1 - w] == 0);
        if (!external) continue;
        const int idx = atomicAdd(&nroots[img], 1);
        if (idx < root_cap) roots[(size_t)img * root_cap + idx] = loc;
    }
}

// ------------------------------------------------------------------------------------ 4
__constant__ int c_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__constant__ int c_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct trace_result {
    long long a2;            // signed shoelace sum (2 * area)
    int xmin, xmax, ymin, ymax;
};

// Follows the outer border starting at the raster-first pixel (x0,y0) with the stepping rule of
// OpenCV's icvFetchContour (outer border: first search clockwise from W, then counter-clockwise
// sweeps from the direction of the previous pixel).  rows != nullptr: also records per-row
// x extents into rows[2*(y-ymin0)] (min) / rows[2*(y-ymin0)+1] (max).
template <typename MapT>
__device__ void trace_border(const MapT &map, int h, int w, int x0, int y0, trace_result &res, int *rows)
{
    auto pix = [&](int x, int y) -> int { return (x >= 0 && x < w && y >= 0 && y < h) ? (int)map[y * w + x] : 0; };
    res.a2 = 0;
    res.xmin = res.xmax = x0;
    res.ymin = res.ymax = y0;
    if (rows) { rows[0] = x0; rows[1] = x0; }
    int s = 4, s_end = 4;
    int nbx = 0, nby = 0;
    do {
        s = (s - 1) & 7;
        nbx = x0 + c_dx[s]; nby = y0 + c_dy[s];
    } while (pix(nbx, nby) == 0 && s != s_end);
    if (s == s_end) return;                               // single pixel
    const int i1x = nbx, i1y = nby;
    int cx = x0, cy = y0;
    const long max_steps = 8L * h * w + 16;
    for (long step = 0; step < max_steps; ++step) {
        int nx = cx, ny = cy;
        s_end = s;
        while (s < 15) {
            ++s;
            nx = cx + c_dx[s & 7]; ny = cy + c_dy[s & 7];
            if (pix(nx, ny) != 0) break;
        }
        s &= 7;
        res.a2 += (long long)cx * ny - (long long)cy * nx;
        if (nx == x0 && ny == y0 && cx == i1x && cy == i1y) break;
        cx = nx; cy = ny;
        res.xmin = min(res.xmin, cx); res.xmax = max(res.xmax, cx);
        res.ymax = max(res.ymax, cy);
        if (rows) {
            int *r = rows + 2 * (cy - y0);
            r[0] = min(r[0], cx); r[1] = max(r[1], cx);
        }
        s = (s + 4) & 7;
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
__device__ void min_area_box(const ipt *hp, int n, float *box8)
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

__global__ __launch_bounds__(256) void trace_boxes_kernel(const unsigned char *__restrict__ fg, int n, int h, int w,
                                                          const int *__restrict__ nroots, const int *__restrict__ roots,
                                                          int root_cap, int *__restrict__ nkept, int *__restrict__ stage,
                                                          int *__restrict__ rows_ws, int cap, int scale, float min_area)
{
    // one block per image (grid-stride over images); lanes take external roots
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        const unsigned char *m = fg + (size_t)img * h * w;
        const int nr = min(nroots[img], root_cap);
        for (int ri = threadIdx.x; ri < nr; ri += blockDim.x) {
            const int loc = roots[(size_t)img * root_cap + ri];
            const int y0 = loc / w, x0 = loc % w;
            trace_result tr;
            trace_border(m, h, w, x0, y0, tr, nullptr);
            const long long a2 = tr.a2 < 0 ? -tr.a2 : tr.a2;
            const double area = (double)a2 * 0.5;
            if (!(area > (double)min_area)) continue;               // utils.py:55 (strict >)
            const int slot = atomicAdd(&nkept[img], 1);
            if (slot >= cap) continue;                              // reported through counts[] > cap
            const int nrows = tr.ymax - y0 + 1;
            // rows scratch of this slot: 2*h ints for extents + the hull is built in a second area
            int *rows = rows_ws + ((size_t)img * cap + slot) * (size_t)(6 * h);
            for (int r = 0; r < nrows; ++r) { rows[2 * r] = 0x7fffffff; rows[2 * r + 1] = -0x7fffffff; }
            trace_result tr2;
            trace_border(m, h, w, x0, y0, tr2, rows);
            ipt *pts = (ipt *)(rows + 2 * h);                       // 4h ints = room for 2h points
            const int nh = hull_from_rows(rows, nrows, y0, pts);
            float box[8];
            min_area_box(pts, nh, box);
            int *st = stage + ((size_t)img * cap + slot) * STAGE_INTS;
            st[0] = loc;
#pragma unroll
            for (int j = 0; j < 8; ++j) st[1 + j] = (int)rintf(box[j] * (float)scale);   // np.round: half to even
            st[9] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------ 5
// Class vote (segmap_manager.py:59-67): mean over the filled contour of softmax(class logits),
// argmax.  The filled contour of an external component = every pixel it encloses; the owner of a
// pixel is found by walking the nesting chain: fg component -> background region north of its
// raster-first pixel -> fg component north of that region's raster-first pixel -> ... until a
// component whose enclosing background is the outside (root 0).
__device__ int owner_root(const unsigned char *m, const int *lab, int w, int loc)
{
    int node = lab[loc + 1];
    for (int guard = 0; guard < 64; ++guard) {
        if (node == 0) return -1;                 // outside background: no owner
        const int p = node - 1;                   // raster-first pixel of this region
        if (p < w) {                              // first row: nothing above
            return m[p] ? p : -1;
        }
        const int up = lab[p - w + 1];            // region north of the raster-first pixel
        if (m[p] && up == 0) return p;            // external foreground component
        node = up;
    }
    return -1;
}

__global__ void class_vote_kernel(const float *__restrict__ logits, int k_out, const unsigned char *__restrict__ fg,
                                  const int *__restrict__ label, long npix, int h, int w,
                                  const int *__restrict__ nkept, const int *__restrict__ stage, int cap,
                                  float *__restrict__ vote)
{
    const int hw = h * w;
    const int n_cls = k_out - 1;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / hw), loc = (int)(p % hw);
        const int nk = min(nkept[img], cap);
        if (nk == 0) continue;
        const int own = owner_root(fg + (size_t)img * hw, label + (size_t)img * (hw + 1), w, loc);
        if (own < 0) continue;
        // find the slot of this owner (few kept objects per image)
        int slot = -1;
        const int *st = stage + (size_t)img * cap * STAGE_INTS;
        for (int s = 0; s < nk; ++s)
            if (st[s * STAGE_INTS] == own) { slot = s; break; }
        if (slot < 0) continue;                   // owner was filtered out by the area test
        const float *lg = logits + p * k_out + 1;
        float mx = lg[0];
        for (int c = 1; c < n_cls; ++c) mx = fmaxf(mx, lg[c]);
        float sum = 0.f;
        for (int c = 0; c < n_cls; ++c) sum += expf(lg[c] - mx);
        float *v = vote + ((size_t)img * cap + slot) * (n_cls + 1);
        for (int c = 0; c < n_cls; ++c) atomicAdd(&v[c], expf(lg[c] - mx) / sum);
    }
}

// ------------------------------------------------------------------------------------ 6
__global__ void emit_kernel(int n, const int *__restrict__ nkept, const int *__restrict__ stage,
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

// ------------------------------------------------------------------------------------ host
extern "C" int ubd_postprocess(ubd_handle *hd, const float *logits, int n, int map_h, int map_w,
                               float logit_threshold, int scale, float min_area, int32_t *binary_map,
                               int32_t *quads, int32_t *classes, int32_t *counts, int cap, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    UBD_REQUIRE(hd && logits && quads && counts && workspace, "ubd_postprocess: null argument");
    UBD_REQUIRE(n > 0 && map_h > 0 && map_w > 0 && cap > 0, "ubd_postprocess: bad sizes n=%d h=%d w=%d cap=%d", n, map_h, map_w, cap);
    UBD_REQUIRE((long)map_h * map_w < (1L << 30), "ubd_postprocess: map too large");
    const int n_cls = hd->cfg.n_classes;
    UBD_REQUIRE(n_cls == 0 || classes, "ubd_postprocess: classes buffer required when n_classes > 0");
    pp_layout L;
    pp_layout_compute(n, map_h, map_w, cap, n_cls, &L);
    UBD_REQUIRE(workspace_bytes >= ubd_postprocess_workspace_bytes(hd, n, map_h, map_w, cap), "ubd_postprocess: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *label = (int *)(ws + L.off_label);
    unsigned char *fg = (unsigned char *)(ws + L.off_fg);
    int *nroots = (int *)(ws + L.off_nroots), *nkept = (int *)(ws + L.off_nkept);
    int *roots = (int *)(ws + L.off_roots), *stage = (int *)(ws + L.off_stage), *rows = (int *)(ws + L.off_rows);
    float *vote = (float *)(ws + L.off_vote);
    const int hw = map_h * map_w;
    const long npix = (long)n * hw;
    // counters are contiguous at the start of the workspace: one memset node
    UBD_CHECK_HIP(hipMemsetAsync(ws, 0, L.off_label, st));
    if (n_cls > 0) UBD_CHECK_HIP(hipMemsetAsync(vote, 0, sizeof(float) * (size_t)n * cap * (n_cls + 1), st));
    int grid = (int)((npix + 255) / 256);
    const int gmax = hd->num_cus * 8;
    if (grid > gmax) grid = gmax;
    hipLaunchKernelGGL(threshold_init_kernel, dim3(grid), dim3(256), 0, st, logits, hd->k_out, logit_threshold, npix, hw, fg, label, binary_map);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(grid), dim3(256), 0, st, fg, label, npix, map_h, map_w);
    hipLaunchKernelGGL(ccl_flatten_img_kernel, dim3(grid), dim3(256), 0, st, label, n, hw);
    hipLaunchKernelGGL(find_roots_kernel, dim3(grid), dim3(256), 0, st, fg, label, npix, map_h, map_w, nroots, roots, L.root_cap);
    hipLaunchKernelGGL(trace_boxes_kernel, dim3(n), dim3(256), 0, st, fg, n, map_h, map_w, nroots, roots, L.root_cap, nkept, stage, rows, cap, scale, min_area);
    if (n_cls > 0)
        hipLaunchKernelGGL(class_vote_kernel, dim3(grid), dim3(256), 0, st, logits, hd->k_out, fg, label, npix, map_h, map_w, nkept, stage, cap, vote);
    hipLaunchKernelGGL(emit_kernel, dim3(n), dim3(256), 0, st, n, nkept, stage, vote, n_cls, cap, quads, classes, counts);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}
