// Training label maps on device: object quads -> 1/scale-resolution polygon fill.
//
// Reference: SegmapManager.build_segmentation_map (semantic_segmentation/segmap_manager.py:81-104) divides every quad by
// `scale`, snaps the corners outward (_proper_round, :106-133) and fills the polygon with PIL's ImageDraw.polygon, later
// objects over earlier ones, value = class id + 1 (or 1).  The fill rule is Pillow's (libImaging/Draw.c polygon_generic;
// third-party, not in the reference tree), restated from its published behaviour and pinned against the installed Pillow by
// fuzzing (oracle/label_raster.py, tests/test_oracle_raster.py): per scanline the x intersections of the non-horizontal
// edges in float32 (x0 + (y - y0) * dx, product and sum rounded separately), an edge's lower end point counted twice, spans
// [round-half-up(left), round-half-down(right)], horizontal edges drawn as they are, and the single pixel of a top / bottom
// corner -- any two edges leaning to the same side that start (last row: end) in one point -- joined to the span of the
// neighbouring row.  Bit-identical to Pillow 12.2 on convex, concave and self-intersecting quadrilaterals alike (0 differences
// on 50 000 arbitrary quads); the one exception, a quad whose opposite corners coincide (four edges in one point), is drawn with the same rule and logged by the host mirror
// (about 4 % of random such quads then differ from Pillow inside one row; strict_markup refuses them).
// Markup is float64 (rescaled / augmented quads are fractional): the division by the scale and _proper_round's comparisons
// and floor / ceil run in double precision exactly as numpy / math do in the reference.
//
// One thread per map pixel; objects are tested in order and the last one that covers the pixel wins (painter's order).
#include "common.h"

struct rq_edge { int x0, y0, x1, y1, xmin, xmax, ymin, ymax; float dx; bool horiz; };

__device__ __forceinline__ int rq_round_up(float f) { return f >= 0.f ? (int)floorf(__fadd_rn(f, 0.5f)) : -(int)floorf(__fadd_rn(fabsf(f), 0.5f)); }
__device__ __forceinline__ int rq_round_down(float f) { return f >= 0.f ? (int)ceilf(__fsub_rn(f, 0.5f)) : -(int)ceilf(__fsub_rn(fabsf(f), 0.5f)); }
__device__ __forceinline__ float rq_x_at(const rq_edge &e, int y) { return __fadd_rn(__fmul_rn((float)(y - e.y0), e.dx), (float)e.x0); }

// segmap_manager.py:96 + :106-133 on float64 markup: bbox / scale (IEEE double division, what numpy does for the reference),
// then floor a coordinate when at least two of the four coordinates on the same axis are strictly larger, else ceil
__device__ void rq_proper_round(const double *bbox, int scale, int *out)
{
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __ddiv_rn(bbox[k], (double)scale);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int larger = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) larger += v[2 * j + (k & 1)] > v[k] ? 1 : 0;
        out[k] = (int)(larger > 1 ? floor(v[k]) : ceil(v[k]));
    }
}

// Is pixel (px, py) inside ImageDraw.polygon(pts) on a canvas of map_h rows?  One scan line of Pillow's polygon fill
// (oracle/label_raster.py fill_polygon is the sequential statement of the same rule).
__device__ bool rq_covers(const int *pts, int px, int py, int map_h)
{
    rq_edge e[4];
    int ymin = map_h - 1, ymax = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rq_edge &d = e[i];
        d.x0 = pts[2 * i]; d.y0 = pts[2 * i + 1]; d.x1 = pts[(2 * i + 2) & 7]; d.y1 = pts[(2 * i + 3) & 7];
        d.xmin = min(d.x0, d.x1); d.xmax = max(d.x0, d.x1); d.ymin = min(d.y0, d.y1); d.ymax = max(d.y0, d.y1);
        d.horiz = d.y0 == d.y1;
        d.dx = d.horiz ? 0.f : __fdiv_rn((float)(d.x1 - d.x0), (float)(d.y1 - d.y0));
        ymin = min(ymin, d.ymin); ymax = max(ymax, d.ymax);
        if (d.horiz && py == d.y0 && px >= d.xmin && px <= d.xmax) return true;       // horizontal edges are drawn as they are
    }
    ymin = max(ymin, 0); ymax = min(ymax, map_h);
    if (py < ymin || py > ymax) return false;
    // intersections of this scan line in edge order; an edge's lower end point counts twice (except in the last row)
    float xx[8];
    int last[4], act[4], na = 0, nx = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (e[i].horiz || py < e[i].ymin || py > e[i].ymax) continue;
        const float x = rq_x_at(e[i], py);
        xx[nx++] = x;
        if (py == e[i].ymax && py < ymax) xx[nx++] = x;
        act[na] = i; last[na] = nx - 1; ++na;
    }
    // "connect discontiguous corners": two edges leaning to the same side that both start in one point of this row (in the
    // last row: both end there) -- the later edge's intersection moves towards the span of the neighbouring row
    for (int bi = 1; bi < na; ++bi) {
        const rq_edge &b = e[act[bi]];
        if (b.dx == 0.f) continue;
        const int bex = b.y0 == py ? b.x0 : b.x1, bey = b.y0 == py ? b.y0 : b.y1;
        for (int ai = 0; ai < bi; ++ai) {
            const rq_edge &a = e[act[ai]];
            if ((b.dx > 0.f && a.dx <= 0.f) || (b.dx < 0.f && a.dx >= 0.f)) continue;
            const bool top = a.ymin == py && b.ymin == py && py < ymax;
            const bool bottom = a.ymax == py && b.ymax == py && py == ymax;
            if (top == bottom) continue;
            const int aex = a.y0 == py ? a.x0 : a.x1, aey = a.y0 == py ? a.y0 : a.y1;
            if (aex != bex || aey != bey) continue;
            const float v = (float)aex;
            const int ya = top ? py + 1 : py - 1;
            const float xa = rq_x_at(a, ya), xb = rq_x_at(b, ya);
            const float lo = fminf(xa, xb), hi = fmaxf(xa, xb);
            if (lo > v) xx[last[bi]] = fmaxf(v, (float)(rq_round_up(lo) - 1));
            else if (hi < v) xx[last[bi]] = fminf(v, __fadd_rn(hi, 1.f));
            break;
        }
    }
    for (int a = 1; a < nx; ++a) {                                   // insertion sort, nx <= 8
        const float v = xx[a];
        int b = a - 1;
        while (b >= 0 && xx[b] > v) { xx[b + 1] = xx[b]; --b; }
        xx[b + 1] = v;
    }
    int x_pos = nx ? (int)xx[0] : 0;
    for (int i = 1; i < nx; i += 2) {
        const int x_end = rq_round_down(xx[i]);
        if (x_end < x_pos) continue;
        int x_start = rq_round_up(xx[i - 1]);
        if (x_pos > x_start) { x_start = x_pos; if (x_end < x_start) continue; }
        if (x_start > x_end) continue;
        if (px >= x_start && px <= x_end) return true;
        x_pos = x_end + 1;
    }
    return false;
}

__global__ __launch_bounds__(256) void build_label_maps_kernel(const double *__restrict__ quads, const int *__restrict__ values,
                                                               const int *__restrict__ counts, int n, int cap, int map_h,
                                                               int map_w, int scale, int *__restrict__ labels)
{
    const long total = (long)n * map_h * map_w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int img = (int)(p / ((long)map_h * map_w));
        const int loc = (int)(p - (long)img * map_h * map_w);
        const int py = loc / map_w, px = loc - py * map_w;
        int cnt = counts[img];
        cnt = cnt < cap ? cnt : cap;
        int label = 0;
        for (int o = 0; o < cnt; ++o) {
            int pts[8];
            rq_proper_round(quads + ((size_t)img * cap + o) * 8, scale, pts);
            if (rq_covers(pts, px, py, map_h)) label = values[(size_t)img * cap + o];
        }
        labels[p] = label;
    }
}

extern "C" int ubd_build_label_maps(const double *quads, const int32_t *values, const int32_t *counts, int n, int cap,
                                    int map_h, int map_w, int scale, int32_t *labels, void *stream)
{
    UBD_REQUIRE(quads && values && counts && labels, "ubd_build_label_maps: null argument");
    UBD_REQUIRE(n > 0 && cap > 0 && map_h > 0 && map_w > 0 && scale > 0, "ubd_build_label_maps: bad shape");
    const long total = (long)n * map_h * map_w;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(build_label_maps_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, quads, values, counts, n, cap, map_h, map_w, scale, labels);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}
