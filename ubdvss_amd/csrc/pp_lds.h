// Device code of the one-launch postprocess front end (one block = one image, union-find forest in LDS) and of the
// per-object box fit, shared by postprocess.hip (stand-alone launches) and forward.hip (the fused stem kernel carries the
// postprocess of the PREVIOUS batch in its first blocks: stem123.h).  Reference call sites and the equivalences behind the
// parallel formulation: see the head of postprocess.hip.  OpenCV-exact float geometry: every function that does float
// arithmetic switches FMA contraction off for its own body, so the results do not depend on the including file's flags.
#pragma once
#include "common.h"

#define CV_PI 3.1415926535897932384626433832795
#define STAGE_INTS 10     // root, quad[8], spare

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
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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

// ---- wave-uniform forms of hull_finish / min_area_box for polygons of at most 64 vertices: vertex v lives in LANE v
// (registers X, Y), every lane runs the same scalar program and fetches vertices with v_readlane (index in an SGPR) instead
// of walking an LDS array on one lane.  Same operations in the same order as the one-lane forms above: the results are
// bit-identical.  Measured (in-kernel stamps, round 3): the box fit of an image's largest object is 64 k cycles this way against
// 70 k on one lane -- a serial chain of dependent instructions costs ~10 cycles per instruction either way; building the hull
// chains in registers too (monotone chain with v_readlane) was SLOWER (119 k).  Kept for the 6 k and because it frees the lane-0
// LDS traffic; the results are bit-identical (tests/test_gpu_postprocess.py runs every case through both -- UBD_PP_SPLIT keeps the LDS form).
__device__ __forceinline__ int pp_rl(int v, int idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx)); }
__device__ __forceinline__ float pp_rlf(float v, int idx)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(idx)));
}

// P = lc[0..nl-1] (top -> bottom) in lanes 0..nl-1 followed by rc REVERSED (bottom -> top) in lanes nl..nl+nr-1 on entry.
__device__ __forceinline__ int hull_finish_wave(int &X, int &Y, int nl, int nr, int lane)
{
    int n = nl + nr;
    auto remove_at = [&](int k) {                              // P[k..n-2] = P[k+1..n-1]
        const int xs = __shfl_down(X, 1, 64), ys = __shfl_down(Y, 1, 64);
        if (lane >= k) { X = xs; Y = ys; }
        --n;
    };
    if (nr > 0 && pp_rl(X, nl - 1) == pp_rl(X, nl) && pp_rl(Y, nl - 1) == pp_rl(Y, nl)) remove_at(nl);      // bottom junction
    if (n > 1 && pp_rl(X, n - 1) == pp_rl(X, 0) && pp_rl(Y, n - 1) == pp_rl(Y, 0)) --n;                    // top junction
    // The chains come out of the gift wrapping without collinear triples; one can only appear where the chains meet.  ALL triples are tested
    // at once (vertex v in lane v, neighbours by lane exchange): the serial removal loop below -- n dependent steps per sweep, at least one
    // sweep, ~10 cycles per dependent instruction on a lone wave -- runs only when some triple IS collinear (it then does what it always did).
    bool changed = false;
    if (n > 2) {
        const int lp = lane == 0 ? n - 1 : lane - 1, ln = lane + 1 >= n ? 0 : lane + 1;
        const ipt a = {__shfl(X, lp, 64), __shfl(Y, lp, 64)}, b = {X, Y}, c = {__shfl(X, ln, 64), __shfl(Y, ln, 64)};
        changed = __ballot(lane < n && cross3(a, b, c) == 0) != 0ull;
    }
    while (changed && n > 2) {
        changed = false;
        for (int k = 0; k < n && n > 2; ++k) {
            const int ka = k == 0 ? n - 1 : k - 1, kc = k + 1 == n ? 0 : k + 1;
            const ipt a = {pp_rl(X, ka), pp_rl(Y, ka)}, b = {pp_rl(X, k), pp_rl(Y, k)}, c = {pp_rl(X, kc), pp_rl(Y, kc)};
            if (cross3(a, b, c) == 0) { remove_at(k); --k; changed = true; }
        }
    }
    if (n == 2) {
        const int x0 = pp_rl(X, 0), y0 = pp_rl(Y, 0), x1 = pp_rl(X, 1), y1 = pp_rl(Y, 1);
        if ((x1 < x0) || (x1 == x0 && y1 < y0)) {               // lexicographic min first
            if (lane == 0) { X = x1; Y = y1; }
            if (lane == 1) { X = x0; Y = y0; }
        }
    } else if (n > 2) {
        int s = 0, sx = pp_rl(X, 0), sy = pp_rl(Y, 0);         // min-x (then min-y) vertex comes first
        for (int k = 1; k < n; ++k) {
            const int xk = pp_rl(X, k), yk = pp_rl(Y, k);
            if (xk < sx || (xk == sx && yk < sy)) { s = k; sx = xk; sy = yk; }
        }
        if (s != 0) {
            int src = lane + s;
            src = src >= n ? src - n : src;
            const int xr = __shfl(X, src, 64), yr = __shfl(Y, src, 64);
            X = xr; Y = yr;
        }
    }
    return n;
}

// cv::minAreaRect + cv::boxPoints as min_area_box, hull vertex v in lane v (n <= 64); every lane returns the same box8.
__device__ __forceinline__ void min_area_box_wave(int X, int Y, int n, int lane, float *box8)
{
#pragma clang fp contract(off)
    float cxr = 0.f, cyr = 0.f, bw = 0.f, bh = 0.f, angle = 0.f;
    const float fx = (float)X, fy = (float)Y;
    if (n > 2) {
        // edge table, one edge per lane (min_area_box::vec)
        float evx, evy, einv;
        {
            int j = lane + 1;
            j = j < n ? j : 0;
            const float xj = (float)__shfl(X, j, 64), yj = (float)__shfl(Y, j, 64);
            const double dx = xj - fx;
            const double dy = yj - fy;
            evx = (float)dx; evy = (float)dy;
            einv = (float)(1. / sqrt(dx * dx + dy * dy));
        }
        float minarea = 3.402823466e+38f;
        int buf_i0 = 0, buf_i5 = 0;
        float buf1 = 0.f, buf2 = 0.f, buf3 = 0.f, buf4 = 0.f;
        int left = 0, bottom = 0, right = 0, top = 0;
        float left_x, right_x, top_y, bottom_y;
        left_x = right_x = pp_rlf(fx, 0);
        top_y = bottom_y = pp_rlf(fy, 0);
        for (int i = 0; i < n; ++i) {
            const float px = pp_rlf(fx, i), py = pp_rlf(fy, i);
            if (px < left_x) left_x = px, left = i;
            if (px > right_x) right_x = px, right = i;
            if (py > top_y) top_y = py, top = i;
            if (py < bottom_y) bottom_y = py, bottom = i;
        }
        float orientation = 0.f;
        {
            double ax = pp_rlf(evx, n - 1), ay = pp_rlf(evy, n - 1);
            for (int i = 0; i < n; ++i) {
                const double bx = pp_rlf(evx, i), by = pp_rlf(evy, i);
                const double convexity = ax * by - ay * bx;
                if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
                ax = bx; ay = by;
            }
        }
        float base_a = orientation, base_b = 0.f;
        // the four calipers: index, edge (vx, vy, 1/length) and vertex of each -- only the one that advances is re-fetched
        int s0 = bottom, s1 = right, s2 = top, s3 = left;
        float vx0 = pp_rlf(evx, s0), vy0 = pp_rlf(evy, s0), in0 = pp_rlf(einv, s0), px0 = pp_rlf(fx, s0), py0 = pp_rlf(fy, s0);
        float vx1 = pp_rlf(evx, s1), vy1 = pp_rlf(evy, s1), in1 = pp_rlf(einv, s1), px1 = pp_rlf(fx, s1), py1 = pp_rlf(fy, s1);
        float vx2 = pp_rlf(evx, s2), vy2 = pp_rlf(evy, s2), in2 = pp_rlf(einv, s2), px2 = pp_rlf(fx, s2), py2 = pp_rlf(fy, s2);
        float vx3 = pp_rlf(evx, s3), vy3 = pp_rlf(evy, s3), in3 = pp_rlf(einv, s3), px3 = pp_rlf(fx, s3), py3 = pp_rlf(fy, s3);
        for (int k = 0; k < n; ++k) {
            const float dp0 = +base_a * vx0 + base_b * vy0;
            const float dp1 = -base_b * vx1 + base_a * vy1;
            const float dp2 = -base_a * vx2 - base_b * vy2;
            const float dp3 = +base_b * vx3 - base_a * vy3;
            float maxcos = dp0 * in0;
            int main_element = 0;
            const float c1 = dp1 * in1; if (c1 > maxcos) { main_element = 1; maxcos = c1; }
            const float c2 = dp2 * in2; if (c2 > maxcos) { main_element = 2; maxcos = c2; }
            const float c3 = dp3 * in3; if (c3 > maxcos) { main_element = 3; maxcos = c3; }
            main_element = __builtin_amdgcn_readfirstlane(main_element);
            {
                const float mvx = main_element == 0 ? vx0 : main_element == 1 ? vx1 : main_element == 2 ? vx2 : vx3;
                const float mvy = main_element == 0 ? vy0 : main_element == 1 ? vy1 : main_element == 2 ? vy2 : vy3;
                const float min_ = main_element == 0 ? in0 : main_element == 1 ? in1 : main_element == 2 ? in2 : in3;
                const float lead_x = mvx * min_;
                const float lead_y = mvy * min_;
                switch (main_element) {
                case 0: base_a = lead_x;  base_b = lead_y;  break;
                case 1: base_a = lead_y;  base_b = -lead_x; break;
                case 2: base_a = -lead_x; base_b = -lead_y; break;
                default: base_a = -lead_y; base_b = lead_x; break;
                }
            }
            {
                int sm = main_element == 0 ? s0 : main_element == 1 ? s1 : main_element == 2 ? s2 : s3;
                sm += 1;
                sm = (sm == n) ? 0 : sm;
                const float nvx = pp_rlf(evx, sm), nvy = pp_rlf(evy, sm), nin = pp_rlf(einv, sm), npx = pp_rlf(fx, sm), npy = pp_rlf(fy, sm);
                if (main_element == 0) { s0 = sm; vx0 = nvx; vy0 = nvy; in0 = nin; px0 = npx; py0 = npy; }
                else if (main_element == 1) { s1 = sm; vx1 = nvx; vy1 = nvy; in1 = nin; px1 = npx; py1 = npy; }
                else if (main_element == 2) { s2 = sm; vx2 = nvx; vy2 = nvy; in2 = nin; px2 = npx; py2 = npy; }
                else { s3 = sm; vx3 = nvx; vy3 = nvy; in3 = nin; px3 = npx; py3 = npy; }
            }
            {
                float dx = px1 - px3;
                float dy = py1 - py3;
                const float width = dx * base_a + dy * base_b;
                dx = px2 - px0;
                dy = py2 - py0;
                const float height = -dx * base_b + dy * base_a;
                const float area = width * height;
                if (area <= minarea) {
                    minarea = area;
                    buf_i0 = s3;
                    buf1 = base_a; buf2 = width; buf3 = base_b; buf4 = height;
                    buf_i5 = s0;
                }
            }
        }
        const float A1 = buf1, B1 = buf3, A2 = -buf3, B2 = buf1;
        const float h0x = pp_rlf(fx, buf_i0), h0y = pp_rlf(fy, buf_i0), h5x = pp_rlf(fx, buf_i5), h5y = pp_rlf(fy, buf_i5);
        const float C1 = A1 * h0x + h0y * B1;
        const float C2 = A2 * h5x + h5y * B2;
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
        const float x0 = pp_rlf(fx, 0), y0 = pp_rlf(fy, 0), x1 = pp_rlf(fx, 1), y1 = pp_rlf(fy, 1);
        cxr = (x0 + x1) * 0.5f;
        cyr = (y0 + y1) * 0.5f;
        const double dx = x1 - x0, dy = y1 - y0;
        bw = (float)sqrt(dx * dx + dy * dy);
        bh = 0.f;
        angle = (float)atan2(dy, dx);
    } else if (n == 1) {
        cxr = pp_rlf(fx, 0); cyr = pp_rlf(fy, 0);
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

// One wave, one kept object: row extents (global, 2 ints per row from row y0) -> hull -> minAreaRect -> boxPoints -> rounded quad
// into st[1..8].  rws: the wave's LDS scratch of 12 * h + 4 ints (row extents | hull points | edge table).  ATOMIC: the
// extents were accumulated by atomics of THIS launch (the fused one-launch front end): read them past the CU's vector L1.
template <bool ATOMIC>
__device__ __forceinline__ void pp_box_object(int *rws, const int *__restrict__ g, int nrows, int y0, int h, int lane, int scale,
                                              int *__restrict__ st, bool serial_tail = false)
{
#pragma clang fp contract(off)
    ipt *pts = (ipt *)(rws + 2 * h);
    float *etab = (float *)(rws + 6 * h);                   // edge table of the hull: 3 x (<= 2h) floats
    for (int r = lane; r < 2 * nrows; r += 64)
        rws[r] = ATOMIC ? __hip_atomic_load(&g[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int cnt[2] = {0, 0};
    ipt *ptsr = pts + h;                               // the right chain is built here (second half of the 2 h point slots)
    {
        // Gift wrapping down BOTH chains at once: lanes 0-31 wrap the left chain (min x), lanes 32-63 the right one (max x).  From
        // vertex row c the next vertex is the later row with the extreme slope dx/dy (min for the left chain, max for the right one;
        // farthest on ties, which drops collinear points).  A chain's candidates are spread over its 32 lanes, fractions compared exactly
        // by int32 cross-multiplication (|dx|, dy < 2^15), then a xor-butterfly inside the half.  A step is ~1.2 k cycles of dependent
        // lane exchanges whoever takes part: with the chains one after the other (round 1-5) the object paid for the SUM of their vertex
        // counts, now for the larger one (the box fit was 75 k of the front end's 170 k cycles, tools/stamps_pp.py).
        const int side = lane >> 5, hl = lane & 31;
        ipt *out = side ? ptsr : pts;
        int nout = 0, c = 0;
        bool done = false;                            // uniform within a half
        for (;;) {
            int xc = 0;
            if (!done) {
                xc = rws[2 * c + side];
                if (hl == 0) out[nout] = (ipt){xc, y0 + c};
                ++nout;
                if (c >= nrows - 1) done = true;
            }
            if (__ballot(!done) == 0ull) break;       // wave-uniform: both chains have reached the last row
            int bn = 0, bd = 0, br = -1;
            if (!done)
                for (int r = c + 1 + hl; r < nrows; r += 32) {
                    const int nn = rws[2 * r + side] - xc, dd = r - c;
                    bool better = true;
                    if (bd != 0) {
                        const int lhs = nn * bd, rhs = bn * dd;
                        better = side ? (lhs >= rhs) : (lhs <= rhs);      // later row wins ties
                    }
                    if (better) { bn = nn; bd = dd; br = r; }
                }
            // The best of the half's 32 candidates (a maximum under a total order -- slope, then row -- so any exchange pattern that
            // joins groups which already agree will do): inside a row of 16 lanes by DPP (quad permutes, then the mirrors of 8 and of 16 --
            // no LDS crossbar round trip), then one exchange between the half's two rows.
            auto join = [&](int on, int od, int orr) {
                bool take;
                if (od == 0) take = false;
                else if (bd == 0) take = true;
                else {
                    const int lhs = on * bd, rhs = bn * od;
                    take = lhs == rhs ? (orr > br) : (side ? (lhs > rhs) : (lhs < rhs));
                }
                if (take) { bn = on; bd = od; br = orr; }
            };
#define PP_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false)
            join(PP_DPP(bn, 0xB1), PP_DPP(bd, 0xB1), PP_DPP(br, 0xB1));            // quad_perm [1,0,3,2]
            join(PP_DPP(bn, 0x4E), PP_DPP(bd, 0x4E), PP_DPP(br, 0x4E));            // quad_perm [2,3,0,1]
            join(PP_DPP(bn, 0x141), PP_DPP(bd, 0x141), PP_DPP(br, 0x141));         // row_half_mirror
            join(PP_DPP(bn, 0x140), PP_DPP(bd, 0x140), PP_DPP(br, 0x140));         // row_mirror
#undef PP_DPP
            join(__shfl_xor(bn, 16, 64), __shfl_xor(bd, 16, 64), __shfl_xor(br, 16, 64));
            if (!done) c = br;
        }
        cnt[0] = __builtin_amdgcn_readlane(nout, 0);
        cnt[1] = __builtin_amdgcn_readlane(nout, 32);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (cnt[0] + cnt[1] <= 64 && !serial_tail) {                  // wave-uniform: the polygon fits the wave's lanes (the usual case)
        int X = 0, Y = 0;
        if (lane < cnt[0] + cnt[1]) {
            const ipt pv = lane < cnt[0] ? pts[lane] : ptsr[cnt[1] - 1 - (lane - cnt[0])];   // the right chain reversed
            X = pv.x; Y = pv.y;
        }
        const int nh = hull_finish_wave(X, Y, cnt[0], cnt[1], lane);
        float box[8];
        min_area_box_wave(X, Y, nh, lane, box);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) st[1 + j] = (int)rintf(box[j] * (float)scale);   // np.round: half to even
            st[9] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        return;
    }
    // the one-lane form wants the right chain right behind the left one: moved down in pieces of 64 (a piece's targets lie below every later source)
    for (int e0 = 0; e0 < cnt[1]; e0 += 64) {
        const int e = e0 + lane;
        ipt t = {0, 0};
        if (e < cnt[1]) t = ptsr[e];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (e < cnt[1]) pts[cnt[0] + e] = t;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
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

// ------------------------------------------------------------------------------------ fused front end (LDS)
// Maps of at most PP_LDS_MAX_HW pixels (128 x 128: the 512 x 512 input of the headline configuration): init, merge,
// flatten, roots, owner, area, keep and extents of ONE image run in ONE block with the union-find forest, the
// foreground bits and the owner map in LDS -- one launch instead of nine, no global atomics on the forest.  Every phase
// is the corresponding kernel above restated on LDS arrays (same links, same external rule, same bit-quad area), so
// the results are identical; the phases are separated by block barriers instead of kernel boundaries.
#define PP_LDS_MAX_HW 16384
#define PP_LDS_THREADS 1024     // stand-alone launch: one 16-wave block per image

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
// the most LDS a job can need (1-pixel-high maps hold the most roots per pixel: root_cap = hw / 2 + 2)
#define PP_LDS_MAX_BYTES ((((2 * (PP_LDS_MAX_HW / 2 + 2)) * 4 + 15) & ~15) + PP_LDS_MAX_HW * 4 + PP_LDS_MAX_HW + 16)
static size_t pp_front_lds_bytes(int hw, int root_cap)
{
    const size_t lab_ints = (size_t)hw + 1 > 2 * (size_t)root_cap ? (size_t)hw + 1 : 2 * (size_t)root_cap;
    return ubd_align_up(lab_ints * 4, 16) + (size_t)hw * 2 * 2 + ubd_align_up(hw, 16) + 16;
}

// Everything the block needs for its image(s); filled by the host (postprocess.hip pp_job_fill), passed by value.
struct pp_lds_args {
    const float *logits;
    int n, k_out, h, w, cap, n_cls, root_cap, poison, scale, serial_tail;   // serial_tail: one-lane LDS form of the box fit (tests)
    float thr, min_area;
    int *binary_map, *g_nroots, *g_nkept, *g_owner, *g_roots, *g_kept, *stage, *ymax, *rows;
    float *vote;
    int *quads, *classes, *counts;
    unsigned long long *stamps;                                // diagnostic build only (nullptr otherwise)
};

// NT threads (a multiple of 64, <= 1024) work on image `img`; `smem`: pp_front_lds_bytes(h * w, root_cap) bytes of LDS, 16-byte aligned.
// TAIL: the block also fits its image's boxes (one wave per kept object, scratch in the dead parts of the forest / root-slot
// arrays), takes the class vote and emits the lists -- the whole postprocess of an image is one call.  A block that works on
// several images in turn puts a __syncthreads() between the calls.
template <int NT, bool TAIL>
__device__ __forceinline__ void pp_image_lds(int *__restrict__ smem, const pp_lds_args &A, const int img)
{
#pragma clang fp contract(off)
    const float *__restrict__ logits = A.logits;
    const int k_out = A.k_out, h = A.h, w = A.w, cap = A.cap, n_cls = A.n_cls, root_cap = A.root_cap, poison = A.poison, scale = A.scale;
    const float thr = A.thr, min_area = A.min_area;
    int *__restrict__ binary_map = A.binary_map, *__restrict__ g_nroots = A.g_nroots, *__restrict__ g_nkept = A.g_nkept;
    int *__restrict__ g_owner = A.g_owner, *__restrict__ g_roots = A.g_roots, *__restrict__ g_kept = A.g_kept;
    int *__restrict__ stage = A.stage, *__restrict__ ymax = A.ymax, *__restrict__ rows = A.rows;
    float *__restrict__ vote = A.vote;
    int *__restrict__ quads = A.quads, *__restrict__ classes = A.classes, *__restrict__ counts = A.counts;
#ifdef UBD_STAMPS
    unsigned long long *__restrict__ stamps = A.stamps;
    int stamp_k = 0;
#define PPSTAMP() do { if (stamps && threadIdx.x == 0) stamps[img * 16 + stamp_k] = __builtin_amdgcn_s_memtime(); ++stamp_k; } while (0)
#else
#define PPSTAMP() do {} while (0)
#endif
    PPSTAMP();
    const int hw = h * w;
    int *lab = smem;
    const int lab_ints = hw + 1 > 2 * root_cap ? hw + 1 : 2 * root_cap;
    short *own16 = (short *)((char *)smem + (((size_t)lab_ints * 4 + 15) & ~(size_t)15));
    short *rs16 = own16 + hw;
    unsigned char *m = (unsigned char *)(rs16 + hw);
    int *ctr = (int *)(m + ((hw + 15) & ~15));               // [0] roots, [1] kept, [2] integrity flag (UBD_PP_POISON)
    int *area2 = lab, *kept = lab + root_cap;                  // aliases, valid after the owner phase
    const int tid = threadIdx.x, lane = tid & 63;
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
        for (int i = tid; i < words; i += NT) smem[i] = 0x7fff7fff;
        __syncthreads();
    }
    // ---- init (pp_init_kernel); the thread's logits are requested up front, PP_LDS_MAX_HW / NT at most
    if (tid < 3) ctr[tid] = 0;                                 // [2]: integrity flag of the test mode
    constexpr int PER_THREAD = PP_LDS_MAX_HW / NT;
    static_assert(NT % 64 == 0 && NT <= 1024 && PP_LDS_MAX_HW % NT == 0, "whole waves, whole rounds");
    float lg[PER_THREAD], lgl[PER_THREAD];                     // lgl: logit left of the wave's first pixel (lane 0 only), requested with the
#pragma unroll                                               // rest: fetched inside the loop it was one dependent memory round trip per iteration
    for (int it = 0; it < PER_THREAD; ++it) {
        const int loc = tid + it * NT;
        lg[it] = loc < hw ? logits[(pbase + loc) * k_out] : 0.f;
        lgl[it] = (lane == 0 && loc < hw && loc > 0) ? logits[(pbase + loc - 1) * k_out] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < PER_THREAD; ++it) {
        const int loc = tid + it * NT;
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
        const int iters = (hw + NT - 1) / NT;
        const bool queued = (size_t)iters * NT * 4 <= (size_t)hw * 4;    // 2 jobs per pixel fit the slice (hw a multiple of NT)
        unsigned short *queue = (unsigned short *)own16 + (size_t)(tid >> 6) * iters * 128;
        int njobs = 0;                                                    // wave-uniform
        for (int loc = tid; loc - lane < hw; loc += NT) {     // wave-uniform trip count
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
#ifdef UBD_PP_RACY_FLATTEN   // diagnostic build only (tools/prove_stress_power.sh): round 2's compressing find, to show that the stress test catches it
    for (int node = tid; node <= hw; node += NT) {
        const int r = uf_find_wg(lab, node);
        __hip_atomic_store(&lab[node], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#else
    // four independent walks per thread, interleaved: a walk is a chain of dependent LDS reads, and the wave waits for the longest
    // of its 64 chains; four chains in flight hide most of that latency.  (Pointer jumping in rounds -- parent := grandparent until
    // nothing changes -- was measured too: 26.6 k cycles against 20.5 k for this, the rounds re-read every node log2(depth) times.)
    for (int node0 = tid; node0 <= hw; node0 += 4 * NT) {
        int cur[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cur[k] = node0 + k * NT; live[k] = cur[k] <= hw; if (!live[k]) cur[k] = 0; }
        for (;;) {
            int nxt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[k] = __hip_atomic_load(&lab[cur[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool moved = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) { moved |= nxt[k] != cur[k]; cur[k] = nxt[k]; }
            if (!moved) break;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (live[k]) __hip_atomic_store(&lab[node0 + k * NT], cur[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#endif
    // the owner / root-slot arrays were the union job queue until the barrier above: no entry is valid yet.  -1 = "no slot", so
    // a read of an entry this launch never wrote cannot return a plausible slot (stale queue jobs are small integers).
    {
        int *z = (int *)own16;                                   // own16 [hw] | rs16 [hw] = hw ints from a 16-byte-aligned base
        for (int i = tid; i < hw; i += NT) z[i] = -1;
    }
    __syncthreads();
    PPSTAMP();

    // ---- roots (pp_roots_kernel)
#pragma unroll 4
    for (int loc = tid; loc < hw; loc += NT) {
        if (!m[loc] || lab[loc + 1] != loc + 1) continue;
        const bool external = (loc < w) || (lab[loc + 1 - w] == 0);
        if (!external) continue;
        const int idx = atomicAdd(&ctr[0], 1);
        g_roots[(size_t)img * root_cap + idx] = loc;
        rs16[loc] = (short)idx;
    }
    __syncthreads();
    PPSTAMP();

    // ---- owner (pp_owner_kernel).  All pixels of a region share its owner, so only the region's raster-first pixel (the root)
    // walks the nesting chain -- two to three dependent LDS reads per step -- and the others copy the root's result after a
    // barrier (two reads): 40 k -> 25 k cycles on noise maps at 512 threads (in-kernel stamps).
    // Roots are sparse (a few hundred per map), so a thread first COLLECTS its roots -- a scan of independent LDS reads -- and then
    // the wave walks them round by round with all their lanes together: walking inside the scan cost every one of the 16 / 32
    // scan iterations a full walk latency for the one or two lanes that had a root there (in-kernel stamps: 23 k of 34 k cycles).
    auto owner_walk = [&](int loc) {
        // the forest is flat (every entry is a root: checked under UBD_PP_POISON in the copy pass below), and a slot is read
        // only at a root's own raster-first pixel
        int node = loc + 1;
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
    };
    for (int skip = 0;; skip += 6) {                                            // six roots per thread and round; more only on adversarial maps
        int r0 = -1, r1 = -1, r2 = -1, r3 = -1, r4 = -1, r5 = -1, nr = 0;
#pragma unroll 8
        for (int loc = tid; loc < hw; loc += NT) {
            const bool is_root = lab[loc + 1] == loc + 1;
            if (is_root) {
                const int j = nr - skip;
                r0 = j == 0 ? loc : r0; r1 = j == 1 ? loc : r1; r2 = j == 2 ? loc : r2;
                r3 = j == 3 ? loc : r3; r4 = j == 4 ? loc : r4; r5 = j == 5 ? loc : r5;
                ++nr;
            }
        }
        if (r0 >= 0) owner_walk(r0);
        if (r1 >= 0) owner_walk(r1);
        if (r2 >= 0) owner_walk(r2);
        if (r3 >= 0) owner_walk(r3);
        if (r4 >= 0) owner_walk(r4);
        if (r5 >= 0) owner_walk(r5);
        if (!__syncthreads_or(nr > skip + 6)) break;                            // block-uniform: somebody has more roots
    }
    __syncthreads();
    // copy pass: root -> every pixel of its region, into the OTHER 16-bit array (the root-slot array is dead now; source and
    // destination cannot alias, so four pixels' reads are in flight together), which then takes over as the owner map
    {
        short *const own_src = own16, *const own_dst = rs16;
        for (int loc0 = tid; loc0 < hw; loc0 += 4 * NT) {
            int node[4], own[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int loc = loc0 + k * NT; node[k] = loc < hw ? lab[loc + 1] : 0; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (poison && lab[node[k]] != node[k]) atomicOr(&ctr[2], 1);    // test mode: a non-root survived the flatten phase
                own[k] = node[k] == 0 ? -1 : own_src[node[k] - 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int loc = loc0 + k * NT;
                if (loc < hw) {
                    own_dst[loc] = (short)own[k];
                    if (!TAIL && g_owner) g_owner[pbase + loc] = own[k];
                }
            }
        }
    }
    __syncthreads();
    { short *t = own16; own16 = rs16; rs16 = t; }                              // owner map <-> free 16-bit array (box scratch of the tail)
    PPSTAMP();
    const int nroots = ctr[0];
    for (int s = tid; s < nroots; s += NT) area2[s] = 0;       // the forest is dead from here on
    __syncthreads();
    PPSTAMP();

    // ---- area (pp_area_kernel).  A thread's pixels lie NT / w rows apart, usually inside the same object: the contributions are
    // summed in a register while the key does not change and flushed with one LDS atomic per run (integer adds: any order gives
    // the same sum).  One object covering the map -- what an untrained network's logits look like -- had every interior pixel
    // add to ONE LDS word: the atomics of a wave serialise.
    {
        int run_key = -1, run_sum = 0;
#pragma unroll 4
        for (int loc = tid; loc < hw; loc += NT) {
            int key = -1, val = 0;
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
            if (val != 0) {
                if (key != run_key) {
                    if (run_sum != 0) atomicAdd(&area2[run_key], run_sum);
                    run_key = key; run_sum = 0;
                }
                run_sum += val;
            }
        }
        if (run_sum != 0) atomicAdd(&area2[run_key], run_sum);
    }
    __syncthreads();
    PPSTAMP();

    // ---- keep (pp_keep_kernel)
    for (int s = tid; s < nroots; s += NT) {
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
        for (int e = tid; e < nk * h; e += NT) {
            const int k = e / h, y = e - k * h;
            r[(size_t)k * (3 * h) + y] = make_int2(0x7fffffff, -1);
        }
    }
    __syncthreads();
    PPSTAMP();

    // ---- extents (pp_extents_kernel): four pixels' neighbourhoods are read together; the last row of an object is taken per
    // thread at the end of a run of equal owners (a thread walks down the map), not by one global atomic per bottom-edge pixel
    {
        int run_k = -1, run_y = 0;
        for (int loc0 = tid; loc0 < hw; loc0 += 4 * NT) {
            int o[4], ol[4], orr[4], yy[4], xx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int loc = loc0 + k * NT;
                o[k] = loc < hw ? own16[loc] : -1;
                yy[k] = row_of(loc); xx[k] = loc - yy[k] * w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int loc = loc0 + k * NT;
                ol[k] = (o[k] >= 0 && xx[k] > 0) ? own16[loc - 1] : -2;
                orr[k] = (o[k] >= 0 && xx[k] < w - 1) ? own16[loc + 1] : -2;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (o[k] < 0) continue;
                const int kk = kept[o[k]];
                if (kk != run_k) {
                    if (run_k >= 0) atomicMax(&ymax[(size_t)img * cap + run_k], run_y);
                    run_k = kk;
                }
                run_y = yy[k];
                if (kk < 0) continue;
                const bool left_end = ol[k] != o[k], right_end = orr[k] != o[k];
                if (!(left_end || right_end)) continue;
                int *r = rows + ((size_t)img * cap + kk) * (size_t)(6 * h);
                if (left_end) atomicMin(&r[2 * yy[k]], xx[k]);
                if (right_end) atomicMax(&r[2 * yy[k] + 1], xx[k]);
            }
        }
        if (run_k >= 0) atomicMax(&ymax[(size_t)img * cap + run_k], run_y);
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
            const int nwv = min(NT / 64, nA + nB);                       // >= 1 (host)
            int *scratch = wid < nA ? lab + 2 * root_cap + wid * S : (int *)rs16 + (wid - nA) * S;
            if (wid < nwv)
                for (int k = wid; k < nk; k += nwv) {
                    int *st = stage + ((size_t)img * cap + k) * STAGE_INTS;
                    const int y0 = row_of(st[0]);
                    const int ym = __hip_atomic_load(&ymax[(size_t)img * cap + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int *g = rows + ((size_t)img * cap + k) * (size_t)(6 * h) + 2 * y0;
                    pp_box_object<true>(scratch, g, ym - y0 + 1, y0, h, lane, scale, st, A.serial_tail != 0);
                }
        }
        // ---- vote (pp_vote_kernel): mean softmax over the filled region
        if (n_cls > 0) {
            for (int loc = tid; loc < hw; loc += NT) {
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
        for (int sidx = tid; sidx < nk; sidx += NT) {
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

