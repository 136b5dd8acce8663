/* Oracle: plain-C restatement of the OpenCV 3.4 routines the ubdvss postprocess
 * calls.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never linked or
 * loaded by the product library.
 *
 * PARITY UNPINNED: opencv-python (requirements.txt:5 pins >=3.4,<4.0) is a
 * third-party dependency that is not vendored in the reference and not
 * installable here; the reference has no tests or golden vectors for this path.
 * The routines below restate the *published* OpenCV 3.4 algorithms
 * (modules/imgproc/src/contours.cpp: cvFindNextContour + icvFetchContour,
 * shapedescr.cpp: contourArea / minAreaRect, convhull.cpp: convexHull +
 * Sklansky_, rotcalipers.cpp: rotatingCalipers, types.cpp: RotatedRect::points)
 * and are anchored on the reference's call sites:
 *   semantic_segmentation/utils.py:51-60        get_contours_and_boxes
 *       findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) -> contourArea > min_area
 *       -> minAreaRect -> boxPoints
 *   semantic_segmentation/segmap_manager.py:54-67  np.round(box*scale).astype(int);
 *       drawContours(fill) mask -> mean softmax prob -> argmax
 *   semantic_segmentation/utils.py:135-138      np_softmax
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, so that float32
 * arithmetic is plain IEEE like OpenCV's generic x86-64 build).
 *
 * ATTRIBUTION.  The operation order, the tie rules and several identifiers below
 * (lnbd, Sklansky_, tl_stack, inv_vect_length, rotatingCalipers ...) follow the
 * OpenCV 3.4 sources named above on purpose: bit-exact quads need OpenCV's exact
 * float32 arithmetic.  OpenCV is distributed under the 3-clause BSD license:
 *   Copyright (C) 2000-2008, Intel Corporation, all rights reserved.
 *   Copyright (C) 2009-2011, Willow Garage Inc., all rights reserved.
 *   Copyright (C) 2009-2016, NVIDIA Corporation, all rights reserved.
 *   Copyright (C) 2010-2013, Advanced Micro Devices, Inc., all rights reserved.
 *   Copyright (C) 2015-2016, OpenCV Foundation, all rights reserved.
 *   Copyright (C) 2015-2016, Itseez Inc., all rights reserved.
 *   Third party copyrights are property of their respective owners.
 * Redistribution and use in source and binary forms, with or without modification,
 * are permitted provided that the following conditions are met: redistributions of
 * source code must retain the above copyright notice, this list of conditions and
 * the following disclaimer; redistributions in binary form must reproduce them in
 * the documentation and/or other materials provided with the distribution; neither
 * the names of the copyright holders nor the names of the contributors may be used
 * to endorse or promote products derived from this software without specific prior
 * written permission.  This software is provided by the copyright holders and
 * contributors "as is" and any express or implied warranties, including, but not
 * limited to, the implied warranties of merchantability and fitness for a
 * particular purpose are disclaimed.  In no event shall the copyright holders or
 * contributors be liable for any direct, indirect, incidental, special, exemplary,
 * or consequential damages however caused and on any theory of liability arising in
 * any way out of the use of this software.
 * The same notice covers the box fit of ubdvss_amd/csrc/postprocess.hip and
 * pp_lds.h, which repeats this arithmetic on the device.
 * An INDEPENDENT cross-check written from Suzuki & Abe's paper, not from OpenCV,
 * is oracle/suzuki_abe.py (tests/test_oracle_hardening.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define CV_PI 3.1415926535897932384626433832795

/* ------------------------------------------------------------------ contours
 * cvFindNextContour (mode = CV_RETR_EXTERNAL) on a copy of the image with a
 * 1-px zero frame (cv::findContours does copyMakeBorder + offset(-1,-1) since
 * 3.2), pixels thresholded to {0,1}; border following = icvFetchContour with
 * nbd = 2, method = CV_CHAIN_APPROX_SIMPLE (a point is emitted when the chain
 * direction changes).
 * Output: contours in DISCOVERY (raster) order; cv2 returns them reversed
 * (cvInsertNodeIntoTree pushes each new contour to the front of the list).
 */
static const int code_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int code_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* returns number of points appended, or -1 on overflow */
static int fetch_contour(signed char *img, int step, int x0, int y0, int offx, int offy,
                         int approx_simple, int *pts, int cap)
{
    const signed char nbd = 2;
    int deltas[16];
    int n = 0;
    for (int i = 0; i < 8; i++) deltas[i] = code_dy[i] * step + code_dx[i];
    memcpy(deltas + 8, deltas, 8 * sizeof(int));

    signed char *i0 = img + y0 * step + x0, *i1, *i3, *i4 = 0;
    int px = x0, py = y0;
    int s_end = 4, s = 4, prev_s;

    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) {                       /* single pixel domain */
        *i0 = (signed char)(nbd | -128);
        if (n >= cap) return -1;
        pts[2 * n] = px + offx; pts[2 * n + 1] = py + offy; n++;
        return n;
    }
    i3 = i0;
    prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        s = s < 15 ? s : 15;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        /* check "right" bound */
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1)
            *i3 = nbd;

        if (s != prev_s || !approx_simple) {
            if (n >= cap) return -1;
            pts[2 * n] = px + offx; pts[2 * n + 1] = py + offy; n++;
            prev_s = s;
        }
        px += code_dx[s];
        py += code_dy[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return n;
}

/* img: h*w bytes (non-zero = foreground).  pts: (x,y) pairs; offsets[ncont+1].
 * Returns the number of contours, -1 on capacity overflow. */
int ubdo_find_external_contours(const uint8_t *img, int h, int w, int approx_simple,
                                int *pts, int pts_cap, int *offsets, int cont_cap)
{
    int step = w + 2, H = h + 2;
    signed char *buf = (signed char *)calloc((size_t)step * H, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            buf[(y + 1) * step + x + 1] = img[y * w + x] ? 1 : 0;
    int ncont = 0, npts = 0;
    offsets[0] = 0;
    /* scanner: rows 1..H-2, x from 1..step-2 (cvStartFindContours sets pt=(1,1),
     * img = img0 + step, width = size.width-1, height = size.height-1) */
    for (int y = 1; y < H - 1; y++) {
        signed char *row = buf + y * step;
        int prev = row[0];      /* frame pixel = 0 */
        int lnbd_x = 0;         /* lnbd = (0, y): frame pixel */
        int x = 1;
        for (; x < step - 1;) {
            int p = 0;
            for (; x < step - 1 && (p = row[x]) == prev; x++) ;
            if (x >= step - 1) break;
            {
                int is_hole = 0;
                if (!(prev == 0 && p == 1)) {       /* not an outer-border start */
                    if (p != 0 || prev < 1) goto resume_scan;
                    if (prev & -2) lnbd_x = x - 1;
                    is_hole = 1;
                }
                /* mode == CV_RETR_EXTERNAL */
                if (is_hole || row[lnbd_x] > 0) goto resume_scan;
                {
                    if (ncont >= cont_cap) { free(buf); return -1; }
                    int n = fetch_contour(buf, step, x, y, -1, -1, approx_simple,
                                          pts + 2 * npts, pts_cap - npts);
                    if (n < 0) { free(buf); return -1; }
                    npts += n;
                    offsets[++ncont] = npts;
                    lnbd_x = x;
                    p = row[x];                      /* scanner re-reads the now marked pixel */
                }
            resume_scan:
                prev = p;
                if (prev & -2) lnbd_x = x;
            }
            x++;
        }
    }
    free(buf);
    return ncont;
}

/* cv::contourArea(contour, oriented=false) for integer points */
double ubdo_contour_area(const int *pts, int n)
{
    if (n == 0) return 0.0;
    double a00 = 0;
    float prevx = (float)pts[2 * (n - 1)], prevy = (float)pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        float x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
        a00 += (double)prevx * y - (double)prevy * x;
        prevx = x; prevy = y;
    }
    return fabs(a00 * 0.5);
}

/* ------------------------------------------------------------------ convex hull
 * cv::convexHull(points, hull, clockwise=true, returnPoints=true) for int points.
 */
typedef struct { int x, y; } pt_t;

static int cmp_pts(const void *a, const void *b)
{
    const pt_t *p = *(const pt_t *const *)a, *q = *(const pt_t *const *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}
#define SGN(v) (((v) > 0) - ((v) < 0))

static int sklansky(pt_t **array, int start, int end, int *stack, int nsign, int sign2)
{
    int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;

    if (start == end || (array[start]->x == array[end]->x && array[start]->y == array[end]->y)) {
        stack[0] = start;
        return 1;
    }
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        int cury = array[pcur]->y, nexty = array[pnext]->y, by = nexty - cury;
        if (SGN(by) != nsign) {
            int ax = array[pcur]->x - array[pprev]->x;
            int bx = array[pnext]->x - array[pcur]->x;
            int ay = cury - array[pprev]->y;
            int convexity = ay * bx - ax * by;
            if (SGN(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pcur = pnext; pnext += incr;
                stack[stacksize] = pnext; stacksize++;
            } else {
                if (pprev == start) {
                    pcur = pnext; stack[1] = pcur;
                    pnext += incr; stack[2] = pnext;
                } else {
                    stack[stacksize - 2] = pnext;
                    pcur = pprev;
                    pprev = stack[stacksize - 4];
                    stacksize--;
                }
            }
        } else {
            pnext += incr;
            stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

/* hull_out: (x,y) pairs, capacity n.  Returns hull size. */
int ubdo_convex_hull(const int *pts, int total, int *hull_out)
{
    if (total == 0) return 0;
    pt_t *data0 = (pt_t *)malloc(sizeof(pt_t) * total);
    pt_t **pointer = (pt_t **)malloc(sizeof(pt_t *) * total);
    int *stack = (int *)malloc(sizeof(int) * (total + 2) * 2);
    int *hullbuf = (int *)malloc(sizeof(int) * (total + 2));
    int nout = 0, miny_ind = 0, maxy_ind = 0, i;
    for (i = 0; i < total; i++) { data0[i].x = pts[2 * i]; data0[i].y = pts[2 * i + 1]; pointer[i] = &data0[i]; }
    qsort(pointer, total, sizeof(pt_t *), cmp_pts);
    for (i = 1; i < total; i++) {
        int y = pointer[i]->y;
        if (pointer[miny_ind]->y > y) miny_ind = i;
        if (pointer[maxy_ind]->y < y) maxy_ind = i;
    }
    if (pointer[0]->x == pointer[total - 1]->x && pointer[0]->y == pointer[total - 1]->y) {
        hullbuf[nout++] = 0;
    } else {
        int *tl_stack = stack;
        int tl_count = sklansky(pointer, 0, maxy_ind, tl_stack, -1, 1);
        int *tr_stack = stack + tl_count;
        int tr_count = sklansky(pointer, total - 1, maxy_ind, tr_stack, -1, -1);
        /* clockwise == true: no swap of the upper halves */
        for (i = 0; i < tl_count - 1; i++) hullbuf[nout++] = (int)(pointer[tl_stack[i]] - data0);
        for (i = tr_count - 1; i > 0; i--) hullbuf[nout++] = (int)(pointer[tr_stack[i]] - data0);
        int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;

        int *bl_stack = stack;
        int bl_count = sklansky(pointer, 0, miny_ind, bl_stack, 1, -1);
        int *br_stack = stack + bl_count;
        int br_count = sklansky(pointer, total - 1, miny_ind, br_stack, 1, 1);
        {   /* clockwise: swap the lower halves */
            int *t = bl_stack; bl_stack = br_stack; br_stack = t;
            int c = bl_count; bl_count = br_count; br_count = c;
        }
        if (stop_idx >= 0) {
            int check_idx = bl_count > 2 ? bl_stack[1]
                          : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
            if (check_idx == stop_idx ||
                (check_idx >= 0 && pointer[check_idx]->x == pointer[stop_idx]->x &&
                 pointer[check_idx]->y == pointer[stop_idx]->y)) {
                bl_count = bl_count < 2 ? bl_count : 2;
                br_count = br_count < 2 ? br_count : 2;
            }
        }
        for (i = 0; i < bl_count - 1; i++) hullbuf[nout++] = (int)(pointer[bl_stack[i]] - data0);
        for (i = br_count - 1; i > 0; i--) hullbuf[nout++] = (int)(pointer[br_stack[i]] - data0);
    }
    /* the stacks hold indices into the SORTED pointer array */
    for (i = 0; i < nout; i++) {
        /* hullbuf already maps to data0 indices except the degenerate single-point case */
        int idx = hullbuf[i];
        if (nout == 1 && total > 0 && idx == 0) idx = (int)(pointer[0] - data0);
        hull_out[2 * i] = data0[idx].x; hull_out[2 * i + 1] = data0[idx].y;
    }
    free(data0); free(pointer); free(stack); free(hullbuf);
    return nout;
}

/* ------------------------------------------------------------------ rotating calipers
 * rotatingCalipers(points, n, CALIPERS_MINAREARECT, out[6]) -- float32 throughout,
 * exactly the operation order of rotcalipers.cpp.
 */
static void rotating_calipers_minarearect(const float *px, const float *py, int n, float *out)
{
    float minarea = FLT_MAX;
    float buf_f[7] = {0}; int buf_i0 = 0, buf_i5 = 0;
    float *inv_vect_length = (float *)malloc(sizeof(float) * n * 3);
    float *vx = inv_vect_length + n, *vy = vx + n;
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4] = {-1, -1, -1, -1};
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    float pt0x = px[0], pt0y = py[0];
    int i, k;
    left_x = right_x = pt0x;
    top_y = bottom_y = pt0y;
    for (i = 0; i < n; i++) {
        double dx, dy;
        if (pt0x < left_x) left_x = pt0x, left = i;
        if (pt0x > right_x) right_x = pt0x, right = i;
        if (pt0y > top_y) top_y = pt0y, top = i;
        if (pt0y < bottom_y) bottom_y = pt0y, bottom = i;
        int nx = (i + 1 < n) ? i + 1 : 0;
        float ptx = px[nx], pty = py[nx];
        dx = ptx - pt0x; dy = pty - pt0y;
        vx[i] = (float)dx; vy[i] = (float)dy;
        inv_vect_length[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0x = ptx; pt0y = pty;
    }
    {   /* hull orientation */
        double ax = vx[n - 1], ay = vy[n - 1];
        for (i = 0; i < n; i++) {
            double bx = vx[i], by = vy[i];
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    for (k = 0; k < n; k++) {
        float dp[4] = {
            +base_a * vx[seq[0]] + base_b * vy[seq[0]],
            -base_b * vx[seq[1]] + base_a * vy[seq[1]],
            -base_a * vx[seq[2]] - base_b * vy[seq[2]],
            +base_b * vx[seq[3]] - base_a * vy[seq[3]],
        };
        float maxcos = dp[0] * inv_vect_length[seq[0]];
        int main_element = 0;
        for (i = 1; i < 4; ++i) {
            float cosalpha = dp[i] * inv_vect_length[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vx[pindex] * inv_vect_length[pindex];
            float lead_y = vy[pindex] * inv_vect_length[pindex];
            switch (main_element) {
            case 0: base_a = lead_x;  base_b = lead_y;  break;
            case 1: base_a = lead_y;  base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            case 3: base_a = -lead_y; base_b = lead_x;  break;
            }
        }
        seq[main_element] += 1;
        seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
        {
            float dx = px[seq[1]] - px[seq[3]];
            float dy = py[seq[1]] - py[seq[3]];
            float width = dx * base_a + dy * base_b;
            dx = px[seq[2]] - px[seq[0]];
            dy = py[seq[2]] - py[seq[0]];
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_i0 = seq[3];
                buf_f[1] = base_a; buf_f[2] = width; buf_f[3] = base_b; buf_f[4] = height;
                buf_i5 = seq[0];
                buf_f[6] = area;
            }
        }
    }
    {
        float A1 = buf_f[1], B1 = buf_f[3];
        float A2 = -buf_f[3], B2 = buf_f[1];
        float C1 = A1 * px[buf_i0] + py[buf_i0] * B1;
        float C2 = A2 * px[buf_i5] + py[buf_i5] * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float ox = (C1 * B2 - C2 * B1) * idet;
        float oy = (A1 * C2 - A2 * C1) * idet;
        out[0] = ox; out[1] = oy;
        out[2] = A1 * buf_f[2]; out[3] = B1 * buf_f[2];
        out[4] = A2 * buf_f[4]; out[5] = B2 * buf_f[4];
    }
    free(inv_vect_length);
}

/* cv::minAreaRect(int points) -> (cx, cy, w, h, angle_deg) */
void ubdo_min_area_rect(const int *pts, int npts, float *rect5)
{
    int *hull = (int *)malloc(sizeof(int) * 2 * (npts > 0 ? npts : 1));
    int n = ubdo_convex_hull(pts, npts, hull);
    float cx = 0, cy = 0, bw = 0, bh = 0, angle = 0;
    float *hx = (float *)malloc(sizeof(float) * 2 * (n > 0 ? n : 1)), *hy = hx + n;
    for (int i = 0; i < n; i++) { hx[i] = (float)hull[2 * i]; hy[i] = (float)hull[2 * i + 1]; }
    if (n > 2) {
        float out[6];
        rotating_calipers_minarearect(hx, hy, n, out);
        cx = out[0] + (out[2] + out[4]) * 0.5f;
        cy = out[1] + (out[3] + out[5]) * 0.5f;
        bw = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        bh = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (n == 2) {
        cx = (hx[0] + hx[1]) * 0.5f;
        cy = (hy[0] + hy[1]) * 0.5f;
        double dx = hx[1] - hx[0], dy = hy[1] - hy[0];
        bw = (float)sqrt(dx * dx + dy * dy);
        bh = 0;
        angle = (float)atan2(dy, dx);
    } else if (n == 1) {
        cx = hx[0]; cy = hy[0];
    }
    angle = (float)(angle * 180 / CV_PI);
    rect5[0] = cx; rect5[1] = cy; rect5[2] = bw; rect5[3] = bh; rect5[4] = angle;
    free(hull); free(hx);
}

/* cv::boxPoints = RotatedRect::points */
void ubdo_box_points(const float *rect5, float *pt8)
{
    float cx = rect5[0], cy = rect5[1], sw = rect5[2], sh = rect5[3], angle = rect5[4];
    double _angle = angle * CV_PI / 180.;
    float b = (float)cos(_angle) * 0.5f;
    float a = (float)sin(_angle) * 0.5f;
    pt8[0] = cx - a * sh - b * sw;
    pt8[1] = cy + b * sh - a * sw;
    pt8[2] = cx + a * sh - b * sw;
    pt8[3] = cy - b * sh - a * sw;
    pt8[4] = 2 * cx - pt8[0];
    pt8[5] = 2 * cy - pt8[1];
    pt8[6] = 2 * cx - pt8[2];
    pt8[7] = 2 * cy - pt8[3];
}

/* np.round (half to even) of float32 value*scale, astype(int) */
static int round_half_even_f32(float v)
{
    return (int)nearbyintf(v);      /* default rounding mode = to nearest even */
}

/* ------------------------------------------------------------------ polygon fill
 * cv2.drawContours(mask, [cnt], -1, 1, thickness=-1): polygon outline (Line between
 * consecutive contour points; segments of a traced border are exact 8-direction
 * runs) plus even-odd scanline interior (FillEdgeCollection).
 */
static void fill_contour(const int *pts, int n, int h, int w, uint8_t *mask)
{
    memset(mask, 0, (size_t)h * w);
    for (int i = 0; i < n; i++) {
        int x0 = pts[2 * i], y0 = pts[2 * i + 1];
        int x1 = pts[2 * ((i + 1) % n)], y1 = pts[2 * ((i + 1) % n) + 1];
        int dx = SGN(x1 - x0), dy = SGN(y1 - y0);
        int steps = abs(x1 - x0) > abs(y1 - y0) ? abs(x1 - x0) : abs(y1 - y0);
        for (int s = 0; s <= steps; s++) {
            int x = x0 + s * dx, y = y0 + s * dy;
            if (x >= 0 && x < w && y >= 0 && y < h) mask[y * w + x] = 1;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            if (mask[y * w + x]) continue;
            int cross = 0;
            for (int i = 0; i < n; i++) {
                int x0 = pts[2 * i], y0 = pts[2 * i + 1];
                int x1 = pts[2 * ((i + 1) % n)], y1 = pts[2 * ((i + 1) % n) + 1];
                if (y0 == y1) continue;
                int ylo = y0 < y1 ? y0 : y1, yhi = y0 < y1 ? y1 : y0;
                if (y < ylo || y >= yhi) continue;
                double xc = x0 + (double)(y - y0) * (x1 - x0) / (double)(y1 - y0);
                if (xc > x) cross++;
            }
            if (cross & 1) mask[y * w + x] = 1;
        }
    }
}

/* ------------------------------------------------------------------ full postprocess
 * SegmapManager.postprocess (segmap_manager.py:41-69) for one image.
 * seg: h*w bytes; class_logits: h*w*n_cls float32 or NULL.
 * quads: [cap][8] int32 in cv2's return order (reverse discovery order);
 * cls: [cap] int32 (argmax of the mean softmax over the filled contour) or untouched.
 * areas (optional, may be NULL): contourArea per kept object.
 * Returns the number of objects, -1 on overflow.
 */
int ubdo_postprocess(const uint8_t *seg, int h, int w, const float *class_logits, int n_cls,
                     int scale, double min_area, int32_t *quads, int32_t *cls, double *areas, int cap)
{
    int pts_cap = 4 * (h + 2) * (w + 2) + 16, cont_cap = h * w / 2 + 16;
    int *pts = (int *)malloc(sizeof(int) * 2 * pts_cap);
    int *offs = (int *)malloc(sizeof(int) * (cont_cap + 1));
    int nc = ubdo_find_external_contours(seg, h, w, 1, pts, pts_cap, offs, cont_cap);
    int nout = 0;
    uint8_t *mask = (class_logits && n_cls > 0) ? (uint8_t *)malloc((size_t)h * w) : NULL;
    if (nc < 0) { free(pts); free(offs); free(mask); return -1; }
    for (int ci = nc - 1; ci >= 0; ci--) {          /* cv2 order: last found first */
        const int *cp = pts + 2 * offs[ci];
        int n = offs[ci + 1] - offs[ci];
        double area = ubdo_contour_area(cp, n);
        if (!(area > min_area)) continue;
        if (nout >= cap) { nout = -1; break; }
        float rect[5], box[8];
        ubdo_min_area_rect(cp, n, rect);
        ubdo_box_points(rect, box);
        for (int j = 0; j < 8; j++) quads[8 * nout + j] = round_half_even_f32(box[j] * (float)scale);
        if (areas) areas[nout] = area;
        if (mask) {
            fill_contour(cp, n, h, w, mask);
            double best = -1; int besti = 0;
            double *acc = (double *)calloc(n_cls, sizeof(double));
            long cnt = 0;
            for (int p = 0; p < h * w; p++) {
                if (!mask[p]) continue;
                const float *lg = class_logits + (size_t)p * n_cls;
                float mx = lg[0];
                for (int c = 1; c < n_cls; c++) mx = lg[c] > mx ? lg[c] : mx;
                double s = 0;
                for (int c = 0; c < n_cls; c++) s += exp((double)(lg[c] - mx));
                for (int c = 0; c < n_cls; c++) acc[c] += exp((double)(lg[c] - mx)) / s;
                cnt++;
            }
            for (int c = 0; c < n_cls; c++) if (acc[c] > best) { best = acc[c]; besti = c; }
            (void)cnt;
            cls[nout] = besti;
            free(acc);
        }
        nout++;
    }
    free(pts); free(offs); free(mask);
    return nout;
}

/* mask of one contour, exposed for tests */
void ubdo_fill_contour(const int *pts, int n, int h, int w, uint8_t *mask) { fill_contour(pts, n, h, w, mask); }
