// Rasteriser of the freeform-mask primitives: the device half of lib/data_factory/ds_ffhq.py:145-217
// (`RandomBrush` / `RandomMask`).  The host (sh-gan_amd/masks.py) makes the random draws in the reference's order and turns
// every mask into a list of 8-word integer records; this kernel draws them exactly as Pillow would:
//   RECT  (0, x0, x1, y0, y1)                     rectangle punched into the keep layer (already clipped, inclusive)
//   DISC  (1, cx, cy, half)                       ImageDraw.ellipse((cx-h, cy-h, cx+h, cy+h), fill=1): row spans from a table
//   QUAD  (2, ymin, ymax) + 4 x EDGE              one segment of ImageDraw.line(width=..): Pillow's ImagingDrawWideLine quad,
//   EDGE  (3, x0, y0, ymin, ymax, dx, xmin, xmax)   filled by its polygon_generic scan-line rule (float32 slopes, ROUND_UP /
//                                                   ROUND_DOWN span ends, corner joining, horizontal edges as plain lines)
//   POINT (4, x, y)                               zero-length segment
//   SEG   (5, x0, y0, x1, y1, width, hypot lo, hypot hi)   the same segment with the quad built HERE (the host only supplies
//                                                   hypot(dx, dy) as a double: libm's value, not the device's)
// mask = keep layer AND NOT (brush layer, flipped up-down / left-right per the two flip flags), 1 = keep, 0 = hole.
// One thread per image row: the row lives in registers as a bit vector (<= 512 columns), records are read with
// wave-uniform addresses.  Integer / float32 arithmetic only, every float product and sum rounded separately (no FMA):
// results are bit-exact with the reference's masks (tests/golden/integer_paths.npz).
#include "shg_common.h"
#include <math.h>

#define MR_WORDS 16          // 512 columns
#define MR_REC 8

struct MaskParams {
    const int* rec;          // [total][8]
    const int* off;          // [B+1] record offsets
    const int* flips;        // [B][2]
    const int* disc;         // [max_half+1][2*max_half+1][2] (l, r) relative to cx - half; l > r = empty row
    int max_half;
    float* mask;             // [B,1,s,s]
    int* holes;              // [B] number of zero pixels
    int B, s;
};

__device__ __forceinline__ void mr_span(unsigned (&row)[MR_WORDS], int x0, int x1, int s) {
    // Pillow's hline8 clipping: [x0, x1] inclusive, clipped to the canvas
    if (x0 < 0) x0 = 0;
    if (x1 >= s) x1 = s - 1;
    if (x0 > x1) return;
#pragma unroll
    for (int w = 0; w < MR_WORDS; ++w) {
        const int lo = max(x0, 32 * w), hi = min(x1, 32 * w + 31);
        if (lo <= hi) {
            const int nb = hi - lo + 1;
            const unsigned m = nb == 32 ? 0xffffffffu : ((1u << nb) - 1u);
            row[w] |= m << (lo - 32 * w);
        }
    }
}

__device__ __forceinline__ int mr_round_up(float f) {       // ROUND_UP of Pillow's Draw.c
    return (int)(f >= 0.0f ? floor((double)(f + 0.5f)) : -floor((double)(fabsf(f) + 0.5f)));
}
__device__ __forceinline__ int mr_round_down(float f) {     // ROUND_DOWN
    return (int)(f >= 0.0f ? ceil((double)(f - 0.5f)) : -ceil((double)(fabsf(f) - 0.5f)));
}

struct MrEdge { int x0, y0, ymin, ymax, xmin, xmax; float dx; };

__device__ __forceinline__ float mr_ex(const MrEdge& e, int y) {
#pragma clang fp contract(off)
    const float a = (float)(y - e.y0) * e.dx;
    return a + (float)e.x0;
}

__device__ __forceinline__ int mr_round_up_d(double f) { return (int)(f >= 0.0 ? floor(f + 0.5) : -floor(fabs(f) + 0.5)); }
__device__ __forceinline__ int mr_round_down_d(double f) { return (int)(f >= 0.0 ? ceil(f - 0.5) : -ceil(fabs(f) - 0.5)); }

__device__ __forceinline__ void mr_add_edge(MrEdge& e, int x0, int y0, int x1, int y1) {      // Pillow's add_edge
    e.xmin = min(x0, x1); e.xmax = max(x0, x1);
    e.ymin = min(y0, y1); e.ymax = max(y0, y1);
    e.dx = y0 == y1 ? 0.f : ((float)(x1 - x0)) / (float)(y1 - y0);
    e.x0 = x0; e.y0 = y0;
}

// Pillow's ImagingDrawWideLine: the quad of a segment of the given width; returns the clamped scan range in ymin / ymax
__device__ __forceinline__ void mr_wide_line(MrEdge (&e)[4], int x0, int y0, int x1, int y1, int width, double big, int s, int& ymin,
                                             int& ymax) {
    const int dx = x1 - x0, dy = y1 - y0;
    const double small = (width - 1) / 2.0;
    const double rmax = (double)mr_round_up_d(small) / big, rmin = (double)mr_round_down_d(small) / big;
    const int dxmin = mr_round_down_d(rmin * dy), dxmax = mr_round_down_d(rmax * dy);
    const int dymin = mr_round_up_d(rmin * dx), dymax = mr_round_up_d(rmax * dx);
    const int vx[4] = {x0 - dxmin, x1 - dxmin, x1 + dxmax, x0 + dxmax};
    const int vy[4] = {y0 + dymax, y1 + dymax, y1 - dymin, y0 - dymin};
#pragma unroll
    for (int k = 0; k < 4; ++k) mr_add_edge(e[k], vx[k], vy[k], vx[(k + 1) & 3], vy[(k + 1) & 3]);
    ymin = max(min(min(vy[0], vy[1]), min(vy[2], vy[3])), 0);
    ymax = min(max(max(vy[0], vy[1]), max(vy[2], vy[3])), s);
}

// Pillow's polygon_generic restricted to one scan line y of a 4-edge polygon; [ymin, YMAX] is the clamped scan range.
__device__ __forceinline__ void mr_quad_row(unsigned (&row)[MR_WORDS], const MrEdge (&e)[4], int y, int YMAX, int s) {
#pragma clang fp contract(off)
    // edge table = the non-horizontal edges in order; tix[t] = edge index of table slot t
    int tix[4], nt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e[k].ymin == e[k].ymax) { if (e[k].ymin == y) mr_span(row, e[k].xmin, e[k].xmax, s); }
        else tix[nt++] = k;
    }
    float xx[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) xx[t] = INFINITY;
    int j = 0;
    auto put = [&](int idx, float v) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t == idx) xx[t] = v;
    };
    auto get = [&](int idx) __attribute__((always_inline)) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t == idx) v = xx[t];
        return v;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= nt) break;
        MrEdge cur = e[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (tix[i] == k) cur = e[k];
        if (y < cur.ymin || y > cur.ymax) continue;
        const float xc = mr_ex(cur, y);
        put(j, xc); ++j;
        if (y == cur.ymax && y < YMAX) { put(j, xc); ++j; }
        else if (cur.dx != 0.f && (j & 1) == 1 && roundf(xc) == xc) {
            for (int k = 0; k < i; ++k) {
                MrEdge oth = e[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (tix[k] == q) oth = e[q];
                if ((cur.dx > 0.f && oth.dx <= 0.f) || (cur.dx < 0.f && oth.dx >= 0.f)) continue;
                if (xc == mr_ex(oth, y)) {
                    const int offy = y == YMAX ? -1 : 1;
                    const float a = mr_ex(cur, y + offy), b = mr_ex(oth, y + offy);
                    float v;
                    if (y == cur.ymax) v = cur.dx > 0.f ? fmaxf(a, b) + 1.f : fminf(a, b) - 1.f;
                    else v = cur.dx > 0.f ? fminf(a, b) : fmaxf(a, b) + 1.f;
                    put(k, v);                     // (Pillow indexes the intersection list with the edge-table index here)
                    break;
                }
            }
        }
    }
    // sort (unused slots hold +inf): odd-even transposition network on 8 values
#pragma unroll
    for (int pass = 0; pass < 8; ++pass)
#pragma unroll
        for (int t = pass & 1; t + 1 < 8; t += 2) {
            const float lo = fminf(xx[t], xx[t + 1]), hi = fmaxf(xx[t], xx[t + 1]);
            xx[t] = lo; xx[t + 1] = hi;
        }
    int x_pos = j == 0 ? -1 : 0;
#pragma unroll
    for (int i = 1; i < 8; i += 2) {
        if (i >= j) break;
        const int x_end = mr_round_down(xx[i]);
        if (x_end < x_pos) continue;
        int x_start = mr_round_up(xx[i - 1]);
        if (x_pos > x_start) {
            x_start = x_pos;
            if (x_end < x_start) continue;
        }
        mr_span(row, x_start, x_end, s);
        x_pos = x_end + 1;
    }
    (void)get;
}

__global__ __launch_bounds__(256) void mask_raster_kernel(const MaskParams p) {
    const int b = blockIdx.x;
    const int y = blockIdx.y * 256 + threadIdx.x;                // output row
    const int s = p.s;
    const bool live = y < s;
    const int f0 = p.flips[2 * b], f1 = p.flips[2 * b + 1];
    const int yb = f0 ? s - 1 - y : y;                           // brush-layer row that lands on output row y
    unsigned keep[MR_WORDS], brush[MR_WORDS];
#pragma unroll
    for (int w = 0; w < MR_WORDS; ++w) { keep[w] = 0u; brush[w] = 0u; }      // keep[] collects the PUNCHED columns
    const int r0 = p.off[b], r1 = p.off[b + 1];
    const int dstride = 2 * p.max_half + 1;
    for (int r = r0; r < r1;) {
        const int* q = p.rec + (long)r * MR_REC;                 // wave-uniform address
        const int type = q[0];
        if (type == 0) {
            if (live && y >= q[3] && y <= q[4]) mr_span(keep, q[1], q[2], s);
            r += 1;
        } else if (type == 1) {
            const int cx = q[1], cy = q[2], h = q[3];
            const int j = yb - (cy - h);
            if (live && j >= 0 && j <= 2 * h) {
                const int* t = p.disc + ((long)h * dstride + j) * 2;
                const int l = t[0], rr = t[1];
                if (l <= rr) mr_span(brush, cx - h + l, cx - h + rr, s);
            }
            r += 1;
        } else if (type == 4) {
            if (live && q[2] == yb) mr_span(brush, q[1], q[1], s);
            r += 1;
        } else if (type == 5) {
            const int x0 = q[1], y0 = q[2], x1 = q[3], y1 = q[4], width = q[5];
            // no row further than `width` from the segment's row range can be touched: skip the geometry for those
            if (live && yb >= min(y0, y1) - width && yb <= max(y0, y1) + width) {
                if (x0 == x1 && y0 == y1) { if (y0 == yb) mr_span(brush, x0, x0, s); }
                else {
                    MrEdge e[4];
                    int ymin, ymax;
                    const double big = __hiloint2double(q[7], q[6]);
                    mr_wide_line(e, x0, y0, x1, y1, width, big, s, ymin, ymax);
                    if (yb >= ymin && yb <= ymax) mr_quad_row(brush, e, yb, ymax, s);
                }
            }
            r += 1;
        } else {                                                 // QUAD header + 4 EDGE records
            const int ymin = q[1], ymax = q[2];
            if (live && yb >= ymin && yb <= ymax) {
                MrEdge e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int* eq = q + (k + 1) * MR_REC;
                    e[k].x0 = eq[1]; e[k].y0 = eq[2]; e[k].ymin = eq[3]; e[k].ymax = eq[4];
                    e[k].dx = __int_as_float(eq[5]); e[k].xmin = eq[6]; e[k].xmax = eq[7];
                }
                mr_quad_row(brush, e, yb, ymax, s);
            }
            r += 5;
        }
    }
    if (!live) return;
    const int nw = s >> 5;
    int holes = 0;
    float* out = p.mask + ((long)b * s + y) * s;
#pragma unroll
    for (int w = 0; w < MR_WORDS; ++w) {
        if (w >= nw) break;
        unsigned bw = brush[w];
        if (f1) {                                                // left-right flip of the brush layer: column x <- s-1-x
            unsigned src = 0u;
#pragma unroll
            for (int t = 0; t < MR_WORDS; ++t) if (t == nw - 1 - w) src = brush[t];
            bw = __brev(src);
        }
        const unsigned hole = keep[w] | bw;                      // punched by a rectangle or painted by the brush
        holes += __popc(hole);
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            float4 v;
            v.x = (hole >> (4 * k4 + 0)) & 1u ? 0.f : 1.f;
            v.y = (hole >> (4 * k4 + 1)) & 1u ? 0.f : 1.f;
            v.z = (hole >> (4 * k4 + 2)) & 1u ? 0.f : 1.f;
            v.w = (hole >> (4 * k4 + 3)) & 1u ? 0.f : 1.f;
            *reinterpret_cast<float4*>(out + 32 * w + 4 * k4) = v;
        }
    }
    atomicAdd(p.holes + b, holes);
}

// records [total][8] int32, offsets [B+1], flips [B][2], disc table [max_half+1][2*max_half+1][2] (all device memory);
// mask [B,1,s,s] float32 out (1 = keep, 0 = hole), holes [B] int32 must be zero on entry (the kernel adds the hole counts).
extern "C" int shg_mask_raster_f32(const int* records, const int* offsets, const int* flips, const int* disc_table, int max_half,
                                   float* mask, int* holes, int B, int s, void* stream) {
    SHG_CHECK_ARG(records && offsets && flips && disc_table && mask && holes, "mask_raster: null pointer");
    SHG_CHECK_ARG(B >= 1 && B <= 65535 && s >= 32 && s <= 512 && s % 32 == 0, "mask_raster: s must be a multiple of 32 in [32, 512]");
    SHG_CHECK_ARG((reinterpret_cast<uintptr_t>(mask) & 15) == 0, "mask_raster: mask must be 16-byte aligned");
    MaskParams p;
    p.rec = records; p.off = offsets; p.flips = flips; p.disc = disc_table; p.max_half = max_half;
    p.mask = mask; p.holes = holes; p.B = B; p.s = s;
    hipLaunchKernelGGL(mask_raster_kernel, dim3(B, shg_cdiv(s, 256)), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
