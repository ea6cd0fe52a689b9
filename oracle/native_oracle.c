/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's one native op and of the integer
 * paths around the generator.  Built by oracle/Makefile into oracle/_build/liboracle.so; loaded only by
 * tests/ (never by the product, never linked into libshgan_hip.so).
 *
 *   orc_upfirdn2d_f32   : the generic gather of lib/model_zoo/stylegan_utils/upfirdn2d.cu:29-92
 *                         (upfirdn2d_kernel_large): per output pixel, the receptive field is found with
 *                         floor division, taps are accumulated in float, the sum is scaled by gain.
 *                         Output size rule: upfirdn2d.cpp:32-33.  The reference's CUDA source cannot be
 *                         compiled here (no nvcc), so this follows its text; it is pinned by the golden
 *                         vectors of tests/golden/upfirdn2d.npz, which come from the reference's own
 *                         `_upfirdn2d_ref` (upfirdn2d.py:98-138).
 *   orc_composite_u8    : lib/experiments/shgan_default.py:257-262 (separately rounded fp32 ops, truncation).
 *   orc_sampler_indices : lib/data_factory/common/ds_sampler.py:58-68 with shuffle=False.
 */
#include <stdint.h>
#include <stdlib.h>

static int floor_div(int a, int b) { /* upfirdn2d.cu:20-24 */
    int t = 1 - a / b;
    return (a + t * b) / b - t;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

int orc_upfirdn2d_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy,
                      int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain) {
    const int OW = (W * upx + padx0 + padx1 - fw + downx) / downx;
    const int OH = (H * upy + pady0 + pady1 - fh + downy) / downy;
    if (OW < 1 || OH < 1) return -1;
    for (int nc = 0; nc < N * C; ++nc) {
        const float* xp = x + (size_t)nc * H * W;
        float* yp = y + (size_t)nc * OH * OW;
        for (int outY = 0; outY < OH; ++outY) {
            /* Y receptive field (upfirdn2d.cu:43-48) */
            const int midY = outY * downy + upy - 1 - pady0;
            const int inY = imin(imax(floor_div(midY, upy), 0), H);
            const int h = imin(imax(floor_div(midY + fh, upy), 0), H) - inY;
            int filterY = midY + fh - (inY + 1) * upy;
            if (flip) filterY = fh - 1 - filterY;
            for (int outX = 0; outX < OW; ++outX) {
                const int midX = outX * downx + upx - 1 - padx0;
                const int inX = imin(imax(floor_div(midX, upx), 0), W);
                const int w = imin(imax(floor_div(midX + fw, upx), 0), W) - inX;
                int filterX = midX + fw - (inX + 1) * upx;
                if (flip) filterX = fw - 1 - filterX;
                const int stepX = flip ? upx : -upx, stepY = flip ? upy : -upy;
                float v = 0.f;
                for (int yy = 0; yy < h; ++yy)
                    for (int xx = 0; xx < w; ++xx)
                        v += xp[(size_t)(inY + yy) * W + inX + xx] * f[(filterY + yy * stepY) * fw + filterX + xx * stepX];
                yp[(size_t)outY * OW + outX] = v * gain;
            }
        }
    }
    return 0;
}

void orc_composite_u8(const float* x4, const float* img, uint8_t* out, int N, int HW) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < 3; ++c)
            for (int p = 0; p < HW; ++p) {
                const float m = x4[((size_t)n * 4) * HW + p] + 0.5f;
                volatile float a = x4[((size_t)n * 4 + 1 + c) * HW + p] * m;      /* volatile: no fused multiply-add */
                volatile float b = img[((size_t)n * 3 + c) * HW + p] * (1.0f - m);
                volatile float v = a + b;
                v = v * 127.5f;
                v = v + 127.5f;
                float r = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                out[((size_t)n * 3 + c) * HW + p] = (uint8_t)(int)r;
            }
}

/* returns the number of indices written (<= cap) */
int orc_sampler_indices(int n_items, int world, int rank, int extend, int64_t* out, int cap) {
    int per = n_items / world;
    if (extend && n_items != per * world) per += 1;
    const int total = per * world;
    int len = n_items;
    if (extend) {
        int extra = total - n_items;
        if (extra > n_items) extra = n_items;      /* list slicing never yields more than the list holds */
        if (extra < 0) extra = 0;
        len = n_items + extra;
    } else if (len > total) len = total;
    int cnt = 0;
    for (int i = rank; i < len; i += world) {
        const int idx = i < n_items ? i : i - n_items;
        if (cnt < cap) out[cnt] = idx;
        ++cnt;
    }
    return cnt;
}
