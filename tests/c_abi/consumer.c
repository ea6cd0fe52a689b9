/* A C consumer of libshgan_hip.so: no Python, no PyTorch -- device memory from the HIP runtime, the kernels through the C-ABI of
 * include/shgan_hip.h, the results checked against the plain-C oracle (oracle/native_oracle.c: upfirdn2d.cu:29-92 restated) and
 * against closed-form expectations.  This is what a non-Python host of the reference's operator boundary would link.
 *
 *   hipcc -x c tests/c_abi/consumer.c oracle/native_oracle.c -Iinclude -Lsh-gan_amd/lib -lshgan_hip -o consumer   (build: no GPU needed)
 *   LD_LIBRARY_PATH=sh-gan_amd/lib ./consumer                                                              (run: MI355X)
 * test infrastructure: uses oracle/ as the checker only. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "shgan_hip.h"

int orc_upfirdn2d_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy,
                      int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain);

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)
#define SHG(e) do { int _r = (e); if (_r != 0) { printf("shg error %d (%s) at line %d\n", _r, shg_last_error(), __LINE__); return 3; } } while (0)

/* IEEE half <-> float for values that are exactly representable (small integers, powers of two): enough for a consumer without __fp16 */
static uint16_t f2h(float v) {
    union { float f; uint32_t u; } c; c.f = v;
    const uint32_t s = (c.u >> 16) & 0x8000u, e = (c.u >> 23) & 0xffu, m = c.u & 0x7fffffu;
    if (e == 0) return (uint16_t)s;
    return (uint16_t)(s | ((e - 127 + 15) << 10) | (m >> 13));
}
static float h2f(uint16_t h) {
    union { float f; uint32_t u; } c;
    const uint32_t s = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    c.u = e == 0 ? s : (s | ((e - 15 + 127) << 23) | (m << 13));
    return c.f;
}

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xffff) / 65536.f - 0.5f; }

int main(void) {
    char arch[64];
    const int cus = shg_device_info(0, arch, sizeof arch);
    printf("abi %d, device 0: %s, %d CUs\n", shg_abi_version(), arch, cus);
    if (cus <= 0) { printf("no device\n"); return 1; }
    unsigned seed = 7u;
    int fails = 0;
    /* 1. upfirdn2d, three geometries: the generic gather kernel, the row-marching pad-2 FIR (host taps), x2 down-sampling */
    const float f1[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    float f[16], taps[8];
    for (int i = 0; i < 4; ++i) { taps[i] = f1[i]; taps[4 + i] = f1[i]; for (int j = 0; j < 4; ++j) f[i * 4 + j] = f1[i] * f1[j]; }
    struct { int N, C, H, W, up, down, p0, p1, sep; } cs[] = {{2, 3, 17, 23, 2, 1, 2, 1, 0}, {1, 5, 64, 64, 1, 1, 2, 2, 1}, {2, 4, 32, 64, 1, 2, 1, 1, 2}};
    for (int k = 0; k < 3; ++k) {
        const int N = cs[k].N, C = cs[k].C, H = cs[k].H, W = cs[k].W, up = cs[k].up, dn = cs[k].down, p0 = cs[k].p0, p1 = cs[k].p1;
        int OH, OW;
        SHG(shg_upfirdn2d_out_size(H, W, 4, 4, up, up, dn, dn, p0, p1, p0, p1, &OH, &OW));
        const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * C * OH * OW;
        float *hx = malloc(nx * 4), *hy = malloc(ny * 4), *ref = malloc(ny * 4), *dx, *dy, *df;
        for (size_t i = 0; i < nx; ++i) hx[i] = frand(&seed);
        CK(hipMalloc((void**)&dx, nx * 4)); CK(hipMalloc((void**)&dy, ny * 4)); CK(hipMalloc((void**)&df, 64));
        CK(hipMemcpy(dx, hx, nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(df, f, 64, hipMemcpyHostToDevice));
        if (cs[k].sep == 1) SHG(shg_fir_pad2_sep_f32(dx, taps, dy, N, C, H, W, 0, 0, 1.5f, NULL));
        else if (cs[k].sep == 2) SHG(shg_fir_resample2_sep_f32(dx, taps, dy, N, C, H, W, 1, 0, 1.5f, NULL));
        else SHG(shg_upfirdn2d_f32(dx, df, dy, N, C, H, W, 4, 4, up, up, dn, dn, p0, p1, p0, p1, 0, 1.5f, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, ny * 4, hipMemcpyDeviceToHost));
        orc_upfirdn2d_f32(hx, f, ref, N, C, H, W, 4, 4, up, up, dn, dn, p0, p1, p0, p1, 0, 1.5f);
        double err = 0, mx = 0;
        for (size_t i = 0; i < ny; ++i) { const double d = fabs(hy[i] - ref[i]); if (d > err) err = d; if (fabs(ref[i]) > mx) mx = fabs(ref[i]); }
        printf("upfirdn2d case %d: %dx%d -> %dx%d  rel err %.2e\n", k, H, W, OH, OW, err / mx);
        if (!(err / mx < 1e-5)) ++fails;
        CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(df)); free(hx); free(hy); free(ref);
    }
    /* 2. the error path: a null pointer must come back as a status + message, not a crash */
    if (shg_upfirdn2d_f32(NULL, NULL, NULL, 1, 1, 4, 4, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.f, NULL) == 0) { printf("null pointers accepted\n"); ++fails; }
    else printf("error path: \"%s\"\n", shg_last_error());
    /* 3. weight preparation + the Winograd F(4x4,3x3) convolution of a constant image with an all-ones kernel: interior = 9 * I */
    {
        const int NB = 1, I = 8, O = 64, H = 32, W = 128;
        const size_t nx = (size_t)NB * I * H * W, ny = (size_t)NB * O * H * W, nw = (size_t)O * I * 9;
        float *hx = malloc(nx * 4), *hw = malloc(nw * 4), *hs = malloc(O * 4), *hy = malloc(ny * 4), *dx, *dw, *ds, *dy, *du;
        for (size_t i = 0; i < nx; ++i) hx[i] = 1.f;
        for (size_t i = 0; i < nw; ++i) hw[i] = 1.f;
        for (int o = 0; o < O; ++o) hs[o] = 1.f;
        const long nu = shg_conv_wino4_weight_elems(O, I);
        CK(hipMalloc((void**)&dx, nx * 4)); CK(hipMalloc((void**)&dw, nw * 4)); CK(hipMalloc((void**)&ds, O * 4));
        CK(hipMalloc((void**)&dy, ny * 4)); CK(hipMalloc((void**)&du, (size_t)nu * 4));
        CK(hipMemcpy(dx, hx, nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw, nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(ds, hs, O * 4, hipMemcpyHostToDevice));
        if (!shg_conv2d_wino4_supported(NB, I, O, H, W)) { printf("wino4 geometry rejected\n"); ++fails; }
        SHG(shg_conv_weight_prep_wino4_f32(dw, ds, du, O, I, O, 0, NULL));
        SHG(shg_conv2d_wino4_f32(dx, du, dy, NB, I, O, O, H, W, NULL, NULL, NULL, NULL, 0, 0.f, 0, 0.f, 1.f, -1.f, NULL, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, ny * 4, hipMemcpyDeviceToHost));
        double err = 0;
        for (int o = 0; o < O; ++o)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int ny_ = (y > 0) + 1 + (y < H - 1), nx_ = (x > 0) + 1 + (x < W - 1);
                    const double d = fabs(hy[((size_t)o * H + y) * W + x] - (double)(I * ny_ * nx_));
                    if (d > err) err = d;
                }
        printf("conv2d_wino4 of ones: max abs err %.2e (values up to %d)\n", err, 9 * I);
        if (!(err < 1e-3)) ++fails;
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(ds)); CK(hipFree(dy)); CK(hipFree(du)); free(hx); free(hw); free(hs); free(hy);
    }
    /* 4. the fp16 route (NHWC halves): pack a 3x3 weight tensor in MFMA operand order, convolve (stride 1, pad 1, fp32 bias), then the
     *    weight gradient of the same geometry -- small integers, so every fp32 accumulation is exact and the check is equality */
    {
        const int N = 2, I = 32, O = 40, H = 9, W = 20, T = 9;
        const size_t nx = (size_t)N * H * W * I, ny = (size_t)N * H * W * O, nw = (size_t)T * O * I;
        uint16_t *hx = malloc(nx * 2), *hw = malloc(nw * 2), *hy = malloc(ny * 2);
        float *xf = malloc(nx * 4), *wf = malloc(nw * 4), *hb = malloc(O * 4), *hdw = malloc(nw * 4);
        for (size_t i = 0; i < nx; ++i) { xf[i] = (float)((int)(frand(&seed) * 8.f)); hx[i] = f2h(xf[i]); }          /* -4 .. 3 */
        for (size_t i = 0; i < nw; ++i) { wf[i] = (float)((int)(frand(&seed) * 4.f)); hw[i] = f2h(wf[i]); }          /* -2 .. 1 */
        for (int o = 0; o < O; ++o) hb[o] = (float)(o % 5 - 2);
        void *dx, *dw, *dwp, *dy, *dws; float *db, *ddw;
        const long npk = shg_conv2d_f16_packed_weight_elems(T, O, I);
        const size_t wsb = shg_conv2d_wgrad_f16_workspace_bytes(N, I, O, H, W, 3);
        CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dwp, (size_t)npk * 2)); CK(hipMalloc(&dy, ny * 2));
        CK(hipMalloc((void**)&db, O * 4)); CK(hipMalloc((void**)&ddw, nw * 4)); CK(hipMalloc(&dws, wsb));
        CK(hipMemcpy(dx, hx, nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw, nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb, O * 4, hipMemcpyHostToDevice));
        SHG(shg_conv2d_f16_pack_weight(dw, dwp, T, O, I, NULL));
        SHG(shg_conv2d_f16(dx, dwp, db, dy, N, I, O, H, W, 3, 1, 1, 0, 0, H, W, NULL));
        SHG(shg_conv2d_wgrad_f16(dx, dy, ddw, N, I, O, H, W, H, W, 3, 1, 1, dws, wsb, NULL));      /* g := y (any half tensor of the output extent) */
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdw, ddw, nw * 4, hipMemcpyDeviceToHost));
        int bad = 0, badw = 0;
        for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int o = 0; o < O; ++o) {
            float acc = hb[o];
            for (int t = 0; t < T; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                for (int i = 0; i < I; ++i) acc += wf[((size_t)t * O + o) * I + i] * xf[(((size_t)n * H + iy) * W + ix) * I + i];
            }
            if (h2f(hy[(((size_t)n * H + y) * W + x) * O + o]) != acc) ++bad;                     /* |acc| <= 9*32*8 = 2304 < 2048+..: exact in half up to 2048 */
        }
        for (int t = 0; t < T; ++t) for (int o = 0; o < O; ++o) for (int i = 0; i < I; ++i) {
            double acc = 0;
            for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)h2f(hy[(((size_t)n * H + y) * W + x) * O + o]) * xf[(((size_t)n * H + iy) * W + ix) * I + i];
            }
            if (fabs(hdw[((size_t)t * O + o) * I + i] - acc) > 1e-6 * (fabs(acc) + 1.0)) ++badw;
        }
        printf("fp16 conv2d (NHWC, packed weights): %d of %zu outputs differ; weight gradient: %d of %zu differ\n", bad, ny, badw, nw);
        if (bad || badw) ++fails;
        if (shg_conv2d_f16(dx, dwp, NULL, dy, N, 24, O, H, W, 3, 1, 1, 0, 0, H, W, NULL) == 0) { printf("I = 24 accepted\n"); ++fails; }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dwp)); CK(hipFree(dy)); CK(hipFree(db)); CK(hipFree(ddw)); CK(hipFree(dws));
        free(hx); free(hw); free(hy); free(xf); free(wf); free(hb); free(hdw);
    }
    printf(fails ? "FAILED (%d)\n" : "consumer ok\n", fails);
    return fails ? 4 : 0;
}
