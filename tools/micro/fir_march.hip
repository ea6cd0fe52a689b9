// Stand-alone timing + correctness harness for the row-marching FIR kernels (sh-gan_amd/csrc/fir_march.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sh-gan_amd/csrc tools/micro/fir_march.hip -o tools/micro/fir_march && tools/micro/fir_march
#include "fir_march.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void launch_down(FirMarchParams p, hipStream_t s) {
    const int K = (p.W + 63) / 64;
    p.LPG = p.W < 64 ? p.W : 64; p.G = 64 / p.LPG;
    const int npg = (p.NC + p.G - 1) / p.G, OH = p.H + 1;
    int nseg = (8192 + npg - 1) / npg; if (nseg > OH / 16) nseg = OH / 16; if (nseg < 1) nseg = 1;
    if (getenv("NSEG")) nseg = atoi(getenv("NSEG"));
    p.R = (OH + nseg - 1) / nseg; p.nseg = (OH + p.R - 1) / p.R; p.nitem = npg * p.nseg;
    dim3 grid((p.nitem + 3) / 4);
    const int dbg = getenv("DBG") ? atoi(getenv("DBG")) : 0;
#define GO(KK, BB) do { if (dbg == 1) hipLaunchKernelGGL((fir_down_march_kernel<KK, BB, 1>), grid, dim3(256), 0, s, p); \
        else if (dbg == 2) hipLaunchKernelGGL((fir_down_march_kernel<KK, BB, 2>), grid, dim3(256), 0, s, p); \
        else if (dbg == 3) hipLaunchKernelGGL((fir_down_march_kernel<KK, 12 - BB>), grid, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((fir_down_march_kernel<KK, BB>), grid, dim3(256), 0, s, p); } while (0)
    if (p.mode == 1 && p.W % 256 == 0 && !getenv("NOV4")) {
        p.nseg = nseg; p.R = (OH + nseg - 1) / nseg; p.nseg = (OH + p.R - 1) / p.R; p.nitem = p.NC * p.nseg;
        dim3 g4((p.nitem + 3) / 4);
        const int b4 = getenv("B4") ? atoi(getenv("B4")) : 0;
#define GO4(KK, BB) do { if (dbg == 1) hipLaunchKernelGGL((fir_down_march4_kernel<KK, BB, 1>), g4, dim3(256), 0, s, p); \
        else if (dbg == 2) hipLaunchKernelGGL((fir_down_march4_kernel<KK, BB, 2>), g4, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((fir_down_march4_kernel<KK, BB>), g4, dim3(256), 0, s, p); } while (0)
        if (p.W == 256) { if (b4 == 4) GO4(1, 4); else GO4(1, 8); } else if (p.W == 512) { if (b4 == 8) GO4(2, 8); else GO4(2, 4); } else { printf("unsupported\n"); exit(1); }
        return;
    }
    if (K == 1) GO(1, 8); else if (K == 2) GO(2, 8); else if (K == 4) GO(4, 8); else if (K == 8) GO(8, 4);
    else { printf("unsupported W\n"); exit(1); }
}


static void launch_up(FirUpParams p, hipStream_t s) {
    const int K = (p.W + 127) / 128;
    p.LPG = p.W / 2 < 64 ? p.W / 2 : 64; p.G = 64 / p.LPG;
    const int npg = (p.NC + p.G - 1) / p.G;
    int nseg = (8192 + npg - 1) / npg; if (nseg > p.H / 8) nseg = p.H / 8; if (nseg < 1) nseg = 1;
    if (getenv("NSEG")) nseg = atoi(getenv("NSEG"));
    p.R = (p.H + nseg - 1) / nseg; p.nseg = (p.H + p.R - 1) / p.R; p.nitem = npg * p.nseg;
    dim3 grid((p.nitem + 3) / 4);
    const int bb = getenv("UB") ? atoi(getenv("UB")) : 0;
    if (K == 1) { if (bb == 1) hipLaunchKernelGGL((fir_up_march_kernel<1, 1>), grid, dim3(256), 0, s, p);
                  else if (bb == 4) hipLaunchKernelGGL((fir_up_march_kernel<1, 4>), grid, dim3(256), 0, s, p);
                  else hipLaunchKernelGGL((fir_up_march_kernel<1, 2>), grid, dim3(256), 0, s, p); }
    else if (K == 2) { if (bb == 1) hipLaunchKernelGGL((fir_up_march_kernel<2, 1>), grid, dim3(256), 0, s, p);
                  else hipLaunchKernelGGL((fir_up_march_kernel<2, 2>), grid, dim3(256), 0, s, p); }
    else { printf("unsupported W\n"); exit(1); }
}

static void up_cases() {
    const float t[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    struct Case { int N, C, H; } cases[] = {{16, 64, 256}, {16, 128, 128}, {16, 256, 64}, {16, 512, 32}, {16, 512, 16}, {16, 512, 8}, {3, 5, 4}, {2, 3, 64}};
    for (auto cs : cases) {
        const int N = cs.N, C = cs.C, NC = N * C, H = cs.H, W = cs.H, OH = 2 * H, OW = 2 * W, PW = W + 1;
        const size_t nm = (size_t)4 * NC * (H + 1) * PW, ny = (size_t)NC * OH * OW, nn = (size_t)N * OH * OW;
        std::vector<float> hm(nm), hr(ny), hn(nn), hs(NC), hb(C);
        unsigned s = 777u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hm) v = rnd();
        for (auto& v : hr) v = rnd();
        for (auto& v : hn) v = rnd();
        for (auto& v : hs) v = 1.f + rnd();
        for (auto& v : hb) v = rnd();
        float *dm, *dr, *dn, *ds, *db, *dy;
        CK(hipMalloc(&dm, nm * 4)); CK(hipMalloc(&dr, ny * 4)); CK(hipMalloc(&dn, nn * 4)); CK(hipMalloc(&ds, NC * 4)); CK(hipMalloc(&db, C * 4));
        CK(hipMalloc(&dy, ny * 4));
        CK(hipMemcpy(dm, hm.data(), nm * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), ny * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dn, hn.data(), nn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), NC * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dy, 0xff, ny * 4));
        FirUpParams p{};
        p.mid = dm; p.y = dy; p.scale = ds; p.bias = db; p.noise = dn; p.residual = dr; p.NC = NC; p.C = C; p.H = H; p.W = W;
        p.noise_mode = 2; p.noise_strength = 0.3f; p.act = 1; p.alpha = 0.2f; p.act_gain = 1.41421356f; p.clamp = 0.6f;
        for (int k = 0; k < 4; ++k) { p.a[k] = t[k] * (1.f + 0.1f * k); p.b[k] = t[k] * (1.f - 0.05f * k); }
        launch_up(p, 0);
        CK(hipDeviceSynchronize());
        const int du = getenv("DBGU") ? atoi(getenv("DBGU")) : 0;
        std::vector<float> hy(ny);
        CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; long bad = 0;
        const int planes[4] = {0, 1, NC / 2, NC - 1};
        for (int pi = 0; pi < 4; ++pi) {
            const int nc = planes[pi], n = nc / C, c = nc % C;
            auto full = [&](int Y, int X) -> double {
                if (Y < 0 || Y > 2 * H || X < 0 || X > 2 * W) return 0.0;
                return hm[(((size_t)((Y & 1) * 2 + (X & 1)) * NC + nc) * (H + 1) + (Y >> 1)) * PW + (X >> 1)];
            };
            for (int Y = 0; Y < OH; ++Y)
                for (int X = 0; X < OW; ++X) {
                    double v = 0;
                    for (int ky = 0; ky < 4; ++ky)
                        for (int kx = 0; kx < 4; ++kx) v += (double)p.b[ky] * p.a[kx] * full(Y + ky - 1, X + kx - 1);
                    v = v * hs[nc] + hn[((size_t)n * OH + Y) * OW + X] * p.noise_strength + hb[c];
                    v = (v < 0 ? v * p.alpha : v) * p.act_gain;
                    v = fmin(fmax(v, -(double)p.clamp), (double)p.clamp);
                    v += hr[((size_t)nc * OH + Y) * OW + X];
                    const double e = fabs(hy[((size_t)nc * OH + Y) * OW + X] - v);
                    if (!(e <= 2e-5)) ++bad;
                    if (e > maxerr || e != e) maxerr = e;
                }
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int it = 20;
        if (du & 1) { p.noise = nullptr; p.noise_mode = 0; }
        if (du & 2) p.residual = nullptr;
        launch_up(p, 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < it; ++i) launch_up(p, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000 / it, bytes = 4.0 * ((double)nm + 2.0 * ny);
        printf("up NC=%5d %3d^2->%3d^2: %8.1f us  %5.2f TB/s  maxerr %.2e bad %ld\n", NC, H, OH, us, bytes / us / 1e6, maxerr, bad);
        CK(hipFree(dm)); CK(hipFree(dr)); CK(hipFree(dn)); CK(hipFree(ds)); CK(hipFree(db)); CK(hipFree(dy));
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* x, float4* y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 v = x[i]; v.x *= 2.f; y[i] = v; }
}
__global__ __launch_bounds__(256) void copy4_kernel(const float* x, float* y, long n) {     // dword copy
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = 2.f * x[i];
}

// epilogue-like store pattern: every wave-instruction writes 8 whole 128-byte lines (8 lanes x 16 B each) that lie `rowB` bytes apart
// (the 4x4-block rows of a convolution tile); NT: nontemporal stores
template <int NT>
__global__ __launch_bounds__(512) void wtile_kernel(char* y, int reps, long rowB, long tileB) {
    const long wg = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; ++r) {
        // wave w, instruction r: lines (lane / 8) of "row" r*8*8 + w*8 + lane/8
        const long line = (long)(r * 64 + wave * 8 + (lane >> 3));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f* d = reinterpret_cast<v4f*>(y + wg * tileB + line * rowB + (lane & 7) * 16);
        const v4f v = {1.f, 2.f, 3.f, (float)r};
        if (NT) __builtin_nontemporal_store(v, d); else *d = v;
    }
}
template <int LB>
__global__ __launch_bounds__(256) void wpat_kernel(char* y, int rows, long pitchB, int order, long nw) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < rows; ++r) {
        const long row = order ? (long)r * nw + w : w * rows + r;
        char* d = y + row * pitchB + lane * LB;
        if (LB == 16) *reinterpret_cast<float4*>(d) = make_float4(1.f, 2.f, 3.f, (float)r);
        else if (LB == 8) *reinterpret_cast<float2*>(d) = make_float2(1.f, (float)r);
        else *reinterpret_cast<float*>(d) = (float)r;
    }
}
int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'u') { up_cases(); return 0; }
    if (argc > 1 && argv[1][0] == 't') {
        // 256 workgroups (one per CU) x 512 threads, each writing a 128 KB "tile" as 16 x 64 lines of 128 B, 2 KB apart
        const long nwg = 8192, rowB = 2048, tileB = 16 * 64 * rowB;
        char* y; CK(hipMalloc(&y, nwg * tileB + 4096));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int nt = 0; nt < 2; ++nt) {
            auto go = [&]() { if (nt) hipLaunchKernelGGL(wtile_kernel<1>, dim3(nwg), dim3(512), 0, 0, y, 16, rowB, tileB);
                              else hipLaunchKernelGGL(wtile_kernel<0>, dim3(nwg), dim3(512), 0, 0, y, 16, rowB, tileB); };
            go();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) go();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)nwg * 16 * 64 * 128;
            printf("wtile nt=%d: %.1f us  %.2f TB/s  (%.1f B/clk/CU at 2.1 GHz)\n", nt, ms * 100, bytes / (ms * 100) / 1e6, bytes / (ms * 1e-4) / 256 / 2.1e9);
        }
        return 0;
    }
    {
        const long nw = 16384; const int rows = 64;
        char* y; CK(hipMalloc(&y, nw * rows * 1152 + 4096));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int offs[6] = {0, 4, 16, 32, 64, 128};
        for (int lb = 4; lb <= 16; lb *= 2)
            for (int pi = 0; pi < 6; ++pi) {
                if (offs[pi] % lb) continue;
                const long pitch = 64 * lb + offs[pi];
                auto go = [&]() {
                    if (lb == 16) hipLaunchKernelGGL(wpat_kernel<16>, dim3(nw / 4), dim3(256), 0, 0, y, rows, pitch, 0, nw);
                    else if (lb == 8) hipLaunchKernelGGL(wpat_kernel<8>, dim3(nw / 4), dim3(256), 0, 0, y, rows, pitch, 0, nw);
                    else hipLaunchKernelGGL(wpat_kernel<4>, dim3(nw / 4), dim3(256), 0, 0, y, rows, pitch, 0, nw);
                };
                go();
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; ++i) go();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("wpat %2d B/lane pitch %4ld (+%3d): %.1f us %.2f TB/s\n", lb, pitch, offs[pi], ms * 100, (double)nw * rows * 64 * lb / (ms * 100) / 1e6);
            }
        CK(hipFree(y));
    }
    {
        const long n = 1024L * 512 * 512 / 4;
        float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 0, n * 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) {
                if (v == 0) hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
                else hipLaunchKernelGGL(copy4_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, 0, (const float*)a, (float*)b, 4 * n);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("copy%d 1 GiB: %.1f us %.2f TB/s\n", v, ms * 50, 2.0 * n * 16 / (ms * 50) / 1e6);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    const float t[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    struct Case { int NC, H; } cases[] = {{1024, 512}, {2048, 256}, {4096, 128}, {8192, 64}, {8192, 32}, {300, 16}, {7, 128}};
    for (auto cs : cases) {
        const int NC = cs.NC, H = cs.H, W = cs.H, OH = H + 1, OW = W + 1;
        std::vector<float> hx((size_t)NC * H * W);
        unsigned s = 12345u;
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; }
        float *dx, *dy;
        CK(hipMalloc(&dx, hx.size() * 4));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; ++mode) {      // 0: plain pitch OW, 1: planar, 2: plain pitch aligned, 3: planar with 128-byte pitch
            FirMarchParams p{};
            p.x = dx; p.NC = NC; p.H = H; p.W = W; p.mode = mode == 1 || mode == 3;
            p.ph2 = H / 2 + 1; p.pitch = mode == 3 ? (W / 2 + 1 + 31) / 32 * 32 : mode == 1 ? (W / 2 + 1 + 3) / 4 * 4 : (mode == 0 ? OW : (OW + 3) / 4 * 4);
            for (int k = 0; k < 4; ++k) { p.a[k] = t[k] * (1.f + 0.1f * k); p.b[k] = t[k] * (1.f - 0.05f * k); }
            const size_t ny = p.mode ? (size_t)4 * NC * p.ph2 * p.pitch : (size_t)NC * OH * p.pitch;
            CK(hipMalloc(&dy, ny * 4));
            CK(hipMemset(dy, 0xff, ny * 4));
            p.y = dy;
            launch_down(p, 0);
            CK(hipDeviceSynchronize());
            std::vector<float> hy(ny);
            CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
            double maxerr = 0; long bad = 0;
            const int planes[4] = {0, 1, NC / 2, NC - 1};
            for (int pi = 0; pi < 4; ++pi) {
                const int nc = planes[pi];
                const float* xp = hx.data() + (size_t)nc * H * W;
                for (int oy = 0; oy < OH; ++oy)
                    for (int ox = 0; ox < OW; ++ox) {
                        double ref = 0;
                        for (int ky = 0; ky < 4; ++ky)
                            for (int kx = 0; kx < 4; ++kx) {
                                const int iy = oy + ky - 2, ix = ox + kx - 2;
                                if (iy >= 0 && iy < H && ix >= 0 && ix < W) ref += (double)p.b[ky] * p.a[kx] * xp[iy * W + ix];
                            }
                        float got;
                        if (p.mode) got = hy[(((size_t)((oy & 1) * 2 + (ox & 1)) * NC + nc) * p.ph2 + (oy >> 1)) * p.pitch + (ox >> 1)];
                        else got = hy[((size_t)nc * OH + oy) * p.pitch + ox];
                        const double e = fabs(got - ref);
                        if (!(e <= 1e-5)) ++bad;
                        if (e > maxerr || e != e) maxerr = e;
                    }
                if (p.mode)   // padding must be zero
                    for (int q = 0; q < 4; ++q)
                        for (int r = 0; r < p.ph2; ++r)
                            for (int c = 0; c < p.pitch; ++c) {
                                const int oy = 2 * r + (q >> 1), ox = 2 * c + (q & 1);
                                if (oy >= OH || ox >= OW) {
                                    const float v = hy[(((size_t)q * NC + nc) * p.ph2 + r) * p.pitch + c];
                                    if (v != 0.f) ++bad;
                                }
                            }
            }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int it = 20;
            launch_down(p, 0);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < it; ++i) launch_down(p, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000 / it, bytes = 4.0 * ((double)NC * H * W + (double)NC * OH * OW);
            printf("down NC=%5d %3d^2 mode %d: %8.1f us  %5.2f TB/s  maxerr %.2e bad %ld\n", NC, H, mode, us, bytes / us / 1e6, maxerr, bad);
            CK(hipFree(dy));
        }
        CK(hipFree(dx));
    }
    return 0;
}
