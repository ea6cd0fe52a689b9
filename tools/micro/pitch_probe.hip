// Micro-benchmark (round 4): does the odd pitch W + 1 of the up path's phase planes cost HBM rate?  A marching reader in the shape of
// fir_up_march: a wave walks the rows of four (H+1) x pitch phase planes (dword loads, 4 x 64 lanes per row = 256 columns + 1) and writes two
// 512-float output rows per step (16-byte stores).  Same bytes, pitch 257 (dense) / 260 / 288 (128-byte aligned rows).
// and with the loads of 1 / 2 / 4 / 8 rows requested before the first store.
// usage: pitch_probe  (planes 16 x 64 as in the 512^2 layer at batch 16)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void k(const float* mid, float* out, int pitch, int H, long pstride, int nplanes) {
    const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int seg = item & 7, pl = item >> 3;                 // 8 segments of 32 rows per plane set
    if (pl >= nplanes) return;
    const float* m[4];
    for (int q = 0; q < 4; ++q) m[q] = mid + ((long)q * nplanes + pl) * pstride;
    float* o = out + (long)pl * 512 * 512;
    float acc = 0.f;
    for (int r0 = seg * 32; r0 < seg * 32 + 32; r0 += U) {
        float v[U][4][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) v[u][q][kx] = m[q][(long)(r0 + u) * pitch + lane + 64 * kx];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u;
            f4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = v[u][0][e] + v[u][1][e] + acc; b[e] = v[u][2][e] + v[u][3][e]; }
            acc = v[u][0][0] * 1e-9f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *(f4*)(o + (long)(2 * r) * 512 + 4 * lane + 256 * h) = a;
                *(f4*)(o + (long)(2 * r + 1) * 512 + 4 * lane + 256 * h) = b;
            }
        }
    }
}
template <int U>
void run(const float* mid, float* out, int pitch, int nplanes) {
    const long pstride = (long)257 * pitch;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<U>, dim3(nplanes * 8 / 4), dim3(256), 0, 0, mid, out, pitch, 256, pstride, nplanes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)nplanes * (4.0 * 256 * 256 + 512.0 * 512) * 4;
    printf("rows in flight %d, pitch %3d: %7.1f us  %.2f TB/s\n", U, pitch, best * 1e3, bytes / (best * 1e-3) / 1e12);
}
int main() {
    const int nplanes = 16 * 64, H = 256;
    float *mid, *out;
    hipMalloc(&mid, (size_t)4 * nplanes * 257 * 288 * 4 + 4096); hipMalloc(&out, (size_t)nplanes * 512 * 512 * 4);
    hipMemset(mid, 0, (size_t)4 * nplanes * 257 * 288 * 4);
    for (int pitch : {257, 288}) {
        run<1>(mid, out, pitch, nplanes); run<2>(mid, out, pitch, nplanes); run<4>(mid, out, pitch, nplanes); run<8>(mid, out, pitch, nplanes);
    }
    return 0;
}
