// Micro-benchmark (round 4): how do the epilogue stores of a Winograd tile interact with the multiply loop of the next one?
// A persistent 512-thread workgroup per CU runs T "tiles": a multiply phase of NM fp32 MFMAs per wave and a store phase that
// writes a 64-channel x 4 x 128-pixel output tile (131 KB) in the pattern of conv_wino4's epilogue (every store instruction =
// two 512-byte runs).  Modes:
//   0 multiply only            1 stores only                 2 multiply, then the 16 stores as one burst (fire and forget)
//   3 like 2, but vmcnt(0) + barrier after every pass of 4 stores (what __syncthreads() does to the present epilogue)
//   4 the 16 stores spread evenly over the multiply phase (data of the previous tile)
//   5 like 2 with the workgroups de-phased by (block % 8) / 8 of a tile period at the start
// usage: store_overlap [grid=256] [NM=384] [T=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* y, int NM, int T, long plane, int W, long long* clk) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = lane * 0.001f, b = 1.0f + lane * 1e-6f;
    const int o_l = tid >> 5, t_l = tid & 31;
    if (MODE == 5) {
        const int d = (blockIdx.x & 7);
        for (int i = 0; i < d * 160; ++i) __builtin_amdgcn_s_sleep(1);      // ~64 cycles each
    }
    long long t0 = clock64();
    for (int t = 0; t < T; ++t) {
        const int tile = blockIdx.x * T + t;          // 4 rows x 128 px, 4 tiles per image row band
        const int tx = tile & 3, ty = (tile >> 2) & 127, n = tile >> 9;
        float* base = y + (long)n * 64 * plane + (long)(ty * 4) * W + tx * 128 + 4 * t_l;
        f32x4 v; v[0] = acc[0][0]; v[1] = acc[1][1]; v[2] = a; v[3] = b;
        auto store = [&](int s) __attribute__((always_inline)) {   // s = pass * 4 + row
            const int pass = s >> 2, i = s & 3;
            float* q = base + (long)(pass * 16 + o_l) * plane + (long)i * W;
            *reinterpret_cast<f32x4*>(q) = v;
        };
        if (MODE != 1) {
            const int per = NM / 16;
            for (int m = 0; m < NM; m += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                if (MODE == 4 && (m % per) == 0 && m / per < 16) store(m / per);
            }
        }
        if (MODE == 1 || MODE == 2 || MODE == 5) {
#pragma unroll
            for (int s = 0; s < 16; ++s) store(s);
        }
        if (MODE == 3) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
                for (int i = 0; i < 4; ++i) store(pass * 4 + i);
                __syncthreads();
            }
        }
    }
    long long t1 = clock64();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    if (acc[0][0] == 123.f) y[0] = acc[3][2] + acc[5][1] + acc[7][7] + acc[2][0] + acc[4][0] + acc[6][0] + acc[1][0];
}

template <int MODE>
void run(float* y, long long* clk, int grid, int NM, int T, long plane, int W) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, y, NM, T, plane, W, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long long h[1024]; hipMemcpy(h, clk, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; ++i) mean += h[i]; mean /= grid;
    const double bytes = (double)grid * T * 131072.0;
    printf("mode %d grid %4d NM %4d T %3d: %8.1f us   %7.0f clocks/tile (clock64 ticks)   stores %.2f TB/s if alone\n", MODE, grid, NM, T, best * 1e3,
           mean / T, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256, NM = argc > 2 ? atoi(argv[2]) : 384, T = argc > 3 ? atoi(argv[3]) : 32;
    const int W = 512; const long plane = 512L * 512;
    float* y; long long* clk;
    hipMalloc(&y, (size_t)16 * 64 * plane * 4 + 4096); hipMalloc(&clk, 1024 * sizeof(long long));
    hipMemset(y, 0, (size_t)16 * 64 * plane * 4);
    run<0>(y, clk, grid, NM, T, plane, W);
    run<1>(y, clk, grid, NM, T, plane, W);
    run<2>(y, clk, grid, NM, T, plane, W);
    run<3>(y, clk, grid, NM, T, plane, W);
    run<4>(y, clk, grid, NM, T, plane, W);
    run<5>(y, clk, grid, NM, T, plane, W);
    return 0;
}
