// Micro-benchmark: do fp32 MFMA and fp32 VALU instructions of different waves on one SIMD overlap on gfx950?
// 512 threads per block (2 waves per SIMD), 256 blocks.  mode 1: every wave issues MFMAs; mode 2: odd SIMD-slot waves issue
// VALU FMAs only; mode 3: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD); mode 4: like 3 with packed VALU;
// mode 5: waves 4-7 LDS reads; mode 6: every wave alternates 1 MFMA + 8 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int mode>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = lane * 0.001f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    f32x2 pv[8];
    for (int i = 0; i < 8; ++i) pv[i] = f32x2{(float)lane, (float)i};
    lds[threadIdx.x] = lane;
    __syncthreads();
    const bool mf = mode == 1 || mode == 6 || ((mode == 3 || mode == 4 || mode == 5) && wave < 4);
    const bool va = mode == 2 || mode == 6 || (mode == 3 && wave >= 4);
    const bool pk = mode == 4 && wave >= 4;
    const bool ld = mode == 5 && wave >= 4;
    float ls = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (mode == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * b + a;
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        if (mf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
        if (va) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * b + a;
        }
        if (pk) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = pv[i] * b + a;
        }
        if (ld) {
#pragma unroll
            for (int r = 0; r < 8; ++r) ls += lds[(lane + r * 64 + it) & 8191];
        }
    }
    float s = ls;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int mode>
void run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<mode>, dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<mode>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (%.1f cycles/iter at 2.4 GHz)\n", mode, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 100000;
    run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters); run<5>(out, iters); run<6>(out, iters);
    return 0;
}
