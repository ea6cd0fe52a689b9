// Micro-probe: what limits a 1-wave-per-SIMD fp32 MFMA stream?  Variants add LDS operand reads, barriers and staging.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NSTEP 36

template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(float* out, const float* gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5, wave = tid >> 6;
    for (int e = tid; e < 20000; e += 256) lds[e] = (float)(e & 15) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const float* wa = lds + (wave & 1) * 64 + l31 + half * 9 * 128;
    const float* xb0 = lds + 9300 + ((wave >> 1) * 64 + l31) * 9 + half;
    const float* xb1 = xb0 + 32 * 9;
    float a0 = 1.f, a1 = 2.f, b0 = 3.f, b1 = 4.f;
    f32x4 gv[9];
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 3) {
#pragma unroll
            for (int k = 0; k < 9; ++k) gv[k] = *reinterpret_cast<const f32x4*>(gsrc + ((size_t)(blockIdx.x * 37 + it * 9 + k) * 1024 + tid * 4) % (1 << 24));
        }
        if (VAR == 0) {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        } else {
            float a[2][2], b[2][2];
            a[0][0] = wa[0]; a[0][1] = wa[32]; b[0][0] = xb0[0]; b[0][1] = xb1[0];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int n = s + 1, t = n / 4, c2 = n % 4;
                if (n < NSTEP) {
                    a[n & 1][0] = wa[((c2 * 2) * 9 + t) * 128]; a[n & 1][1] = wa[((c2 * 2) * 9 + t) * 128 + 32];
                    b[n & 1][0] = xb0[t * 9 * 3 + c2 * 2]; b[n & 1][1] = xb1[t * 9 * 3 + c2 * 2];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][0], b[s & 1][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][0], b[s & 1][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][1], b[s & 1][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][1], b[s & 1][1], acc[3], 0, 0, 0);
                if (VAR >= 3 && s == NSTEP / 2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4*>(lds + 12000 + (tid + k * 256) * 4 % 8000) = gv[k];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (VAR >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR> void run(const char* name, float* out, float* src) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 90000);
    hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), 90000, 0, out, src, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), 90000, 0, out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 32 * 32 * 2 * 4 * NSTEP * (double)iters * grid * 4;
    printf("%-40s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA @2.4GHz)\n", name, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * NSTEP * iters));
}

int main() {
    float *out, *src; hipMalloc(&out, 1 << 20); hipMalloc(&src, (1 << 24) * 4 + 65536); hipMemset(src, 0, (1 << 24) * 4);
    run<0>("V0 pure MFMA (regs)", out, src);
    run<1>("V1 + LDS operand prefetch", out, src);
    run<2>("V2 + barrier per 144 MFMA", out, src);
    run<3>("V3 + global loads + mid ds_write", out, src);
    return 0;
}
