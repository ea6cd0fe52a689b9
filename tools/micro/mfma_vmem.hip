// Micro-benchmark: what does a global load cost inside a stream of fp32 MFMAs on gfx950?  512 threads (2 waves / SIMD), 256 blocks.
//  mode 0: 8 MFMAs per iteration, nothing else                  mode 1: + 8 global_load_dword (64-bit VGPR address), one behind each MFMA
//  mode 2: + 8 loads with SGPR base + 32-bit VGPR offset         mode 3: + 2 global_load_dwordx4 (same bytes)
//  mode 4: + 8 ds_read_b32                                       mode 5: loads refill the A operand of the MFMA issued 2 earlier (ring)
//  mode 6: like 5, refill right behind the reader               mode 7: mode 1 with ONE wave per SIMD (256 threads)
//  mode 8: mode 1 with the accumulators in AGPRs (inline asm)   mode 9: mode 0 with the accumulators in AGPRs   mode 10: mode 4 + AGPR
//  mode 11: the 8 loads of an iteration in one burst ahead of its 8 MFMAs   mode 12: 8 MFMAs + 8 loads whose results are never used as
//  MFMA operands but land in registers (mode 1) vs. loads into ONE register (same destination, write-after-write)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a[8], b = 1.0001f;
    for (int j = 0; j < 8; ++j) a[j] = lane * 0.001f + j;
    lds[threadIdx.x] = lane; lds[threadIdx.x + 512] = lane;
    __syncthreads();
    const float* pl = src + (blockIdx.x * 8 + wave) * 4096 + lane;     // per-lane 64-bit pointer
    const float* pu = src + (blockIdx.x * 8 + wave) * 4096;            // uniform base
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 7) * 512;
        float ld[8];
        f32x4 l4[2];
        if (MODE == 11) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ld[j] = pl[o + j * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE >= 8 && MODE <= 10) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[j]), "v"(b));
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc[j], 0, 0, 0);
            if (MODE == 1 || MODE == 7 || MODE == 8) ld[j] = pl[o + j * 64];
            if (MODE == 2) ld[j] = pu[o + j * 64 + lane];
            if (MODE == 3 && (j & 3) == 0) l4[j >> 2] = *reinterpret_cast<const f32x4*>(pu + o + (j >> 2) * 256 + lane * 4);
            if (MODE == 4 || MODE == 10) ld[j] = lds[(lane + j * 64 + o) & 4095];
            if (MODE == 5 && j >= 2) a[j - 2] = pl[o + j * 64];
            if (MODE == 6) a[j] = pl[o + j * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 5) { a[6] = pl[o + 6 * 64 + 1]; a[7] = pl[o + 7 * 64 + 1]; }
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 7 || MODE == 8 || MODE == 10 || MODE == 11) for (int j = 0; j < 8; ++j) sink += ld[j];
        if (MODE == 3) sink += l4[0][0] + l4[1][3];
    }
    float s = sink;
    for (int j = 0; j < 8; ++j) { s += a[j]; for (int r = 0; r < 16; ++r) s += acc[j][r]; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* src, float* out, int iters) {
    const int threads = MODE == 7 ? 256 : 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 8 * (threads / 64) / 4;     // MFMAs per SIMD
    printf("mode %d: %.3f ms  %.1f ns per MFMA per SIMD (64 cycles = %.1f ns at 2.4 GHz)\n", MODE, ms, ms * 1e6 / mf, 64 / 2.4);
}
int main() {
    float *src, *out;
    hipMalloc(&src, (size_t)256 * 8 * 4096 * 4 + 65536); hipMemset(src, 0, (size_t)256 * 8 * 4096 * 4 + 65536);
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 50000;
    run<0>(src, out, iters); run<1>(src, out, iters); run<2>(src, out, iters); run<3>(src, out, iters);
    run<4>(src, out, iters); run<5>(src, out, iters); run<6>(src, out, iters); run<7>(src, out, iters);
    run<8>(src, out, iters); run<9>(src, out, iters); run<10>(src, out, iters); run<11>(src, out, iters);
    return 0;
}
