// Micro-benchmark (round 5): the multiply loop of conv_wino4 (36 Winograd positions x 64 output channels x 32 tiles per
// workgroup, 8 waves x 9 accumulator tiles of 32x32) with the fp32 products replaced by a bf16 x 3 split -- a = hi + mid + lo (three bf16
// pieces, 24 mantissa bits), six products per tile and 16 input channels (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- against today's v_mfma_f32_32x32x2_f32 loop.  What is measured is the LOOP RATE only (no
// transforms, no epilogue): cycles per 16 input channels and workgroup, every CU busy, the weight-side operand (U: one consumer wave per
// element) streaming from L2 as in the kernel.
//   V0  fp32 today:        U: one dword per lane and MFMA straight from L2 into registers, V: ds_read_b32 from LDS        (2 chunks of 8 channels)
//   V1  bf16x3, U via LDS: U pieces by per-wave LDS-DMA ring (buffer_load ... lds, 6 KiB ahead), V pieces by ds_read_b128 (BOTH operands through LDS)
//   V2  bf16x3, U direct:  U pieces by global_load_dwordx4 straight into registers (one tile ahead), V pieces by ds_read_b128
//   V3  bf16x3, no U traffic: the same MFMA / LDS stream with U held in registers (ceiling of the loop)
// Operand volumes per 16 channels and workgroup: fp32 U 147 KB / V 74 KB; bf16x3 U 221 KB / V 111 KB (single-buffered here: a real kernel could
// not double-buffer it beside the U ring).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bf16x3_loop tools/micro/bf16x3_loop.hip ; run: tools/micro/bf16x3_loop [chunks of 16 channels = 16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16x __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 srd) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd) : "memory");
}

// U layouts (per workgroup-independent, shared by all workgroups: L2 resident): fp32 [chunk8][k-step 4][wave 8][tile 9][64 lanes] floats;
// bf16x3 [chunk16][wave 8][tile 9][piece 3][64 lanes][8 bf16]
template <int V>
__global__ __launch_bounds__(512, 2) void loop_kernel(const void* U, float* out, long long* clk, int nch16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // V operand image: fp32 [36 positions][k 8][64 lanes] floats; bf16x3 [36 positions][3 pieces][64 lanes][16 B] = 108 KiB; a wave uses positions 4 w .. 4 w + 4
    for (int i = tid; i < 108 * 1024 / 4; i += 512) ((float*)lds)[i] = (float)((i * 7 + 3) & 255) * 0.001f;
    __syncthreads();
    f16x acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    if constexpr (V == 0) {
        const float* u = (const float*)U + (size_t)wave * 9 * 64 + lane;
        const float* vl = (const float*)lds + wave * 4 * 8 * 64 + lane;
        // chunks of 8 channels = 4 k-steps of 2; the operands of chunk c + 1 are requested before the products of chunk c (the kernel's
        // register ring: 36 dwords per lane in flight)
        float a[2][4][9];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) a[0][k][t] = u[((size_t)k * 8 * 9 + t) * 64];
        for (int c2 = 0; c2 < nch16; ++c2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 2 * c2 + h, cn = c + 1 < 2 * nch16 ? c + 1 : c;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < 9; ++t) a[h ^ 1][k][t] = u[((size_t)(cn * 4 + k) * 8 * 9 + t) * 64];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float b = vl[((t >> 1) * 8 + h * 4 + k) * 64];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][k][t], b, acc[t], 0, 0, 0);
                    }
            }
        }
    } else {
        const unsigned char* vl = lds + wave * 4 * 3 * 1024 + lane * 16;           // this wave's V pieces
        if constexpr (V == 1) {
            // per-wave ring of 6 x 1 KiB U pieces behind the V image: piece j of the stream = (chunk, tile, piece); up to 6 requests in flight
            const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + 108 * 1024 + wave * 6 * 1024;
            const unsigned char* ring = lds + 108 * 1024 + wave * 6 * 1024 + lane * 16;
            const unsigned long long b = (unsigned long long)U;
            i32x4 srd; srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b); srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            srd[2] = (int)((size_t)nch16 * 8 * 27 * 1024); srd[3] = 0x00020000;
            const int total = nch16 * 27;
            auto req = [&](int j) __attribute__((always_inline)) {
                const int c = j / 27, r = j - c * 27;
                dma16(ring0 + (j % 6) * 1024, (unsigned)(((c * 8 + wave) * 27 + r) * 1024 + lane * 16), srd);
            };
            for (int j = 0; j < 6 && j < total; ++j) req(j);
            for (int c = 0; c < nch16; ++c) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int j = c * 27 + t * 3;
                    // the three pieces of tile t have landed when at most 3 younger requests are outstanding (6 in flight: j .. j+5)
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    const s8 ah = *(const s8*)(ring + ((j + 0) % 6) * 1024), am = *(const s8*)(ring + ((j + 1) % 6) * 1024), al = *(const s8*)(ring + ((j + 2) % 6) * 1024);
                    const s8 bh = *(const s8*)(vl + ((t >> 1) * 3 + 0) * 1024), bm = *(const s8*)(vl + ((t >> 1) * 3 + 1) * 1024), bl = *(const s8*)(vl + ((t >> 1) * 3 + 2) * 1024);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 6 < total) req(j + 6);
                    if (j + 7 < total) req(j + 7);
                    if (j + 8 < total) req(j + 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const s8* u = (const s8*)U + (size_t)wave * 27 * 64 + lane;
            // V2: the pieces of tile t + 1 are requested before the products of tile t (a ring of two tiles in registers); V3: three tiles'
            // pieces loaded once
            s8 a[3][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { a[0][q] = u[(size_t)q * 64]; a[1][q] = u[(size_t)(3 + q) * 64]; a[2][q] = u[(size_t)(6 + q) * 64]; }
            for (int c = 0; c < nch16; ++c) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if constexpr (V == 2) {
                        const int tn = t + 1 < 9 ? t + 1 : 0, cn = t + 1 < 9 ? c : (c + 1 < nch16 ? c + 1 : c);
#pragma unroll
                        for (int q = 0; q < 3; ++q) a[(t + 1) % 3][q] = u[((size_t)cn * 8 * 27 + tn * 3 + q) * 64];
                    }
                    const s8 bh = *(const s8*)(vl + ((t >> 1) * 3 + 0) * 1024), bm = *(const s8*)(vl + ((t >> 1) * 3 + 1) * 1024), bl = *(const s8*)(vl + ((t >> 1) * 3 + 2) * 1024);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][0], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][0], bm, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][1], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][1], bm, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][0], bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3][2], bh, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[tid] = s;
    if (tid == 448) clk[blockIdx.x] = t1 - t0;                  // (wave 7: the younger wave of its SIMD finishes last)
}

template <int V>
static void run(const void* U, float* out, long long* clk, int nch16, const char* what) {
    const int grid = 256, lds = V == 1 ? 108 * 1024 + 8 * 6 * 1024 : 108 * 1024;
    hipFuncSetAttribute((const void*)loop_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(loop_kernel<V>, dim3(grid), dim3(512), lds, 0, U, out, clk, nch16);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(loop_kernel<V>, dim3(grid), dim3(512), lds, 0, U, out, clk, nch16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid); hipMemcpy(h.data(), clk, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= grid;
    // direct-form-equivalent rate of the whole chip for this loop: per workgroup and 16 channels 36 positions x 64 x 32 x 16 MACs
    const double us = ms * 1e3 / reps, flop = 2.0 * 36 * 64 * 32 * 16 * nch16 * grid;
    printf("%-58s %8.1f us  %9.0f cycles per 16 channels and workgroup  %7.1f TFLOP/s of Winograd-domain products (err %d)\n", what, us, cyc / nch16,
           flop / us * 1e-6, (int)hipGetLastError());
}

int main(int argc, char** argv) {
    const int nch16 = argc > 1 ? atoi(argv[1]) : 16;
    const size_t bytes = (size_t)nch16 * 8 * 27 * 1024 + (size_t)nch16 * 2 * 4 * 8 * 9 * 256 + 4096;
    void* U; float* out; long long* clk;
    hipMalloc(&U, bytes); hipMalloc(&out, 4096); hipMalloc(&clk, 256 * sizeof(long long));
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x1ff));    // finite small values in either reading (fp32 pairs / bf16)
    hipMemcpy(U, h.data(), bytes, hipMemcpyHostToDevice);
    printf("multiply loop of conv_wino4's tile (36 positions x 64 x 32), %d input channels, 256 workgroups of 8 waves (one per CU)\n", nch16 * 16);
    run<0>(U, out, clk, nch16, "V0 fp32 32x32x2 (today): U L2 -> registers, V LDS b32");
    run<3>(U, out, clk, nch16, "V3 bf16x3 32x32x16, U held in registers (ceiling), V LDS b128");
    run<2>(U, out, clk, nch16, "V2 bf16x3, U L2 -> registers (dwordx4), V LDS b128");
    run<1>(U, out, clk, nch16, "V1 bf16x3, U L2 -> LDS-DMA ring -> b128, V LDS b128");
    return 0;
}
