// Floor probe for the Spectral Hint Unit's three launches: kernels with the SAME grid, workgroup size, LDS
// footprint and global load / store pattern as shu_rfft2_shift_kernel / shu_spectral_kernel / shu_split_irfft2_kernel
// (sh-gan_amd/csrc/shu.hip) and NO arithmetic between the load and the store phase -- what a launch of that shape costs whatever
// the transform inside does: launch + dependency on the previous launch + one global -> LDS pass + barrier(s) + the stores.
// Study code: built by tools/shu_floor.py into tools/_variants/, never linked into libshgan_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SHU_N 64
#define SHU_NH 33

// rfft2 skeleton: one workgroup = one plane; 16 KB in, LDS 16.6 + 16.9 KB, 2 x 64 x 33 floats out with the real kernel's store indices
extern "C" __global__ __launch_bounds__(256) void rfft2_floor_kernel(const float* x, long xbs, float* T, int C) {
    __shared__ float xs[SHU_N][SHU_N + 1];
    __shared__ float Rr[SHU_N][SHU_NH], Ri[SHU_N][SHU_NH];
    const int c = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const float* xp = x + (long)n * xbs + (long)c * SHU_N * SHU_N;
    for (int e = tid; e < SHU_N * SHU_N; e += 256) xs[e >> 6][e & 63] = xp[e];
    __syncthreads();
    // (pass 1 would run here) -> Rr / Ri written once
    for (int e = tid; e < SHU_N * SHU_NH; e += 256) { Rr[e / SHU_NH][e % SHU_NH] = xs[e >> 6][e & 63]; Ri[e / SHU_NH][e % SHU_NH] = xs[(e >> 6) ^ 1][e & 63]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ur = mr & 63, row = (ur + 31) & 63;
        T[(((long)n * 2 * C + (mr >= 64 ? C : 0) + c) * SHU_N + row) * SHU_NH + l31] = (mr >= 64 ? Ri : Rr)[ur][l31];
    }
    if (tid < 2 * SHU_N) {
        const int u = tid & 63;
        const bool imrow = tid >= 64;
        T[(((long)n * 2 * C + (imrow ? C : 0) + c) * SHU_N + ((u + 31) & 63)) * SHU_NH + 32] = Rr[u][32];
    }
}

// spectral skeleton: workgroup = one sample x 64 positions; 16 KB T tile in (float4 rows), 32 KB LDS, 64 x 64 floats out,
// + the 98 KB + 16 KB of packed weights every workgroup streams from L2 (read once here, 4 bytes per lane per k-step like the real loop)
extern "C" __global__ __launch_bounds__(256) void spectral_floor_kernel(const float* T, const float* w0p, const float* w1p, const float* cw, float* S, int P, int B) {
    __shared__ __attribute__((aligned(16))) float Tl[64][64];
    __shared__ __attribute__((aligned(16))) float tl[64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int mo = wave >> 1, nt = wave & 1;
    const int n = blockIdx.y, p0 = blockIdx.x * 64;
    const float* Tn = T + (long)n * 64 * P + p0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 256 + tid, ch = e >> 4, c4 = e & 15;
        *reinterpret_cast<float4*>(&Tl[ch][4 * c4]) = *reinterpret_cast<const float4*>(Tn + (long)ch * P + 4 * c4);
    }
    float wsum = 0.f;
    for (int k = 0; k < B; ++k) wsum += cw[(long)k * P + p0 + nt * 32 + l31];
    __syncthreads();
    const float* ap = w0p + mo * 64 + lane;
    for (int ks = 0; ks < 32; ++ks) wsum += ap[ks * 128];
    for (int r = 0; r < 16; ++r) {
        const int row = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        tl[row][nt * 32 + l31] = Tl[row][nt * 32 + l31];
    }
    __syncthreads();
    const float* bp = w1p + mo * 64 + lane;
    for (int ks = 0; ks < B * 32; ++ks) wsum += bp[ks * 128];
    float* Sn = S + (long)n * 64 * P + p0 + nt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Sn[(long)row * P] = tl[row][nt * 32 + l31] + (wsum == 12345.f ? 1.f : 0.f);
    }
}

// split + irfft2 skeleton: one workgroup = one plane; 2 x 8.4 KB in, 2 x 16.9 KB LDS, the five levels' outputs (r*r floats each,
// read-modify-write: the encoder accumulates into the skip features) with one barrier per level like the real kernel
struct SplitFloorParams { const float* Y; float* out[5]; long obs[5]; int C; };
extern "C" __global__ __launch_bounds__(256) void split_floor_kernel(const SplitFloorParams p) {
    __shared__ float2 S[SHU_N][SHU_NH];
    __shared__ float2 Z[SHU_N][SHU_NH];
    const int c = blockIdx.x, n = blockIdx.y, C = p.C;
    const long plane = SHU_N * SHU_NH;
    const float* yre = p.Y + ((long)n * 2 * C + c) * plane;
    const float* yim = p.Y + ((long)n * 2 * C + C + c) * plane;
    for (int e = threadIdx.x; e < plane; e += 256) S[e / SHU_NH][e % SHU_NH] = make_float2(yre[e], yim[e]);
    __syncthreads();
    for (int l = 0; l < 5; ++l) {
        const int r = 4 << l, rh = r / 2 + 1;
        for (int e = threadIdx.x; e < r * rh; e += 256) Z[e / rh][e % rh] = S[SHU_N / 2 - r / 2 + e / rh][e % rh];
        __syncthreads();
        float* op = p.out[l] + (long)n * p.obs[l] + (long)c * r * r;
        for (int e = threadIdx.x; e < r * r; e += 256) op[e] = op[e] + Z[e / r][(e % r) >> 1].x;
        __syncthreads();
    }
}

extern "C" int floor_rfft2(const float* x, long xbs, float* T, int N, int C, void* stream) {
    hipLaunchKernelGGL(rfft2_floor_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, xbs, T, C);
    return (int)hipGetLastError();
}
extern "C" int floor_spectral(const float* T, const float* w0p, const float* w1p, const float* cw, float* S, int N, int P, int B, void* stream) {
    hipLaunchKernelGGL(spectral_floor_kernel, dim3(P / 64, N), dim3(256), 0, (hipStream_t)stream, T, w0p, w1p, cw, S, P, B);
    return (int)hipGetLastError();
}
extern "C" int floor_split(const float* Y, float* const* out, const long* obs, int N, int C, void* stream) {
    SplitFloorParams p;
    p.Y = Y; p.C = C;
    for (int l = 0; l < 5; ++l) { p.out[l] = out[l]; p.obs[l] = obs[l]; }
    hipLaunchKernelGGL(split_floor_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
// an empty launch of the same grid: the launch + dependency cost alone
extern "C" __global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
extern "C" int floor_empty(int gx, int gy, void* stream) {
    hipLaunchKernelGGL(empty_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int*)nullptr);
    return (int)hipGetLastError();
}
