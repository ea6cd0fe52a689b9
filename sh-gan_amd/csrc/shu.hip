// Spectral Hint Unit kernels (gfx950): rFFT2 + row shift, and band-weighted sum + Gaussian split +
// row unshift + irFFT2 at five resolutions.  Reference: lib/model_zoo/shgan.py:312-336 (SHU.forward),
// :143-160 (heterogeneous_filter), :281-310 (Gaussian-split tables); index conventions in
// SURVEY.md appendix B.  The two 1x1 convolutions in between (64->64 +bias+ReLU, 64->384) and the band sum of
// the heterogeneous filter are one kernel (shu_spectral_kernel) that never writes the [N,384,64,33] tensor;
// other geometries run them on the MFMA implicit-GEMM kernel (conv_mfma.hip) as single-tap convolutions.
//
// The transform is always 64x64 (shu_input_res = 64), so a plane fits in LDS many times over: one
// workgroup owns one (sample, channel) plane, keeps the whole spectrum in LDS and never goes back to
// HBM between the row pass, the column pass and the five inverse transforms.  At N = 64 a direct
// DFT with an LDS twiddle table (64 complex MACs per output) costs < 1 MFLOP per plane.
#include "shg_common.h"

#define SHU_N 64          // transform size (shu_input_res)
#define SHU_NH 33         // half spectrum width

__device__ __forceinline__ void shu_build_twiddles(float2* tw) {
    if (threadIdx.x < SHU_N) {
        float s, c;
        sincospif((float)threadIdx.x / 32.0f, &s, &c);   // angle = 2*pi*m/64
        tw[threadIdx.x] = make_float2(c, s);
    }
}

typedef float shu_f32x16 __attribute__((ext_vector_type(16)));

// x: channel planes [C][64][64] per sample at x + n*xbs.  T: [N, 2C, 64, 33]; ch c = Re, ch C+c = Im,
// rows shifted so DC sits on row 31 (shgan.py:313-319), scaled by 1/4096 (norm='forward').
//
// Both passes are small dense matrix products with the DFT matrices and run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32 products / fp32 accumulation -- same arithmetic as a scalar DFT loop):
//   pass 1 (along w, real input):   P[m][h] = sum_w A1[m][w] x[h][w],  m = 0..32 -> cos rows (Re R[h][m]),
//                                   m = 33..63 -> -sin rows of k = m-32 = 1..31 (Im R[h][k]; Im is 0 at k = 0, 32)
//   pass 2 (along h, complex):      D[m][k] = sum_kk A2[m][kk] B[kk][k],  B = [Re R ; Im R] (128 x 33),
//                                   m < 64: Re T[u=m] = [cos | sin],  m >= 64: Im T[u=m-64] = [-sin | cos]
// One workgroup = one plane, 4 waves = 4 output tiles of 32x32 per pass; column k = 32 of pass 2 (the 33rd) is
// done by 128 threads on the VALU.
__global__ __launch_bounds__(256) void shu_rfft2_shift_kernel(const float* x, long xbs, float* T, int C) {
    __shared__ float xs[SHU_N][SHU_N + 1];
    __shared__ float Rr[SHU_N][SHU_NH], Ri[SHU_N][SHU_NH];
    __shared__ float2 tw[SHU_N];
    const int c = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    shu_build_twiddles(tw);
    const float* xp = x + (long)n * xbs + (long)c * SHU_N * SHU_N;
    for (int e = tid; e < SHU_N * SHU_N; e += 256) xs[e >> 6][e & 63] = xp[e];
    if (tid < SHU_N) { Ri[tid][0] = 0.f; Ri[tid][32] = 0.f; }
    __syncthreads();
    {   // ---- pass 1: tile (mt, nt) = (wave / 2, wave % 2); K = w
        const int m = (wave >> 1) * 32 + l31, h = (wave & 1) * 32 + l31;
        const int km = m < SHU_NH ? m : m - 32;             // frequency of this A row
        shu_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int idx = (km * half) & 63;                          // (km * w) mod 64 for w = 2*ks + half
#pragma unroll 8
        for (int ks = 0; ks < SHU_N / 2; ++ks) {
            const float2 t = tw[idx];
            const float av = m < SHU_NH ? t.x : -t.y;
            const float bv = xs[h][2 * ks + half];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            idx = (idx + 2 * km) & 63;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (mr < SHU_NH) Rr[h][mr] = acc[r]; else Ri[h][mr - 32] = acc[r];
        }
    }
    __syncthreads();
    const float sc = 1.0f / (SHU_N * SHU_N);
    {   // ---- pass 2, columns k = 0..31: m-tile = wave (rows m = wave*32 ..), K = 128 (Re R rows, then Im R rows)
        const int m = wave * 32 + l31, u = m & 63;
        const bool imrow = m >= 64;
        shu_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int idx = (u * half) & 63;                           // (u * h) mod 64, h = (2*ks + half) mod 64
#pragma unroll 8
        for (int ks = 0; ks < SHU_N; ++ks) {
            const int kk = 2 * ks + half, h = kk & 63;
            const bool second = kk >= 64;                     // B row from Im R
            const float2 t = tw[idx];
            // Re T = cos*Rr + sin*Ri ;  Im T = -sin*Rr + cos*Ri      (e^{-i theta})
            const float av = imrow ? (second ? t.x : -t.y) : (second ? t.y : t.x);
            const float bv = second ? Ri[h][l31] : Rr[h][l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            idx = (idx + 2 * u) & 63;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int ur = mr & 63, row = (ur + 31) & 63;
            T[(((long)n * 2 * C + (mr >= 64 ? C : 0) + c) * SHU_N + row) * SHU_NH + l31] = acc[r] * sc;
        }
    }
    if (tid < 2 * SHU_N) {   // ---- pass 2, column k = 32 (Im R = 0 there)
        const int u = tid & 63;
        const bool imrow = tid >= 64;
        float v = 0.f;
        for (int h = 0; h < SHU_N; ++h) {
            const float2 t = tw[(u * h) & 63];
            v += (imrow ? -t.y : t.x) * Rr[h][32];
        }
        T[(((long)n * 2 * C + (imrow ? C : 0) + c) * SHU_N + ((u + 31) & 63)) * SHU_NH + 32] = v * sc;
    }
}

extern "C" int shg_shu_rfft2_shift_f32(const float* x, long x_batch_stride, float* T, int N, int C, void* stream) {
    SHG_CHECK_ARG(x && T, "shu_rfft2: null pointer");
    SHG_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1, "shu_rfft2: bad shape");
    hipLaunchKernelGGL(shu_rfft2_shift_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, x_batch_stride, T, C);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---- fused spectral stage: S[n,o,p] = sum_k cw[k,p] * sum_i W1[o*B+k, i] * relu(sum_j W0[i,j] T[n,j,p] + b0[i])
// (SHU conv0 + ReLU `shgan.py:320-321`, heterogeneous_filter `shgan.py:143-160`: 1x1 conv to O*B channels, view [O,B], weighted sum
// over the B bands with the position-dependent table cw).  cw depends on the position only, so it moves into the B operand:
//   S[o,p] = sum_{(k,i)} W1[o*B+k, i] * (cw[k,p] * t[i,p])          -- ONE GEMM with K = B*64 instead of B GEMMs and a reduction.
// Workgroup = one sample x 64 consecutive spectral positions (64*33 = 33 such tiles), 4 waves = the 2 x 2 tiles of 32 x 32 of both
// products on v_mfma_f32_32x32x2_f32; T tile and t in LDS (32 KB), weights pre-packed per k-step in MFMA A-operand order
// (w0p [32][2][64], w1p [B*32][2][64]: lane l of row block mo holds W[mo*32 + (l & 31)][2*ks + (l >> 5)]).
struct ShuSpectralParams {
    const float* T; const float* w0p; const float* b0; const float* w1p; const float* cw; float* S;
    int P, B;       // positions per plane (64*33), bands
};

__global__ __launch_bounds__(256) void shu_spectral_kernel(const ShuSpectralParams p) {
    __shared__ __attribute__((aligned(16))) float Tl[64][64];       // [channel][position]
    __shared__ __attribute__((aligned(16))) float tl[64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int mo = wave >> 1, nt = wave & 1;
    const int n = blockIdx.y, p0 = blockIdx.x * 64;
    const float* Tn = p.T + (long)n * 64 * p.P + p0;
    // T tile: 64 channels x 64 positions, rows of 256 bytes
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 256 + tid, ch = e >> 4, c4 = e & 15;
        *reinterpret_cast<float4*>(&Tl[ch][4 * c4]) = *reinterpret_cast<const float4*>(Tn + (long)ch * p.P + 4 * c4);
    }
    float cwv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cwv[k] = k < p.B ? p.cw[(long)k * p.P + p0 + nt * 32 + l31] : 0.f;
    __syncthreads();
    shu_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {   // t = relu(W0 T + b0)
        const float* ap = p.w0p + mo * 64 + lane;
#pragma unroll 8
        for (int ks = 0; ks < 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ks * 128], Tl[2 * ks + half][nt * 32 + l31], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            tl[row][nt * 32 + l31] = fmaxf(acc[r] + p.b0[row], 0.f);
            acc[r] = 0.f;
        }
    }
    __syncthreads();
    {   // S = [W1_0 | W1_1 | ...] (cw (.) t)
        const float* ap = p.w1p + mo * 64 + lane;
        for (int k = 0; k < p.B; ++k) {
            const float cwk = cwv[0];
#pragma unroll 8
            for (int ks = 0; ks < 32; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(k * 32 + ks) * 128], cwk * tl[2 * ks + half][nt * 32 + l31], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 7; ++q) cwv[q] = cwv[q + 1];     // (rotate: keeps the band index static)
        }
    }
    float* Sn = p.S + (long)n * 64 * p.P + p0 + nt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Sn[(long)row * p.P] = acc[r];
    }
}

// T [N,64,P] (P = 64*33 positions, a multiple of 64), w0p / w1p: MFMA-ordered weights (see above), b0 [64], cw [B,P], B <= 8
// -> S [N,64,P].  The shipped SHU geometry only (2C = 64 spectral channels); other shapes: shg_conv2d_f32 twice.
extern "C" int shg_shu_spectral_f32(const float* T, const float* w0p, const float* b0, const float* w1p, const float* cw, float* S,
                                    int N, int C2, int P, int bands, void* stream) {
    SHG_CHECK_ARG(T && w0p && b0 && w1p && cw && S, "shu_spectral: null pointer");
    SHG_CHECK_ARG(C2 == 64 && P % 64 == 0 && bands >= 1 && bands <= 8 && N >= 1 && N <= 65535,
                  "shu_spectral: built for 64 spectral channels, P %% 64 == 0, at most 8 bands");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(S)) & 15) == 0, "shu_spectral: T / S must be 16-byte aligned");
    ShuSpectralParams p{T, w0p, b0, w1p, cw, S, P, bands};
    hipLaunchKernelGGL(shu_spectral_kernel, dim3(P / 64, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

struct ShuSplitParams {
    const float* Y;        // [N, 2C*B, 64, 33] when B > 1 (bands to reduce with cw) or [N, 2C, 64, 33] when B == 1
    const float* cw;       // [B, 64, 33]
    const float* gauss[5]; // level l: [r, r/2+1], r = 4 << l
    float* out[5];         // level l: planes [C][r][r] per sample at out[l] + n*obs[l]
    long obs[5];
    int C, B, accumulate;
};

__global__ __launch_bounds__(256) void shu_split_irfft2_kernel(const ShuSplitParams p) {
    __shared__ float2 S[SHU_N][SHU_NH];
    __shared__ float2 Z[SHU_N][SHU_NH];
    __shared__ float2 tw[SHU_N];
    __shared__ float gl[12 + 40 + 144 + 544];   // Gaussian-split tables of the four scalar levels (r = 4 .. 32: r * (r/2 + 1) floats each)
    const int c = blockIdx.x, n = blockIdx.y;
    const int C = p.C, B = p.B;
    shu_build_twiddles(tw);
    {   // the scalar levels read their table once per (row, column, j): from LDS, not from global memory inside the DFT loops
        int off = 0;
        for (int l = 0; l < 4; ++l) {
            const int cnt = (4 << l) * ((2 << l) + 1);
            if (p.out[l]) for (int e = threadIdx.x; e < cnt; e += 256) gl[off + e] = p.gauss[l][e];
            off += cnt;
        }
    }
    // heterogeneous band sum: flat conv-output channel = o*B + k  (shgan.py:157-160)
    const long plane = SHU_N * SHU_NH;
    const float* yre = p.Y + ((long)n * 2 * C + c) * B * plane;
    const float* yim = p.Y + ((long)n * 2 * C + C + c) * B * plane;
    for (int e = threadIdx.x; e < plane; e += 256) {
        float re = 0.f, im = 0.f;
        for (int k = 0; k < B; ++k) {
            const float wgt = B > 1 ? p.cw[k * plane + e] : 1.f;
            re += yre[k * plane + e] * wgt;
            im += yim[k * plane + e] * wgt;
        }
        S[e / SHU_NH][e % SHU_NH] = make_float2(re, im);
    }
    __syncthreads();
    for (int l = 0; l < 4; ++l) {          // r = 4 .. 32: scalar DFTs (together 1/7 of the work of the 64 x 64 level)
        const int r = 4 << l, rh = r / 2 + 1, tstep = SHU_N / r;
        if (!p.out[l]) continue;
        const float* g = gl + (l == 0 ? 0 : l == 1 ? 12 : l == 2 ? 52 : 196);
        // complex inverse DFT over rows of the cropped, weighted, un-shifted block (shgan.py:328-334)
        for (int e = threadIdx.x; e < r * rh; e += 256) {
            const int y = e / rh, w = e - y * rh;
            float re = 0.f, im = 0.f;
            for (int j = 0; j < r; ++j) {
                const int a = (j + r / 2 - 1) & (r - 1);          // row inside the crop
                const float2 v = S[SHU_N / 2 - r / 2 + a][w];
                const float gw = g[a * rh + w];
                const float2 t = tw[(j * y * tstep) & 63];
                re += gw * (v.x * t.x - v.y * t.y);               // (a+bi)(c + si)
                im += gw * (v.x * t.y + v.y * t.x);
            }
            Z[y][w] = make_float2(re, im);
        }
        __syncthreads();
        // half-complex -> real along x (imaginary parts of the DC and Nyquist bins are ignored, as in c2r)
        float* op = p.out[l] + (long)n * p.obs[l] + (long)c * r * r;
        for (int e = threadIdx.x; e < r * r; e += 256) {
            const int y = e / r, x = e - y * r;
            float v = Z[y][0].x + ((x & 1) ? -Z[y][r / 2].x : Z[y][r / 2].x);
            for (int w = 1; w < r / 2; ++w) {
                const float2 t = tw[(w * x * tstep) & 63];
                v += 2.f * (Z[y][w].x * t.x - Z[y][w].y * t.y);
            }
            op[e] = p.accumulate ? op[e] + v : v;
        }
        __syncthreads();
    }
    if (!p.out[4]) return;
    // ---- r = 64 level on the matrix cores (same scheme as shu_rfft2_shift_kernel, inverse signs):
    //   V[j][w] = g[a][w] * S[a][w], a = (j + 31) & 63                     (weight + row un-shift, shgan.py:328-334)
    //   pass A: [Re Z ; Im Z][m][w] = sum_kk A[m][kk] [Re V ; Im V][kk][w]   A = [[cos, -sin], [sin, cos]](theta j y)
    //   pass B: out[y][x] = sum_kk Zc[y][kk] Bm[kk][x],  kk < 33: Re Z[y][w=kk] * c_w cos(theta w x) (c_0 = c_32 = 1, else 2),
    //           kk >= 33: Im Z[y][w=kk-32] * (-2 sin(theta w x))            (c2r ignores Im of the DC / Nyquist bins)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    float* Vr = reinterpret_cast<float*>(&Z[0][0]);          // [64][33]
    float* Vi = Vr + SHU_N * SHU_NH;
    {
        const float* g = p.gauss[4];
        for (int e = tid; e < SHU_N * SHU_NH; e += 256) {
            const int j = e / SHU_NH, w = e - j * SHU_NH;
            const int a = (j + 31) & 63;
            const float2 v = S[a][w];
            const float gw = g[a * SHU_NH + w];
            Vr[e] = gw * v.x; Vi[e] = gw * v.y;
        }
    }
    __syncthreads();
    float* Zr = reinterpret_cast<float*>(&S[0][0]);          // [64][33]  (S is no longer needed)
    float* Zi = Zr + SHU_N * SHU_NH;
    {   // pass A, columns w = 0..31: m-tile = wave, K = 128 (Re V rows, then Im V rows)
        const int m = wave * 32 + l31, y = m & 63;
        const bool imrow = m >= 64;
        shu_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int idx = (y * half) & 63;
#pragma unroll 8
        for (int ks = 0; ks < SHU_N; ++ks) {
            const int kk = 2 * ks + half, j = kk & 63;
            const bool second = kk >= 64;
            const float2 t = tw[idx];
            // Re Z = cos*Vr - sin*Vi ;  Im Z = sin*Vr + cos*Vi      (e^{+i theta})
            const float av = imrow ? (second ? t.x : t.y) : (second ? -t.y : t.x);
            const float bv = (second ? Vi : Vr)[j * SHU_NH + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            idx = (idx + 2 * y) & 63;
        }
        // column w = 32 on the VALU (threads 0..127: row m = tid)
        float v32 = 0.f;
        if (tid < 2 * SHU_N) {
            const int yy = tid & 63;
            const bool im2 = tid >= 64;
            for (int j = 0; j < SHU_N; ++j) {
                const float2 t = tw[(yy * j) & 63];
                const float vr = Vr[j * SHU_NH + 32], vi = Vi[j * SHU_NH + 32];
                v32 += im2 ? (t.y * vr + t.x * vi) : (t.x * vr - t.y * vi);
            }
        }
        __syncthreads();                 // everyone is done reading S-derived data? (S itself was last read above) -> Zr/Zi may be written
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            (mr >= 64 ? Zi : Zr)[(mr & 63) * SHU_NH + l31] = acc[r];
        }
        if (tid < 2 * SHU_N) (tid >= 64 ? Zi : Zr)[(tid & 63) * SHU_NH + 32] = v32;
    }
    __syncthreads();
    {   // pass B: tile (mt, nt) = (wave / 2, wave % 2), rows m = y, columns n = x, K = 64
        const int y = (wave >> 1) * 32 + l31, x = (wave & 1) * 32 + l31;
        shu_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
        for (int ks = 0; ks < SHU_N / 2; ++ks) {
            const int kk = 2 * ks + half;
            const bool second = kk >= SHU_NH;
            const int w = second ? kk - 32 : kk;
            const float2 t = tw[(w * x) & 63];
            const float coef = (w == 0 || w == 32) ? 1.f : 2.f;
            const float av = (second ? Zi : Zr)[y * SHU_NH + w];
            const float bv = second ? -2.f * t.y : coef * t.x;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        float* op = p.out[4] + (long)n * p.obs[4] + (long)c * SHU_N * SHU_N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int yr = (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = op + yr * SHU_N + x;
            *dst = p.accumulate ? *dst + acc[r] : acc[r];
        }
    }
}

// gauss / out / out_batch_stride: arrays of 5 entries for r = 4, 8, 16, 32, 64 (out[l] may be null to skip).
extern "C" int shg_shu_split_irfft2_f32(const float* Y, const float* cw, const float* const* gauss, float* const* out,
                                        const long* out_batch_stride, int N, int C, int bands, int accumulate, void* stream) {
    SHG_CHECK_ARG(Y && gauss && out && out_batch_stride, "shu_split: null pointer");
    SHG_CHECK_ARG(bands >= 1 && (bands == 1 || cw), "shu_split: cw required when bands > 1");
    SHG_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1, "shu_split: bad shape");
    ShuSplitParams p;
    p.Y = Y; p.cw = cw; p.C = C; p.B = bands; p.accumulate = accumulate;
    for (int l = 0; l < 5; ++l) {
        p.gauss[l] = gauss[l]; p.out[l] = out[l]; p.obs[l] = out_batch_stride[l];
        SHG_CHECK_ARG(!out[l] || gauss[l], "shu_split: missing gaussian table for level %d", l);
    }
    hipLaunchKernelGGL(shu_split_irfft2_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---- adjoint of the split stage (training rows: the gradient of shgan.py:326-336 w.r.t. the filtered spectrum).
// out_r = irfft2( unshift( crop_r(S) * gauss_r ) ) is linear in S; with G_r = dL/dout_r its transpose is
//   dL/dS[32 - r/2 + j][kx] += gauss_r[j][kx] * c_kx * DFT2_r(G_r)[ky][kx],   j = (ky + r/2 - 1) mod r,  c_0 = c_{r/2} = 1, else 2
// (the c2r pass counts the interior columns twice and ignores Im of the DC / Nyquist columns -- whose adjoint terms vanish by
// themselves: the forward DFT of a real row is real there).  One workgroup = one (sample, channel) plane; the five levels are scalar
// DFTs from LDS (together < 1.5 MFLOP per plane; this kernel runs once per training step on 32 channels), accumulated in LDS.
struct ShuSplitAdjParams {
    const float* g[5];     // level l: planes [C][r][r] per sample at g[l] + n*gbs[l]; null = no gradient for that level
    long gbs[5];
    const float* gauss[5];
    float* GS;             // [N, 2C, 64, 33]: Re planes, then Im planes
    int C;
};

__global__ __launch_bounds__(256) void shu_split_adjoint_kernel(const ShuSplitAdjParams p) {
    __shared__ float2 acc[SHU_N][SHU_NH];
    __shared__ float2 T1[SHU_N][SHU_NH];
    __shared__ float gl[SHU_N * SHU_N];
    __shared__ float2 tw[SHU_N];
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    shu_build_twiddles(tw);
    for (int e = tid; e < SHU_N * SHU_NH; e += 256) acc[e / SHU_NH][e % SHU_NH] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int l = 0; l < 5; ++l) {
        if (!p.g[l]) continue;
        const int r = 4 << l, rh = r / 2 + 1, tstep = SHU_N / r;
        const float* gp = p.g[l] + (long)n * p.gbs[l] + (long)c * r * r;
        for (int e = tid; e < r * r; e += 256) gl[e] = gp[e];
        __syncthreads();
        for (int e = tid; e < r * rh; e += 256) {               // along x: real -> half complex, e^{-i theta}
            const int y = e / rh, kx = e - y * rh;
            float re = 0.f, im = 0.f;
            for (int x = 0; x < r; ++x) {
                const float2 t = tw[(kx * x * tstep) & 63];
                const float v = gl[y * r + x];
                re += v * t.x; im -= v * t.y;
            }
            T1[y][kx] = make_float2(re, im);
        }
        __syncthreads();
        const float* gw = p.gauss[l];
        for (int e = tid; e < r * rh; e += 256) {               // along y: complex, e^{-i theta}; weight, shift, embed
            const int ky = e / rh, kx = e - ky * rh;
            float re = 0.f, im = 0.f;
            for (int y = 0; y < r; ++y) {
                const float2 t = tw[(ky * y * tstep) & 63];
                const float2 v = T1[y][kx];
                re += v.x * t.x + v.y * t.y;                     // (a + bi)(c - si)
                im += v.y * t.x - v.x * t.y;
            }
            const int j = (ky + r / 2 - 1) & (r - 1);
            const float wgt = gw[j * rh + kx] * ((kx == 0 || kx == r / 2) ? 1.f : 2.f);
            float2& a = acc[SHU_N / 2 - r / 2 + j][kx];
            a.x += wgt * re; a.y += wgt * im;
        }
        __syncthreads();
    }
    const long plane = SHU_N * SHU_NH;
    float* ore = p.GS + ((long)n * 2 * p.C + c) * plane;
    float* oim = p.GS + ((long)n * 2 * p.C + p.C + c) * plane;
    for (int e = tid; e < plane; e += 256) {
        const float2 a = acc[e / SHU_NH][e % SHU_NH];
        ore[e] = a.x; oim[e] = a.y;
    }
}

// g / g_batch_stride / gauss: arrays of 5 entries for r = 4, 8, 16, 32, 64 (g[l] may be null); GS [N, 2C, 64, 33] is overwritten.
extern "C" int shg_shu_split_adjoint_f32(const float* const* g, const long* g_batch_stride, const float* const* gauss, float* GS, int N, int C,
                                         void* stream) {
    SHG_CHECK_ARG(g && g_batch_stride && gauss && GS, "shu_split_adjoint: null pointer");
    SHG_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1, "shu_split_adjoint: bad shape");
    ShuSplitAdjParams p;
    p.GS = GS; p.C = C;
    for (int l = 0; l < 5; ++l) {
        p.g[l] = g[l]; p.gbs[l] = g_batch_stride[l]; p.gauss[l] = gauss[l];
        SHG_CHECK_ARG(!g[l] || gauss[l], "shu_split_adjoint: missing gaussian table for level %d", l);
    }
    hipLaunchKernelGGL(shu_split_adjoint_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
