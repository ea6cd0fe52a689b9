// Weight gradient of a 2-D convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN call inside Conv2dGradWeight of the reference's conv2d_gradfix.py:140-146
// (`aten::cudnn_convolution_backward_weight` / `..._transpose_backward_weight`), i.e. for y = conv2d(x, w, stride s, padding p)
//     dw[o, i, ky, kx] = sum_{n, oy, ox} g[n, o, oy, ox] * x[n, i, oy*s - p + ky, ox*s - p + kx]        (zero outside x).
// The transposed convolution's weight gradient is the same sum with the roles of the two tensors exchanged
// (dw[ci, co, ky, kx] = sum x[n,ci,y,x] * g[n,co, y*s - p + ky, x*s - p + kx]): the host passes g as `x` and x as `g`.
//
// GEMM view: M = 32 output channels, N = 32 input channels (x kh*kw taps), K = all output pixels of the batch.  One workgroup
// owns a 32 x 32 x taps tile of dw for one slice of K; its four waves take every fourth k-step of a 64-pixel chunk (one piece of an
// output row), keep all taps in registers (9 accumulator tiles for a 3x3 kernel) and are summed through LDS in a fixed order.
// Per chunk the 32 x 64 piece of g and the 32 x kh x (64 s + kw - 1) window of x are staged in LDS (odd row pitches: the operand
// reads of 32 different channels hit 32 different banks).  K slices are summed by a second tiny kernel in slice order, so the
// result is deterministic (no atomics).
#include "shg_common.h"

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const float* x;      // [NB, I, H, W]
    const float* g;      // [NB, O, OH, OW]
    float* out;          // dw [O, I, kh, kw] (nslice == 1) or partials [nslice][O][I][kh*kw]
    int NB, I, O, H, W, OH, OW, stride, pad;
    int chunks_x;        // 64-pixel pieces per output row
    int nchunk;          // NB * OH * chunks_x
    int nslice;
};

template <int KH, int KW, int S>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TAPS = KH * KW, PX = 64, XW = PX * S + KW - 1;
    constexpr int GP = PX + 1;                                   // pitch of a g row (odd)
    constexpr int XP = (KH * XW) | 1;                            // pitch of one input channel's window (odd)
    __shared__ float Gs[32 * GP];
    __shared__ float Xs[32 * XP];
    __shared__ float red[3 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32, slice = blockIdx.z;
    const int per = (p.nchunk + p.nslice - 1) / p.nslice;
    const int c_begin = slice * per, c_end = min(p.nchunk, c_begin + per);
    wg_f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int c = c_begin; c < c_end; ++c) {
        const int cx = c % p.chunks_x, oy = (c / p.chunks_x) % p.OH, n = c / (p.chunks_x * p.OH);
        const int ox0 = cx * PX;
        __syncthreads();                                         // previous chunk fully consumed
        for (int e = tid; e < 32 * PX; e += 256) {               // g piece: 32 channels x 64 pixels of output row oy
            const int o = e / PX, px = e % PX;
            const bool ok = o0 + o < p.O && ox0 + px < p.OW;
            Gs[o * GP + px] = ok ? p.g[(((long)n * p.O + o0 + o) * p.OH + oy) * p.OW + ox0 + px] : 0.f;
        }
        for (int e = tid; e < 32 * KH * XW; e += 256) {          // x window: 32 channels x KH rows x (64 s + KW - 1) columns
            const int i = e / (KH * XW), r = (e / XW) % KH, cc = e % XW;
            const int iy = oy * S - p.pad + r, ix = ox0 * S - p.pad + cc;
            const bool ok = i0 + i < p.I && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            Xs[i * XP + r * XW + cc] = ok ? p.x[(((long)n * p.I + i0 + i) * p.H + iy) * p.W + ix] : 0.f;
        }
        __syncthreads();
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {                            // this wave's k-steps: 4q + wave
            const int k = 2 * (4 * q + wave) + half;             // pixel of this lane's operand row
            const float a = Gs[l31 * GP + k];
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[l31 * XP + (t / KW) * XW + k * S + (t % KW)], acc[t], 0, 0, 0);
        }
    }
    // sum the four waves in a fixed order through LDS (32 x 32 floats per tap at a time), wave 0 writes the result
    __syncthreads();
    float* dst = p.out + (p.nslice > 1 ? (long)slice * p.O * p.I * TAPS : 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave - 1) * 1024 + r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = ((acc[t][r] + red[r * 64 + lane]) + red[1024 + r * 64 + lane]) + red[2048 + r * 64 + lane];
                const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half, i = i0 + l31;
                if (o < p.O && i < p.I) dst[((long)o * p.I + i) * TAPS + t] = v;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, long n, int nslice) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float v = 0.f;
    for (int s = 0; s < nslice; ++s) v += part[(long)s * n + e];
    dw[e] = v;
}

static int wgrad_slices(int NB, int I, int O, int OH, int OW) {
    const long tiles = (long)shg_cdiv(I, 32) * shg_cdiv(O, 32);
    const long nchunk = (long)NB * OH * shg_cdiv(OW, 64);
    long s = (4 * 256 + tiles - 1) / tiles;                      // about four workgroups per CU
    if (s > nchunk) s = nchunk;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

// bytes of scratch shg_conv2d_wgrad_f32 needs for this problem (0: none)
extern "C" size_t shg_conv2d_wgrad_workspace_bytes(int NB, int I, int O, int OH, int OW, int kh, int kw) {
    const int s = wgrad_slices(NB, I, O, OH, OW);
    return s > 1 ? (size_t)s * O * I * kh * kw * sizeof(float) : 0;
}

// dw [O, I, kh, kw] = weight gradient of y = conv2d(x [NB,I,H,W], w, stride, pad) given g = dL/dy [NB,O,OH,OW];
// kh x kw = 3x3 or 1x1, stride 1 or 2.  workspace: shg_conv2d_wgrad_workspace_bytes.
extern "C" int shg_conv2d_wgrad_f32(const float* x, const float* g, float* dw, int NB, int I, int O, int H, int W, int OH, int OW,
                                    int kh, int kw, int stride, int pad, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && g && dw, "conv2d_wgrad: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1, "conv2d_wgrad: bad shape");
    SHG_CHECK_ARG((kh == 3 && kw == 3) || (kh == 1 && kw == 1), "conv2d_wgrad: 3x3 and 1x1 kernels only");
    SHG_CHECK_ARG(stride == 1 || stride == 2, "conv2d_wgrad: stride 1 or 2");
    SHG_CHECK_ARG(pad >= 0 && (OH - 1) * stride - pad + kh - 1 < H + pad && (OW - 1) * stride - pad + kw - 1 < W + pad,
                  "conv2d_wgrad: output extent does not match x, stride and padding");
    WgradParams p{};
    p.x = x; p.g = g; p.NB = NB; p.I = I; p.O = O; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.stride = stride; p.pad = pad;
    p.chunks_x = shg_cdiv(OW, 64);
    p.nchunk = NB * OH * p.chunks_x;
    p.nslice = wgrad_slices(NB, I, O, OH, OW);
    const size_t need = p.nslice > 1 ? (size_t)p.nslice * O * I * kh * kw * sizeof(float) : 0;
    SHG_CHECK_ARG(need == 0 || (workspace && ws_bytes >= need), "conv2d_wgrad: workspace too small (shg_conv2d_wgrad_workspace_bytes)");
    p.out = p.nslice > 1 ? (float*)workspace : dw;
    const dim3 grid(shg_cdiv(I, 32), shg_cdiv(O, 32), p.nslice);
    hipStream_t s = (hipStream_t)stream;
    if (kh == 3 && stride == 1) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3, 1>), grid, dim3(256), 0, s, p);
    else if (kh == 3) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3, 2>), grid, dim3(256), 0, s, p);
    else if (stride == 1) hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2>), grid, dim3(256), 0, s, p);
    SHG_CHECK_LAUNCH();
    if (p.nslice > 1) {
        const long n = (long)O * I * kh * kw;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(shg_cdiv(n, 256)), dim3(256), 0, s, (const float*)workspace, dw, n, p.nslice);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
