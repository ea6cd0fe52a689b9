// upfirdn2d for gfx950: pad -> zero-insert upsample -> 2-D FIR -> decimate, NCHW fp32.
//
// Native counterpart of the reference's only CUDA op:
//   lib/model_zoo/stylegan_utils/upfirdn2d.cpp:16-94  (host entry, output-size rule :32-33)
//   lib/model_zoo/stylegan_utils/upfirdn2d.cu:29-200  (large / small kernels)
// Semantics are those of upfirdn2d.py:98-138 (`_upfirdn2d_ref`): the filter is applied as a true
// convolution unless `flip`, and multiplied by `gain`.
//
// MI355X design: HBM-bound streaming op.  Two kernels:
//  * fir_same_kernel<FH,FW,TY>: up = down = 1 (every FIR the generator issues on activations).
//    One lane per output column (coalesced 256 B rows per wave), TY vertically adjacent outputs
//    per lane so the (TY+FH-1) x FW window is loaded once into registers and reused; an optional
//    fused epilogue (per-(n,c) scale, noise, bias, lrelu_agc, residual) makes the synthesis
//    `convT -> FIR -> +noise -> +bias -> act -> +skip` tail (stylegan.py:295-304,
//    comodgan.py:326-327) a single pass over HBM.
//  * upfirdn_generic_kernel: any up/down/padding/filter (RGB skip upsample, D's down path, tests).
#include "shg_common.h"

struct UfdParams {
    const float* x;
    const float* f;
    float* y;
    int NC, C, H, W, OH, OW;
    int fh, fw, upx, upy, dnx, dny, px0, py0, flip;
    float gain;
    // fused epilogue (all optional)
    const float* scale;      // [NC] multiplies the filtered value (demodulation coefficient)
    const float* bias;       // [C]
    const float* noise;      // noise_mode 1: [OH,OW]; 2: [N,OH,OW]
    const float* residual;   // [NC,OH,OW], added after the activation
    int noise_mode;
    float noise_strength;
    int act;                 // 0 none, 1 lrelu_agc
    float alpha, act_gain, clamp;
    int has_epilogue;
};

__device__ __forceinline__ float ufd_epilogue(const UfdParams& p, float v, int nc, int oy, int ox) {
    if (!p.has_epilogue) return v;
    const int c = nc % p.C, n = nc / p.C;
    if (p.scale) v *= p.scale[nc];
    if (p.noise_mode == 1) v += p.noise[oy * p.OW + ox] * p.noise_strength;
    else if (p.noise_mode == 2) v += p.noise[((long)n * p.OH + oy) * p.OW + ox] * p.noise_strength;
    if (p.bias) v += p.bias[c];
    if (p.act) v = shg_lrelu_agc(v, p.alpha, p.act_gain, p.clamp);
    if (p.residual) v += p.residual[((long)nc * p.OH + oy) * p.OW + ox];
    return v;
}

// up = down = 1, FH x FW filter.  One workgroup = a 16 x 64 output tile of one (n,c) plane at a time: the
// (16+FH-1) x (64+FW-1) input window is staged in LDS with coalesced row loads (1.2 loads per output instead
// of FH*FW), then every lane produces 4 vertically adjacent outputs from a register window.
// Per-plane part of the epilogue, evaluated once per (n,c) plane instead of once per output.
struct UfdPlane { float sc, bs; const float* nz; const float* res; };
__device__ __forceinline__ UfdPlane ufd_plane(const UfdParams& p, int nc) {
    UfdPlane q;
    const int n = nc / p.C, c = nc - n * p.C;
    q.sc = (p.has_epilogue && p.scale) ? p.scale[nc] : 1.f;
    q.bs = (p.has_epilogue && p.bias) ? p.bias[c] : 0.f;
    q.nz = !p.has_epilogue || p.noise_mode == 0 ? nullptr : (p.noise_mode == 1 ? p.noise : p.noise + (long)n * p.OH * p.OW);
    q.res = (p.has_epilogue && p.residual) ? p.residual + (long)nc * p.OH * p.OW : nullptr;
    return q;
}
__device__ __forceinline__ float ufd_finish(const UfdParams& p, const UfdPlane& q, float v, int pix) {
    if (!p.has_epilogue) return v;
    v *= q.sc;
    if (q.nz) v += q.nz[pix] * p.noise_strength;
    v += q.bs;
    if (p.act) v = shg_lrelu_agc(v, p.alpha, p.act_gain, p.clamp);
    if (q.res) v += q.res[pix];
    return v;
}

template <int FH, int FW>
__global__ __launch_bounds__(256) void fir_same_kernel(const UfdParams p) {
    constexpr int TH = 16, TW = 64, IH = TH + FH - 1, IW = TW + FW - 1, PITCH = IW + 1;
    __shared__ float tile[IH * PITCH];
    __shared__ float sf[FH * FW];
    if (threadIdx.x < FH * FW) {
        // stored so that sf[ky][kx] multiplies x[oy + ky - py0][ox + kx - px0]
        const int ky = threadIdx.x / FW, kx = threadIdx.x % FW;
        const int sy = p.flip ? ky : FH - 1 - ky, sx = p.flip ? kx : FW - 1 - kx;
        sf[threadIdx.x] = p.f[sy * FW + sx] * p.gain;
    }
    __syncthreads();
    float fr[FH * FW];
#pragma unroll
    for (int k = 0; k < FH * FW; ++k) fr[k] = sf[k];
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int tx = threadIdx.x & 63, ty = (threadIdx.x >> 6) * 4;
    for (int nc = blockIdx.z; nc < p.NC; nc += gridDim.z) {
        const float* xp = p.x + (long)nc * p.H * p.W;
        const UfdPlane pl = ufd_plane(p, nc);
        for (int e = threadIdx.x; e < IH * IW; e += 256) {
            const int r = e / IW, c = e - r * IW;
            const int iy = oy0 + r - p.py0, ix = ox0 + c - p.px0;
            tile[r * PITCH + c] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? xp[(long)iy * p.W + ix] : 0.f;
        }
        __syncthreads();
        float win[4 + FH - 1][FW];
#pragma unroll
        for (int r = 0; r < 4 + FH - 1; ++r)
#pragma unroll
            for (int k = 0; k < FW; ++k) win[r][k] = tile[(ty + r) * PITCH + tx + k];
        const int ox = ox0 + tx;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oy = oy0 + ty + t;
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < FH; ++ky)
#pragma unroll
                for (int kx = 0; kx < FW; ++kx) v += win[t + ky][kx] * fr[ky * FW + kx];
            if (oy < p.OH && ox < p.OW) p.y[((long)nc * p.OH + oy) * p.OW + ox] = ufd_finish(p, pl, v, oy * p.OW + ox);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void upfirdn_generic_kernel(const UfdParams p) {
    extern __shared__ float sfg[];
    for (int k = threadIdx.x; k < p.fh * p.fw; k += 256) {
        const int ky = k / p.fw, kx = k - ky * p.fw;
        const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
        sfg[k] = p.f[sy * p.fw + sx] * p.gain;
    }
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= p.OW || oy >= p.OH) return;
    // y[oy,ox] = sum_k xup[oy*dn + ky - pad0] * fflip[ky], xup[u] = x[u/up] when up | u
    for (int nc = blockIdx.z; nc < p.NC; nc += gridDim.z) {
        const float* xp = p.x + (long)nc * p.H * p.W;
        float v = 0.f;
        for (int ky = 0; ky < p.fh; ++ky) {
            const int uy = oy * p.dny + ky - p.py0;
            if (uy < 0 || (uy % p.upy) != 0) continue;
            const int iy = uy / p.upy;
            if (iy >= p.H) continue;
            for (int kx = 0; kx < p.fw; ++kx) {
                const int ux = ox * p.dnx + kx - p.px0;
                if (ux < 0 || (ux % p.upx) != 0) continue;
                const int ix = ux / p.upx;
                if (ix >= p.W) continue;
                v += xp[(long)iy * p.W + ix] * sfg[ky * p.fw + kx];
            }
        }
        p.y[((long)nc * p.OH + oy) * p.OW + ox] = ufd_epilogue(p, v, nc, oy, ox);
    }
}

// FIR after the all-phase transposed convolution (conv_mfma.hip, out_mode 1): the (2H+1)x(2W+1)
// intermediate lives as four phase planes mid[(a*2+b)][nc][u][v] = full[2u+a][2v+b], each (H+1)x(W+1).
// y[Y,X] = sum_{ky,kx<4} fk[ky][kx] * full[Y+ky-1][X+kx-1]   (pad [1,1,1,1], conv2d_resample.py:138).
// One workgroup = an 8 x 128 tile of low-resolution pixels (16 x 256 outputs) of one (n,c) plane at a time:
// the four 10 x 130 phase windows are staged in LDS with coalesced row loads; each lane owns two horizontally
// adjacent low-res pixels, assembles their 5 x 7 neighbourhood from LDS and produces 2 x 4 outputs, so that the
// skip tensor, the noise and the result move as 16-byte accesses.
template <bool VEC>
__global__ __launch_bounds__(256) void fir_up_planar_kernel(const UfdParams p) {
    constexpr int TU = 8, TV = 128, PU = TU + 2, PV = TV + 2, PITCH = PV + 1;
    __shared__ float tile[4][PU * PITCH];
    __shared__ float sf[16];
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        const int sy = p.flip ? ky : 3 - ky, sx = p.flip ? kx : 3 - kx;
        sf[threadIdx.x] = p.f[sy * 4 + sx] * p.gain;
    }
    __syncthreads();
    float fr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) fr[k] = sf[k];
    const int v0 = blockIdx.x * TV, u0 = blockIdx.y * TU;
    const int tv = (threadIdx.x & 63) * 2, tu0 = (threadIdx.x >> 6) * 2;   // lane: low-res cols tv, tv+1; rows tu0, tu0+1
    const int PWg = p.W + 1;
    const long P = (long)(p.H + 1) * PWg;
    for (int nc = blockIdx.z; nc < p.NC; nc += gridDim.z) {
        const UfdPlane plq = ufd_plane(p, nc);
        // plane (a,b) window: rows u0-a .. , cols v0-b ..   (full row Y = 2u+a, col X = 2v+b)
        for (int e = threadIdx.x; e < 4 * PU * PV; e += 256) {
            const int pl = e / (PU * PV), rem = e - pl * (PU * PV);
            const int r = rem / PV, c = rem - r * PV;
            const int a = pl >> 1, b = pl & 1;
            const int u = u0 - a + r, v = v0 - b + c;
            // valid full-resolution rows: Y = 2u+a in [0, 2H]  <=>  u in [0, H-a]
            const bool ok = u >= 0 && u <= p.H - a && v >= 0 && v <= p.W - b;
            tile[pl][r * PITCH + c] = ok ? p.x[((long)pl * p.NC + nc) * P + (long)u * PWg + v] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int tu = tu0 + q;
            const int u = u0 + tu, v = v0 + tv;
            // neighbourhood rows Y = 2u-1 .. 2u+3 and cols X = 2v-1 .. 2v+5; element (r,c) lives in plane
            // (a,b) = ((r+1)&1, (c+1)&1) at window position (tu + (r>>1) + ..., tv + ...)
            float m[5][7];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int a = (r + 1) & 1;
                const int ur = tu + ((r + 1) >> 1) - 1 + a;
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    const int b = (c + 1) & 1;
                    const int vc = tv + ((c + 1) >> 1) - 1 + b;
                    m[r][c] = tile[a * 2 + b][ur * PITCH + vc];
                }
            }
            if (u < p.H && v < p.W) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    float o4[4];
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) {
                        float acc = 0.f;
#pragma unroll
                        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 4; ++kx) acc += m[dy + ky][dx + kx] * fr[ky * 4 + kx];
                        o4[dx] = acc;
                    }
                    const int pix = (2 * u + dy) * p.OW + 2 * v;
                    float* yp = p.y + (long)nc * p.OH * p.OW + pix;
                    if (VEC && v + 1 < p.W) {
                        // 16-byte path: scale / noise / bias / activation / skip on four outputs at once
                        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.has_epilogue && plq.nz) nz = *reinterpret_cast<const float4*>(plq.nz + pix);
                        if (p.has_epilogue && plq.res) rs = *reinterpret_cast<const float4*>(plq.res + pix);
                        const float nzv[4] = {nz.x, nz.y, nz.z, nz.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                        float o[4];
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx) {
                            float t = o4[dx];
                            if (p.has_epilogue) {
                                t = t * plq.sc + nzv[dx] * p.noise_strength + plq.bs;
                                if (p.act) t = shg_lrelu_agc(t, p.alpha, p.act_gain, p.clamp);
                                t += rsv[dx];
                            }
                            o[dx] = t;
                        }
                        *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx)
                            if (2 * v + dx < p.OW) yp[dx] = ufd_finish(p, plq, o4[dx], pix + dx);
                    }
                }
            }
        }
        __syncthreads();
    }
}

static int ufd_launch(UfdParams& p, hipStream_t s) {
    const int gz = p.NC < 32768 ? p.NC : 32768;
    if (p.upx == 1 && p.upy == 1 && p.dnx == 1 && p.dny == 1 && p.fh == 4 && p.fw == 4) {
        dim3 grid(shg_cdiv(p.OW, 64), shg_cdiv(p.OH, 16), gz);
        hipLaunchKernelGGL((fir_same_kernel<4, 4>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid(shg_cdiv(p.OW, 64), shg_cdiv(p.OH, 4), gz);
        hipLaunchKernelGGL(upfirdn_generic_kernel, grid, dim3(256), sizeof(float) * p.fh * p.fw, s, p);
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

static int ufd_fill(UfdParams& p, const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                    int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain) {
    SHG_CHECK_ARG(x && f && y, "upfirdn2d: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1, "upfirdn2d: x must be rank 4 and non-empty");
    SHG_CHECK_ARG(fh >= 1 && fw >= 1, "upfirdn2d: f must be at least 1x1");                       // upfirdn2d.cpp:26
    SHG_CHECK_ARG(fh * fw <= 4096, "upfirdn2d: filter too large");
    SHG_CHECK_ARG(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");       // :27
    SHG_CHECK_ARG(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1"); // :28
    SHG_CHECK_ARG((long)N * C * H * W <= 2147483647L, "upfirdn2d: x is too large");               // :22
    const int OW = (W * upx + padx0 + padx1 - fw + downx) / downx;                                // :32
    const int OH = (H * upy + pady0 + pady1 - fh + downy) / downy;                                // :33
    SHG_CHECK_ARG(OW >= 1 && OH >= 1, "upfirdn2d: output must be at least 1x1");                  // :34
    SHG_CHECK_ARG((long)N * C * OH * OW <= 2147483647L, "upfirdn2d: output is too large");        // :36
    p = UfdParams{};
    p.x = x; p.f = f; p.y = y; p.NC = N * C; p.C = C; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
    p.fh = fh; p.fw = fw; p.upx = upx; p.upy = upy; p.dnx = downx; p.dny = downy; p.px0 = padx0; p.py0 = pady0;
    p.flip = flip ? 1 : 0; p.gain = gain;
    return SHG_OK;
}

// Drop-in for upfirdn2d_plugin.upfirdn2d (upfirdn2d.cpp:16); the caller allocates y with the
// extent given by shg_upfirdn2d_out_size.
extern "C" int shg_upfirdn2d_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                                 int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                                 float gain, void* stream) {
    UfdParams p;
    int rc = ufd_fill(p, x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain);
    if (rc != SHG_OK) return rc;
    return ufd_launch(p, (hipStream_t)stream);
}

extern "C" int shg_upfirdn2d_out_size(int H, int W, int fh, int fw, int upx, int upy, int downx, int downy, int padx0,
                                      int padx1, int pady0, int pady1, int* OH, int* OW) {
    SHG_CHECK_ARG(OH && OW, "upfirdn2d_out_size: null pointer");
    *OW = (W * upx + padx0 + padx1 - fw + downx) / downx;
    *OH = (H * upy + pady0 + pady1 - fh + downy) / downy;
    return SHG_OK;
}

// FIR + fused synthesis-layer tail: y = act(FIR(x)*gain*scale[n,c] + noise*strength + bias[c]) + residual.
extern "C" int shg_upfirdn2d_epilogue_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                                          int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                                          int flip, float gain, const float* scale, const float* bias, const float* noise,
                                          int noise_mode, float noise_strength, int act, float alpha, float act_gain,
                                          float clamp, const float* residual, void* stream) {
    UfdParams p;
    int rc = ufd_fill(p, x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain);
    if (rc != SHG_OK) return rc;
    SHG_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2, "upfirdn2d_epilogue: bad noise_mode");
    p.scale = scale; p.bias = bias; p.noise = noise; p.residual = residual;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.act_gain = act_gain; p.clamp = clamp; p.has_epilogue = 1;
    return ufd_launch(p, (hipStream_t)stream);
}

// Second half of the up-sampling synthesis layer: 4x4 FIR (pad 1, `gain`) over the four phase planes written by
// shg_conv2d_f32(mode 2, out_mode 1) -> y [N,C,2H,2W], fused with scale/noise/bias/lrelu_agc/residual
// (stylegan.py:295-304, comodgan.py:326-327).  H, W are the LOW-resolution extents.
extern "C" int shg_upfir_planar_f32(const float* mid, const float* f, float* y, int N, int C, int H, int W, int flip, float gain,
                                    const float* scale, const float* bias, const float* noise, int noise_mode,
                                    float noise_strength, int act, float alpha, float act_gain, float clamp,
                                    const float* residual, void* stream) {
    SHG_CHECK_ARG(mid && f && y, "upfir_planar: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1, "upfir_planar: bad shape");
    SHG_CHECK_ARG(4L * N * C * (H + 1) * (W + 1) <= 2147483647L && 4L * N * C * H * W <= 2147483647L, "upfir_planar: tensor too large");
    SHG_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2, "upfir_planar: bad noise_mode");
    UfdParams p{};
    p.x = mid; p.f = f; p.y = y; p.NC = N * C; p.C = C; p.H = H; p.W = W; p.OH = 2 * H; p.OW = 2 * W;
    p.fh = 4; p.fw = 4; p.upx = p.upy = p.dnx = p.dny = 1; p.px0 = p.py0 = 1; p.flip = flip ? 1 : 0; p.gain = gain;
    p.scale = scale; p.bias = bias; p.noise = noise; p.residual = residual;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.act_gain = act_gain; p.clamp = clamp; p.has_epilogue = 1;
    const int gz = p.NC < 32768 ? p.NC : 32768;
    // 16-byte path needs 16-byte aligned rows of y / residual / noise: OW = 2W multiple of 4 and aligned bases
    const bool vec = (W % 2 == 0) && (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)noise) % 16 == 0);
    dim3 grid(shg_cdiv(W, 128), shg_cdiv(H, 8), gz);
    if (vec) hipLaunchKernelGGL(fir_up_planar_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(fir_up_planar_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
