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
#include "fir_march.h"
#include <stdlib.h>

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
    // fir_same only: polyphase-planar output for the stride-2 convolution (conv_wino_poly.hip): when pl_pp > 0 the result
    // element (oy, ox) goes to plane (oy&1)*2 + (ox&1), row oy>>1, column ox>>1 of y [4][NC][pl_ph2][pl_pp]
    int pl_ph2, pl_pp;
    // timing studies, -DSHG_ABLATE build only (env SHG_FIR_DBG: 1 skip window loads, 2 skip FIR math, 4 skip stores)
#ifdef SHG_ABLATE
    int dbg;
#else
    static constexpr int dbg = 0;
#endif
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
__device__ __attribute__((aligned(16))) float shg_ufd_zeros[64];   // stand-in for an absent noise operand (unconditional loads)

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

// VEC4: the input rows are 16-byte aligned (W % 4 == 0, aligned base) and 0 <= px0 <= 4: the window is fetched as aligned
// float4 (columns ox0-4 .. ox0+67, each float4 entirely inside the image or entirely padding) -- 630 loads per tile
// instead of 2345 dword loads.
template <int FH, int FW, bool VEC4>
__global__ __launch_bounds__(256, 4) void fir_same_kernel(const UfdParams p) {
    constexpr int TH = 32, TW = 64, IH = TH + FH - 1, IW = TW + FW - 1;
    constexpr int PITCH = VEC4 ? 72 : IW + 1;  // VEC4: 18 float4 per row, LDS column = ix - (ox0 - 4)
    constexpr int RPT = (IH + 3) / 4;          // window rows staged per thread (4 row groups x 64 columns)
    constexpr int NX = IH * (FW - 1);          // elements of the FW-1 extra window columns
    constexpr int NV4 = IH * 18, V4PT = (NV4 + 255) / 256;
    static_assert(NX <= 256, "extra window columns must fit one pass");
    static_assert(!VEC4 || (FW == 4 && FH == 4), "the float4 window assumes a 4x4 filter");
    __shared__ __attribute__((aligned(16))) float tile[IH * PITCH];
    // taps: uniform addresses -> scalar loads; fr[ky][kx] multiplies x[oy + ky - py0][ox + kx - px0]
    float fr[FH * FW];
#pragma unroll
    for (int k = 0; k < FH * FW; ++k) {
        const int ky = k / FW, kx = k % FW;
        const int sy = p.flip ? ky : FH - 1 - ky, sx = p.flip ? kx : FW - 1 - kx;
        fr[k] = p.f[sy * FW + sx] * p.gain;
    }
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int tx = tid & 63, rg = tid >> 6;          // compute: column tx, rows rg*8 .. rg*8+7; staging: column tx, rows rg, rg+4, ...
    const int xr = tid / (FW - 1), xc = TW + tid % (FW - 1);
    const bool xact = tid < NX;
    const int ix0 = ox0 - p.px0 + tx, iy0 = oy0 - p.py0;
    const int ixe = ox0 - p.px0 + xc, iye = iy0 + xr;
    const bool xok = xact && iye >= 0 && iye < p.H && ixe >= 0 && ixe < p.W;
    const long plane = (long)p.H * p.W;
    const int coff = VEC4 ? 4 - p.px0 : 0;           // LDS column of window column 0
    float tvr[VEC4 ? 1 : RPT + 1];
    float4 tv4[VEC4 ? V4PT : 1];
    // unconditional loads from clamped 32-bit byte offsets off the uniform plane base, masked afterwards
    auto load_tile = [&](int nc) __attribute__((always_inline)) {
        const char* pb = reinterpret_cast<const char*>(p.x + nc * plane);
        if constexpr (VEC4) {
            int t_i = tid;
            asm volatile("" : "+v"(t_i));                  // keep the per-load offsets / masks out of the loop-invariant set
#pragma unroll
            for (int k = 0; k < V4PT; ++k) {
                const int e = t_i + 256 * k;
                const int r = e / 18, q = e - r * 18;
                const int iy = iy0 + r, ix = ox0 - 4 + 4 * q;
                const bool ok = e < NV4 && iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W;
                const unsigned boff = ok ? (unsigned)(iy * p.W + ix) * 4u : 0u;
                const float4 val = *reinterpret_cast<const float4*>(pb + boff);
                tv4[k] = ok ? val : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            int ix_i = ix0, rg_i = rg;
            asm volatile("" : "+v"(ix_i), "+v"(rg_i));
            const bool okx = ix_i >= 0 && ix_i < p.W;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = rg_i + 4 * k;
                const int iy = iy0 + r;
                const bool ok = okx && r < IH && iy >= 0 && iy < p.H;
                const unsigned boff = ok ? (unsigned)(iy * p.W + ix_i) * 4u : 0u;
                const float val = *reinterpret_cast<const float*>(pb + boff);
                tvr[k] = ok ? val : 0.f;
            }
            const unsigned eoff = xok ? (unsigned)(iye * p.W + ixe) * 4u : 0u;
            const float val = *reinterpret_cast<const float*>(pb + eoff);
            tvr[RPT] = xok ? val : 0.f;
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
        if constexpr (VEC4) {
#pragma unroll
            for (int k = 0; k < V4PT; ++k) {
                const int e = tid + 256 * k;
                if (e < NV4) reinterpret_cast<float4*>(tile)[e] = tv4[k];     // row r = e/18 at r*72 floats: contiguous in e
            }
        } else {
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (rg + 4 * k < IH) tile[(rg + 4 * k) * PITCH + tx] = tvr[k];
            if (xact) tile[xr * PITCH + xc] = tvr[RPT];
        }
    };
    int nc = blockIdx.z;
    if (nc < p.NC) load_tile(nc);
    for (; nc < p.NC; nc += gridDim.z) {
        const UfdPlane pl = ufd_plane(p, nc);
        store_tile();
        __syncthreads();
        if (nc + (int)gridDim.z < p.NC) load_tile(nc + gridDim.z);      // next plane's window, in flight during the FIR
        const int ox = ox0 + tx;
        float* yp = p.y + (long)nc * p.OH * p.OW;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __builtin_amdgcn_sched_barrier(0);
            const int ty = rg * 8 + half * 4;
            float win[4 + FH - 1][FW];
#pragma unroll
            for (int r = 0; r < 4 + FH - 1; ++r)
#pragma unroll
                for (int k = 0; k < FW; ++k) win[r][k] = tile[(ty + r) * PITCH + tx + k + coff];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int oy = oy0 + ty + t;
                float v = 0.f;
#pragma unroll
                for (int ky = 0; ky < FH; ++ky)
#pragma unroll
                    for (int kx = 0; kx < FW; ++kx) v += win[t + ky][kx] * fr[ky * FW + kx];
                if (p.pl_pp > 0) {
                    // every entry of the four planes is written: positions beyond the filtered image (pitch padding, the
                    // last row / column of the odd planes) get ZERO -- the Winograd transforms of the consumer mix a
                    // patch's columns, so stale memory there would leak into valid outputs
                    if ((oy >> 1) < p.pl_ph2 && (ox >> 1) < p.pl_pp)
                        p.y[((long)((oy & 1) * 2 + (ox & 1)) * p.NC + nc) * p.pl_ph2 * p.pl_pp + (oy >> 1) * p.pl_pp + (ox >> 1)] =
                            (oy < p.OH && ox < p.OW) ? v : 0.f;
                } else if (oy < p.OH && ox < p.OW) yp[oy * p.OW + ox] = ufd_finish(p, pl, v, oy * p.OW + ox);
            }
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
template <bool VEC, int TV>
__global__ __launch_bounds__(256, 4) void fir_up_planar_kernel(const UfdParams p) {
    // tile = TU x TV low-resolution pixels (TV = 128 / 64 / 32 for wide / medium / narrow images), 2 x 2 of them per lane
    constexpr int LR = TV / 2, TU = 2 * (256 / LR), PU = TU + 2, PV = TV + 2, PITCH = PV + 1;
    constexpr int G = 256 / TV;                // staging row groups
    constexpr int NLD = 4 * PU / G;            // window rows staged per thread
    constexpr int NXE = 8 * PU, NXS = (NXE + 255) / 256;   // elements of the two extra window columns, slots per thread
    static_assert(4 * PU % G == 0, "window rows must split evenly over the staging groups");
    __shared__ float tile[4][PU * PITCH];
    // filter taps: uniform addresses -> scalar loads, the 16 taps live in SGPRs
    float fr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ky = k >> 2, kx = k & 3;
        const int sy = p.flip ? ky : 3 - ky, sx = p.flip ? kx : 3 - kx;
        fr[k] = p.f[sy * 4 + sx] * p.gain;
    }
    const int tid = threadIdx.x;
    const int v0 = blockIdx.x * TV, u0 = blockIdx.y * TU;
    const int tv = (tid % LR) * 2, tu0 = (tid / LR) * 2;   // compute role: low-res cols tv, tv+1; rows tu0, tu0+1
    const int PWg = p.W + 1;
    const long P = (long)(p.H + 1) * PWg;
    // staging role: thread (g, col) fetches window column `col` of window rows g*NLD .. g*NLD+NLD-1, the rows of the
    // four planes being numbered pl*PU + r.  Plane (a,b) holds full rows Y = 2u+a, cols X = 2v+b, and its window
    // starts at (u0-a, v0-b).  The plane of every staged row is wave-uniform.  The two extra window columns are
    // fetched one element per slot.
    const int g = tid / TV, col = tid % TV;
    const int gu = __builtin_amdgcn_readfirstlane(TV >= 64 ? g : (g >> 1));     // TV = 32: lanes of a wave span groups 2w, 2w+1
    auto row_plane = [&](int k) __attribute__((always_inline)) {               // plane of staged row k (uniform)
        return TV == 128 ? 2 * gu + k / PU : gu;
    };
    auto row_r = [&](int k) __attribute__((always_inline)) {                   // its row inside the plane window
        return TV == 128 ? k % PU : (TV == 64 ? k : (g & 1) * NLD + k);
    };
    float tvr[NLD + NXS];
    // unconditional loads from clamped 32-bit byte offsets off a uniform plane base (scalar base + vector offset
    // addressing), masked afterwards: no divergent branches, one address register per load
    auto load_tile = [&](int nc) __attribute__((always_inline)) {
        // (opaque copies: keeps the per-load offsets and masks from being hoisted out of the plane loop, where they
        // would occupy ~60 registers for the whole kernel)
        int col_i = col, g_i = g, t_i = tid;
        asm volatile("" : "+v"(col_i), "+v"(g_i), "+v"(t_i));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int pl = row_plane(k);
            const int r = TV == 32 ? (g_i & 1) * NLD + k : row_r(k);
            const int pa = pl >> 1, pb_ = pl & 1;
            const char* pb = reinterpret_cast<const char*>(p.x + ((long)pl * p.NC + nc) * P);   // uniform
            const int u = u0 - pa + r, v = v0 - pb_ + col_i;
            // valid full-resolution rows: Y = 2u+a in [0, 2H]  <=>  u in [0, H-a]
            const bool ok = v >= 0 && v <= p.W - pb_ && u >= 0 && u <= p.H - pa;
            const unsigned boff = ok ? (unsigned)(u * PWg + v) * 4u : 0u;
            const float val = (p.dbg & 1) ? 1.f : *reinterpret_cast<const float*>(pb + boff);
            tvr[k] = ok ? val : 0.f;
        }
#pragma unroll
        for (int x = 0; x < NXS; ++x) {
            const int e = t_i + 256 * x;                 // extra element: window row e/2 (of 4*PU), column TV + (e&1)
            const int R = e >> 1, pl = R / PU, r = R - pl * PU, c = TV + (e & 1);
            const int u = u0 - (pl >> 1) + r, v = v0 - (pl & 1) + c;
            const bool ok = e < NXE && u >= 0 && u <= p.H - (pl >> 1) && v <= p.W - (pl & 1);
            const unsigned long eoff = ok ? ((unsigned long)((long)pl * p.NC + nc) * P + (unsigned)(u * PWg + v)) * 4ul : 0ul;
            const float val = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.x) + eoff);
            tvr[NLD + x] = ok ? val : 0.f;
        }
    };
    int nc = blockIdx.z;
    if (nc < p.NC) load_tile(nc);
    for (; nc < p.NC; nc += gridDim.z) {
        const UfdPlane plq = ufd_plane(p, nc);
#pragma unroll
        for (int k = 0; k < NLD; ++k) tile[row_plane(k)][row_r(k) * PITCH + col] = tvr[k];
#pragma unroll
        for (int x = 0; x < NXS; ++x) {
            const int e = tid + 256 * x;
            const int R = e >> 1, pl = R / PU, r = R - pl * PU;
            if (e < NXE) tile[pl][r * PITCH + TV + (e & 1)] = tvr[NLD + x];
        }
        __syncthreads();
        const int u = u0 + tu0, v = v0 + tv;
        // vmcnt retires in order and counts stores: a noise row requested behind the window prefetch (or behind a store) can only
        // be waited for by draining both.  The noise rows are therefore requested unconditionally (an absent operand reads
        // zeros), two rows ahead, and the next plane's window only after the last of them (after output row 1).
        const bool vecp = VEC && v + 1 < p.W;
        const float* nzb = (p.has_epilogue && plq.nz) ? plq.nz : shg_ufd_zeros;
        const bool nzp = vecp && p.has_epilogue && plq.nz;
        auto nz_row = [&](int dy) __attribute__((always_inline)) {
            const int yy = 2 * u + dy < p.OH ? 2 * u + dy : p.OH - 1;
            return *reinterpret_cast<const float4*>(nzb + (nzp ? yy * p.OW + 2 * v : 0));
        };
        float4 nzq[2];
        if (VEC) { nzq[0] = nz_row(0); nzq[1] = nz_row(1); }
        else if (nc + (int)gridDim.z < p.NC) load_tile(nc + gridDim.z);
        // neighbourhood of the lane's 2 x 2 low-res pixels: rows Y = 2u-1 .. 2u+5, cols X = 2v-1 .. 2v+5; element (r,c)
        // lives in plane (ra,cb) = ((r+1)&1, (c+1)&1)
        float m[7][7];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int ra = (r + 1) & 1;
            const int ur = tu0 + ((r + 1) >> 1) - 1 + ra;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const int cb = (c + 1) & 1;
                const int vc = tv + ((c + 1) >> 1) - 1 + cb;
                m[r][c] = tile[ra * 2 + cb][ur * PITCH + vc];
            }
        }
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            __builtin_amdgcn_sched_barrier(0);     // one output row at a time: keeps the live set small
            if (v < p.W && 2 * u + dy < p.OH) {
                float o4[4];
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    float acc = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) acc += m[dy + ky][dx + kx] * fr[ky * 4 + kx];
                    o4[dx] = (p.dbg & 2) ? m[dy][dx] : acc;
                }
                const int pix = (2 * u + dy) * p.OW + 2 * v;
                float* yp = p.y + (long)nc * p.OH * p.OW + pix;
                if (vecp) {
                    // 16-byte path: scale / noise / bias / activation / skip on four outputs at once
                    const float4 nz = nzq[dy & 1];
                    const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
                    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.has_epilogue && plq.res) rs = *reinterpret_cast<const float4*>(plq.res + pix);
                    const float rsv[4] = {rs.x, rs.y, rs.z, rs.w};
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
                    if (dy < 2) nzq[dy & 1] = nz_row(dy + 2);                    // ahead of this row's store
                    if (!(p.dbg & 4) || o[0] == 12345.f) *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx)
                        if (2 * v + dx < p.OW) yp[dx] = ufd_finish(p, plq, o4[dx], pix + dx);
                }
            }
            // next plane's window: every thread stages its part -- outside the per-lane conditions, behind the last noise request
            if (VEC && dy == 1 && nc + (int)gridDim.z < p.NC) load_tile(nc + gridDim.z);
        }
        __syncthreads();
    }
}

static int ufd_launch(UfdParams& p, hipStream_t s) {
    const int gz = p.NC < 32768 ? p.NC : 32768;
    if (p.upx == 1 && p.upy == 1 && p.dnx == 1 && p.dny == 1 && p.fh == 4 && p.fw == 4) {
        // each workgroup walks several planes with the next window prefetched; ~8k workgroups in total
        // (planar output: the grid covers the whole 2*pl_ph2 x 2*pl_pp extent of the planes, padding included)
        const int cov_w = p.pl_pp > 0 ? 2 * p.pl_pp : p.OW, cov_h = p.pl_pp > 0 ? 2 * p.pl_ph2 : p.OH;
        const int tiles = shg_cdiv(cov_w, 64) * shg_cdiv(cov_h, 32);
        int gzs = 8192 / tiles; if (gzs < 1) gzs = 1; if (gzs > p.NC) gzs = p.NC;
        dim3 grid(shg_cdiv(cov_w, 64), shg_cdiv(cov_h, 32), gzs);
        const bool vec4 = p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && p.px0 >= 0 && p.px0 <= 4;
        if (vec4) hipLaunchKernelGGL((fir_same_kernel<4, 4, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fir_same_kernel<4, 4, false>), grid, dim3(256), 0, s, p);
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

// ---------------------------------------------------------------------------------------------
// The plugin's full operand range (upfirdn2d.cpp:38-59: any strides, AT_DISPATCH_FLOATING_TYPES_AND_HALF): x and y described by their
// element strides, dtype float32 / float16 / float64; accumulation in float (float64: in double), `v *= gain` last, as
// upfirdn2d.cu:38,104-121.  The streaming kernels above serve the two dense layouts the networks use (float32 NCHW, float16 NHWC); this
// one serves everything else the reference's op accepts -- float64 tensors, float32 channels_last, float16 NCHW, arbitrary views --
// without a conversion pass.  One lane per output element, lanes ordered along the unit-stride axis of y.
// ---------------------------------------------------------------------------------------------
struct UfdStridedP {
    const void* x; const float* f; void* y;
    int N, C, H, W, OH, OW;
    long sx[4], sy[4];           // element strides of x / y: n, c, h, w
    int fh, fw; long fsy, fsx;   // filter extent and element strides
    int upx, upy, dnx, dny, px0, py0, flip;
    float gain;
    int c_minor;                 // y has unit channel stride: lanes run over c first
    long total;
};

template <typename T, typename A>
__global__ __launch_bounds__(256) void upfirdn_strided_kernel(const UfdStridedP p) {
    const T* x = (const T*)p.x;
    T* y = (T*)p.y;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < p.total; e += (long)gridDim.x * 256) {
        int n, c, oy, ox;
        long r = e;
        if (p.c_minor) { c = (int)(r % p.C); r /= p.C; ox = (int)(r % p.OW); r /= p.OW; oy = (int)(r % p.OH); n = (int)(r / p.OH); }
        else { ox = (int)(r % p.OW); r /= p.OW; oy = (int)(r % p.OH); r /= p.OH; c = (int)(r % p.C); n = (int)(r / p.C); }
        const T* xp = x + n * p.sx[0] + c * p.sx[1];
        A v = (A)0;
        for (int ky = 0; ky < p.fh; ++ky) {
            const int uy = oy * p.dny + ky - p.py0;
            if (uy < 0 || (uy % p.upy) != 0) continue;
            const int iy = uy / p.upy;
            if (iy >= p.H) continue;
            const int sy_ = p.flip ? ky : p.fh - 1 - ky;
            for (int kx = 0; kx < p.fw; ++kx) {
                const int ux = ox * p.dnx + kx - p.px0;
                if (ux < 0 || (ux % p.upx) != 0) continue;
                const int ix = ux / p.upx;
                if (ix >= p.W) continue;
                const int sx_ = p.flip ? kx : p.fw - 1 - kx;
                v += (A)xp[iy * p.sx[2] + ix * p.sx[3]] * (A)p.f[sy_ * p.fsy + sx_ * p.fsx];
            }
        }
        v *= (A)p.gain;
        y[n * p.sy[0] + c * p.sy[1] + oy * p.sy[2] + ox * p.sy[3]] = (T)v;
    }
}

// dtype: 0 float32, 1 float16, 2 float64.  x_strides / y_strides: four element strides (n, c, h, w) each; f [fh, fw] float32 with element
// strides f_stride_y / f_stride_x.  y must not alias x.  Same size rule and argument meaning as shg_upfirdn2d_f32.
extern "C" int shg_upfirdn2d_strided(const void* x, const float* f, void* y, int dtype, int N, int C, int H, int W, const long* x_strides,
                                     const long* y_strides, int fh, int fw, long f_stride_y, long f_stride_x, int upx, int upy, int downx,
                                     int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream) {
    SHG_CHECK_ARG(x && f && y && x_strides && y_strides, "upfirdn2d_strided: null pointer");
    SHG_CHECK_ARG(dtype >= 0 && dtype <= 2, "upfirdn2d_strided: dtype must be 0 (float32), 1 (float16) or 2 (float64)");
    SHG_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1, "upfirdn2d: x must be rank 4 and non-empty");
    SHG_CHECK_ARG(fh >= 1 && fw >= 1, "upfirdn2d: f must be at least 1x1");                       // upfirdn2d.cpp:26
    SHG_CHECK_ARG(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");       // :27
    SHG_CHECK_ARG(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1"); // :28
    SHG_CHECK_ARG((long)N * C * H * W <= 2147483647L, "upfirdn2d: x is too large");               // :22
    const int OW = (W * upx + padx0 + padx1 - fw + downx) / downx;                                // :32
    const int OH = (H * upy + pady0 + pady1 - fh + downy) / downy;                                // :33
    SHG_CHECK_ARG(OW >= 1 && OH >= 1, "upfirdn2d: output must be at least 1x1");                  // :34
    SHG_CHECK_ARG((long)N * C * OH * OW <= 2147483647L, "upfirdn2d: output is too large");        // :36
    for (int k = 0; k < 4; ++k) SHG_CHECK_ARG(x_strides[k] >= 0 && y_strides[k] >= 0, "upfirdn2d_strided: negative stride");
    UfdStridedP p{};
    p.x = x; p.f = f; p.y = y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
    for (int k = 0; k < 4; ++k) { p.sx[k] = x_strides[k]; p.sy[k] = y_strides[k]; }
    p.fh = fh; p.fw = fw; p.fsy = f_stride_y; p.fsx = f_stride_x;
    p.upx = upx; p.upy = upy; p.dnx = downx; p.dny = downy; p.px0 = padx0; p.py0 = pady0; p.flip = flip ? 1 : 0; p.gain = gain;
    p.c_minor = (C > 1 && y_strides[1] == 1) ? 1 : 0;
    p.total = (long)N * C * OH * OW;
    long blocks = (p.total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL((upfirdn_strided_kernel<float, float>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (dtype == 1) hipLaunchKernelGGL((upfirdn_strided_kernel<_Float16, float>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((upfirdn_strided_kernel<double, double>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
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

// FIR pre-filter of the stride-2 convolution (conv2d_resample.py:116-120: upfirdn2d with padding 2 for the 4x4 filter) written
// as the four polyphase planes of the (H+1) x (W+1) result: y [4][N*C][H/2+1][PP], plane (a,b) element (u,v) = xf[2u+a][2v+b]
// (PP >= W/2+1, a multiple of 4 floats; entries outside a plane's extent are written as zeros).
extern "C" int shg_fir_down_planar_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int PP, int flip,
                                       float gain, void* stream) {
    UfdParams p;
    int rc = ufd_fill(p, x, f, y, N, C, H, W, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, flip, gain);
    if (rc != SHG_OK) return rc;
    SHG_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "fir_down_planar: H and W must be even");
    SHG_CHECK_ARG(PP % 4 == 0 && PP >= W / 2 + 1, "fir_down_planar: plane pitch must be a multiple of 4 and >= W/2 + 1");
    SHG_CHECK_ARG(4L * N * C * (H / 2 + 1) * PP <= 2147483647L, "fir_down_planar: output is too large");
    p.pl_ph2 = H / 2 + 1; p.pl_pp = PP;
    return ufd_launch(p, (hipStream_t)stream);
}

// Row-marching form of the same pad-2 4x4 pre-filter (fir_march.h) for a SEPARABLE filter f[ky][kx] = fy[ky] * fx[kx] -- every
// filter upfirdn2d.setup_filter builds from a 1-D kernel (upfirdn2d.py:61-95).  taps_host = {fx[0..3], fy[0..3]} in HOST memory
// (they travel as kernel arguments).  PP == 0: y [N,C,H+1,W+1]; PP > 0: the polyphase planes of shg_fir_down_planar_f32 with
// pitch PP (a multiple of 32 floats makes every store a whole 128-byte line for W >= 256).
extern "C" int shg_fir_pad2_sep_supported(int H, int W, int PP) {
    if (H < 2 || W < 8 || W > 512 || !(W % 64 == 0 ? (W == 64 || W == 128 || W == 256 || W == 512) : 64 % W == 0)) return 0;
    if (PP > 0 && (H % 2 || W % 2 || PP % 4 || PP < W / 2 + 1)) return 0;
    return 1;
}

extern "C" int shg_fir_pad2_sep_f32(const float* x, const float* taps_host, float* y, int N, int C, int H, int W, int PP, int flip,
                                    float gain, void* stream) {
    SHG_CHECK_ARG(x && taps_host && y, "fir_pad2_sep: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1, "fir_pad2_sep: bad shape");
    SHG_CHECK_ARG(shg_fir_pad2_sep_supported(H, W, PP), "fir_pad2_sep: unsupported geometry (shg_fir_pad2_sep_supported)");
    SHG_CHECK_ARG((long)N * C * (H + 1) * (W + 1) <= 2147483647L && (PP == 0 || 4L * N * C * (H / 2 + 1) * PP <= 2147483647L),
                  "fir_pad2_sep: tensor too large");
    FirMarchParams p{};
    p.x = x; p.y = y; p.NC = N * C; p.H = H; p.W = W;
    p.mode = PP > 0; p.pitch = PP > 0 ? PP : W + 1; p.ph2 = H / 2 + 1;
    for (int k = 0; k < 4; ++k) {
        p.a[k] = taps_host[flip ? k : 3 - k];
        p.b[k] = taps_host[4 + (flip ? k : 3 - k)] * gain;
    }
    const int OH = H + 1;
    const bool wide = p.mode && W % 256 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
                      PP % 2 == 0;
    p.LPG = wide ? 64 : (W < 64 ? W : 64);
    p.G = wide ? 1 : 64 / p.LPG;
    const int npg = shg_cdiv(p.NC, p.G);
    int nseg = shg_cdiv(8192, npg);                              // ~8k waves; at least 16 rows each (3 halo rows are re-read per segment)
    if (nseg > OH / 16) nseg = OH / 16;
    if (nseg < 1) nseg = 1;
    p.R = shg_cdiv(OH, nseg); p.nseg = shg_cdiv(OH, p.R); p.nitem = npg * p.nseg;
    const dim3 grid(shg_cdiv(p.nitem, 4));
    hipStream_t s = (hipStream_t)stream;
    if (wide) {
        if (W == 256) hipLaunchKernelGGL((fir_down_march4_kernel<1, 8>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fir_down_march4_kernel<2, 8>), grid, dim3(256), 0, s, p);
    } else {
        const int K = shg_cdiv(W, 64);
        if (K == 1) hipLaunchKernelGGL((fir_down_march_kernel<1, 8>), grid, dim3(256), 0, s, p);
        else if (K == 2) hipLaunchKernelGGL((fir_down_march_kernel<2, 8>), grid, dim3(256), 0, s, p);
        else if (K == 4) hipLaunchKernelGGL((fir_down_march_kernel<4, 8>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fir_down_march_kernel<8, 4>), grid, dim3(256), 0, s, p);
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
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
#ifdef SHG_ABLATE
    { const char* d = getenv("SHG_FIR_DBG"); p.dbg = d ? atoi(d) : 0; }
#endif
    // each workgroup walks several (n,c) planes with the next window prefetched; ~8k workgroups keep 256 CUs busy
    const int TV = W >= 96 ? 128 : (W >= 48 ? 64 : 32), TU = 1024 / TV;
    const int tiles = shg_cdiv(W, TV) * shg_cdiv(H, TU);
    int gz = 8192 / tiles; if (gz < 1) gz = 1; if (gz > p.NC) gz = p.NC;
    // 16-byte path needs 16-byte aligned rows of y / residual / noise: OW = 2W multiple of 4 and aligned bases
    const bool vec = (W % 2 == 0) && (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)noise) % 16 == 0);
    dim3 grid(shg_cdiv(W, TV), shg_cdiv(H, TU), gz);
    void (*kern)(const UfdParams) =
        TV == 128 ? (vec ? fir_up_planar_kernel<true, 128> : fir_up_planar_kernel<false, 128>)
      : TV == 64  ? (vec ? fir_up_planar_kernel<true, 64> : fir_up_planar_kernel<false, 64>)
                  : (vec ? fir_up_planar_kernel<true, 32> : fir_up_planar_kernel<false, 32>);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// Row-marching form of shg_upfir_planar_f32 (fir_march.h) for a SEPARABLE filter; taps_host = {fx[0..3], fy[0..3]} in HOST memory.
extern "C" int shg_upfir_planar_sep_supported(int H, int W) {
    if (H < 1 || W < 4 || W > 256 || W % 2) return 0;
    return (W >= 128 ? W % 128 == 0 : 64 % (W / 2) == 0) ? 1 : 0;
}

extern "C" int shg_upfir_planar_sep_f32(const float* mid, const float* taps_host, float* y, int N, int C, int H, int W, int flip,
                                        float gain, const float* scale, const float* bias, const float* noise, int noise_mode,
                                        float noise_strength, int act, float alpha, float act_gain, float clamp,
                                        const float* residual, void* stream) {
    SHG_CHECK_ARG(mid && taps_host && y, "upfir_planar_sep: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1, "upfir_planar_sep: bad shape");
    SHG_CHECK_ARG(shg_upfir_planar_sep_supported(H, W), "upfir_planar_sep: unsupported geometry (shg_upfir_planar_sep_supported)");
    SHG_CHECK_ARG(4L * N * C * (H + 1) * (W + 1) <= 2147483647L && 4L * N * C * H * W <= 2147483647L, "upfir_planar_sep: tensor too large");
    SHG_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2, "upfir_planar_sep: bad noise_mode");
    SHG_CHECK_ARG((((uintptr_t)y | (uintptr_t)residual | (uintptr_t)noise) & 15) == 0, "upfir_planar_sep: y / residual / noise must be 16-byte aligned");
    FirUpParams p{};
    p.mid = mid; p.y = y; p.scale = scale; p.bias = bias; p.noise = noise; p.residual = residual;
    p.NC = N * C; p.C = C; p.H = H; p.W = W;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.act_gain = act_gain; p.clamp = clamp;
    for (int k = 0; k < 4; ++k) {
        p.a[k] = taps_host[flip ? k : 3 - k];
        p.b[k] = taps_host[4 + (flip ? k : 3 - k)] * gain;
    }
    p.LPG = W / 2 < 64 ? W / 2 : 64; p.G = 64 / p.LPG;
    const int npg = shg_cdiv(p.NC, p.G);
    int nseg = shg_cdiv(8192, npg);                              // ~8k waves, at least 8 low-resolution rows each
    if (nseg > H / 8) nseg = H / 8;
    if (nseg < 1) nseg = 1;
    p.R = shg_cdiv(H, nseg); p.nseg = shg_cdiv(H, p.R); p.nitem = npg * p.nseg;
    const dim3 grid(shg_cdiv(p.nitem, 4));
    hipStream_t s = (hipStream_t)stream;
    if (W <= 128) hipLaunchKernelGGL((fir_up_march_kernel<1, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((fir_up_march_kernel<2, 1>), grid, dim3(256), 0, s, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// The two x2 resampling FIRs of the training rows as row-marching kernels (fir_march.h), SEPARABLE 4x4 filter, taps_host =
// {fx[0..3], fy[0..3]} in HOST memory:
//   up == 1:  y [N,C,H/2,W/2] = upfirdn2d(x, f, down=2, padding=1)          (conv2d_resample.py:104-108; backward of the up FIR)
//   up == 2:  y [N,C,2H,2W]   = upfirdn2d(x, f, up=2, padding=[2,1,2,1])     (upsample2d, upfirdn2d.py:288-305; backward of the down FIR)
extern "C" int shg_fir_resample2_sep_supported(int H, int W, int up) {
    if (up == 1) return (H >= 2 && H % 2 == 0 && W >= 8 && W % 4 == 0 && (W <= 256 ? 64 % (W / 4) == 0 : W == 512)) ? 1 : 0;
    if (up == 2) return (H >= 1 && W >= 4 && W % 2 == 0 && (W <= 128 ? 64 % (W / 2) == 0 : W == 256)) ? 1 : 0;
    return 0;
}

extern "C" int shg_fir_resample2_sep_f32(const float* x, const float* taps_host, float* y, int N, int C, int H, int W, int up, int flip,
                                         float gain, void* stream) {
    SHG_CHECK_ARG(x && taps_host && y, "fir_resample2_sep: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1, "fir_resample2_sep: bad shape");
    SHG_CHECK_ARG(shg_fir_resample2_sep_supported(H, W, up), "fir_resample2_sep: unsupported geometry (shg_fir_resample2_sep_supported)");
    SHG_CHECK_ARG((long)N * C * H * W * (up == 2 ? 4 : 1) <= 2147483647L, "fir_resample2_sep: tensor too large");
    SHG_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "fir_resample2_sep: x / y must be 16-byte aligned");
    FirRsParams p{};
    p.x = x; p.y = y; p.NC = N * C; p.H = H; p.W = W;
    for (int k = 0; k < 4; ++k) {
        p.a[k] = taps_host[flip ? k : 3 - k];
        p.b[k] = taps_host[4 + (flip ? k : 3 - k)] * gain;
    }
    const int lanes = up == 1 ? W / 4 : W / 2;                   // lanes per row
    p.LPG = lanes < 64 ? lanes : 64; p.G = 64 / p.LPG;
    const int K = shg_cdiv(lanes, 64);
    const int npg = shg_cdiv(p.NC, p.G);
    const int rows = up == 1 ? H / 2 : H;                        // marched rows (dn2: output rows, up2: input rows)
    int nseg = shg_cdiv(8192, npg);
    if (nseg > rows / 8) nseg = rows / 8;
    if (nseg < 1) nseg = 1;
    p.R = shg_cdiv(rows, nseg); p.nseg = shg_cdiv(rows, p.R); p.nitem = npg * p.nseg;
    const dim3 grid(shg_cdiv(p.nitem, 4));
    hipStream_t s = (hipStream_t)stream;
    if (up == 1) {
        if (K == 1) hipLaunchKernelGGL((fir_dn2_march_kernel<1>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fir_dn2_march_kernel<2>), grid, dim3(256), 0, s, p);
    } else {
        if (K == 1) hipLaunchKernelGGL((fir_up2_march_kernel<1>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((fir_up2_march_kernel<2>), grid, dim3(256), 0, s, p);
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
