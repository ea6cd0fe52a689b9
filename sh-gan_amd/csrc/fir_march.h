// Row-marching 4x4 FIR kernels for gfx950 (included by upfirdn2d.hip and tools/micro/fir_march.hip).
//
// The FIR passes around the resampling convolutions (conv2d_resample.py:116-120 pad-2 pre-filter of the stride-2 conv,
// :133-138 pad-1 post-filter of the transposed conv; upfirdn2d.py:98-138 semantics) are pure HBM streams: 8 bytes per output.
// A tiled kernel (stage a window in LDS, barrier, filter, store) reaches ~3.3 TB/s on this chip because every workgroup touches
// 32-64 short row pieces a full row pitch apart and nothing is in flight while it waits at its barrier.  These kernels stream
// instead, the way an elementwise kernel does:
//   * one WAVE owns whole image rows (lane l holds columns l, l+64, ...; images narrower than 64 put several planes side by side
//     in a wave) and marches down a segment of rows, so it reads and writes contiguous memory.  Accesses are DWORD per lane on
//     purpose: the result rows have odd lengths (W+1, or plane pitches that are not multiples of 128 B), and on this chip a
//     wave-wide 8/16-byte store that does not start on a 128-byte boundary runs at 2.9-3.5 TB/s where a dword store keeps
//     6.0 TB/s (tools/micro/fir_march.hip, `wpat`);
//   * the filter is separable (f = fy (x) fx, checked by the host): the horizontal pass takes its three neighbour values from the
//     adjacent lanes with DPP wavefront shifts (no LDS, no barrier), the vertical pass keeps the last four horizontally filtered
//     rows in registers;
//   * rows are taken in batches of B: all B row loads are issued back to back (unconditional, clamped addresses; B rows per wave
//     in flight), then the batch is filtered and stored.  Nothing is carried across the batch loop except
//     computed values, so a conservative s_waitcnt costs nothing; the other waves of the SIMD fill the gaps.
#pragma once
#include <hip/hip_runtime.h>

struct FirMarchParams {
    const float* x;          // [NC, H, W]
    float* y;
    int NC, H, W;
    int mode;                // 0: y [NC, H+1, pitch] rows;  1: polyphase planes y [4][NC][ph2][pitch]
    int pitch, ph2;
    int R, nseg;             // output rows per wave, row segments per plane
    int LPG, G;              // lanes per plane row (min(W, 64)), planes per wave
    int nitem;               // plane groups * nseg
    float a[4], b[4];        // horizontal / vertical taps: out[oy][ox] = sum b[ky] a[kx] x[oy+ky-2][ox+kx-2]
};

template <int V> struct FmIC { static constexpr int value = V; };

__device__ __forceinline__ float fm_shr1(float v, float fill) {   // lane i <- lane i-1, lane 0 <- fill
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float fm_shl1(float v, float fill) {   // lane i <- lane i+1, lane 63 <- fill
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float fm_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// pad-2 4x4 FIR, (H+1) x (W+1) result.  K = ceil(W / 64) columns per lane; W < 64: W must divide 64.
template <int K, int B, int DBG = 0>     // DBG (tools/micro only): 1 no loads, 2 no stores
__global__ __launch_bounds__(256) void fir_down_march_kernel(const FirMarchParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.nitem) return;
    const int seg = item % p.nseg, pg = item / p.nseg;
    const int g = lane / p.LPG, li = lane - g * p.LPG;
    const int OH = p.H + 1;
    const int r0 = seg * p.R, r1 = min(r0 + p.R, OH);
    const int plane_raw = pg * p.G + g;
    const bool lane_on = plane_raw < p.NC;
    const int plane = lane_on ? plane_raw : 0;
    const bool grp = K == 1 && p.LPG < 64;                       // several planes per wave: neighbours across a group edge are padding
    const bool e0 = grp && li == 0, e1 = grp && li <= 1, eL = grp && li == p.LPG - 1;
    const int last_li = (p.W - 1) & 63;                          // lane of the last column (in k = K-1)
    bool con[K];                                                 // column in range
#pragma unroll
    for (int k = 0; k < K; ++k) con[k] = lane_on && li + 64 * k < p.W;
    const float* xb = p.x + (long)plane * p.H * p.W + li;
    const float a0 = p.a[0], a1 = p.a[1], a2 = p.a[2], a3 = p.a[3];
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2], b3 = p.b[3];

    float pf[B][K];                    // rows in flight
    float hr[4][K];                    // last four horizontally filtered rows
    float he[4];                       // ... their column W (valid in the last lane of a row)
    auto load_row = [&](float (&dst)[K], int iy) __attribute__((always_inline)) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const float* src = xb + (long)iyc * p.W;
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = (DBG & 1) ? (float)iyc : src[con[k] ? 64 * k : 0];
    };
    const long plane_elems = p.mode ? (long)p.ph2 * p.pitch : (long)OH * p.pitch;
    const int w2 = p.W >> 1;

    auto step = [&](auto J, int i) __attribute__((always_inline)) {
        constexpr int jb = decltype(J)::value, j = jb & 3;
        const int iy = r0 - 2 + i;
        const bool rok = iy >= 0 && iy < p.H;
        float c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = (rok && con[k]) ? pf[jb][k] : 0.f;
        float s1l = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float f1 = k > 0 ? fm_lane(c[k - 1], 63) : 0.f, f2 = k > 0 ? fm_lane(c[k - 1], 62) : 0.f;
            const float fn = k + 1 < K ? fm_lane(c[k + 1], 0) : 0.f;
            float s1 = fm_shr1(c[k], f1);
            float s2 = fm_shr1(s1, f2);
            float n1 = fm_shl1(c[k], fn);
            s1 = e0 ? 0.f : s1; s2 = e1 ? 0.f : s2; n1 = eL ? 0.f : n1;
            hr[j][k] = a0 * s2 + a1 * s1 + a2 * c[k] + a3 * n1;
            if (k == K - 1) s1l = s1;
        }
        he[j] = a0 * s1l + a1 * c[K - 1];                        // column W = a0 x[W-2] + a1 x[W-1], in the lane holding column W-1
        const int oy = r0 + i - 3;
        if (i >= 3 && oy < r1 && (!(DBG & 2) || hr[j][0] == 12345.f)) {   // uniform
            constexpr int j0 = (j + 1) & 3, j1 = (j + 2) & 3, j2 = (j + 3) & 3;
            float o[K];
#pragma unroll
            for (int k = 0; k < K; ++k) o[k] = b0 * hr[j0][k] + b1 * hr[j1][k] + b2 * hr[j2][k] + b3 * hr[j][k];
            const float oe = b0 * he[j0] + b1 * he[j1] + b2 * he[j2] + b3 * he[j];
            if (p.mode) {
                // polyphase planes: element (oy, ox) -> plane (oy&1)*2 + (ox&1), row oy>>1, column ox>>1
                float* pe = p.y + ((long)((oy & 1) * 2) * p.NC + plane) * plane_elems + (long)(oy >> 1) * p.pitch;
                float* pl = pe + (li & 1) * (long)p.NC * plane_elems + (li >> 1);      // this lane's plane (column parity = lane parity)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (con[k]) pl[32 * k] = o[k];
                if (lane_on && li == last_li) pe[w2] = oe;
                // zero padding up to the pitch: even plane from column W/2+1, odd plane from W/2
                for (int t = li; t < 2 * (p.pitch - w2); t += p.LPG) {
                    const int zc = w2 + (t >> 1) + 1 - (t & 1);
                    if (lane_on && zc < p.pitch) pe[(t & 1) * (long)p.NC * plane_elems + zc] = 0.f;
                }
            } else {
                float* py = p.y + (long)plane * plane_elems + (long)oy * p.pitch + li;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (con[k]) py[64 * k] = o[k];
                if (lane_on && li == last_li) py[p.W - li] = oe;
            }
        }
    };
    const int nin = r1 - r0 + 3;
    static_assert(B == 4 || B == 8, "batch of 4 or 8 rows");
    for (int i = 0; i < nin; i += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) load_row(pf[j], r0 - 2 + i + j);
        step(FmIC<0>{}, i);
        step(FmIC<1>{}, i + 1);
        step(FmIC<2>{}, i + 2);
        step(FmIC<3>{}, i + 3);
        if constexpr (B == 8) {
            step(FmIC<4>{}, i + 4);
            step(FmIC<5>{}, i + 5);
            step(FmIC<6>{}, i + 6);
            step(FmIC<7>{}, i + 7);
        }
    }
    // planar form: row ph2-1 of the two odd-row planes lies beyond the filtered image -> zeros (written by the last segment)
    if (p.mode && seg == p.nseg - 1 && lane_on) {
        float* pz = p.y + ((long)2 * p.NC + plane) * plane_elems + (long)(p.ph2 - 1) * p.pitch;
        for (int q = 0; q < 2; ++q, pz += (long)p.NC * plane_elems)
            for (int c = li; c < p.pitch; c += p.LPG) pz[c] = 0.f;
    }
}

// The same filter for wide planes in polyphase-planar form (W a multiple of 256): a lane holds K float4 of a row (16-byte loads), and
// its four results go out as two 8-byte stores, (x, z) to the even-column plane and (y, w) to the odd-column one -- 512 contiguous
// bytes per plane and wave.  With a plane pitch that is a multiple of 32 floats every store covers whole 128-byte lines, the tail
// line of a row (column W/2 of the even plane, zero padding) included.
template <int K, int B, int DBG = 0>
__global__ __launch_bounds__(256) void fir_down_march4_kernel(const FirMarchParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.nitem) return;
    const int seg = item % p.nseg, plane = item / p.nseg;
    const int OH = p.H + 1;
    const int r0 = seg * p.R, r1 = min(r0 + p.R, OH);
    const float* xb = p.x + (long)plane * p.H * p.W + 4 * lane;
    const float a0 = p.a[0], a1 = p.a[1], a2 = p.a[2], a3 = p.a[3];
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2], b3 = p.b[3];
    float4 pf[B][K];
    float4 hr[4][K];
    float he[4];
    auto load_row = [&](float4 (&dst)[K], int iy) __attribute__((always_inline)) {
        const int iyc = min(max(iy, 0), p.H - 1);
        const float* src = xb + (long)iyc * p.W;
#pragma unroll
        for (int k = 0; k < K; ++k)
            dst[k] = (DBG & 1) ? make_float4(1.f, 2.f, 3.f, (float)iyc) : *reinterpret_cast<const float4*>(src + 256 * k);
    };
    const long plane_elems = (long)p.ph2 * p.pitch;
    const int w2 = p.W >> 1, npad = p.pitch - w2;
    auto step = [&](auto J, int i) __attribute__((always_inline)) {
        constexpr int jb = decltype(J)::value, j = jb & 3;
        const int iy = r0 - 2 + i;
        const bool rok = iy >= 0 && iy < p.H;
        float4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c[k].x = rok ? pf[jb][k].x : 0.f; c[k].y = rok ? pf[jb][k].y : 0.f;
            c[k].z = rok ? pf[jb][k].z : 0.f; c[k].w = rok ? pf[jb][k].w : 0.f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float fz = k > 0 ? fm_lane(c[k - 1].z, 63) : 0.f, fw = k > 0 ? fm_lane(c[k - 1].w, 63) : 0.f;
            const float fx = k + 1 < K ? fm_lane(c[k + 1].x, 0) : 0.f;
            const float pz = fm_shr1(c[k].z, fz), pw = fm_shr1(c[k].w, fw), nx = fm_shl1(c[k].x, fx);
            hr[j][k].x = a0 * pz + a1 * pw + a2 * c[k].x + a3 * c[k].y;
            hr[j][k].y = a0 * pw + a1 * c[k].x + a2 * c[k].y + a3 * c[k].z;
            hr[j][k].z = a0 * c[k].x + a1 * c[k].y + a2 * c[k].z + a3 * c[k].w;
            hr[j][k].w = a0 * c[k].y + a1 * c[k].z + a2 * c[k].w + a3 * nx;
        }
        he[j] = a0 * fm_lane(c[K - 1].z, 63) + a1 * fm_lane(c[K - 1].w, 63);      // column W (uniform)
        const int oy = r0 + i - 3;
        if (i >= 3 && oy < r1 && (!(DBG & 2) || hr[j][0].x == 12345.f)) {      // uniform
            constexpr int j0 = (j + 1) & 3, j1 = (j + 2) & 3, j2 = (j + 3) & 3;
            float* pe = p.y + ((long)((oy & 1) * 2) * p.NC + plane) * plane_elems + (long)(oy >> 1) * p.pitch;
            float* po = pe + (long)p.NC * plane_elems;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float4 o;
                o.x = b0 * hr[j0][k].x + b1 * hr[j1][k].x + b2 * hr[j2][k].x + b3 * hr[j][k].x;
                o.y = b0 * hr[j0][k].y + b1 * hr[j1][k].y + b2 * hr[j2][k].y + b3 * hr[j][k].y;
                o.z = b0 * hr[j0][k].z + b1 * hr[j1][k].z + b2 * hr[j2][k].z + b3 * hr[j][k].z;
                o.w = b0 * hr[j0][k].w + b1 * hr[j1][k].w + b2 * hr[j2][k].w + b3 * hr[j][k].w;
                const int c2 = 2 * (lane + 64 * k);
                *reinterpret_cast<float2*>(pe + c2) = make_float2(o.x, o.z);
                *reinterpret_cast<float2*>(po + c2) = make_float2(o.y, o.w);
            }
            const float oe = b0 * he[j0] + b1 * he[j1] + b2 * he[j2] + b3 * he[j];
            // tail of the row: column W/2 of the even plane, zeros up to the pitch in both
            for (int t = lane; t < npad; t += 64) {
                pe[w2 + t] = t == 0 ? oe : 0.f;
                po[w2 + t] = 0.f;
            }
        }
    };
    const int nin = r1 - r0 + 3;
    for (int i = 0; i < nin; i += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) load_row(pf[j], r0 - 2 + i + j);
        step(FmIC<0>{}, i);
        step(FmIC<1>{}, i + 1);
        step(FmIC<2>{}, i + 2);
        step(FmIC<3>{}, i + 3);
        if constexpr (B == 8) {
            step(FmIC<4>{}, i + 4);
            step(FmIC<5>{}, i + 5);
            step(FmIC<6>{}, i + 6);
            step(FmIC<7>{}, i + 7);
        }
    }
    if (seg == p.nseg - 1) {
        float* pz = p.y + ((long)2 * p.NC + plane) * plane_elems + (long)(p.ph2 - 1) * p.pitch;
        for (int q = 0; q < 2; ++q, pz += (long)p.NC * plane_elems)
            for (int c = lane; c < p.pitch; c += 64) pz[c] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// pad-1 4x4 FIR after the transposed convolution (conv2d_resample.py:133-138), fused with the synthesis-layer tail
// (stylegan.py:295-304, comodgan.py:326-327).  Input: the four phase planes mid[(a*2+b)][nc][u][v] = full[2u+a][2v+b] of the
// (2H+1) x (2W+1) transposed-conv result, each (H+1) x (W+1) with pitch W+1; output y [NC, 2H, 2W]:
//     y[Y][X] = act((sum_{ky,kx} b[ky] a[kx] full[Y+ky-1][X+kx-1]) * scale[nc] + noise * strength + bias[c]) + residual.
// A lane owns 4 output columns = 2 low-resolution columns (v0 = 2c, v1 = v0+1) and marches down the low-resolution rows: per step it
// loads row r of the four planes (dword loads: the pitch is odd), filters it horizontally into er (full row 2r) and or (row 2r+1)
// with three DPP neighbour values each, and emits output rows 2(r-1), 2(r-1)+1 from (or[r-2], er[r-1], or[r-1], er[r], or[r]) --
// 16-byte stores and 16-byte noise / skip loads on 128-byte aligned rows.
struct FirUpParams {
    const float* mid; float* y;
    const float* scale; const float* bias; const float* noise; const float* residual;
    int NC, C, H, W;          // low-resolution extent
    int noise_mode;           // 0 none, 1 [2H,2W], 2 [N,2H,2W]
    float noise_strength;
    int act; float alpha, act_gain, clamp;
    int R, nseg, LPG, G, nitem;
    float a[4], b[4];
};

__device__ __forceinline__ float fm_act(float v, int act, float alpha, float gain, float clamp) {
    if (!act) return v;
    v = (v < 0.f ? v * alpha : v) * gain;
    return clamp >= 0.f ? fminf(fmaxf(v, -clamp), clamp) : v;
}

template <int K, int B>
__global__ __launch_bounds__(256) void fir_up_march_kernel(const FirUpParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.nitem) return;
    const int seg = item % p.nseg, pg = item / p.nseg;           // (handing a workgroup the same segment of four planes instead -- shared
    const int g = lane / p.LPG, li = lane - g * p.LPG;           //  noise rows -- measures the same)
    const int u0 = seg * p.R, u1 = min(u0 + p.R, p.H);
    const int plane_raw = pg * p.G + g;
    const bool lane_on = plane_raw < p.NC;
    const int plane = lane_on ? plane_raw : 0;
    const bool grp = K == 1 && p.LPG < 64;
    const bool gfirst = grp && li == 0, glast = grp && li == p.LPG - 1;
    const int PW = p.W + 1, OW = 2 * p.W, OH = 2 * p.H;
    const long P = (long)(p.H + 1) * PW;
    const float* mb = p.mid + (long)plane * P + 2 * li;                    // plane q adds q * NC * P
    const long qs = (long)p.NC * P;
    const int n = plane / p.C, ch = plane - n * p.C;
    const float sc = p.scale ? p.scale[plane] : 1.f, bs = p.bias ? p.bias[ch] : 0.f;
    const float* nzb = p.noise_mode == 0 ? nullptr : (p.noise_mode == 1 ? p.noise : p.noise + (long)n * OH * OW);
    const float* rsb = p.residual ? p.residual + (long)plane * OH * OW : nullptr;
    float* yb = p.y + (long)plane * OH * OW;
    const float a0 = p.a[0], a1 = p.a[1], a2 = p.a[2], a3 = p.a[3];
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2], b3 = p.b[3];

    struct Row { float e0[2][K], e1[2][K], o0[2][K], o1[2][K], ew[2]; float4 nz[2][K], rs[2][K]; };   // [a] = row parity
    Row in[B];
    float4 orA[K], erB[K], orB[K];                                         // or[r-2], er[r-1], or[r-1]
#pragma unroll
    for (int k = 0; k < K; ++k) orA[k] = erB[k] = orB[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_step = [&](Row& d, int r) __attribute__((always_inline)) {
        const int rc = min(max(r, 0), p.H);
        const float* src = mb + (long)rc * PW;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float* se = src + (2 * a) * qs;
            const float* so = se + qs;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                d.e0[a][k] = se[128 * k]; d.e1[a][k] = se[128 * k + 1];      // (an 8-byte load from the 4-byte aligned address is slower)
                d.o0[a][k] = so[128 * k]; d.o1[a][k] = so[128 * k + 1];
            }
            d.ew[a] = se[p.W - 2 * li];                                   // E[W], the column past the last lane
        }
        // noise / skip rows of the outputs this step completes: Y = 2(r-1), 2(r-1)+1
        const int yc = min(max(2 * (r - 1), 0), OH - 2);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const long off = (long)(yc + dy) * OW + 4 * (li + 64 * k);
                d.nz[dy][k] = nzb ? *reinterpret_cast<const float4*>(nzb + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                d.rs[dy][k] = rsb ? *reinterpret_cast<const float4*>(rsb + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    auto hfilt = [&](const Row& d, int a, bool ok, float4 (&h)[K]) __attribute__((always_inline)) {
        float e0[K], e1[K], o0[K], o1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            e0[k] = ok ? d.e0[a][k] : 0.f; e1[k] = ok ? d.e1[a][k] : 0.f;
            o0[k] = ok ? d.o0[a][k] : 0.f; o1[k] = ok ? d.o1[a][k] : 0.f;
        }
        const float ew = ok ? d.ew[a] : 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float fo = k > 0 ? fm_lane(o1[k - 1], 63) : 0.f;
            float om1 = fm_shr1(o1[k], fo);
            float e2 = k + 1 < K ? fm_shl1(e0[k], fm_lane(e0[k + 1], 0)) : fm_shl1(e0[k], ew);
            float o2 = fm_shl1(o0[k], k + 1 < K ? fm_lane(o0[k + 1], 0) : 0.f);
            om1 = gfirst ? 0.f : om1; e2 = glast ? ew : e2; o2 = glast ? 0.f : o2;
            h[k].x = a0 * om1 + a1 * e0[k] + a2 * o0[k] + a3 * e1[k];
            h[k].y = a0 * e0[k] + a1 * o0[k] + a2 * e1[k] + a3 * o1[k];
            h[k].z = a0 * o0[k] + a1 * e1[k] + a2 * o1[k] + a3 * e2;
            h[k].w = a0 * e1[k] + a1 * o1[k] + a2 * e2 + a3 * o2;
        }
    };
    auto finish = [&](float v, float nz, float rs) __attribute__((always_inline)) {
        return fm_act(v * sc + nz * p.noise_strength + bs, p.act, p.alpha, p.act_gain, p.clamp) + rs;
    };
    auto step = [&](const Row& d, int t) __attribute__((always_inline)) {
        const int r = u0 - 1 + t;
        float4 ern[K], orn[K];
        hfilt(d, 0, r >= 0 && r <= p.H, ern);
        hfilt(d, 1, r >= 0 && r <= p.H - 1, orn);
        const int u = r - 1;
        if (t >= 2 && u < u1) {                                            // uniform
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float4 y0, y1;
                y0.x = b0 * orA[k].x + b1 * erB[k].x + b2 * orB[k].x + b3 * ern[k].x;
                y0.y = b0 * orA[k].y + b1 * erB[k].y + b2 * orB[k].y + b3 * ern[k].y;
                y0.z = b0 * orA[k].z + b1 * erB[k].z + b2 * orB[k].z + b3 * ern[k].z;
                y0.w = b0 * orA[k].w + b1 * erB[k].w + b2 * orB[k].w + b3 * ern[k].w;
                y1.x = b0 * erB[k].x + b1 * orB[k].x + b2 * ern[k].x + b3 * orn[k].x;
                y1.y = b0 * erB[k].y + b1 * orB[k].y + b2 * ern[k].y + b3 * orn[k].y;
                y1.z = b0 * erB[k].z + b1 * orB[k].z + b2 * ern[k].z + b3 * orn[k].z;
                y1.w = b0 * erB[k].w + b1 * orB[k].w + b2 * ern[k].w + b3 * orn[k].w;
                y0.x = finish(y0.x, d.nz[0][k].x, d.rs[0][k].x); y0.y = finish(y0.y, d.nz[0][k].y, d.rs[0][k].y);
                y0.z = finish(y0.z, d.nz[0][k].z, d.rs[0][k].z); y0.w = finish(y0.w, d.nz[0][k].w, d.rs[0][k].w);
                y1.x = finish(y1.x, d.nz[1][k].x, d.rs[1][k].x); y1.y = finish(y1.y, d.nz[1][k].y, d.rs[1][k].y);
                y1.z = finish(y1.z, d.nz[1][k].z, d.rs[1][k].z); y1.w = finish(y1.w, d.nz[1][k].w, d.rs[1][k].w);
                if (lane_on) {
                    float* yr = yb + (long)(2 * u) * OW + 4 * (li + 64 * k);
                    *reinterpret_cast<float4*>(yr) = y0;
                    *reinterpret_cast<float4*>(yr + OW) = y1;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { orA[k] = orB[k]; erB[k] = ern[k]; orB[k] = orn[k]; }
    };
    const int nt = u1 - u0 + 2;
    for (int t = 0; t < nt; t += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) load_step(in[j], u0 - 1 + t + j);
#pragma unroll
        for (int j = 0; j < B; ++j) step(in[j], t + j);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The two resampling FIRs of the training rows, same marching scheme, separable 4-tap filters:
//   fir_dn2_march_kernel:  y [NC,H/2,W/2] = upfirdn2d(x, f, down=2, padding=1)   -- the pre-filter of the discriminator's 1x1
//       stride-2 skip convolutions (conv2d_resample.py:104-108) and the backward of the up-sampling FIR (upfirdn2d.py:174-192);
//   fir_up2_march_kernel:  y [NC,2h,2w]   = upfirdn2d(g, f, up=2, padding=[2,1,2,1]) -- upsample2d of the running RGB image
//       (upfirdn2d.py:288-305, gain 4 in the taps) and the backward of the above.
//     dn2:  y[oy][ox]   = sum b[ky] a[kx] x[2oy+ky-1][2ox+kx-1]
//     up2:  y[2u][2v]   = (b0 r[u-1] + b2 r[u])[2v],  y[2u+1] = (b1 r[u] + b3 r[u+1]),   r[u][2v] = a0 g[u][v-1] + a2 g[u][v],
//           r[u][2v+1] = a1 g[u][v] + a3 g[u][v+1]                    (zero insertion: every output sees 2 x 2 inputs)
struct FirRsParams {
    const float* x; float* y;
    int NC, H, W;             // INPUT extent
    int R, nseg, LPG, G, nitem;
    float a[4], b[4];
};

template <int K>   // K float4 per lane per input row (W = 256 K for K > 1)
__global__ __launch_bounds__(256) void fir_dn2_march_kernel(const FirRsParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.nitem) return;
    const int seg = item % p.nseg, pg = item / p.nseg;
    const int g = lane / p.LPG, li = lane - g * p.LPG;
    const int OH = p.H >> 1, OW = p.W >> 1;
    const int r0 = seg * p.R, r1 = min(r0 + p.R, OH);           // output rows
    const int plane_raw = pg * p.G + g;
    const bool lane_on = plane_raw < p.NC;
    const int plane = lane_on ? plane_raw : 0;
    const bool grp = K == 1 && p.LPG < 64;
    const bool gfirst = grp && li == 0, glast = grp && li == p.LPG - 1;
    const float* xb = p.x + (long)plane * p.H * p.W + 4 * li;
    float* yb = p.y + (long)plane * OH * OW + 2 * li;
    const float a0 = p.a[0], a1 = p.a[1], a2 = p.a[2], a3 = p.a[3];
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2], b3 = p.b[3];
    auto hrow = [&](int iy, float2 (&h)[K]) __attribute__((always_inline)) {
        const bool ok = iy >= 0 && iy < p.H;
        const float* src = xb + (long)min(max(iy, 0), p.H - 1) * p.W;
        float4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + 256 * k);
            c[k].x = ok ? v.x : 0.f; c[k].y = ok ? v.y : 0.f; c[k].z = ok ? v.z : 0.f; c[k].w = ok ? v.w : 0.f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float pw = fm_shr1(c[k].w, k > 0 ? fm_lane(c[k - 1].w, 63) : 0.f);
            float nx = fm_shl1(c[k].x, k + 1 < K ? fm_lane(c[k + 1].x, 0) : 0.f);
            pw = gfirst ? 0.f : pw; nx = glast ? 0.f : nx;
            h[k].x = a0 * pw + a1 * c[k].x + a2 * c[k].y + a3 * c[k].z;
            h[k].y = a0 * c[k].y + a1 * c[k].z + a2 * c[k].w + a3 * nx;
        }
    };
    float2 hm1[K], h0[K], h1[K], h2[K];                       // rows 2oy-1, 2oy, 2oy+1, 2oy+2
    hrow(2 * r0 - 1, hm1);
    hrow(2 * r0, h0);
    for (int oy = r0; oy < r1; ++oy) {
        hrow(2 * oy + 1, h1);
        hrow(2 * oy + 2, h2);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float2 o;
            o.x = b0 * hm1[k].x + b1 * h0[k].x + b2 * h1[k].x + b3 * h2[k].x;
            o.y = b0 * hm1[k].y + b1 * h0[k].y + b2 * h1[k].y + b3 * h2[k].y;
            if (lane_on) *reinterpret_cast<float2*>(yb + (long)oy * OW + 128 * k) = o;
            hm1[k] = h1[k]; h0[k] = h2[k];
        }
    }
}

template <int K>   // K float2 per lane per input row (w = 128 K for K > 1)
__global__ __launch_bounds__(256) void fir_up2_march_kernel(const FirRsParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.nitem) return;
    const int seg = item % p.nseg, pg = item / p.nseg;
    const int g = lane / p.LPG, li = lane - g * p.LPG;
    const int OW = 2 * p.W;
    const int u0 = seg * p.R, u1 = min(u0 + p.R, p.H);           // input rows
    const int plane_raw = pg * p.G + g;
    const bool lane_on = plane_raw < p.NC;
    const int plane = lane_on ? plane_raw : 0;
    const bool grp = K == 1 && p.LPG < 64;
    const bool gfirst = grp && li == 0, glast = grp && li == p.LPG - 1;
    const float* xb = p.x + (long)plane * p.H * p.W + 2 * li;
    float* yb = p.y + (long)plane * (2 * p.H) * OW + 4 * li;
    const float a0 = p.a[0], a1 = p.a[1], a2 = p.a[2], a3 = p.a[3];
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2], b3 = p.b[3];
    auto hrow = [&](int u, float4 (&h)[K]) __attribute__((always_inline)) {
        const bool ok = u >= 0 && u < p.H;
        const float* src = xb + (long)min(max(u, 0), p.H - 1) * p.W;
        float2 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 v = *reinterpret_cast<const float2*>(src + 128 * k);
            c[k].x = ok ? v.x : 0.f; c[k].y = ok ? v.y : 0.f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float py = fm_shr1(c[k].y, k > 0 ? fm_lane(c[k - 1].y, 63) : 0.f);
            float nx = fm_shl1(c[k].x, k + 1 < K ? fm_lane(c[k + 1].x, 0) : 0.f);
            py = gfirst ? 0.f : py; nx = glast ? 0.f : nx;
            h[k].x = a0 * py + a2 * c[k].x;          // columns 2v, 2v+1, 2v+2, 2v+3 (v = 2 (li + 64 k))
            h[k].y = a1 * c[k].x + a3 * c[k].y;
            h[k].z = a0 * c[k].x + a2 * c[k].y;
            h[k].w = a1 * c[k].y + a3 * nx;
        }
    };
    float4 rm1[K], r0_[K], rp1[K];
    hrow(u0 - 1, rm1);
    hrow(u0, r0_);
    for (int u = u0; u < u1; ++u) {
        hrow(u + 1, rp1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 e, o;
            e.x = b0 * rm1[k].x + b2 * r0_[k].x; e.y = b0 * rm1[k].y + b2 * r0_[k].y;
            e.z = b0 * rm1[k].z + b2 * r0_[k].z; e.w = b0 * rm1[k].w + b2 * r0_[k].w;
            o.x = b1 * r0_[k].x + b3 * rp1[k].x; o.y = b1 * r0_[k].y + b3 * rp1[k].y;
            o.z = b1 * r0_[k].z + b3 * rp1[k].z; o.w = b1 * r0_[k].w + b3 * rp1[k].w;
            if (lane_on) {
                float* yr = yb + (long)(2 * u) * OW + 256 * k;
                *reinterpret_cast<float4*>(yr) = e;
                *reinterpret_cast<float4*>(yr + OW) = o;
            }
            rm1[k] = r0_[k]; r0_[k] = rp1[k];
        }
    }
}
