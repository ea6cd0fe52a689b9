// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the generator forward path, what the reference delegates to cuDNN through
//   lib/model_zoo/stylegan_utils/conv2d_gradfix.py:35-43,109-116  (F.conv2d / F.conv_transpose2d)
// as called from conv2d_resample.py:26-51 and stylegan.py:103-193 (modulated_conv2d).
//
// Formulation (MI355X-first, not a translation of the grouped-conv trick of stylegan.py:187-190):
//   y[n,o,P] = act( out_scale[n,o] * sum_{i,t} Wt[i,t,o] * (x[n,i,P*S + d_t] * in_scale[n,i])
//                   + noise + bias[o] ) + residual
// i.e. the *non-fused* modulation algebra (stylegan.py:172-181): the weight matrix is shared by the
// whole batch, so one [I*taps, O] operand stays L2/LDS resident while all samples stream through.
// GEMM view: M = O (A operand = weights), N = pixels (B operand = im2col of x, built in LDS from a
// halo patch), K = I*taps.
//
// Kernel variants (one template):
//   * tap-list conv  (UP = false): stride-1 'same', stride-2, single-tap 1x1;
//   * all-phase transposed stride-2 3x3 conv (UP = true): one pass over the low-resolution input
//     produces the four sub-pixel phases of the (2H+1)x(2W+1) result at once -- each of the nine taps
//     feeds exactly one phase, so the MFMA work equals the reference's conv_transpose2d and the input
//     and weights are staged once instead of four times.
// Tiling: workgroup = WO x WP waves; each wave owns (MO*32) x (NP*32) outputs = MO*NP accumulators of
// 32x32 (x4 phases for UP).  K is consumed in chunks of KC input channels: weights [KC*NTAPS][BO] and
// the input patch [PATCH][KC+1] live in LDS; every MFMA covers two channels of one tap (lanes 0-31:
// channel c, lanes 32-63: channel c+1).  Operand reads are conflict-free ds_read_b32 with immediate
// offsets.  The next chunk is prefetched into registers (global loads in flight) while the current
// one is multiplied, so one workgroup alone keeps its MFMA pipes fed; 2 workgroups per CU cover the
// barrier gaps.  Small grids are split along K (partial sums to a workspace + a fused reduce/epilogue
// kernel) so that 4x4..16x16 layers still fill 256 CUs.
#include "shg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* x;          // [NB, I, H, W]
    const float* wt;         // prepped weights [wgroups][I][NTAPS][OP]
    float* y;                // output (see out_mode)
    float* part;             // split-K partial sums [ksplit][...same indexing as y...] or null
    const float* in_scale;   // [NB, I] or null
    const float* out_scale;  // [NB, O] or null
    const float* bias;       // [O] or null
    const float* noise;      // see noise_mode
    const float* residual;   // like y, added after the activation
    int NB, I, O, OP;
    int H, W;
    int OHp, OWp;            // output grid computed by this launch
    int OHt, OWt;            // full output tensor extent
    int S;                   // input stride
    int dy0, dx0;            // patch origin = (oy0*S + dy0, ox0*S + dx0)
    int PH, PW, PATCH;       // patch rows/cols per image; PATCH = TN*PH*PW
    int tw_log2, th_log2, tn_log2;
    int tiles_x, tiles_y, n_ptiles, n_otiles;
    int wgroups;
    long wstride;
    int ksplit, i_per_slice;
    long part_stride;        // elements per split-K slice
    int out_mode;            // UP only: 0 = interleaved [NB,O,2H+1,2W+1]; 1 = planar [4][NB,O,H+1,W+1]
    int noise_mode;          // 0 none, 1 [OHt,OWt], 2 [NB,OHt,OWt]
    float noise_strength;
    int act;
    float alpha, gain, clamp;
    int tap_off[9];          // LDS patch offset (in patch elements) of tap t
};

// Bijective XCD-aware remap (blocks b, b+8, ... share an XCD and its L2): every XCD walks a
// contiguous range of the o-tile-major work list, so its L2 holds one weight slice at a time.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float conv_epilogue(const ConvParams& p, float v, int n, int o, long idx, float nz) {
    if (p.out_scale) v *= p.out_scale[n * p.O + o];
    v += nz;
    if (p.bias) v += p.bias[o];
    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
    if (p.residual) v += p.residual[idx];
    return v;
}

template <int NTAPS, int KC, int MO, int NP, int WO, int WP, int XQ, bool UP>
__global__ __launch_bounds__(WO * WP * 64, 2) void conv_mfma_kernel(const ConvParams p) {
    constexpr int BO = MO * 32 * WO;
    constexpr int NT = WO * WP * 64;
    constexpr int ROWS = KC * NTAPS;
    constexpr int XP = KC + 1;       // odd pitch: conflict-free staging writes and B reads
    constexpr int NPH = UP ? 4 : 1;
    constexpr int V4 = ROWS * BO / 4;
    constexpr int PER = (V4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                // [ROWS][BO]
    float* Xl = smem + ROWS * BO;    // [PATCH][XP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;

    const int nwork = p.n_ptiles * p.n_otiles;
    const int kslice = blockIdx.x / nwork;
    const int work = xcd_remap(blockIdx.x - kslice * nwork, nwork);
    const int otile = work / p.n_ptiles;
    const int ptile = work - otile * p.n_ptiles;
    const int txb = ptile % p.tiles_x;
    const int tyb = (ptile / p.tiles_x) % p.tiles_y;
    const int tnb = ptile / (p.tiles_x * p.tiles_y);
    const int TN = 1 << p.tn_log2;
    const int n0 = tnb << p.tn_log2, oy0 = tyb << p.th_log2, ox0 = txb << p.tw_log2;
    const int o0 = otile * BO;
    const int HW = p.H * p.W;
    const int PHW = p.PH * p.PW;
    const int i_begin = kslice * p.i_per_slice;
    const int i_end = min(p.I, i_begin + p.i_per_slice);

    // ---- per-lane staging assignment: patch elements q = tid + k*NT -> global offset (or -1 = padding)
    int xoff[XQ], xsn[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int q = tid + k * NT;
        const int tn = q / PHW;
        const int rem = q - tn * PHW;
        const int py = rem / p.PW, px = rem - py * p.PW;
        const int n = n0 + tn;
        const int iy = oy0 * p.S + p.dy0 + py, ix = ox0 * p.S + p.dx0 + px;
        const bool ok = (q < p.PATCH) && (n < p.NB) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        xoff[k] = ok ? (n * p.I * HW + iy * p.W + ix) : -1;
        xsn[k] = n * p.I;
    }

    // ---- per-lane B-fragment base inside the patch (pixel j of the tile -> patch element)
    const float* xbase[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        const int tx = j & ((1 << p.tw_log2) - 1);
        const int ty = (j >> p.tw_log2) & ((1 << p.th_log2) - 1);
        int tn = j >> (p.tw_log2 + p.th_log2);
        tn = tn < TN ? tn : 0;   // out-of-tile lanes read image 0 of the tile (masked on store)
        xbase[np] = Xl + (tn * PHW + ty * p.S * p.PW + tx * p.S) * XP + half;
    }
    const float* wa = Wl + wo * MO * 32 + l31 + half * NTAPS * BO;

    f32x16 acc[NPH][MO][NP];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][mo][np][r] = 0.f;

    const float* wbase = p.wt + (long)(p.wgroups > 1 ? (n0 % p.wgroups) : 0) * p.wstride;

    float4 wv[PER];
    float xv[XQ][KC];
    auto load_chunk = [&](int i0) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * NT;
            const int row = e / (BO / 4), c4 = e - row * (BO / 4);
            const int o = o0 + c4 * 4;
            const long grow = (long)i0 * NTAPS + row;
            wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < V4 && grow < (long)i_end * NTAPS && o < p.OP)
                wv[k] = *reinterpret_cast<const float4*>(wbase + grow * p.OP + o);
        }
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
#pragma unroll
            for (int ic = 0; ic < KC; ++ic) {
                const int i = i0 + ic;
                float v = 0.f;
                if (xoff[k] >= 0 && i < i_end) {
                    v = p.x[xoff[k] + i * HW];
                    if (p.in_scale) v *= p.in_scale[xsn[k] + i];
                }
                xv[k][ic] = v;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * NT;
            if (e < V4) *reinterpret_cast<float4*>(Wl + e * 4) = wv[k];
        }
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            const int q = tid + k * NT;
            if (q < p.PATCH) {
#pragma unroll
                for (int ic = 0; ic < KC; ++ic) Xl[q * XP + ic] = xv[k][ic];
            }
        }
    };

    load_chunk(i_begin);
    store_chunk();
    __syncthreads();

    for (int i0 = i_begin; i0 < i_end; i0 += KC) {
        const bool more = i0 + KC < i_end;
        if (more) load_chunk(i0 + KC);      // global loads stay in flight across the MFMA block

        if constexpr (!UP) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int toff = p.tap_off[t] * XP;
#pragma unroll
                for (int c2 = 0; c2 < KC / 2; ++c2) {
                    float a[MO], b[NP];
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo) a[mo] = wa[((c2 * 2) * NTAPS + t) * BO + mo * 32];
#pragma unroll
                    for (int np = 0; np < NP; ++np) b[np] = xbase[np][toff + c2 * 2];
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                        for (int np = 0; np < NP; ++np)
                            acc[0][mo][np] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mo], b[np], acc[0][mo][np], 0, 0, 0);
                }
            }
        } else {
            // tap t = ky*3+kx feeds phase (ky&1, kx&1) from the input pixel shifted by (ky==2 ? -1 : 0, kx==2 ? -1 : 0);
            // the patch origin is (u0-1, v0-1), so shift (sy,sx) sits at patch offset (1+sy)*PW + (1+sx)
            const int s00 = (p.PW + 1) * XP, s01 = p.PW * XP, s10 = XP, s11 = 0;
#pragma unroll
            for (int c2 = 0; c2 < KC / 2; ++c2) {
                float b[4][NP];
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    b[0][np] = xbase[np][s00 + c2 * 2];   // ( 0, 0)
                    b[1][np] = xbase[np][s01 + c2 * 2];   // ( 0,-1)
                    b[2][np] = xbase[np][s10 + c2 * 2];   // (-1, 0)
                    b[3][np] = xbase[np][s11 + c2 * 2];   // (-1,-1)
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    constexpr int PHASE[9] = {0, 1, 0, 2, 3, 2, 0, 1, 0};
                    constexpr int SHIFT[9] = {0, 0, 1, 0, 0, 1, 2, 2, 3};
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo) {
                        const float a = wa[((c2 * 2) * 9 + t) * BO + mo * 32];
#pragma unroll
                        for (int np = 0; np < NP; ++np)
                            acc[PHASE[t]][mo][np] =
                                __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[SHIFT[t]][np], acc[PHASE[t]][mo][np], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: D[row = out channel][col = pixel]; row = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        const int tx = j & ((1 << p.tw_log2) - 1);
        const int ty = (j >> p.tw_log2) & ((1 << p.th_log2) - 1);
        const int tn = j >> (p.tw_log2 + p.th_log2);
        const int n = n0 + tn, oy = oy0 + ty, ox = ox0 + tx;
        if (tn >= TN || n >= p.NB || oy >= p.OHp || ox >= p.OWp) continue;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            long plane = (long)p.OHt * p.OWt;    // elements per (n,o) plane of the output
            long base;                            // offset of (n, o=0, pixel)
            int pix;
            if constexpr (UP) {
                const int a = ph >> 1, b = ph & 1;
                if (a == 1 && oy >= p.H) continue;                    // Y = 2u+a <= 2H
                if (b == 1 && ox >= p.W) continue;                    // X = 2v+b <= 2W
                if (p.out_mode == 1) {
                    plane = (long)(p.H + 1) * (p.W + 1);
                    pix = oy * (p.W + 1) + ox;
                    base = ((long)ph * p.NB + n) * p.O * plane + pix;
                } else {
                    pix = (2 * oy + a) * p.OWt + 2 * ox + b;
                    base = (long)n * p.O * plane + pix;
                }
            } else {
                pix = oy * p.OWt + ox;
                base = (long)n * p.O * plane + pix;
            }
            float nz = 0.f;
            if (p.ksplit == 1) {
                if (p.noise_mode == 1) nz = p.noise[pix] * p.noise_strength;
                else if (p.noise_mode == 2) nz = p.noise[(long)n * plane + pix] * p.noise_strength;
            }
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + (wo * MO + mo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o >= p.O) continue;
                    const long idx = base + (long)o * plane;
                    const float v = acc[ph][mo][np][r];
                    if (p.ksplit > 1) p.part[(long)kslice * p.part_stride + idx] = v;
                    else p.y[idx] = conv_epilogue(p, v, n, o, idx, nz);
                }
            }
        }
    }
}

// split-K tail: y = epilogue(sum_s part[s]) over an [NB, O, plane] tensor
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p, long total, long plane) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.part[(long)s * p.part_stride + e];
        const long no = e / plane;
        const int pix = (int)(e - no * plane);
        const int n = (int)(no / p.O), o = (int)(no - (long)n * p.O);
        float nz = 0.f;
        if (p.noise_mode == 1) nz = p.noise[pix] * p.noise_strength;
        else if (p.noise_mode == 2) nz = p.noise[(long)n * plane + pix] * p.noise_strength;
        p.y[e] = conv_epilogue(p, v, n, o, e, nz);
    }
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct ConvPlan { int tw, th, tn, n_ptiles, n_otiles, ksplit, i_per_slice; };

// Tile shape and split-K factor for a launch: BP pixels per tile as TW x TH x TN (powers of two).
static ConvPlan conv_plan(int NB, int I, int O, int OHp, int OWp, int BO, int BP, int KC, int wgroups, bool allow_split) {
    ConvPlan c;
    int tw = 32; while (tw > 1 && tw / 2 >= OWp) tw >>= 1;
    if (tw > BP) tw = BP;
    int th = BP / tw; while (th > 1 && th / 2 >= OHp) th >>= 1;
    int tn = BP / (tw * th);
    if (wgroups > 1) tn = 1;                    // per-slot weights: one image per tile
    while (tn > 1 && tn / 2 >= NB) tn >>= 1;
    c.tw = tw; c.th = th; c.tn = tn;
    c.n_ptiles = shg_cdiv(OWp, tw) * shg_cdiv(OHp, th) * shg_cdiv(NB, tn);
    c.n_otiles = shg_cdiv(O, BO);
    const int chunks = shg_cdiv(I, KC);
    int ks = 1;
    if (allow_split) {
        const int grid = c.n_ptiles * c.n_otiles;
        // aim at >= 2 workgroups per CU (512) while keeping >= 4 chunks per slice
        while (grid * ks < 512 && chunks / (ks * 2) >= 4 && ks < 64) ks *= 2;
    }
    c.i_per_slice = shg_cdiv(chunks, ks) * KC;
    c.ksplit = shg_cdiv(I, c.i_per_slice);
    return c;
}

template <int NTAPS, int KC, int MO, int NP, int WO, int WP, int XQ, bool UP>
static int launch_conv(ConvParams& p, void* workspace, size_t ws_bytes, hipStream_t s) {
    constexpr int BO = MO * 32 * WO, BP = NP * 32 * WP, NT = WO * WP * 64;
    const long out_elems = (UP && p.out_mode == 1) ? 4L * p.NB * p.O * (p.H + 1) * (p.W + 1) : (long)p.NB * p.O * p.OHt * p.OWt;
    ConvPlan c = conv_plan(p.NB, p.I, p.O, p.OHp, p.OWp, BO, BP, KC, p.wgroups, workspace != nullptr && !UP);
    if (c.ksplit > 1 && (size_t)c.ksplit * out_elems * sizeof(float) > ws_bytes) {   // workspace too small: no split
        c.ksplit = 1; c.i_per_slice = shg_cdiv(p.I, KC) * KC;
    }
    p.tw_log2 = ilog2(c.tw); p.th_log2 = ilog2(c.th); p.tn_log2 = ilog2(c.tn);
    p.tiles_x = shg_cdiv(p.OWp, c.tw); p.tiles_y = shg_cdiv(p.OHp, c.th);
    p.n_ptiles = c.n_ptiles; p.n_otiles = c.n_otiles;
    p.ksplit = c.ksplit; p.i_per_slice = c.i_per_slice;
    p.part = c.ksplit > 1 ? (float*)workspace : nullptr;
    p.part_stride = out_elems;
    // patch extents: PH/PW hold the tap spans on entry
    const int span_y = p.PH, span_x = p.PW;
    p.PH = (c.th - 1) * p.S + span_y;
    p.PW = (c.tw - 1) * p.S + span_x;
    p.PATCH = c.tn * p.PH * p.PW;
    if (p.PATCH > XQ * NT) { shg_set_error("conv: patch of %d elements exceeds the staging capacity %d", p.PATCH, XQ * NT); return SHG_ERR_UNSUPPORTED; }
    if (!UP)
        for (int t = 0; t < NTAPS; ++t) {
            const int dyr = p.tap_off[t] >> 6, dxr = p.tap_off[t] & 63;   // packed (dy-dy0, dx-dx0)
            p.tap_off[t] = dyr * p.PW + dxr;
        }
    const size_t lds = sizeof(float) * ((size_t)KC * NTAPS * BO + (size_t)(KC + 1) * p.PATCH);
    auto kern = conv_mfma_kernel<NTAPS, KC, MO, NP, WO, WP, XQ, UP>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) { shg_set_error("conv: LDS request %zu exceeds 160 KiB", lds); return SHG_ERR_UNSUPPORTED; }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { shg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return SHG_ERR_LAUNCH; }
    }
    const int grid = p.n_ptiles * p.n_otiles * p.ksplit;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p);
    SHG_CHECK_LAUNCH();
    if (p.ksplit > 1) {
        const long plane = (long)p.OHt * p.OWt;
        int rg = shg_cdiv(out_elems, 256); if (rg > 2048) rg = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, p, out_elems, plane);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}

static int conv_dispatch(ConvParams& p, int K, int S, bool up, void* workspace, size_t ws_bytes, hipStream_t s) {
    const bool narrow = p.O <= 64;    // 64 x 256 tile instead of 128 x 128
    if (up) return launch_conv<9, 8, 2, 1, 1, 4, 1, true>(p, workspace, ws_bytes, s);   // 64 channels x 128 low-res pixels x 4 phases
    if (K == 9 && S == 1) return narrow ? launch_conv<9, 8, 2, 2, 1, 4, 2, false>(p, workspace, ws_bytes, s)
                                        : launch_conv<9, 8, 2, 2, 2, 2, 2, false>(p, workspace, ws_bytes, s);
    if (K == 9 && S == 2) return narrow ? launch_conv<9, 8, 2, 2, 1, 4, 5, false>(p, workspace, ws_bytes, s)
                                        : launch_conv<9, 8, 2, 2, 2, 2, 3, false>(p, workspace, ws_bytes, s);
    if (K == 1 && S == 1) return narrow ? launch_conv<1, 32, 2, 2, 1, 4, 1, false>(p, workspace, ws_bytes, s)
                                        : launch_conv<1, 32, 2, 2, 2, 2, 1, false>(p, workspace, ws_bytes, s);
    shg_set_error("conv2d: 1x1 stride-2 convolution is not implemented (decimate with upfirdn2d first)");
    return SHG_ERR_UNSUPPORTED;
}

static int conv_fill(ConvParams& p, const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W, int kh,
                     int kw, int mode, int pad, int wgroups, long wstride, const float* in_scale, const float* out_scale,
                     const float* bias, const float* noise, int noise_mode, float noise_strength, int act, float alpha,
                     float gain, float clamp, const float* residual, int out_mode) {
    SHG_CHECK_ARG(x && wt && y, "conv2d: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1, "conv2d: empty tensor");
    SHG_CHECK_ARG((kh == 3 && kw == 3) || (kh == 1 && kw == 1), "conv2d: only 3x3 and 1x1 kernels (got %dx%d)", kh, kw);
    SHG_CHECK_ARG(OP % 4 == 0 && OP >= O, "conv2d: OP must be a multiple of 4 and >= O");
    SHG_CHECK_ARG(mode >= 0 && mode <= 2, "conv2d: bad mode %d", mode);
    SHG_CHECK_ARG(mode != 2 || (kh == 3 && pad == 0), "conv2d: transposed mode supports 3x3, padding 0");
    SHG_CHECK_ARG(pad >= 0 && pad < 32, "conv2d: bad padding");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L, "conv2d: x is too large");
    p = ConvParams{};
    p.x = x; p.wt = wt; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.bias = bias;
    p.noise = noise_mode ? noise : nullptr; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    p.wgroups = wgroups < 1 ? 1 : wgroups; p.wstride = wstride;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.out_mode = out_mode; p.ksplit = 1;
    if (mode == 2) {
        // transposed stride 2 (conv2d_resample.py:130-137): Y = 2u+a, X = 2v+b over u in [0,H], v in [0,W]
        p.OHt = 2 * H + 1; p.OWt = 2 * W + 1; p.OHp = H + 1; p.OWp = W + 1; p.S = 1;
        p.dy0 = -1; p.dx0 = -1; p.PH = 2; p.PW = 2;    // taps reach rows u-1..u, cols v-1..v
        SHG_CHECK_ARG(4L * NB * O * (H + 1) * (W + 1) < 2147483647L, "conv2d: y is too large");
        return SHG_OK;
    }
    const int S = mode == 0 ? 1 : 2;
    const int OH = (H + 2 * pad - kh) / S + 1, OW = (W + 2 * pad - kw) / S + 1;
    SHG_CHECK_ARG(OH >= 1 && OW >= 1, "conv2d: output must be at least 1x1");
    SHG_CHECK_ARG((long)NB * O * OH * OW < 2147483647L, "conv2d: y is too large");
    p.OHp = p.OHt = OH; p.OWp = p.OWt = OW; p.S = S;
    p.dy0 = -pad; p.dx0 = -pad; p.PH = kh; p.PW = kw;
    for (int t = 0; t < kh * kw; ++t) p.tap_off[t] = ((t / kw) << 6) | (t % kw);
    return SHG_OK;
}

// mode: 0 = stride 1, symmetric padding `pad`        (conv2d_resample.py:145-147)
//       1 = stride 2, symmetric padding `pad`        (conv2d_resample.py:116-120, strided conv)
//       2 = transposed stride 2, padding 0 -> [2H+1, 2W+1]   (conv2d_resample.py:122-137)
// out_mode (mode 2 only): 0 = interleaved result; 1 = four phase planes [4][NB,O,H+1,W+1] for
// shg_upfir_planar_f32.  workspace (optional, caller owned) enables split-K for small grids; its size comes
// from shg_conv2d_workspace_bytes.
extern "C" int shg_conv2d_f32(const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W, int kh, int kw,
                              int mode, int pad, int wgroups, long wstride, const float* in_scale, const float* out_scale,
                              const float* bias, const float* noise, int noise_mode, float noise_strength, int act, float alpha,
                              float gain, float clamp, const float* residual, int out_mode, void* workspace, size_t ws_bytes,
                              void* stream) {
    ConvParams p;
    int rc = conv_fill(p, x, wt, y, NB, I, O, OP, H, W, kh, kw, mode, pad, wgroups, wstride, in_scale, out_scale, bias, noise,
                       noise_mode, noise_strength, act, alpha, gain, clamp, residual, out_mode);
    if (rc != SHG_OK) return rc;
    return conv_dispatch(p, kh * kw, p.S, mode == 2, workspace, ws_bytes, (hipStream_t)stream);
}

// Bytes of split-K workspace that shg_conv2d_f32 can use for this problem (0 = it will not split).
extern "C" size_t shg_conv2d_workspace_bytes(int NB, int I, int O, int H, int W, int kh, int kw, int mode, int pad, int wgroups) {
    if (mode == 2 || NB < 1 || I < 1 || O < 1) return 0;
    const int S = mode == 0 ? 1 : 2;
    const int OH = (H + 2 * pad - kh) / S + 1, OW = (W + 2 * pad - kw) / S + 1;
    if (OH < 1 || OW < 1) return 0;
    const bool narrow = O <= 64;
    const int KC = kh * kw == 9 ? 8 : 32;
    ConvPlan c = conv_plan(NB, I, O, OH, OW, narrow ? 64 : 128, narrow ? 256 : 128, KC, wgroups < 1 ? 1 : wgroups, true);
    return c.ksplit > 1 ? (size_t)c.ksplit * NB * O * OH * OW * sizeof(float) : 0;
}

// ------------------------------------------------------------------------------------------------
// Weight preparation: [O,I,KH,KW] -> GEMM layout [I][taps][OP] (o contiguous, zero padded to OP)
// ------------------------------------------------------------------------------------------------

// scale[o] = gain * rsqrt(mean_{i,k} w^2)  (stylegan.py:146) when demod, else gain.
__global__ __launch_bounds__(256) void weight_scale_kernel(const float* w, float* scale, int IK, int demod, float gain) {
    const int o = blockIdx.x;
    float acc = 0.f;
    if (demod) {
        for (int k = threadIdx.x; k < IK; k += 256) { const float v = w[(long)o * IK + k]; acc += v * v; }
        __shared__ float red[256];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        acc = red[0];
    }
    if (threadIdx.x == 0) scale[o] = demod ? gain * rsqrtf(acc / (float)IK) : gain;
}

// wt[(i*KK + t')*OP + o], t' = flip ? KK-1-t : t   (flip = w.flip([2,3]), conv2d_resample.py:32-33)
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* w, const float* scale, float* wt, int O, int I,
                                                               int KK, int OP, int flip) {
    __shared__ float tile[32][33];
    const int IK = I * KK;
    const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, k = k0 + tx;
        tile[r][tx] = (o < O && k < IK) ? w[(long)o * IK + k] * scale[o] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, o = o0 + tx;
        if (k >= IK || o >= OP) continue;
        const int i = k / KK, t = k - i * KK;
        const int tt = flip ? KK - 1 - t : t;
        wt[((long)i * KK + tt) * OP + o] = tile[tx][r];
    }
}

// wsq[i][o] = sum_t wt[i][t][o]^2   (for the demodulation coefficients, stylegan.py:155)
__global__ __launch_bounds__(256) void weight_sq_kernel(const float* wt, float* wsq, int I, int KK, int OP) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)I * OP) return;
    const int i = (int)(e / OP), o = (int)(e - (long)i * OP);
    float acc = 0.f;
    for (int t = 0; t < KK; ++t) { const float v = wt[((long)i * KK + t) * OP + o]; acc += v * v; }
    wsq[e] = acc;
}

// w: [O,I,KH,KW] fp32.  wt: [I*KK*OP] out.  wscale: [O] scratch/out.  wsq: [I*OP] out or null.
// demod=1 reproduces stylegan.py:146 (per-output-channel RMS normalisation) times `gain`;
// demod=0 multiplies by `gain` (conv2d_layer weight_gain, stylegan.py:227).
// One layout serves all three convolution modes (the transposed kernel indexes taps as ky*3+kx).
extern "C" int shg_conv_weight_prep_f32(const float* w, float* wt, float* wscale, float* wsq, int O, int I, int KH, int KW,
                                        int OP, int demod, float gain, int flip, void* stream) {
    SHG_CHECK_ARG(w && wt && wscale, "weight_prep: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && KH >= 1 && KW >= 1, "weight_prep: empty weight");
    SHG_CHECK_ARG(OP % 4 == 0 && OP >= O, "weight_prep: OP must be a multiple of 4 and >= O");
    hipStream_t s = (hipStream_t)stream;
    const int KK = KH * KW, IK = I * KK;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(O), dim3(256), 0, s, w, wscale, IK, demod, gain);
    SHG_CHECK_LAUNCH();
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(shg_cdiv(IK, 32), shg_cdiv(OP, 32)), dim3(256), 0, s, w, wscale, wt, O, I, KK,
                       OP, flip);
    SHG_CHECK_LAUNCH();
    if (wsq) {
        hipLaunchKernelGGL(weight_sq_kernel, dim3(shg_cdiv((long)I * OP, 256)), dim3(256), 0, s, wt, wsq, I, KK, OP);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
