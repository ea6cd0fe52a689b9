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
// halo patch), K = I*taps.  A "tap list" generalises stride-1 'same', stride-2 and the four
// sub-pixel phases of the stride-2 transposed convolution.
//
// Tiling: workgroup = WO x WP waves; each wave owns (MO*32) x (NP*32) outputs = MO*NP accumulators
// of 32x32 (16 VGPRs each).  K is consumed in chunks of KC input channels (KC*NTAPS k-values):
// weights [KC*NTAPS][BO] and the input patch [PATCH][KC+1] are staged in LDS, then every
// MFMA 32x32x2 consumes two channels of one tap (lanes 0-31: channel c, lanes 32-63: channel c+1).
// LDS reads are ds_read_b32, conflict-free (32 consecutive floats per half wave).
#include "shg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* x;          // [NB, I, H, W]
    const float* wt;         // prepped weights [wgroups][I][NTAPS][OP]
    float* y;                // [NB, O, OHt, OWt]
    const float* in_scale;   // [NB, I] or null
    const float* out_scale;  // [NB, O] or null
    const float* bias;       // [O] or null
    const float* noise;      // see noise_mode
    const float* residual;   // [NB, O, OHt, OWt] or null (added after the activation)
    int NB, I, O, OP;
    int H, W;
    int OHp, OWp;            // output grid computed by this launch (phase grid for transposed)
    int OHt, OWt;            // full output tensor extent
    int os, oa, ob;          // output position = (oy*os + oa, ox*os + ob)
    int S;                   // input stride
    int dy0, dx0;            // min tap offsets: patch origin = (oy0*S + dy0, ox0*S + dx0)
    int PH, PW, PATCH;       // patch rows/cols per image; PATCH = TN*PH*PW floats per channel
    int tw_log2, th_log2, tn_log2;
    int tiles_x, tiles_y, n_ptiles, n_otiles;
    int wgroups;             // weight set of slot b = wt + (b % wgroups) * wstride
    long wstride;
    int noise_mode;          // 0 none, 1 [OHt,OWt] shared over (n,o), 2 [NB,OHt,OWt]
    float noise_strength;
    int act;                 // 0: y*gain, 1: lrelu_agc(alpha, gain, clamp)
    float alpha, gain, clamp;
    int tap_off[9];          // LDS patch offset of tap t: (dy-dy0)*PW + (dx-dx0)
};

// Bijective XCD-aware remap (blocks b, b+8, ... share an XCD and its L2): every XCD walks a
// contiguous range of the o-tile-major work list, so its L2 holds one weight slice at a time.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int NTAPS, int KC, int MO, int NP, int WO, int WP>
__global__ __launch_bounds__(WO * WP * 64, 3) void conv_mfma_kernel(const ConvParams p) {
    constexpr int BO = MO * 32 * WO;
    constexpr int NT = WO * WP * 64;
    constexpr int ROWS = KC * NTAPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                      // [ROWS][BO]
    constexpr int XP = KC + 1;             // odd pitch: conflict-free for both the staging writes and B reads
    float* Xl = smem + ROWS * BO;          // [PATCH][XP]  (channel fastest -> the channel offset is an immediate)
    int* tab = (int*)(Xl + XP * p.PATCH);  // [PATCH] global pixel offset (without channel) or -1
    int* tabn = tab + p.PATCH;             // [PATCH] n*I for the in_scale lookup

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;

    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int otile = work / p.n_ptiles;
    const int ptile = work - otile * p.n_ptiles;
    const int txb = ptile % p.tiles_x;
    const int tyb = (ptile / p.tiles_x) % p.tiles_y;
    const int tnb = ptile / (p.tiles_x * p.tiles_y);
    const int TN = 1 << p.tn_log2;
    const int n0 = tnb << p.tn_log2, oy0 = tyb << p.th_log2, ox0 = txb << p.tw_log2;
    const int o0 = otile * BO;
    const int HW = p.H * p.W;

    // ---- patch table: where each patch element lives in global memory (or -1 = zero padding)
    const int PHW = p.PH * p.PW;
    for (int q = tid; q < p.PATCH; q += NT) {
        const int tn = q / PHW;
        const int rem = q - tn * PHW;
        const int py = rem / p.PW, px = rem - py * p.PW;
        const int n = n0 + tn;
        const int iy = oy0 * p.S + p.dy0 + py, ix = ox0 * p.S + p.dx0 + px;
        const bool ok = (n < p.NB) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        tab[q] = ok ? (n * p.I * HW + iy * p.W + ix) : -1;
        tabn[q] = n * p.I;
    }

    // ---- per-lane B-fragment offsets inside the patch (pixel j of the tile -> patch element)
    int boff[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        const int tx = j & ((1 << p.tw_log2) - 1);
        const int ty = (j >> p.tw_log2) & ((1 << p.th_log2) - 1);
        int tn = j >> (p.tw_log2 + p.th_log2);
        tn = tn < TN ? tn : 0;   // out-of-tile lanes read image 0 of the tile (results masked on store)
        boff[np] = (tn * PHW + ty * p.S * p.PW + tx * p.S) * XP + half;
    }
    const float* wa = Wl + wo * MO * 32 + l31 + half * NTAPS * BO;   // + compile-time (c2, t, mo) offsets

    f32x16 acc[MO][NP];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mo][np][r] = 0.f;

    const float* wbase = p.wt + (long)(p.wgroups > 1 ? (n0 % p.wgroups) : 0) * p.wstride;
    __syncthreads();   // tab visible

    for (int i0 = 0; i0 < p.I; i0 += KC) {
        // ---- stage weights: rows (i0*NTAPS + r) of the [I*NTAPS][OP] matrix, columns o0..o0+BO
        {
            constexpr int V4 = ROWS * BO / 4;
            constexpr int PER = (V4 + NT - 1) / NT;
            constexpr int WB = PER < 5 ? PER : 5;      // float4 loads in flight per lane
#pragma unroll 1
            for (int kb = 0; kb < PER; kb += WB) {
                float4 wv[WB];
#pragma unroll
                for (int k = 0; k < WB; ++k) {
                    const int e = tid + (kb + k) * NT;
                    const int row = e / (BO / 4), c4 = e - row * (BO / 4);
                    const int o = o0 + c4 * 4;
                    const long grow = (long)i0 * NTAPS + row;
                    wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < V4 && grow < (long)p.I * NTAPS && o < p.OP)
                        wv[k] = *reinterpret_cast<const float4*>(wbase + grow * p.OP + o);
                }
#pragma unroll
                for (int k = 0; k < WB; ++k) {
                    const int e = tid + (kb + k) * NT;
                    if (e < V4) *reinterpret_cast<float4*>(Wl + e * 4) = wv[k];
                }
            }
            // ---- stage the input patch (scaled by the per-sample style when modulated),
            //      XB channels in flight per lane
            constexpr int XB = KC < 8 ? KC : 8;
            for (int q = tid; q < p.PATCH; q += NT) {
                const int off = tab[q];
                const int sn = tabn[q];
#pragma unroll 1
                for (int icb = 0; icb < KC; icb += XB) {
                    float xv[XB];
#pragma unroll
                    for (int ic = 0; ic < XB; ++ic) {
                        const int i = i0 + icb + ic;
                        float v = 0.f;
                        if (off >= 0 && i < p.I) {
                            v = p.x[off + i * HW];
                            if (p.in_scale) v *= p.in_scale[sn + i];
                        }
                        xv[ic] = v;
                    }
#pragma unroll
                    for (int ic = 0; ic < XB; ++ic) Xl[q * XP + icb + ic] = xv[ic];
                }
            }
        }
        __syncthreads();

        // ---- MFMA: every instruction covers 2 input channels of one tap
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int toff = p.tap_off[t] * XP;
            const float* xb[NP];
#pragma unroll
            for (int np = 0; np < NP; ++np) xb[np] = Xl + boff[np] + toff;
#pragma unroll
            for (int c2 = 0; c2 < KC / 2; ++c2) {
                float a[MO], b[NP];
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) a[mo] = wa[((c2 * 2) * NTAPS + t) * BO + mo * 32];
#pragma unroll
                for (int np = 0; np < NP; ++np) b[np] = xb[np][c2 * 2];
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                    for (int np = 0; np < NP; ++np)
                        acc[mo][np] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mo], b[np], acc[mo][np], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = out channel][col = pixel]; row = (r&3) + 8*(r>>2) + 4*half
    const int OHWt = p.OHt * p.OWt;
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        const int tx = j & ((1 << p.tw_log2) - 1);
        const int ty = (j >> p.tw_log2) & ((1 << p.th_log2) - 1);
        const int tn = j >> (p.tw_log2 + p.th_log2);
        const int n = n0 + tn, oy = oy0 + ty, ox = ox0 + tx;
        if (tn >= TN || n >= p.NB || oy >= p.OHp || ox >= p.OWp) continue;
        const int pix = (oy * p.os + p.oa) * p.OWt + ox * p.os + p.ob;
        float nz = 0.f;
        if (p.noise_mode == 1) nz = p.noise[pix] * p.noise_strength;
        else if (p.noise_mode == 2) nz = p.noise[n * OHWt + pix] * p.noise_strength;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + (wo * MO + mo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                float v = acc[mo][np][r];
                if (p.out_scale) v *= p.out_scale[n * p.O + o];
                v += nz;
                if (p.bias) v += p.bias[o];
                v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
                const long idx = ((long)n * p.O + o) * OHWt + pix;
                if (p.residual) v += p.residual[idx];
                p.y[idx] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host launcher
// ------------------------------------------------------------------------------------------------

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int NTAPS, int KC, int MO, int NP, int WO, int WP>
static int launch_conv(ConvParams& p, hipStream_t s) {
    constexpr int BO = MO * 32 * WO, BP = NP * 32 * WP;
    // pixel tile: TW x TH x TN with TW*TH*TN == BP (all powers of two)
    int tw = 32; while (tw > 1 && tw / 2 >= p.OWp) tw >>= 1;
    int th = BP / tw; while (th > 1 && th / 2 >= p.OHp) th >>= 1;
    int tn = BP / (tw * th);
    if (p.wgroups > 1) tn = 1;                    // per-slot weights: one image per tile
    while (tn > 1 && tn / 2 >= p.NB) tn >>= 1;
    p.tw_log2 = ilog2(tw); p.th_log2 = ilog2(th); p.tn_log2 = ilog2(tn);
    p.tiles_x = shg_cdiv(p.OWp, tw);
    p.tiles_y = shg_cdiv(p.OHp, th);
    const int tiles_n = shg_cdiv(p.NB, tn);
    p.n_ptiles = p.tiles_x * p.tiles_y * tiles_n;
    p.n_otiles = shg_cdiv(p.O, BO);
    // patch extents were stored as tap spans by the caller: PH/PW currently hold (dymax-dy0+1)
    const int span_y = p.PH, span_x = p.PW;
    p.PH = (th - 1) * p.S + span_y;
    p.PW = (tw - 1) * p.S + span_x;
    p.PATCH = tn * p.PH * p.PW;
    for (int t = 0; t < NTAPS; ++t) {
        // tap_off currently holds packed (dy-dy0, dx-dx0) as dyr*64 + dxr
        const int dyr = p.tap_off[t] >> 6, dxr = p.tap_off[t] & 63;
        p.tap_off[t] = dyr * p.PW + dxr;
    }
    const size_t lds = sizeof(float) * ((size_t)KC * NTAPS * BO + (size_t)(KC + 1) * p.PATCH) + sizeof(int) * 2 * (size_t)p.PATCH;
    if (lds > 160 * 1024) { shg_set_error("conv: LDS request %zu exceeds 160 KiB", lds); return SHG_ERR_UNSUPPORTED; }
    auto kern = conv_mfma_kernel<NTAPS, KC, MO, NP, WO, WP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { shg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return SHG_ERR_LAUNCH; }
    }
    const int grid = p.n_ptiles * p.n_otiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WO * WP * 64), lds, s, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

template <int NTAPS, int KC>
static int launch_conv_tile(ConvParams& p, hipStream_t s) {
    if (p.O <= 64) return launch_conv<NTAPS, KC, 2, 2, 1, 4>(p, s);   // 64 x 256 tile
    return launch_conv<NTAPS, KC, 2, 2, 2, 2>(p, s);                    // 128 x 128 tile
}

// mode: 0 = stride 1, symmetric padding `pad`        (conv2d_resample.py:145-147)
//       1 = stride 2, symmetric padding `pad`        (conv2d_resample.py:116-120, strided conv)
//       2 = transposed stride 2, padding 0, output (2H+kh-2)... i.e. (H-1)*2+kh  (conv2d_resample.py:122-137)
// wt must come from shg_conv_weight_prep_f32 with the matching layout (0 for modes 0/1, 1 for mode 2).
extern "C" int shg_conv2d_f32(const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W,
                              int kh, int kw, int mode, int pad, int wgroups, long wstride,
                              const float* in_scale, const float* out_scale, const float* bias,
                              const float* noise, int noise_mode, float noise_strength, int act, float alpha,
                              float gain, float clamp, const float* residual, void* stream) {
    SHG_CHECK_ARG(x && wt && y, "conv2d: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1, "conv2d: empty tensor");
    SHG_CHECK_ARG((kh == 3 && kw == 3) || (kh == 1 && kw == 1), "conv2d: only 3x3 and 1x1 kernels (got %dx%d)", kh, kw);
    SHG_CHECK_ARG(OP % 4 == 0 && OP >= O, "conv2d: OP must be a multiple of 4 and >= O");
    SHG_CHECK_ARG(mode >= 0 && mode <= 2, "conv2d: bad mode %d", mode);
    SHG_CHECK_ARG(mode != 2 || (kh == 3 && pad == 0), "conv2d: transposed mode supports 3x3, padding 0");
    SHG_CHECK_ARG(pad >= 0 && pad < 32, "conv2d: bad padding");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L, "conv2d: x is too large");
    hipStream_t s = (hipStream_t)stream;
    ConvParams p{};
    p.x = x; p.wt = wt; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.bias = bias;
    p.noise = noise_mode ? noise : nullptr; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    p.wgroups = wgroups < 1 ? 1 : wgroups; p.wstride = wstride;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    const int K = kh * kw;
    if (mode == 0 || mode == 1) {
        const int S = mode == 0 ? 1 : 2;
        const int OH = (H + 2 * pad - kh) / S + 1, OW = (W + 2 * pad - kw) / S + 1;
        SHG_CHECK_ARG(OH >= 1 && OW >= 1, "conv2d: output must be at least 1x1");
        SHG_CHECK_ARG((long)NB * O * OH * OW < 2147483647L, "conv2d: y is too large");
        p.OHp = p.OHt = OH; p.OWp = p.OWt = OW; p.os = 1; p.oa = 0; p.ob = 0; p.S = S;
        p.dy0 = -pad; p.dx0 = -pad; p.PH = kh; p.PW = kw;
        for (int t = 0; t < K; ++t) p.tap_off[t] = ((t / kw) << 6) | (t % kw);
        if (K == 9) return launch_conv_tile<9, 8>(p, s);
        return launch_conv_tile<1, 64>(p, s);
    }
    // transposed stride 2: out[Y,X] = sum_{ky,kx: (Y-ky),(X-kx) even} x[(Y-ky)/2,(X-kx)/2] * W[ky,kx]
    // phase (a,b) = (Y&1, X&1): a==0 -> ky in {0,2} reading rows u, u-1 ; a==1 -> ky = 1 reading row u.
    const int OH = 2 * H + 1, OW = 2 * W + 1;
    SHG_CHECK_ARG((long)NB * O * OH * OW < 2147483647L, "conv2d: y is too large");
    p.OHt = OH; p.OWt = OW; p.os = 2; p.S = 1;
    long woff = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            ConvParams q = p;
            q.wt = wt + woff;
            q.oa = a; q.ob = b;
            q.OHp = a == 0 ? H + 1 : H;
            q.OWp = b == 0 ? W + 1 : W;
            const int nty = a == 0 ? 2 : 1, ntx = b == 0 ? 2 : 1;
            q.dy0 = a == 0 ? -1 : 0; q.dx0 = b == 0 ? -1 : 0;
            q.PH = nty; q.PW = ntx;   // tap spans
            // tap order must match the weight-prep layout: ky ascending, kx ascending
            int t = 0;
            for (int iy = 0; iy < nty; ++iy)
                for (int ix = 0; ix < ntx; ++ix) {
                    // ky = 0 -> dy = 0 (row u), ky = 2 -> dy = -1 (row u-1); ky = 1 -> dy = 0
                    const int dy = a == 0 ? (iy == 0 ? 0 : -1) : 0;
                    const int dx = b == 0 ? (ix == 0 ? 0 : -1) : 0;
                    q.tap_off[t++] = ((dy - q.dy0) << 6) | (dx - q.dx0);
                }
            const int ntaps = nty * ntx;
            int rc;
            if (ntaps == 4) rc = launch_conv_tile<4, 16>(q, s);
            else if (ntaps == 2) rc = launch_conv_tile<2, 32>(q, s);
            else rc = launch_conv_tile<1, 64>(q, s);
            if (rc != SHG_OK) return rc;
            woff += (long)I * ntaps * OP;
        }
    return SHG_OK;
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

// layout 0: wt[(i*KK + t')*OP + o], t' = flip ? KK-1-t : t
// layout 1 (transposed stride-2 phases, 3x3 only): four blocks [(a,b)][i][t_local][OP], block sizes I*{4,2,2,1}*OP
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* w, const float* scale, float* wt, int O, int I,
                                                               int KK, int OP, int layout, int flip) {
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
        long dst;
        if (layout == 0) {
            const int tt = flip ? KK - 1 - t : t;
            dst = ((long)i * KK + tt) * OP + o;
        } else {
            int ky = t / 3, kx = t - ky * 3;
            if (flip) { ky = 2 - ky; kx = 2 - kx; }   // w.flip([2,3]) keeps every tap in its parity class
            const int a = ky == 1, b = kx == 1;
            const int nty = a ? 1 : 2, ntx = b ? 1 : 2;
            const int ly = a ? 0 : (ky >> 1), lx = b ? 0 : (kx >> 1);
            const int tl = ly * ntx + lx;
            // block offsets in taps: (0,0):0, (0,1):4, (1,0):6, (1,1):8
            const int boff = a == 0 ? (b == 0 ? 0 : 4) : (b == 0 ? 6 : 8);
            dst = ((long)boff * I + (long)i * (nty * ntx) + tl) * OP + o;
        }
        wt[dst] = tile[tx][r];
    }
}

// wsq[i][o] = sum_t wt[i][t][o]^2   (for the demodulation coefficients, stylegan.py:155)
__global__ __launch_bounds__(256) void weight_sq_kernel(const float* wt, float* wsq, int I, int KK, int OP, int layout) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)I * OP) return;
    const int i = (int)(e / OP), o = (int)(e - (long)i * OP);
    float acc = 0.f;
    if (layout == 0) {
        for (int t = 0; t < KK; ++t) { const float v = wt[((long)i * KK + t) * OP + o]; acc += v * v; }
    } else {
        const int nt[4] = {4, 2, 2, 1}, bo[4] = {0, 4, 6, 8};
        for (int ph = 0; ph < 4; ++ph)
            for (int t = 0; t < nt[ph]; ++t) {
                const float v = wt[((long)bo[ph] * I + (long)i * nt[ph] + t) * OP + o];
                acc += v * v;
            }
    }
    wsq[e] = acc;
}

// w: [O,I,KH,KW] fp32.  wt: [I*KK*OP] out.  wscale: [O] scratch/out.  wsq: [I*OP] out or null.
// demod=1 reproduces stylegan.py:146 (per-output-channel RMS normalisation) times `gain`;
// demod=0 multiplies by `gain` (conv2d_layer weight_gain, stylegan.py:227).
extern "C" int shg_conv_weight_prep_f32(const float* w, float* wt, float* wscale, float* wsq, int O, int I, int KH, int KW,
                                        int OP, int demod, float gain, int layout, int flip, void* stream) {
    SHG_CHECK_ARG(w && wt && wscale, "weight_prep: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && KH >= 1 && KW >= 1, "weight_prep: empty weight");
    SHG_CHECK_ARG(OP % 4 == 0 && OP >= O, "weight_prep: OP must be a multiple of 4 and >= O");
    SHG_CHECK_ARG(layout == 0 || (layout == 1 && KH == 3 && KW == 3), "weight_prep: layout 1 needs a 3x3 kernel");
    hipStream_t s = (hipStream_t)stream;
    const int KK = KH * KW, IK = I * KK;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(O), dim3(256), 0, s, w, wscale, IK, demod, gain);
    SHG_CHECK_LAUNCH();
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(shg_cdiv(IK, 32), shg_cdiv(OP, 32)), dim3(256), 0, s, w, wscale, wt, O, I, KK,
                       OP, layout, flip);
    SHG_CHECK_LAUNCH();
    if (wsq) {
        hipLaunchKernelGGL(weight_sq_kernel, dim3(shg_cdiv((long)I * OP, 256)), dim3(256), 0, s, wt, wsq, I, KK, OP, layout);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
