// fp16 route of the training / config-5 rows (reference: the `use_fp16` branches -- stylegan.py:136-138,486,660-667,
// comodgan.py:40-47,305 -- whose convolutions are cuDNN half kernels with fp32 accumulation, and upfirdn2d.cu's half
// instantiation, upfirdn2d.cpp:59).  gfx950 only.
//
// Data layout: fp16 activations live in HBM channels-LAST ([N,H,W,C], torch.channels_last on a [N,C,H,W] tensor): the
// v_mfma_f32_32x32x16_f16 operands are 8 consecutive K values per lane, and with K = input channels that is ONE 16-byte load
// per lane from an NHWC tensor (an NCHW tensor would need a transposing gather for every operand).  Weights are prepared as
// [tap][O][I] (I fastest), accumulation is fp32, results are rounded to fp16 once.
//
//   conv_f16_kernel      y[n,oy,ox,o] = sum_t sum_i w[t][o][i] * x[n, oy'*s_in + dy_t, ox'*s_in + dx_t, i]  with the output pixel
//                        (oy,ox) = (oy'*s_out + oy0, ox'*s_out + ox0): stride-1 / stride-2 convolutions (s_out = 1) and, launched once
//                        per sub-pixel phase, the stride-2 transposed convolution (s_out = 2, each phase has its own 1 / 2 / 4 taps:
//                        no multiplies by inserted zeros).  Implicit GEMM: M = output channels (A = weights, global/L2 -> registers),
//                        N = a tile of 8x16 output pixels (B = the input patch of a 32-channel chunk in LDS, one ds_read_b128 per
//                        operand, pixel stride 80 B = conflict-free), K = taps x channels.
//   conv_wgrad_f16       dw[t][o][i] = sum_{n,oy,ox} g[n,oy,ox,o] * x[n, oy*s + dy_t, ox*s + dx_t, i]: K = pixels, so an operand is 8
//                        consecutive pixels of ONE channel -- the NHWC tiles are transposed while they are staged ([channel][pixel]
//                        LDS images) and every operand is an aligned ds_read_b128; the three kx taps of a row share their reads
//                        (shift by one half = v_alignbit, by two = registers; stride 2: even / odd column halves); 64 x 64 (o,i) tile
//                        per workgroup, all taps per wave, split over pixel slices with fp32 partials + a fixed-order reduction
//                        (deterministic).
//   upfirdn2d_f16        the generic gather of upfirdn2d.cu:29-92 on NHWC halves, 8 channels per lane, fp32 accumulation.
//   bias_act_f16 (+bwd)  x + bias[c] -> lrelu_agc, and its gradient from the saved output (common/utils.py:135-143).
#include <type_traits>
#include "shg_common.h"
#include "conv_f16_p.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

#ifndef SHG_F16_ADEPTH
#define SHG_F16_ADEPTH 1
#endif
#ifndef SHG_F16_EG
#define SHG_F16_EG 4
#endif
#ifndef SHG_F16_ABL
#define SHG_F16_ABL 0   // timing-study builds only (tools/_variants, tools/f16_abl.sh): 1 no B reads, 2 no A loads, 4 no staging, 8 no epilogue, 16 no loop barriers
#endif

#ifdef SHG_F16_TRACE
// timeline study (python sh-gan_amd/build.py --variant=f16trace -DSHG_F16_TRACE=1, tools/f16_trace.py): every 61st workgroup of channel group 0
// records clock64() of wave 0 at: entry | first chunk in LDS | first chunk multiplied | all chunks multiplied | stores issued | stores drained
__device__ long long shg_f16_trace_buf[256 * 8];
extern "C" int shg_f16_trace_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(shg_f16_trace_buf), sizeof(shg_f16_trace_buf));
}
#define F16_TRACE(slot) do { if (blockIdx.y == 0 && blockIdx.x % 61 == 0 && blockIdx.x / 61 < 256 && threadIdx.x == 0) shg_f16_trace_buf[(blockIdx.x / 61) * 8 + (slot)] = clock64(); } while (0)
#else
#define F16_TRACE(slot) do { } while (0)
#endif

namespace f16 {

constexpr int TH = 8, TW = 16;       // output-pixel tile of a workgroup (4 waves x 2 rows x 16 columns)
constexpr int KC = 32;               // channels per LDS chunk (two MFMA k-steps)
constexpr int PSTR = 40;             // halves per staged pixel: 32 + 8 padding -> 80-byte stride, conflict-free ds_read_b128


template <int MB, int NT, int NB, bool WLDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MB == 2 && NB == 2 && !WLDS) ? 3 : 1))) void conv_f16_kernel(const ConvP p) {
    // tile = TH x (16 NB) output pixels: wave w owns rows 2w, 2w+1; with NB = 1 its 32 lanes-of-a-block are 2 rows x 16 columns, with
    // NB = 2 block nb is row 2w + nb and the lanes are its 32 columns -- every weight operand then feeds two MFMAs
    extern __shared__ __attribute__((aligned(16))) _Float16 patch[];
    constexpr int TWK = TW * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kg = lane >> 5;
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y, n = tile / p.tiles_y;
    const int ob0 = blockIdx.y * MB;                                                   // first 32-channel block of this workgroup
    const int ry = NB == 1 ? wave * 2 + (j >> 4) : wave * 2, rx = NB == 1 ? (j & 15) : j;   // lane's pixel inside the tile (block 0)
    const int iy_base = ty * TH * p.s_in + p.org_y, ix_base = tx * TWK * p.s_in + p.org_x;
    f16x acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][b][r] = 0.f;
    const int npix = p.PH * p.PW;
    const _Float16* xn = p.x + (long)n * p.H * p.W * p.I;
    const int c16n = p.I >> 4;
    // weights in MFMA operand order: [32-channel block][tap slot][16-channel k-step][lane][8] -- one coalesced 1 KiB load per operand
    const _Float16* wl = p.w + (long)lane * 8;
    // Patch staging, software-pipelined: 4 lanes x 16 bytes per pixel; the pixel -> address map does not depend on the channel chunk,
    // so it is computed once; the loads of chunk c+1 are issued BEFORE the multiply loop of chunk c and land in LDS after it.
    constexpr int SIT = NB == 2 ? 6 : 9;                            // ceil(max patch pixels / 64): 10 x 34 (NB = 2), 17 x 33 (stride 2)
    int goff[SIT];
    const int q8 = (tid & 3) * 8;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int pp = (tid >> 2) + it * 64, py = pp / p.PW, px = pp - py * p.PW;
        const int iy = iy_base + py, ix = ix_base + px;
        goff[it] = (pp < npix && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
    }
    h8 stage[SIT];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (goff[it] >= 0) v = *(const h8*)(xn + (long)goff[it] * p.I + c0 + q8);
            stage[it] = v;
        }
        if (p.in_scale) {                                           // `x * styles.to(x.dtype)` (stylegan.py:173): half x half -> half
            const float* sp = p.in_scale + (long)n * p.I + c0 + q8;
            _Float16 sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) sh[k] = (_Float16)sp[k];
#pragma unroll
            for (int it = 0; it < SIT; ++it)
#pragma unroll
                for (int k = 0; k < 8; ++k) stage[it][k] = stage[it][k] * sh[k];
        }
    };
    // WLDS: the weight slab of a chunk ([MB blocks][NT taps][2 k-steps] pieces of 1 KiB, already in operand order) is staged ONCE per
    // workgroup through the same register pipeline and read from LDS lane-linearly (conflict-free) -- instead of every wave loading every
    // operand from L2 one tap ahead (half the wave cycles parked on those loads, profiles/r03_f16_pmc_summary.txt)
    constexpr int WPC = MB * NT * 2, WIT = WLDS ? (WPC + 3) / 4 : 1;      // pieces per chunk; a pass of the 256 threads moves 4 pieces
    _Float16* wlds = patch + p.wlds_off;
    h8 wstage[WIT];
    auto fetch_w = [&](int c0) __attribute__((always_inline)) {
        if constexpr (WLDS) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pi = it * 4 + wave;                          // piece = (m * NT + t) * 2 + ks
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (pi < WPC) {
                    const int ks = pi & 1, mt = pi >> 1, m = mt / NT, t = mt - m * NT;
                    v = *(const h8*)(wl + ((((long)(ob0 + m) * p.wslots + p.tw[t]) * c16n + (c0 >> 4) + ks) << 9));
                }
                wstage[it] = v;
            }
        }
    };
    F16_TRACE(0);
    fetch(0);
    fetch_w(0);
    for (int c0 = 0; c0 < p.I; c0 += KC) {
        if (!(SHG_F16_ABL & 16)) __syncthreads();
        if (!(SHG_F16_ABL & 4) || c0 == 0)
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int pp = (tid >> 2) + it * 64;
            if (pp < npix) *(h8*)(patch + pp * PSTR + q8) = stage[it];
        }
        if constexpr (WLDS) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pi = it * 4 + wave;
                if (pi < WPC) *(h8*)(wlds + pi * 512 + lane * 8) = wstage[it];
            }
        }
        if (!(SHG_F16_ABL & 16)) __syncthreads();
        if (c0 == 0) F16_TRACE(1);
        if (c0 + KC < p.I && !(SHG_F16_ABL & 4)) { fetch(c0 + KC); fetch_w(c0 + KC); }
        // Weight operands: unconditional loads (the packed tensor is zero-padded to whole MB groups of blocks and I % 32 == 0), those of
        // tap t+1 requested before the MFMAs of tap t -- a conditional load would be waited for on the spot (vmcnt(0) per MFMA pair).
        const _Float16* wc = wl + ((long)ob0 * p.wslots * c16n + (c0 >> 4)) * 512;
        constexpr int AD = SHG_F16_ADEPTH;                          // operand prefetch distance in taps (ring of AD + 1 slots)
        h8 a[AD + 1][2][MB];
        if constexpr (!WLDS) {
#pragma unroll
            for (int t = 0; t < AD && t < NT; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < MB; ++m) a[t][ks][m] = *(const h8*)(wc + (((long)m * p.wslots + p.tw[t]) * c16n + ks) * 512);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (WLDS) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < MB; ++m) a[t % (AD + 1)][ks][m] = *(const h8*)(wlds + ((m * NT + t) * 2 + ks) * 512 + lane * 8);
            } else if (t + AD < NT && !((SHG_F16_ABL & 2) && c0)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < MB; ++m) a[(t + AD) % (AD + 1)][ks][m] = *(const h8*)(wc + (((long)m * p.wslots + p.tw[t + AD]) * c16n + ks) * 512);
            }
            // the compiler otherwise sinks the operand requests to just before their use (measured: 3-8 % on the 128 / 256-channel layers)
            __builtin_amdgcn_sched_barrier(0);
            const _Float16* bp = patch + ((ry * p.s_in + p.tdy[t]) * p.PW + rx * p.s_in + p.tdx[t]) * PSTR + kg * 8;
            h8 b[2][NB];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < NB; ++q) b[ks][q] = (SHG_F16_ABL & 1) ? a[0][ks][0] : *(const h8*)(bp + q * p.s_in * p.PW * PSTR + ks * 16);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int q = 0; q < NB; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t % (AD + 1)][ks][m], b[ks][q], acc[m][q], 0, 0, 0);
        }
        if (c0 == 0) F16_TRACE(2);
    }
    F16_TRACE(3);
    // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  A lane holds 4-channel runs of ONE
    // pixel: stored directly, every store instruction would touch 64 different 128-byte lines with 8 bytes each.  The tile is therefore
    // transposed through LDS (pixel-major, 16 bytes of padding per pixel) and leaves as whole 16-byte pieces of contiguous channel runs.
    constexpr int OPS = MB * 32 + 8;
    if (SHG_F16_ABL & 8) {
        float sacc = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[m][q][r];
        if (sacc != 1234.5f) return;
    }
    float bv[MB][4][4];                                             // bias of a lane's channels (no-tail launches), requested before the barrier
    const bool pre_bias = p.bias && !p.tail;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = ob0 * 32 + m * 32 + qq * 8 + kg * 4 + e;
                bv[m][qq][e] = pre_bias ? p.bias[o < p.O ? o : p.O - 1] : 0.f;
            }
    __syncthreads();
    F16_TRACE(6);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int pl = NB == 1 ? ry * TW + rx : (ry + q) * TWK + rx;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int ol = m * 32 + qq * 8 + kg * 4;
                h4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[m][q][qq * 4 + e] + bv[m][qq][e]);
                *(h4*)(patch + pl * OPS + ol) = v;
            }
    }
    __syncthreads();
    F16_TRACE(7);
    constexpr int PCS = MB * 4;                                   // 16-byte pieces per pixel
    if ((p.O & 7) == 0) {
        // whole 8-channel pieces: all LDS reads first, every operand load unconditional (clamped address), only the store predicated -- a
        // rolled loop with early exits spent 550 cycles per piece (tools/f16_trace.py: LDS latency + address arithmetic in series)
        constexpr int EIT = TH * TWK * PCS / 256, EG = EIT < SHG_F16_EG ? EIT : SHG_F16_EG;    // pieces per thread, in groups of EG (registers)
        for (int i0 = 0; i0 < EIT; i0 += EG) {
        h8 vv[EG];
#pragma unroll
        for (int it = 0; it < EG; ++it) {
            const int e = tid + (i0 + it) * 256, pl = e / PCS, pc = e - pl * PCS;
            vv[it] = *(const h8*)(patch + pl * OPS + pc * 8);
        }
#pragma unroll
        for (int it = 0; it < EG; ++it) {
            const int e = tid + (i0 + it) * 256, pl = e / PCS, pc = e - pl * PCS, row = pl / TWK, col = pl - row * TWK;
            const int gy = ty * TH + row, gx = tx * TWK + col;
            const int oy = gy * p.s_out + p.oy0, ox = gx * p.s_out + p.ox0, o = ob0 * 32 + pc * 8;
            const bool ok = gy < p.GH && gx < p.GW && oy >= 0 && oy < p.OHt && ox >= 0 && ox < p.OWt && o < p.O;
            const int oyc = oy < 0 ? 0 : (oy < p.OHt ? oy : p.OHt - 1), oxc = ox < 0 ? 0 : (ox < p.OWt ? ox : p.OWt - 1), oc = o < p.O ? o : p.O - 8;
            const long pix = (long)oyc * p.OWt + oxc, off = ((long)n * p.OHt * p.OWt + pix) * p.O + oc;
            h8 v = vv[it];
            if (p.tail) {
                const float nz = p.noise_mode == 0 ? 0.f : p.noise[(p.noise_mode == 2 ? (long)n * p.OHt * p.OWt : 0) + pix] * p.noise_strength;
                const float4 d0 = p.out_scale ? *(const float4*)(p.out_scale + (long)n * p.O + oc) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 d1 = p.out_scale ? *(const float4*)(p.out_scale + (long)n * p.O + oc + 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 b0 = p.bias ? *(const float4*)(p.bias + oc) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 b1 = p.bias ? *(const float4*)(p.bias + oc + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                h8 rs = {0, 0, 0, 0, 0, 0, 0, 0};
                if (p.residual) rs = *(const h8*)(p.residual + off);
                const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float z = __builtin_fmaf((float)v[k], dd[k], nz) + bb[k];     // (spelled out: conv_f16_ring.hip's tail is the same expression)
                    z = p.act ? shg_lrelu_agc(z, p.alpha, p.gain, p.clamp) : z * p.gain;
                    v[k] = (_Float16)((float)(_Float16)z + (float)rs[k]);
                }
            }
            if (ok) *(h8*)(p.y + off) = v;
        }
        }
    } else
    for (int e = tid; e < TH * TWK * PCS; e += 256) {
        const int pl = e / PCS, pc = e - pl * PCS, row = pl / TWK, col = pl - row * TWK;
        const int gy = ty * TH + row, gx = tx * TWK + col;
        if (gy >= p.GH || gx >= p.GW) continue;
        const int oy = gy * p.s_out + p.oy0, ox = gx * p.s_out + p.ox0;
        if (oy < 0 || oy >= p.OHt || ox < 0 || ox >= p.OWt) continue;
        const int o = ob0 * 32 + pc * 8;
        _Float16* yp = p.y + (((long)n * p.OHt + oy) * p.OWt + ox) * p.O + o;
        h8 v = *(const h8*)(patch + pl * OPS + pc * 8);
        if (p.tail) {                                               // the layer tail on the half-rounded convolution result, as the reference applies it
            const long pix = (long)oy * p.OWt + ox;
            const float nz = p.noise_mode == 0 ? 0.f : p.noise[(p.noise_mode == 2 ? (long)n * p.OHt * p.OWt : 0) + pix] * p.noise_strength;
            const bool full = (p.O & 7) == 0 && o + 7 < p.O;
            float dd[8], bb[8];
            h8 rs = {0, 0, 0, 0, 0, 0, 0, 0};
            if (full) {                                             // 8 consecutive channels: vector loads of the per-channel operands
                const float4 d0 = p.out_scale ? *(const float4*)(p.out_scale + (long)n * p.O + o) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 d1 = p.out_scale ? *(const float4*)(p.out_scale + (long)n * p.O + o + 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 b0 = p.bias ? *(const float4*)(p.bias + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 b1 = p.bias ? *(const float4*)(p.bias + o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                dd[0] = d0.x; dd[1] = d0.y; dd[2] = d0.z; dd[3] = d0.w; dd[4] = d1.x; dd[5] = d1.y; dd[6] = d1.z; dd[7] = d1.w;
                bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
                if (p.residual) rs = *(const h8*)(p.residual + (yp - p.y));
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int oc = o + k < p.O ? o + k : p.O - 1;
                    dd[k] = p.out_scale ? p.out_scale[(long)n * p.O + oc] : 1.f;
                    bb[k] = p.bias ? p.bias[oc] : 0.f;
                    if (p.residual && o + k < p.O) rs[k] = p.residual[(yp - p.y) + k];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float z = __builtin_fmaf((float)v[k], dd[k], nz) + bb[k];     // (spelled out: conv_f16_ring.hip's tail is the same expression)
                z = p.act ? shg_lrelu_agc(z, p.alpha, p.gain, p.clamp) : z * p.gain;
                v[k] = (_Float16)((float)(_Float16)z + (float)rs[k]);
            }
        }
        if ((p.O & 7) == 0 && o + 7 < p.O) *(h8*)yp = v;
        else
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (o + k < p.O) yp[k] = v[k];
    }
#ifdef SHG_F16_TRACE
    F16_TRACE(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F16_TRACE(5);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int WR = 4;                // output rows per staged block (the 3x3 stride-1 kernel stages 8: wgrad_rows)
constexpr int wgrad_rows(int k, int s) { return (k == 3 && s == 1) ? 8 : WR; }
constexpr int WC = 16;               // output columns per staged block = one MFMA k-step

struct WgradP {
    const _Float16* x;               // [N,H,W,I]
    const _Float16* g;               // [N,OH,OW,O]
    float* part;                     // [slices][NT][OP][IP] fp32 partial sums (OP, IP = O, I rounded up to 64)
    int N, I, O, H, W, OH, OW;
    int s, pad, k;                   // stride, padding, kernel size (1 or 3)
    int bx, by;                      // blocks per image along x / y
    long nblocks;                    // N * by * bx
    int slices, OP, IP;
    int XR, XC;                      // staged input rows / columns per block
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS images are TRANSPOSED while staging ([channel][pixel], 2-byte scatter writes of the 16-byte NHWC loads), so that an MFMA operand --
// 8 consecutive pixels of one channel -- is ONE aligned ds_read_b128 instead of eight 2-byte gathers:
//   gT[64 o][4 rows x 16 px (+8)]                       A operand of row r, k-group kg: gT[o][r*16 + kg*8 ..]
//   xT[64 i][XR rows][RP (+8 per channel)]              stride 1: RP = 24, columns as they come (18 used); the three kx taps of a row are
//                                                        the aligned 8 columns, the same shifted by one half (v_alignbit) and by two (registers)
//                                                        from one b128 + one b32 read;  stride 2: RP = 48 = even columns | odd columns,
//                                                        kx = 0 / 2 from the even half (aligned / shifted by one), kx = 1 from the odd half.
// Channel pitches (144 / 304 / 880 / 208 bytes) put the 16 lanes of a b128 group on 16 different 16-byte slots: conflict-free.
template <int K, int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_wgrad_f16_kernel(const WgradP p) {
    constexpr int NT = K * K, WR = wgrad_rows(K, S);         // (8 rows at 3x3 stride 1: 72 MFMAs per wave between two barriers instead of 36)
    constexpr int XR = (WR - 1) * S + K, XC = (WC - 1) * S + K, RP = S == 1 ? 24 : 48, GP = WR * WC + 8, XP = XR * RP + 8;
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    _Float16* gT = sm;                               // [64][GP]
    _Float16* xT = sm + 64 * GP;                     // [64][XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kg = lane >> 5;
    const int it = blockIdx.x % (p.IP / 64), ot = blockIdx.x / (p.IP / 64);
    const int wo = (wave >> 1) * 32, wi = (wave & 1) * 32;                 // this wave's 32 x 32 corner of the 64 x 64 tile
    f16x acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging is software-pipelined: the NHWC loads of block b+1 are issued before the multiply loop of block b and are written
    // (transposed) into LDS after it.  A thread stages FOUR ADJACENT PIXELS of one 8-channel group: per channel they are 8 contiguous bytes
    // of the [channel][pixel] image -- one ds_write_b64 (stride 2: the even and the odd pair, two ds_write_b32) instead of four 2-byte
    // scatter writes (48 ds_write_b16 per thread and block kept the LDS pipe busier than the matrix pipe: bank-conflict rate 0.54,
    // profiles/r05_f16_pmc_base_summary.txt).  Gradient tile: 16 quads x 8 groups (threads 0-127; K = 1: the input tile has the same shape
    // and takes threads 128-255); input tile: XR rows x QPR quads (the last one partial).
    constexpr int GITEMS = WR * 4 * 8;               // 128 (4 rows) or 256 (8 rows)
    constexpr int QPR = (XC + 3) / 4, XITEMS = XR * QPR * 8, XROUNDS = K == 1 ? 1 : (XITEMS + 255) / 256;
    h8 gv[4], xv[XROUNDS][4];
    const int sq = tid & 7;                          // 8-channel group of this thread's items
    auto fetch = [&](long blk) __attribute__((always_inline)) {
        const int bxi = (int)(blk % p.bx);
        const long rest = blk / p.bx;
        const int byi = (int)(rest % p.by), n = (int)(rest / p.by);
        const int oy0 = byi * WR, ox0 = bxi * WC;
        if (K != 1 || tid < 128) {
            const int quad = (tid >> 3) & (WR * 4 - 1), r = quad >> 2, c = (quad & 3) * 4;
            const int oy = oy0 + r, ch = ot * 64 + sq * 8;
            const _Float16* gp = p.g + (((long)n * p.OH + oy) * p.OW + ox0 + c) * p.O + ch;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (tid < GITEMS && oy < p.OH && ox0 + c + m < p.OW && ch < p.O) v = *(const h8*)(gp + (long)m * p.O);
                gv[m] = v;
            }
        }
        const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
#pragma unroll
        for (int u = 0; u < XROUNDS; ++u) {
            const int e = K == 1 ? tid - 128 : tid + u * 256, quad = e >> 3, r = quad / QPR, c = (quad - r * QPR) * 4;
            const int iy = iy0 + r, ch = it * 64 + sq * 8;
            const bool rok = e >= 0 && e < XITEMS && iy >= 0 && iy < p.H && ch < p.I;
            const _Float16* xp = p.x + (((long)n * p.H + iy) * p.W + ix0 + c) * p.I + ch;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                const int ix = ix0 + c + m;
                if (rok && c + m < XC && ix >= 0 && ix < p.W) v = *(const h8*)(xp + (long)m * p.I);
                xv[u][m] = v;
            }
        }
    };
    if ((long)blockIdx.y < p.nblocks) fetch(blockIdx.y);
    for (long blk = blockIdx.y; blk < p.nblocks; blk += p.slices) {
        __syncthreads();
        // LDS row = (ch % 8) * 8 + ch / 8: the 8 lanes of a pixel quad hit 8 different 16-byte slots
        if (tid < GITEMS) {
            const int quad = tid >> 3, pp = (quad >> 2) * WC + (quad & 3) * 4;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h4 w4 = {gv[0][k], gv[1][k], gv[2][k], gv[3][k]};
                *(h4*)(gT + (k * 8 + sq) * GP + pp) = w4;
            }
        }
#pragma unroll
        for (int u = 0; u < XROUNDS; ++u) {
            const int e = K == 1 ? tid - 128 : tid + u * 256, quad = e >> 3, r = quad / QPR, c = (quad - r * QPR) * 4;
            if (e >= 0 && e < XITEMS) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    _Float16* row = xT + (k * 8 + sq) * XP + r * RP;
                    if constexpr (S == 1) {
                        h4 w4 = {xv[u][0][k], xv[u][1][k], xv[u][2][k], xv[u][3][k]};
                        *(h4*)(row + c) = w4;
                    } else {                                         // columns c, c + 2 -> even half, c + 1, c + 3 -> odd half (at + 24)
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 ev = {xv[u][0][k], xv[u][2][k]}, od = {xv[u][1][k], xv[u][3][k]};
                        *(h2*)(row + (c >> 1)) = ev;
                        *(h2*)(row + 24 + (c >> 1)) = od;
                    }
                }
            }
        }
        __syncthreads();
        if (blk + p.slices < p.nblocks) fetch(blk + p.slices);
#pragma unroll 1
        for (int r = 0; r < WR; ++r) {
            const h8 a = *(const h8*)(gT + (wo + j) * GP + r * WC + kg * 8);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const _Float16* row = xT + (wi + j) * XP + (r * S + ky) * RP + kg * 8;
                const u32x4 e0 = *(const u32x4*)row;
                if constexpr (K == 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, e0), acc[0], 0, 0, 0);
                } else {
                    const unsigned e4 = *(const unsigned*)(row + 8);
                    u32x4 sh;                                              // the same 8 pixels shifted by one
                    sh[0] = __builtin_amdgcn_alignbit(e0[1], e0[0], 16); sh[1] = __builtin_amdgcn_alignbit(e0[2], e0[1], 16);
                    sh[2] = __builtin_amdgcn_alignbit(e0[3], e0[2], 16); sh[3] = __builtin_amdgcn_alignbit(e4, e0[3], 16);
                    u32x4 third;
                    if constexpr (S == 1) { third[0] = e0[1]; third[1] = e0[2]; third[2] = e0[3]; third[3] = e4; }       // shifted by two
                    else third = *(const u32x4*)(row + 24);                                                             // the odd columns
                    const h8 b0 = __builtin_bit_cast(h8, e0), b1 = __builtin_bit_cast(h8, S == 1 ? sh : third), b2 = __builtin_bit_cast(h8, S == 1 ? third : sh);
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc[ky * 3 + 2], 0, 0, 0);
                }
            }
        }
    }
    // partial sums: row (o) = (r & 3) + 8 (r >> 2) + 4 kg, column (i) = j
    float* pp = p.part + (long)blockIdx.y * NT * p.OP * p.IP;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (LDS row R of a tile holds channel (R % 8) * 8 + R / 8, see the staging)
            const int ro = wo + (r & 3) + 8 * (r >> 2) + 4 * kg, ri = wi + j;
            const int o = ot * 64 + (ro & 7) * 8 + (ro >> 3), i = it * 64 + (ri & 7) * 8 + (ri >> 3);
            pp[((long)t * p.OP + o) * p.IP + i] = acc[t][r];
        }
}

// dw[t][o][i] = sum over slices in a fixed order: G slice groups per workgroup (256 / G elements each), thread (g, el) adds slices g, g + G, ...
// in four interleaved accumulators, the groups are combined in group order through LDS (deterministic for a given shape).  With one thread per
// element over all slices a 64-channel layer (36 864 weights x 512 slices) ran on 144 workgroups of serial strided loads.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, int slices, int NT, int O, int I, int OP, int IP) {
    __shared__ float red[256];
    constexpr int EL = 256 / G;
    const int el = threadIdx.x % EL, g = threadIdx.x / EL;
    const long total = (long)NT * O * I, e = (long)blockIdx.x * EL + el, sl = (long)NT * OP * IP;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (e < total) {
        const int i = (int)(e % I);
        const long r = e / I;
        const int o = (int)(r % O), t = (int)(r / O);
        const float* pe = part + ((long)t * OP + o) * IP + i;
        int k = g;
        for (; k + 3 * G < slices; k += 4 * G) {
            v0 += pe[k * sl]; v1 += pe[(k + G) * sl]; v2 += pe[(k + 2 * G) * sl]; v3 += pe[(k + 3 * G) * sl];
        }
        for (; k < slices; k += G) v0 += pe[k * sl];
    }
    float v = (v0 + v1) + (v2 + v3);
    if (G > 1) {
        red[threadIdx.x] = v;
        __syncthreads();
        if (g == 0)
            for (int k = 1; k < G; ++k) v += red[k * EL + el];
    }
    if (g == 0 && e < total) dw[e] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// FIR resampling, NHWC halves (upfirdn2d.cu:29-92 generic gather; fp32 accumulation as its `scalar_t` -> float path)
// ---------------------------------------------------------------------------------------------------------------------------
struct UfdH {
    const _Float16* x;
    const float* f;
    _Float16* y;
    int N, C, H, W, OH, OW, fh, fw, upx, upy, dnx, dny, px0, py0, flip;
    float gain;
};

__global__ __launch_bounds__(256) void upfirdn2d_f16_kernel(const UfdH p) {
    __shared__ float sf[64];
    for (int k = threadIdx.x; k < p.fh * p.fw; k += 256) {
        const int ky = k / p.fw, kx = k - ky * p.fw;
        const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
        sf[k] = p.f[sy * p.fw + sx] * p.gain;
    }
    __syncthreads();
    const int c8n = p.C >> 3;
    const long total = (long)p.N * p.OH * p.OW * c8n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % c8n);
        long r = e / c8n;
        const int ox = (int)(r % p.OW);
        r /= p.OW;
        const int oy = (int)(r % p.OH), n = (int)(r / p.OH);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
                const h8 xv = *(const h8*)(p.x + (((long)n * p.H + iy) * p.W + ix) * p.C + c8 * 8);
                const float fk = sf[ky * p.fw + kx];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += (float)xv[q] * fk;
            }
        }
        h8 out;
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q] = (_Float16)v[q];
        *(h8*)(p.y + e * 8) = out;
    }
}

// up = down = 1 (the pad-2 pre-filter of the stride-2 layers, the pad-1 post-filter of the transposed convolutions and both their
// gradients -- all but the x2 resampling of the skip paths): a lane produces 2 x 4 adjacent pixels x 8 channels, so that each loaded
// input vector feeds up to 2 x 4 outputs (35 loads per 8 outputs of a 4x4 filter instead of 16 each) and nothing is divided.
// 4x4 filter with up = 2 or down = 2 (the skip branches of the half blocks and their gradients): the factors are compile-time constants (the generic
// kernel above divides by run-time up / down factors per tap and waits for every guarded load), every load is issued unconditionally from a clamped
// address and masked afterwards -- 16 loads per output for down = 2, the 4 taps that hit a real sample for up = 2.
template <int UP, int DN>
__global__ __launch_bounds__(256) void updn4_f16_kernel(const UfdH p) {
    static_assert((UP == 1 && DN == 2) || (UP == 2 && DN == 1), "one factor of two");
    __shared__ float sf[16];
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        sf[threadIdx.x] = p.f[(p.flip ? ky : 3 - ky) * 4 + (p.flip ? kx : 3 - kx)] * p.gain;
    }
    __syncthreads();
    const unsigned c8n = p.C >> 3, total = (unsigned)p.N * p.OH * p.OW * c8n;                  // (host: < 2^31)
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned c8 = e % c8n;
        unsigned r = e / c8n;
        const int ox = r % p.OW;
        r /= p.OW;
        const int oy = r % p.OH, n = r / p.OH;
        const _Float16* xb = p.x + (long)n * p.H * p.W * p.C + c8 * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (DN == 2) {
            h8 xin[4][4];
            float m[4][4];
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int iy = oy * 2 + ky - p.py0, iyc = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ix = ox * 2 + kx - p.px0, ixc = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
                    xin[ky][kx] = *(const h8*)(xb + ((long)iyc * p.W + ixc) * p.C);
                    m[ky][kx] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? sf[ky * 4 + kx] : 0.f;
                }
            }
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += (float)xin[ky][kx][q] * m[ky][kx];
        } else {
            // sample (oy + ky - py0) / 2 exists for the two ky of the right parity
            const int ky0 = (p.py0 - oy) & 1, kx0 = (p.px0 - ox) & 1;
            h8 xin[2][2];
            float m[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ky = ky0 + 2 * a, uy = oy + ky - p.py0, iy = uy >> 1, iyc = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int kx = kx0 + 2 * b, ux = ox + kx - p.px0, ix = ux >> 1, ixc = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
                    xin[a][b] = *(const h8*)(xb + ((long)iyc * p.W + ixc) * p.C);
                    m[a][b] = (uy >= 0 && iy < p.H && ux >= 0 && ix < p.W) ? sf[ky * 4 + kx] : 0.f;
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += (float)xin[a][b][q] * m[a][b];
        }
        h8 out;
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q] = (_Float16)v[q];
        *(h8*)(p.y + (long)e * 8) = out;
    }
}

template <int FS, int RB>  // FS = 4: the 4x4 filter, fully unrolled with unconditional (clamped) loads; FS = 0: any size, guarded loop; RB output rows per lane
__global__ __launch_bounds__(256) void fir_same_f16_kernel(const UfdH p) {
    __shared__ float sf[64];
    for (int k = threadIdx.x; k < p.fh * p.fw; k += 256) {
        const int ky = k / p.fw, kx = k - ky * p.fw;
        const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
        sf[k] = p.f[sy * p.fw + sx] * p.gain;
    }
    __syncthreads();
    // a lane = 2 rows x 4 columns of outputs x 8 channels: (fh + 1) x (fw + 3) loads feed 8 outputs (4.4 per output for a 4x4 filter)
    const int c8n = p.C >> 3, oxg = (p.OW + 3) >> 2, oyg = (p.OH + RB - 1) / RB;
    const long total = (long)p.N * oyg * oxg * c8n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % c8n);
        long r = e / c8n;
        const int ox0 = (int)(r % oxg) * 4;
        r /= oxg;
        const int oy0 = (int)(r % oyg) * RB, n = (int)(r / oyg);
        float v[RB][4][8];
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[b][a][q] = 0.f;
        if constexpr (FS == 4) {
            // every load is issued (addresses clamped, out-of-range vectors zeroed afterwards): 35 independent loads in flight per lane
            // instead of one guarded load at a time -- the guarded loop was 72 % parked on memory at 2.0 TB/s
            h8 xin[3 + RB][7];
#pragma unroll
            for (int ry = 0; ry < 3 + RB; ++ry) {
                const int iy = oy0 + ry - p.py0, iyc = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
                const _Float16* row = p.x + ((long)n * p.H + iyc) * p.W * p.C + c8 * 8;
#pragma unroll
                for (int cx = 0; cx < 7; ++cx) {
                    const int ix = ox0 + cx - p.px0, ixc = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
                    xin[ry][cx] = *(const h8*)(row + (long)ixc * p.C);
                }
            }
#pragma unroll
            for (int ry = 0; ry < 3 + RB; ++ry) {
                const int iy = oy0 + ry - p.py0;
#pragma unroll
                for (int cx = 0; cx < 7; ++cx) {
                    const int ix = ox0 + cx - p.px0;
                    const float m = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 1.f : 0.f;
                    float xf[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xf[q] = (float)xin[ry][cx][q] * m;
#pragma unroll
                    for (int b = 0; b < RB; ++b) {
                        const int ky = ry - b;
                        if (ky < 0 || ky >= 4) continue;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int kx = cx - a;
                            if (kx < 0 || kx >= 4) continue;
                            const float fk = sf[ky * 4 + kx];
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[b][a][q] += xf[q] * fk;
                        }
                    }
                }
            }
        } else
        for (int ry = 0; ry < p.fh + RB - 1; ++ry) {                  // input row oy0 + ry - py0 feeds output row b with tap ky = ry - b
            const int iy = oy0 + ry - p.py0;
            if (iy < 0 || iy >= p.H) continue;
            const _Float16* row = p.x + ((long)n * p.H + iy) * p.W * p.C + c8 * 8;
            for (int cx = 0; cx < p.fw + 3; ++cx) {                   // input column ox0 + cx - px0 feeds output a with tap kx = cx - a
                const int ix = ox0 + cx - p.px0;
                if (ix < 0 || ix >= p.W) continue;
                const h8 xv = *(const h8*)(row + (long)ix * p.C);
                float xf[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xf[q] = (float)xv[q];
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int ky = ry - b;
                    if (ky < 0 || ky >= p.fh) continue;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int kx = cx - a;
                        if (kx < 0 || kx >= p.fw) continue;
                        const float fk = sf[ky * p.fw + kx];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[b][a][q] += xf[q] * fk;
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if (oy0 + b >= p.OH) break;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (ox0 + a >= p.OW) break;
                h8 out;
#pragma unroll
                for (int q = 0; q < 8; ++q) out[q] = (_Float16)v[b][a][q];
                *(h8*)(p.y + ((((long)n * p.OH + oy0 + b) * p.OW + ox0 + a) * p.C) + c8 * 8) = out;
            }
        }
    }
}

// Same-size 4x4 FIR, marching: a lane owns TWO adjacent output columns x 8 channels and walks down `rows` output rows.  Per input row it loads
// 5 pixels (2.5 sixteen-byte loads per output piece instead of 4.4), forms the two horizontal 4-tap sums in fp32 and keeps the last four of them in
// registers; an output row is the vertical 4-tap sum of that ring -- 8 multiply-adds per output value instead of 16, written on float2 so that the
// compiler emits packed fp32 math.  The 4x4 filter of the model is an outer product (upfirdn2d.setup_filter of [1,3,3,1]); every thread factors the
// taps itself (pivot row / column) and a filter that is NOT rank one takes the plain 16-tap loop below, same launch.  Column masks are folded into the
// horizontal coefficients once per lane, out-of-range rows are loaded from a clamped address and multiplied by zero.  The 2 x 4-output
// lanes of fir_same_f16_kernel spent 206 VALU operations per output piece (105 us of pure issue at 64 ch x 513^2 x 8) and ran at 2.3 TB/s.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fir4_march_f16_kernel(const UfdH p, int rows) {
    const int c8n = p.C >> 3, oxg = (p.OW + 1) >> 1, nstrip = (p.OH + rows - 1) / rows;
    const unsigned total = (unsigned)p.N * nstrip * oxg * c8n;
    unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= total) return;
    const int c8 = e % c8n;
    e /= c8n;
    const int ox0 = (e % oxg) * 2;
    e /= oxg;
    const int oy0 = (e % nstrip) * rows, n = e / nstrip;
    // taps as the gather form uses them: output (oy, ox) = sum_k t[ky][kx] * x[oy + ky - py0][ox + kx - px0]
    float t[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) t[ky][kx] = p.f[(p.flip ? ky : 3 - ky) * 4 + (p.flip ? kx : 3 - kx)] * p.gain;
    int pky = 0, pkx = 0;
    float piv = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
            if (fabsf(t[ky][kx]) > fabsf(piv)) { piv = t[ky][kx]; pky = ky; pkx = kx; }
    float fr[4], fc[4];
    bool sep = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) { fr[k] = t[pky][k]; fc[k] = piv != 0.f ? t[k][pkx] / piv : 0.f; }
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) sep = sep && fabsf(t[ky][kx] - fc[ky] * fr[kx]) <= 1e-6f * fabsf(piv);
    const _Float16* xb = p.x + (long)n * p.H * p.W * p.C + c8 * 8;
    _Float16* yb = p.y + (long)n * p.OH * p.OW * p.C + c8 * 8;
    if (!sep) {                                                     // not an outer product: 16 guarded taps per output
        for (int oy = oy0; oy < oy0 + rows && oy < p.OH; ++oy)
            for (int a = 0; a < 2 && ox0 + a < p.OW; ++a) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int ky = 0; ky < 4; ++ky)
                    for (int kx = 0; kx < 4; ++kx) {
                        const int iy = oy + ky - p.py0, ix = ox0 + a + kx - p.px0;
                        if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
                        const h8 xv = *(const h8*)(xb + ((long)iy * p.W + ix) * p.C);
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[q] += (float)xv[q] * t[ky][kx];
                    }
                h8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (_Float16)acc[q];
                *(h8*)(yb + ((long)oy * p.OW + ox0 + a) * p.C) = o;
            }
        return;
    }
    float cx[2][4];
    int ixc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int ix = ox0 + j - p.px0;
        ixc[j] = (ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix)) * p.C;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ix = ox0 + a + k - p.px0;
            cx[a][k] = (ix >= 0 && ix < p.W) ? fr[k] : 0.f;
        }
    f2 h[4][2][4];                                                  // ring of horizontal sums: [input row & 3][column][channel pair]
    auto step = [&](int rr, auto SLOT) __attribute__((always_inline)) {
        constexpr int S = decltype(SLOT)::value;
        const int iy = oy0 + rr - p.py0;
        const float my = (iy >= 0 && iy < p.H) ? 1.f : 0.f;
        const _Float16* row = xb + (long)(iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy)) * p.W * p.C;
        h8 xv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[j] = *(const h8*)(row + ixc[j]);
        f2 xf[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) { hh2 v = {xv[j][2 * q], xv[j][2 * q + 1]}; xf[j][q] = __builtin_convertvector(v, f2); }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f2 sum = xf[a][q] * cx[a][0];
#pragma unroll
                for (int k = 1; k < 4; ++k) sum += xf[a + k][q] * cx[a][k];
                h[S][a][q] = sum * my;
            }
        const int oy = oy0 + rr - 3;                                // rows oy - py0 + 0..3 = ring slots S+1, S+2, S+3, S
        if (rr >= 3 && oy < p.OH) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                h8 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f2 v = h[(S + 1) & 3][a][q] * fc[0] + h[(S + 2) & 3][a][q] * fc[1] + h[(S + 3) & 3][a][q] * fc[2] + h[S][a][q] * fc[3];
                    const hh2 r = __builtin_convertvector(v, hh2);
                    o[2 * q] = r[0]; o[2 * q + 1] = r[1];
                }
                if (ox0 + a < p.OW) *(h8*)(yb + ((long)oy * p.OW + ox0 + a) * p.C) = o;
            }
        }
    };
    for (int rb = 0; rb < rows + 3; rb += 4) {                      // rows % 4 == 0: the last group has three steps
        step(rb, std::integral_constant<int, 0>{});
        step(rb + 1, std::integral_constant<int, 1>{});
        step(rb + 2, std::integral_constant<int, 2>{});
        if (rb + 3 < rows + 3) step(rb + 3, std::integral_constant<int, 3>{});
    }
}

// y = lrelu_agc(x + bias[c]) on NHWC halves (arithmetic in fp32, one rounding), and its gradient from the saved output
// Streaming passes: the per-channel operands are read as two float4 ONCE per thread when 256 % (C/8) == 0 (a thread then keeps its channels for
// the whole grid-stride loop) -- eight scalar operand loads and a 64-bit modulo per 16-byte piece ran at 2.0-3.1 TB/s of read + write traffic, this
// form at 5.4-6.3 (tools/conv_f16_bench.py).  EU > 1 requests several pieces before the first use: no gain (4.8 TB/s at 4), the pass is not
// short of loads in flight.
#ifndef SHG_F16_EU
#define SHG_F16_EU 1
#endif
constexpr int EU = SHG_F16_EU;

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// CONSTC: 256 % (C/8) == 0 -- a thread keeps its 8 channels for the whole grid-stride loop and holds their bias in registers
template <bool CONSTC>
__global__ __launch_bounds__(256) void bias_act_f16_kernel(const _Float16* x, const float* bias, _Float16* y, long total8, int C, int act,
                                                           float alpha, float gain, float clamp) {
    const int c8n = C >> 3;
    const long stride = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (CONSTC && bias) load8f(bias + (int)(e0 % c8n) * 8, bb);
    for (long eb = e0; eb < total8; eb += EU * stride) {
        h8 v[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const long e = eb + u * stride;
            v[u] = *(const h8*)(x + (e < total8 ? e : total8 - 1) * 8);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const long e = eb + u * stride;
            if (!CONSTC && bias) load8f(bias + (int)((e < total8 ? e : total8 - 1) % c8n) * 8, bb);
            h8 out;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = (float)v[u][q] + bb[q];
                t = act ? shg_lrelu_agc(t, alpha, gain, clamp) : t * gain;
                out[q] = (_Float16)t;
            }
            if (e < total8) *(h8*)(y + e * 8) = out;
        }
    }
}

// (three streams and no per-channel operand: the plain loop already runs at 5+ TB/s, grouping the loads measured 5 % slower)
__global__ __launch_bounds__(256) void bias_act_backward_f16_kernel(const _Float16* g, const _Float16* y, _Float16* dx, long total8, int act,
                                                                    float alpha, float gain, float clamp) {
    const float gp = gain, gn = act ? alpha * gain : gain;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long)gridDim.x * 256) {
        const h8 gv = *(const h8*)(g + e * 8), yv = *(const h8*)(y + e * 8);
        h8 out;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = (float)yv[q];
            const float slope = (act && clamp >= 0.f && fabsf(v) >= clamp) ? 0.f : (v > 0.f ? gp : gn);
            out[q] = (_Float16)((float)gv[q] * slope);
        }
        *(h8*)(dx + e * 8) = out;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Modulation tail of a half layer in ONE pass each way (stylegan.py:176-181 `fma(x, dcoefs, noise)` + :298-304 bias / lrelu_agc, and
// with act = 0 / no noise / no bias the input scaling `x * styles` of :173):
//     y = A(t * d[n,c] + noise[n?,h,w] + bias[c])
// backward (first order): gz = gy * A'(y) read from the saved output, gt = gz * d, and the three reductions of the same pass --
// sum_{hw} gz t (-> d), sum_{hw} gz (-> bias) per (n, c) as per-workgroup partials [n][block][2][C] (summed in a fixed order by the
// caller: deterministic), sum_c gz per pixel (-> noise).  A lane holds 8 channels of one pixel; the C/8 lanes of a pixel are adjacent.
// ---------------------------------------------------------------------------------------------------------------------------
struct TailP {
    const _Float16* t; const _Float16* y_in; const _Float16* gy;
    const float* d; const float* noise; const float* bias;
    const _Float16* u; const float* e;  // backward: optional second product, out = A'(y) (gy d + u e)
    _Float16* out;                    // forward: y; backward: gt
    float* part; float* gnoise;
    int N, HW, C, noise_mode, act, nblk;
    float alpha, gain, clamp;
};

template <bool CONSTC>
__global__ __launch_bounds__(256) void modtail_f16_kernel(const TailP p) {
    const int c8n = p.C >> 3, n = blockIdx.y;
    const long total = (long)p.HW * c8n;
    const _Float16* tp = p.t + (long)n * p.HW * p.C;
    _Float16* yp = p.out + (long)n * p.HW * p.C;
    const float* np_ = p.noise_mode == 0 ? nullptr : p.noise + (p.noise_mode == 2 ? (long)n * p.HW : 0);
    const long stride = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
    float dd[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (CONSTC) {
        const int c = (int)(e0 % c8n) * 8;
        if (p.d) load8f(p.d + (long)n * p.C + c, dd);
        if (p.bias) load8f(p.bias + c, bb);
    }
    for (long eb = e0; eb < total; eb += EU * stride) {
        h8 v[EU];
        float nz[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const long e = eb + u * stride, ec = e < total ? e : total - 1;
            v[u] = *(const h8*)(tp + ec * 8);
            nz[u] = np_ ? np_[ec / c8n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const long e = eb + u * stride;
            if (!CONSTC) {
                const int c = (int)((e < total ? e : total - 1) % c8n) * 8;
                if (p.d) load8f(p.d + (long)n * p.C + c, dd);
                if (p.bias) load8f(p.bias + c, bb);
            }
            h8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float z = (float)v[u][q] * dd[q] + nz[u] + bb[q];
                z = p.act ? shg_lrelu_agc(z, p.alpha, p.gain, p.clamp) : z * p.gain;
                o[q] = (_Float16)z;
            }
            if (e < total) *(h8*)(yp + e * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void modtail_backward_f16_kernel(const TailP p) {
    __shared__ float red[2][256][8];
    const int c8n = p.C >> 3, n = blockIdx.y, tid = threadIdx.x;
    const long total = (long)p.HW * c8n, off = (long)n * p.HW * p.C;
    const float gp = p.gain, gn = p.act ? p.alpha * p.gain : p.gain;
    float s1[8], s0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s1[q] = s0[q] = 0.f;
    // (256 and the grid stride are multiples of C/8 <= 64: a thread keeps its channel group for the whole loop)
    const int c = (int)(((long)blockIdx.x * 256 + tid) % c8n) * 8;
    float dd[8], ee[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { dd[q] = p.d ? p.d[(long)n * p.C + c + q] : 1.f; ee[q] = (p.u && p.e) ? p.e[(long)n * p.C + c + q] : 1.f; }
    const long stride = (long)gridDim.x * 256;
    const int iters = (int)((total + stride - 1) / stride);                                        // uniform trip count: whole waves stay in the shuffle
    for (int it = 0; it < iters; ++it) {
        const long e = (long)blockIdx.x * 256 + tid + it * stride;
        const bool ok = e < total;
        float pix_sum = 0.f;
        if (ok) {
            const h8 g = *(const h8*)(p.gy + off + e * 8), yv = *(const h8*)(p.y_in + off + e * 8);
            h8 tv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (p.t) tv = *(const h8*)(p.t + off + e * 8);
            h8 uv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (p.u) uv = *(const h8*)(p.u + off + e * 8);
            h8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float yq = (float)yv[q];
                const float slope = (p.act && p.clamp >= 0.f && fabsf(yq) >= p.clamp) ? 0.f : (yq > 0.f || !p.act ? gp : gn);
                const float gz = (float)g[q] * slope;
                o[q] = p.u ? (_Float16)__builtin_fmaf((float)uv[q] * slope, ee[q], gz * dd[q]) : (_Float16)(gz * dd[q]);
                s1[q] += gz * (float)tv[q];
                s0[q] += gz;
                pix_sum += gz;
            }
            *(h8*)(p.out + off + e * 8) = o;
        }
        if (p.gnoise) {                                           // sum over the C/8 adjacent lanes of this pixel
            for (int m = 1; m < c8n; m <<= 1) pix_sum += __shfl_xor(pix_sum, m, 64);
            if (ok && (e % c8n) == 0) p.gnoise[(long)n * p.HW + e / c8n] = pix_sum;
        }
    }
    if (!p.part) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) { red[0][tid][q] = s1[q]; red[1][tid][q] = s0[q]; }
    __syncthreads();
    // threads tid, tid + c8n, tid + 2 c8n, ... share a channel group
    if (tid < c8n) {
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float a = 0.f;
                for (int j = tid; j < 256; j += c8n) a += red[k][j][q];
                p.part[(((long)n * p.nblk + blockIdx.x) * 2 + k) * p.C + tid * 8 + q] = a;
            }
    }
}

template <int MB, int NB>
static int launch_conv(ConvP p, hipStream_t st) {
    const dim3 grid((unsigned)((long)p.N * p.tiles_x * p.tiles_y), (unsigned)shg_cdiv(p.OB, MB));
    size_t lds = (size_t)p.PH * p.PW * PSTR * sizeof(_Float16);
    const size_t out_tile = (size_t)TH * TW * NB * (MB * 32 + 8) * sizeof(_Float16);       // the epilogue's transposed output tile reuses the patch
    if (out_tile > lds) lds = out_tile;
    // Thick 3x3 layers (I >= 384: >= 12 chunks per tile) stage their weight slab in LDS (MB * 9 * 2 KiB behind the patch): 206 -> 181 us at
    // 512 channels.  Thin layers lose (64 ch: 256 -> 312 us, 128 ch: 196 -> 214: two chunks cannot amortise the second staging stream and the
    // 232-register kernel), MB = 4 would need 72 KB + the patch: both keep the L2 route with one-tap-ahead operand prefetch.
    const bool wl = p.ntaps == 9 && MB <= 2 && p.I >= 384;
    p.wlds_off = (int)(lds / sizeof(_Float16));
    if (wl) lds += (size_t)MB * 9 * 2 * 1024;
    switch (p.ntaps) {
        case 1: hipLaunchKernelGGL((conv_f16_kernel<MB, 1, NB, false>), grid, dim3(256), lds, st, p); break;
        case 2: hipLaunchKernelGGL((conv_f16_kernel<MB, 2, NB, false>), grid, dim3(256), lds, st, p); break;
        case 4: hipLaunchKernelGGL((conv_f16_kernel<MB, 4, NB, false>), grid, dim3(256), lds, st, p); break;
        case 9:
            if (wl) hipLaunchKernelGGL((conv_f16_kernel<MB, 9, NB, (MB <= 2)>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((conv_f16_kernel<MB, 9, NB, false>), grid, dim3(256), lds, st, p);
            break;
        default: shg_set_error("conv2d_f16: %d taps", p.ntaps); return SHG_ERR_UNSUPPORTED;
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// one launch of the gather convolution for a tap list; dy/dx are INPUT offsets relative to (oy'*s_in, ox'*s_in)
static int conv_taps(ConvP p, int ntaps, const int* dy, const int* dx, const int* slot, int GH, int GW, hipStream_t st) {
    if (GH <= 0 || GW <= 0) return SHG_OK;
    int mny = dy[0], mxy = dy[0], mnx = dx[0], mxx = dx[0];
    for (int t = 1; t < ntaps; ++t) {
        mny = dy[t] < mny ? dy[t] : mny; mxy = dy[t] > mxy ? dy[t] : mxy;
        mnx = dx[t] < mnx ? dx[t] : mnx; mxx = dx[t] > mxx ? dx[t] : mxx;
    }
    p.ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) { p.tdy[t] = dy[t] - mny; p.tdx[t] = dx[t] - mnx; p.tw[t] = slot[t]; }
    p.org_y = mny; p.org_x = mnx;
    p.GH = GH; p.GW = GW;
    if (conv_ring_eligible(p, mxy - mny + 1, mxx - mnx + 1)) return conv_ring_launch(p, mxy - mny + 1, mxx - mnx + 1, st);
    if (mny == mnx && mxy - mny == 2 && mxx - mnx == 2 && conv_down_eligible(p)) return conv_down_launch(p, -mny, st);
    // wide grids of stride-1 reads: 8 x 32 pixel tiles, two pixel blocks per wave (each weight operand feeds two MFMAs); stride-2 reads
    // (a 4x larger patch) and narrow grids keep 8 x 16
    const int nb = (p.s_in == 1 && GW > TW) ? 2 : 1;
    p.PH = (TH - 1) * p.s_in + (mxy - mny) + 1;
    p.PW = (TW * nb - 1) * p.s_in + (mxx - mnx) + 1;
    p.GH = GH; p.GW = GW;
    p.tiles_y = shg_cdiv(GH, TH); p.tiles_x = shg_cdiv(GW, TW * nb);
    if (nb == 2) {                   // (4 channel blocks x 2 pixel blocks need 290 VGPRs = one wave per SIMD: 2 x 2 keeps three)
        if (p.OB > 1) return launch_conv<2, 2>(p, st);
        return launch_conv<1, 2>(p, st);
    }
    if (p.OB > 2) return launch_conv<4, 1>(p, st);
    if (p.OB > 1) return launch_conv<2, 1>(p, st);
    return launch_conv<1, 1>(p, st);
}

// ---- block-boundary casts (stylegan.py:486-495,659-663; comodgan.py:39-43,305-312: `x.to(dtype)`): float32 NCHW <-> float16 NHWC as one
// transposing pass through LDS.  Tile = 64 channels x 64 pixels; global accesses are 256-byte runs along the pixels on the float32 side and
// whole 16-byte pieces of 8 channels (128 bytes per pixel and tile) on the float16 side.  torch's `.to(dtype, memory_format)` reached
// 1.6 TB/s on these tensors (63 us for [8, 512, 64, 64]).
__global__ __launch_bounds__(256) void relayout_to_half_kernel(const float* x, _Float16* y, int C, int HW) {
    __shared__ float t[64][65];
    const int tid = threadIdx.x, n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const float* xn = x + (long)n * C * HW;
    _Float16* yn = y + (long)n * C * HW;
    const int pl = tid & 63, cw = tid >> 6;
#pragma unroll 4
    for (int c = cw; c < 64; c += 4) t[c][pl] = (c0 + c < C && p0 + pl < HW) ? xn[(long)(c0 + c) * HW + p0 + pl] : 0.f;
    __syncthreads();
    const int cq = (tid & 7) * 8;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pp = (tid >> 3) + 32 * it;
        if (p0 + pp < HW && c0 + cq < C) {                       // C % 8 == 0: a piece is inside or outside as a whole
            h8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (_Float16)t[cq + k][pp];
            *(h8*)(yn + (long)(p0 + pp) * C + c0 + cq) = v;
        }
    }
}

__global__ __launch_bounds__(256) void relayout_to_float_kernel(const _Float16* x, float* y, int C, int HW) {
    __shared__ float t[64][65];
    const int tid = threadIdx.x, n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const _Float16* xn = x + (long)n * C * HW;
    float* yn = y + (long)n * C * HW;
    const int cq = (tid & 7) * 8;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pp = (tid >> 3) + 32 * it;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p0 + pp < HW && c0 + cq < C) v = *(const h8*)(xn + (long)(p0 + pp) * C + c0 + cq);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[cq + k][pp] = (float)v[k];
    }
    __syncthreads();
    const int pl = tid & 63, cw = tid >> 6;
#pragma unroll 4
    for (int c = cw; c < 64; c += 4)
        if (c0 + c < C && p0 + pl < HW) yn[(long)(c0 + c) * HW + p0 + pl] = t[c][pl];
}

// thin tensors (the 4-channel network inputs, RGB): a lane owns a pixel, C <= 16 planes on the float32 side, 2 C contiguous bytes on the other
template <bool TO_HALF>
__global__ __launch_bounds__(256) void relayout_thin_kernel(const void* src, void* dst, int C, long HW, long total) {
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const long n = p / HW, q = p - n * HW;
        if (TO_HALF) {
            const float* x = (const float*)src + n * C * HW + q;
            _Float16* y = (_Float16*)dst + p * C;
            for (int c = 0; c < C; ++c) y[c] = (_Float16)x[(long)c * HW];
        } else {
            const _Float16* x = (const _Float16*)src + p * C;
            float* y = (float*)dst + n * C * HW + q;
            for (int c = 0; c < C; ++c) y[(long)c * HW] = (float)x[c];
        }
    }
}

}  // namespace f16

// x [N,H,W,I] halves, bias fp32 [O] or null, y halves.  w: k*k tap slots (cross-correlation taps in row-major (ky,kx) order) packed in MFMA operand
// order by shg_conv2d_f16_pack_weight: [ceil(O/32)][k*k][I/16][64][8] halves.
//   mode 0: y[N,OH,OW,O] = conv2d(x, w, stride, pad)                      OH = (H + 2 pad - k) / stride + 1
//   mode 1: y[N,OH,OW,O] = rows / columns [crop, crop + OH) of conv_transpose2d(x, w, stride 2) (3x3; w[t][o][i] = torch weight[i][o][ky][kx]),
//           zero where the (2H+1) x (2W+1) result ends earlier.  I % 32 == 0.
struct shg_f16_tail_ { const float* in_scale; const float* out_scale; const float* noise; int noise_mode; float noise_strength; int act; float alpha, gain, clamp; const void* residual; };

static int conv2d_f16_impl(const void* x, const void* w, const float* bias, void* y, int N, int I, int O, int H, int W, int k, int stride,
                           int pad, int mode, int crop, int OH, int OW, const shg_f16_tail_* tl, void* stream) {
    SHG_CHECK_ARG(x && w && y, "conv2d_f16: null pointer");
    SHG_CHECK_ARG(N >= 1 && I >= 32 && (I % 32) == 0 && O >= 1 && H >= 1 && W >= 1, "conv2d_f16: bad shape (I must be a multiple of 32)");
    SHG_CHECK_ARG((k == 1 || k == 3) && (stride == 1 || stride == 2) && pad >= 0 && pad <= k, "conv2d_f16: 1x1 / 3x3 kernels, stride 1 / 2");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                  "conv2d_f16: x, w and y must be 16-byte aligned (vector loads / stores)");
    f16::ConvP p{};
    p.x = (const _Float16*)x; p.w = (const _Float16*)w; p.bias = bias; p.y = (_Float16*)y;
    p.N = N; p.I = I; p.O = O; p.H = H; p.W = W; p.OHt = OH; p.OWt = OW;
    p.OB = (O + 31) / 32; p.wslots = k * k;
    if (tl) {
        p.in_scale = tl->in_scale; p.out_scale = tl->out_scale; p.noise = tl->noise_mode ? tl->noise : nullptr; p.noise_mode = tl->noise ? tl->noise_mode : 0;
        p.noise_strength = tl->noise_strength; p.act = tl->act; p.alpha = tl->alpha; p.gain = tl->gain; p.clamp = tl->clamp;
        p.residual = (const _Float16*)tl->residual;
        p.tail = (tl->out_scale || p.noise_mode || tl->act || tl->gain != 1.f || tl->residual) ? 1 : 0;
    }
    p.gain = p.tail ? p.gain : 1.f;
    hipStream_t st = (hipStream_t)stream;
    int dy[9], dx[9], slot[9];
    if (mode == 0) {
        SHG_CHECK_ARG(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1 && OH >= 1 && OW >= 1, "conv2d_f16: output extent");
        p.s_in = stride; p.s_out = 1; p.oy0 = 0; p.ox0 = 0;
        for (int t = 0; t < k * k; ++t) { dy[t] = t / k - pad; dx[t] = t % k - pad; slot[t] = t; }
        return f16::conv_taps(p, k * k, dy, dx, slot, OH, OW, st);
    }
    SHG_CHECK_ARG(mode == 1 && k == 3 && stride == 2 && crop >= 0 && OH >= 1 && OW >= 1, "conv2d_f16: the transposed form is 3x3 stride 2");
    // full[oy][ox] = sum_{ky,kx} x[(oy-ky)/2][(ox-kx)/2] w[ky][kx] over even (oy-ky), (ox-kx); phase (py,px) = parity of (oy,ox)
    if (f16::convt_upring_eligible(p)) return f16::convt_upring_launch(p, crop, st);      // all four phases from one pass over the input
    p.s_in = 1; p.s_out = 2;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            int nt = 0;
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2) { dy[nt] = -(ky / 2); dx[nt] = -(kx / 2); slot[nt] = ky * 3 + kx; ++nt; }
            p.oy0 = py - crop; p.ox0 = px - crop;
            // grid rows oy' with 0 <= 2 oy' + py < 2H+1 intersected with the crop window [crop, crop + OH)
            const int GH = py ? H : H + 1, GW = px ? W : W + 1;
            const int rc = f16::conv_taps(p, nt, dy, dx, slot, GH, GW, st);
            if (rc != SHG_OK) return rc;
        }
    return SHG_OK;
}


extern "C" int shg_conv2d_f16(const void* x, const void* w, const float* bias, void* y, int N, int I, int O, int H, int W, int k, int stride,
                              int pad, int mode, int crop, int OH, int OW, void* stream) {
    return conv2d_f16_impl(x, w, bias, y, N, I, O, H, W, k, stride, pad, mode, crop, OH, OW, nullptr, stream);
}

// The same convolution with the layer tail of the INFERENCE route fused (half layers under no_grad): x * in_scale[n,i] (half x half) while the
// patch is staged -- both modes --, and for mode 0: y = A(conv * out_scale[n,o] + noise * noise_strength + bias[o]) + residual applied to the
// half-rounded convolution result in the store pass (stylegan.py:173-181,298-304; comodgan.py:320-327).  in_scale [N,I], out_scale [N,O], noise
// [OH,OW] (noise_mode 1) / [N,OH,OW] (2), bias [O]: fp32, each optional; residual: halves like y; act = 0: (..) * gain.
extern "C" int shg_conv2d_f16_fused(const void* x, const void* w, void* y, int N, int I, int O, int H, int W, int k, int stride, int pad, int mode,
                                    int crop, int OH, int OW, const float* in_scale, const float* out_scale, const float* noise, int noise_mode,
                                    float noise_strength, const float* bias, int act, float alpha, float gain, float clamp, const void* residual,
                                    void* stream) {
    SHG_CHECK_ARG(mode == 0 || !(out_scale || noise || act || residual || gain != 1.f), "conv2d_f16_fused: the transposed form takes in_scale only (its tail follows the FIR)");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(out_scale) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "conv2d_f16: x, w, y, residual, out_scale and bias must be 16-byte aligned (vector loads)");
    const shg_f16_tail_ tl{in_scale, out_scale, noise, noise_mode, noise_strength, act, alpha, gain, clamp, residual};
    return conv2d_f16_impl(x, w, bias, y, N, I, O, H, W, k, stride, pad, mode, crop, OH, OW, &tl, stream);
}

// w [T][O][I] halves (T tap slots) -> MFMA operand order [ceil(O/32)][T][I/16][64][8], rows beyond O zero
__global__ __launch_bounds__(256) void pack_weight_f16_kernel(const _Float16* w, _Float16* wp, int T, int O, int I, long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int c16n = I >> 4, c16 = (int)(r % c16n);
        r /= c16n;
        const int t = (int)(r % T), ob = (int)(r / T);
        const int o = ob * 32 + (lane & 31), i = c16 * 16 + (lane >> 5) * 8 + el;
        wp[e] = o < O ? w[((long)t * O + o) * I + i] : (_Float16)0.f;
    }
}

// (32-channel blocks rounded up to a multiple of 4: the kernel loads the operands of whole groups of MB <= 4 blocks unconditionally)
extern "C" long shg_conv2d_f16_packed_weight_elems(int T, int O, int I) { return (long)(((O + 31) / 32 + 3) / 4 * 4) * T * (I / 16) * 512; }

extern "C" int shg_conv2d_f16_pack_weight(const void* w, void* wp, int T, int O, int I, void* stream) {
    SHG_CHECK_ARG(w && wp && T >= 1 && O >= 1 && I >= 32 && (I % 32) == 0, "conv2d_f16_pack_weight: bad arguments (I must be a multiple of 32)");
    const long total = shg_conv2d_f16_packed_weight_elems(T, O, I);
    int grid = shg_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weight_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w, (_Float16*)wp, T, O, I, total);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// The same operand order straight from a torch-layout weight: src [A][B][T] halves (T = k*k), logical W[o][i][t] = src[o][i][t] (A = O, B = I)
// or, `transposed`, src[i][o][t] (A = I, B = O: the conv_transpose2d layout / the channel-transposed weight of an input gradient); `flip`
// reverses the taps (t -> T-1-t: the 180-degree rotation of an input gradient).  Input channels beyond I read zero (Ip = I rounded up to 32).
// One gather pass instead of flip + permute-copy + pack (three launches per fp16 data-gradient convolution).
__global__ __launch_bounds__(256) void pack_weight_oihw_f16_kernel(const _Float16* src, _Float16* wp, int T, int O, int I, int Ip, int transposed, int flip,
                                                                   long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int c16n = Ip >> 4, c16 = (int)(r % c16n);
        r /= c16n;
        const int t = (int)(r % T), ob = (int)(r / T);
        const int o = ob * 32 + (lane & 31), i = c16 * 16 + (lane >> 5) * 8 + el;
        const int ts = flip ? T - 1 - t : t;
        _Float16 v = (_Float16)0.f;
        if (o < O && i < I) v = transposed ? src[((long)i * O + o) * T + ts] : src[((long)o * I + i) * T + ts];
        wp[e] = v;
    }
}

extern "C" int shg_conv2d_f16_pack_weight_oihw(const void* src, void* wp, int T, int O, int I, int transposed, int flip, void* stream) {
    SHG_CHECK_ARG(src && wp && (T == 1 || T == 9) && O >= 1 && I >= 1, "conv2d_f16_pack_weight_oihw: bad arguments");
    const int Ip = (I + 31) / 32 * 32;
    const long total = shg_conv2d_f16_packed_weight_elems(T, O, Ip);
    int grid = shg_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weight_oihw_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (_Float16*)wp, T, O, I, Ip,
                       transposed, flip, total);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// the transposed form writes every pixel of its crop window only where the (2H+1) x (2W+1) result exists: callers zero y first when
// crop + OH > 2H + 1 (shg_conv2d_f16_needs_clear says so)
extern "C" int shg_conv2d_f16_needs_clear(int H, int W, int crop, int OH, int OW) {
    return (crop + OH > 2 * H + 1 || crop + OW > 2 * W + 1) ? 1 : 0;
}

// pixel slices of the fp16 weight gradient: workgroups in total / (64 x 64 tiles)
// (measured at 256 / 512 / 768 / 1024 workgroups: 3x3 best at 512 = two resident per CU in one round, 467 / 531 / 467 / 478 TFLOP/s at 64 channels
// stride 1, 372 / 449 / 381 at stride 2 -- whose 260-register build held ONE workgroup per CU (346 TFLOP/s) until amdgpu_waves_per_eu(2) brought it to
// 255 without spills; the load-bound 1x1 layers at 1024: 126 / 211 / 237 / 253 at 512 channels)
static long wgrad_f16_slices(long tiles, long nblocks, long wgs) {
#ifdef SHG_ABLATE
    if (const char* e = getenv("SHG_WGRAD16_WGS")) wgs = atol(e);       // study switch (python sh-gan_amd/build.py --ablate: tools/_variants)
#endif
    long slices = (wgs + tiles - 1) / tiles;
    if (slices > nblocks) slices = nblocks;
    return slices < 1 ? 1 : slices;
}

extern "C" size_t shg_conv2d_wgrad_f16_workspace_bytes(int N, int I, int O, int OH, int OW, int k) {
    const int OP = (O + 63) / 64 * 64, IP = (I + 63) / 64 * 64;
    const long nblocks = (long)N * shg_cdiv(OH, f16::WR) * shg_cdiv(OW, f16::WC);
    return (size_t)wgrad_f16_slices((long)(OP / 64) * (IP / 64), nblocks, k == 1 ? 1024 : 512) * k * k * OP * IP * sizeof(float);
}

// dw [k*k][O][I] fp32 = sum_{n,oy,ox} g[n,oy,ox,o] x[n, oy*stride - pad + ky, ox*stride - pad + kx, i]; x [N,H,W,I], g [N,OH,OW,O] halves
extern "C" int shg_conv2d_wgrad_f16(const void* x, const void* g, float* dw, int N, int I, int O, int H, int W, int OH, int OW, int k, int stride,
                                    int pad, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && g && dw && workspace, "conv2d_wgrad_f16: null pointer");
    SHG_CHECK_ARG(N >= 1 && I >= 8 && (I % 8) == 0 && O >= 8 && (O % 8) == 0, "conv2d_wgrad_f16: I and O must be multiples of 8");
    SHG_CHECK_ARG((k == 1 || k == 3) && (stride == 1 || stride == 2) && pad >= 0, "conv2d_wgrad_f16: 1x1 / 3x3 kernels, stride 1 / 2");
    SHG_CHECK_ARG(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1, "conv2d_wgrad_f16: output extent");
    SHG_CHECK_ARG(ws_bytes >= shg_conv2d_wgrad_f16_workspace_bytes(N, I, O, OH, OW, k), "conv2d_wgrad_f16: workspace too small");
    f16::WgradP p{};
    p.x = (const _Float16*)x; p.g = (const _Float16*)g; p.part = (float*)workspace;
    p.N = N; p.I = I; p.O = O; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.s = stride; p.pad = pad; p.k = k;
    p.OP = (O + 63) / 64 * 64; p.IP = (I + 63) / 64 * 64;
    const int wr = f16::wgrad_rows(k, stride);
    p.by = shg_cdiv(OH, wr); p.bx = shg_cdiv(OW, f16::WC);
    p.nblocks = (long)N * p.by * p.bx;
    const long tiles = (long)(p.OP / 64) * (p.IP / 64);
    const long slices = wgrad_f16_slices(tiles, p.nblocks, k == 1 ? 1024 : 512);
    p.slices = (int)slices;
    SHG_CHECK_ARG(!(k == 1 && stride == 2), "conv2d_wgrad_f16: 1x1 stride-2 (the forward decimates with upfirdn2d first)");
    p.XR = (wr - 1) * stride + k; p.XC = (f16::WC - 1) * stride + k;
    const size_t lds = ((size_t)64 * (wr * f16::WC + 8) + (size_t)64 * (p.XR * (stride == 1 ? 24 : 48) + 8)) * sizeof(_Float16);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)tiles, (unsigned)slices);
    if (k == 3 && stride == 1) hipLaunchKernelGGL((f16::conv_wgrad_f16_kernel<3, 1>), grid, dim3(256), lds, st, p);
    else if (k == 3) hipLaunchKernelGGL((f16::conv_wgrad_f16_kernel<3, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((f16::conv_wgrad_f16_kernel<1, 1>), grid, dim3(256), lds, st, p);
    SHG_CHECK_LAUNCH();
    const long total = (long)k * k * O * I;
    int G = 1;
    while (G < 16 && (total + 256 / G - 1) / (256 / G) < 2048 && p.slices >= 8 * G) G *= 2;
    const dim3 rgrid((unsigned)((total + 256 / G - 1) / (256 / G)));
    const float* part = (const float*)workspace;
    switch (G) {
        case 1: hipLaunchKernelGGL(f16::wgrad_reduce_kernel<1>, rgrid, dim3(256), 0, st, part, dw, p.slices, k * k, O, I, p.OP, p.IP); break;
        case 2: hipLaunchKernelGGL(f16::wgrad_reduce_kernel<2>, rgrid, dim3(256), 0, st, part, dw, p.slices, k * k, O, I, p.OP, p.IP); break;
        case 4: hipLaunchKernelGGL(f16::wgrad_reduce_kernel<4>, rgrid, dim3(256), 0, st, part, dw, p.slices, k * k, O, I, p.OP, p.IP); break;
        case 8: hipLaunchKernelGGL(f16::wgrad_reduce_kernel<8>, rgrid, dim3(256), 0, st, part, dw, p.slices, k * k, O, I, p.OP, p.IP); break;
        default: hipLaunchKernelGGL(f16::wgrad_reduce_kernel<16>, rgrid, dim3(256), 0, st, part, dw, p.slices, k * k, O, I, p.OP, p.IP); break;
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// upfirdn2d on NHWC halves: the semantics of shg_upfirdn2d_f32 / upfirdn2d.cu:29-92; C % 8 == 0
extern "C" int shg_upfirdn2d_f16(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy, int downx,
                                 int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream) {
    SHG_CHECK_ARG(x && f && y, "upfirdn2d_f16: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 8 && (C % 8) == 0 && H >= 1 && W >= 1, "upfirdn2d_f16: bad shape (C must be a multiple of 8)");
    SHG_CHECK_ARG(fh >= 1 && fw >= 1 && fh * fw <= 64 && upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "upfirdn2d_f16: bad filter / factors");
    const int OW = (W * upx + padx0 + padx1 - fw + downx) / downx, OH = (H * upy + pady0 + pady1 - fh + downy) / downy;
    SHG_CHECK_ARG(OW >= 1 && OH >= 1, "upfirdn2d_f16: empty output");
    f16::UfdH p{(const _Float16*)x, f, (_Float16*)y, N, C, H, W, OH, OW, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain};
    const bool same = upx == 1 && upy == 1 && downx == 1 && downy == 1;
    const int rb = 2;                                        // (one row per lane was measured too: 287 vs 221 us at 64 ch x 513^2)
    const long total = same ? (long)N * ((OH + rb - 1) / rb) * ((OW + 3) / 4) * (C / 8) : (long)N * OH * OW * (C / 8);
    int grid = shg_cdiv(total, 256);
    if (grid > 256 * 32) grid = 256 * 32;
#ifdef SHG_ABLATE
    const bool march = !getenv("SHG_F16_FIR_OLD");                     // study switch: the 2 x 4-pixel kernel instead
#else
    const bool march = true;
#endif
    if (same && fh == 4 && fw == 4 && march) {
        // marching strips: the longest of 32 / 16 / 8 / 4 rows that still leaves >= 8 workgroups per CU
        int rows = 32;
        auto lanes = [&](int r) { return (long)N * ((OH + r - 1) / r) * ((OW + 1) / 2) * (C / 8); };
        while (rows > 4 && lanes(rows) < 256L * 256 * 8) rows /= 2;
        SHG_CHECK_ARG(lanes(rows) < (1L << 31), "upfirdn2d_f16: tensor too large");
        hipLaunchKernelGGL(f16::fir4_march_f16_kernel, dim3((unsigned)shg_cdiv(lanes(rows), 256)), dim3(256), 0, (hipStream_t)stream, p, rows);
    } else if (same && fh == 4 && fw == 4) hipLaunchKernelGGL((f16::fir_same_f16_kernel<4, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (same) hipLaunchKernelGGL((f16::fir_same_f16_kernel<0, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (fh == 4 && fw == 4 && upx == 1 && upy == 1 && downx == 2 && downy == 2 && total < (1L << 31))
        hipLaunchKernelGGL((f16::updn4_f16_kernel<1, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (fh == 4 && fw == 4 && upx == 2 && upy == 2 && downx == 1 && downy == 1 && total < (1L << 31))
        hipLaunchKernelGGL((f16::updn4_f16_kernel<2, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(f16::upfirdn2d_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// src float32 [N,C,HW] (NCHW) -> dst float16 [N,HW,C] (NHWC) when to_half, the reverse otherwise; C % 8 == 0.
extern "C" int shg_relayout_f32_f16(const void* src, void* dst, int N, int C, long HW, int to_half, void* stream) {
    SHG_CHECK_ARG(src && dst && N >= 1 && C >= 1 && (C % 8 == 0 || C <= 16) && HW >= 1 && HW <= 0x7fffffffL && N <= 65535, "relayout: C % 8 == 0 or C <= 16");
    if (C % 8) {
        const long total = (long)N * HW;
        long grid = (total + 255) / 256;
        if (grid > 65536) grid = 65536;
        if (to_half) hipLaunchKernelGGL(f16::relayout_thin_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, total);
        else hipLaunchKernelGGL(f16::relayout_thin_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, total);
        SHG_CHECK_LAUNCH();
        return SHG_OK;
    }
    SHG_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "relayout: 16-byte aligned tensors");
    const dim3 grid((unsigned)((HW + 63) / 64), (C + 63) / 64, N);
    if (to_half) hipLaunchKernelGGL(f16::relayout_to_half_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (_Float16*)dst, C, (int)HW);
    else hipLaunchKernelGGL(f16::relayout_to_float_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (float*)dst, C, (int)HW);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

extern "C" int shg_bias_act_f16(const void* x, const float* bias, void* y, long pixels, int C, int act, float alpha, float gain, float clamp,
                                void* stream) {
    SHG_CHECK_ARG(x && y, "bias_act_f16: null pointer");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "bias_act_f16: x, y and bias must be 16-byte aligned");
    SHG_CHECK_ARG(pixels >= 0 && C >= 8 && (C % 8) == 0, "bias_act_f16: C must be a multiple of 8");
    const long total8 = pixels * (C / 8);
    if (total8 == 0) return SHG_OK;
    int grid = (int)shg_cdiv(total8, 256L * f16::EU);
    if (grid > 256 * 16) grid = 256 * 16;
    if (256 % (C / 8) == 0)
        hipLaunchKernelGGL(f16::bias_act_f16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, bias, (_Float16*)y, total8, C,
                           act, alpha, gain, clamp);
    else
        hipLaunchKernelGGL(f16::bias_act_f16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, bias, (_Float16*)y, total8, C,
                           act, alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

extern "C" int shg_bias_act_backward_f16(const void* g, const void* y, void* dx, long total, int act, float alpha, float gain, float clamp,
                                         void* stream) {
    SHG_CHECK_ARG(g && y && dx, "bias_act_backward_f16: null pointer");
    SHG_CHECK_ARG(total >= 0 && (total % 8) == 0, "bias_act_backward_f16: element count must be a multiple of 8");
    if (total == 0) return SHG_OK;
    int grid = shg_cdiv(total / 8, 256);
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(f16::bias_act_backward_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)g, (const _Float16*)y,
                       (_Float16*)dx, total / 8, act, alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// y = A(t * d[n,c] + noise + bias[c]) on NHWC halves: d fp32 [N,C] or NULL, noise fp32 [HW] (noise_mode 1) / [N,HW] (2) / none (0), bias fp32
// [C] or NULL; act = 0: (..) * gain.  C a multiple of 8.
extern "C" int shg_modtail_f16(const void* t, const float* d, const float* noise, int noise_mode, const float* bias, void* y, int N, long HW, int C,
                               int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(t && y && N >= 1 && HW >= 1 && C >= 8 && (C % 8) == 0, "modtail_f16: bad arguments (C must be a multiple of 8)");
    SHG_CHECK_ARG(HW < 2147483647L, "modtail_f16: H*W must fit 31 bits");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "modtail_f16: t, y, d and bias must be 16-byte aligned");
    f16::TailP p{};
    p.t = (const _Float16*)t; p.d = d; p.noise = noise_mode ? noise : nullptr; p.noise_mode = noise ? noise_mode : 0; p.bias = bias;
    p.out = (_Float16*)y; p.N = N; p.HW = (int)HW; p.C = C; p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    long blocks = (HW * (C / 8) + 256 * f16::EU - 1) / (256 * f16::EU);
    if (blocks > 2048) blocks = 2048;
    if (256 % (C / 8) == 0) hipLaunchKernelGGL(f16::modtail_f16_kernel<true>, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(f16::modtail_f16_kernel<false>, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// workgroups per sample of shg_modtail_backward_f16 (rows of its `part` buffer)
extern "C" int shg_modtail_backward_f16_blocks(long HW, int C) {
    long blocks = (HW * (C / 8) + 255) / 256;
    return (int)(blocks > 256 ? 256 : blocks);
}

// gt = gy * A'(y) * d [+ u * A'(y) * e] (halves); part [N][blocks][2][C] fp32 = per-workgroup sums over pixels of gz*t and gz (NULL: skipped; t may be
// NULL when only sum gz is wanted); gnoise [N,HW] fp32 = sum over channels of gz (NULL: skipped); u (halves like gy) / e [N,C] fp32: the optional second
// product of the tail's double backward.  C in {8,16,...,512} with C/8 a power of two.
extern "C" int shg_modtail_backward_f16(const void* gy, const void* y, const void* t, const float* d, const void* u, const float* e, void* gt, float* part,
                                        float* gnoise, int N, long HW, int C, int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(gy && y && gt && N >= 1 && HW >= 1, "modtail_backward_f16: null pointer / empty");
    SHG_CHECK_ARG(HW < 2147483647L, "modtail_backward_f16: H*W must fit 31 bits");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(gt) |
                    reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(u)) & 15) == 0, "modtail_backward_f16: gy, y, t, u, gt and d must be 16-byte aligned (vector accesses)");
    const int c8n = C / 8;
    SHG_CHECK_ARG(C >= 8 && (C % 8) == 0 && c8n <= 64 && (c8n & (c8n - 1)) == 0, "modtail_backward_f16: C/8 must be a power of two <= 64");
    f16::TailP p{};
    p.gy = (const _Float16*)gy; p.y_in = (const _Float16*)y; p.t = (const _Float16*)t; p.d = d; p.out = (_Float16*)gt; p.part = part;
    p.u = (const _Float16*)u; p.e = e;
    p.gnoise = gnoise; p.N = N; p.HW = (int)HW; p.C = C; p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.nblk = shg_modtail_backward_f16_blocks(HW, C);
    hipLaunchKernelGGL(f16::modtail_backward_f16_kernel, dim3(p.nblk, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
