// Weight gradient of the stride-1 3x3 'same' convolutions in the Winograd domain, on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// Replaces, for these layers, the direct kernel of conv_wgrad.hip (reference: the cuDNN call inside Conv2dGradWeight, conv2d_gradfix.py:140-146).
// The transpose of F(4x4, 3x3) (conv_wino4.hip: Y = A^T [(G w G^T) .* (B^T d B)] A per 4x4 output tile and 6x6 input patch d):
//      dL/dw = G^T [ sum over tiles  (A dY A^T) .* (B^T d B) ] G
// -- 36 multiplications per tile and (o, i) pair instead of the 144 of the direct sum.  Both operands are transformed in the kernel.
//
// GEMM view: 36 independent products M_p[o, i] = sum_tiles Z_p[o, tile] * V_p[i, tile]; k = tiles.  A workgroup (8 waves) owns 64 output x 32
// input channels for one slice of the tiles, wave (os, is) the 16 x 16 block (os, is) of it with ALL 36 positions in registers (36 accumulator
// tiles of 4 registers): lane (c, t) of a k-step holds channel c of both sides and tile t of four -- it reads ITS 4x4 gradient tile and 6x6
// input patch from LDS, transforms them in registers (one pass down the columns, then row by row) and the results ARE the MFMA operands
// (A[o = c][k = t], B[k = t][i = c]): no exchange of transformed data between waves, no cross-wave reduction, one barrier per chunk.
// The price is that a gradient tile is transformed by the two waves that share its channels and an input patch by four (244 VALU
// instructions per 36 MFMAs of 32 cycles -- the two waves of a SIMD overlap one's transforms with the other's products).
// A chunk = 8 tiles (1 x 8 or 2 x 4; images narrower than 16 columns: 4 tiles, 2 x 2 or 4 x 1) of one image: the raw 4 TY x 4 TX gradient window of 64 channels and the (4 TY + 2) x (4 TX + 8) input
// window of 32 channels arrive by 16-byte LDS-DMA (`buffer_load ... lds`; the descriptor's range check writes the zero padding), double
// buffered, 16 channels interleaved per piece so that the 16 channels of a ds_read_b128 group hit 16 consecutive 16-byte slots (the
// interleave is done on the GLOBAL side of the DMA: lane = (piece, channel)).  The epilogue applies G^T . G per accumulator element (all 36 positions of an (o, i) pair live in one lane) and writes
// 9 partial sums; tile slices are added by conv_wgrad.hip's fixed-order reduction (deterministic).
#include "shg_common.h"

typedef float ww_f4 __attribute__((ext_vector_type(4)));
typedef int ww_i4 __attribute__((ext_vector_type(4)));

void shg_launch_wgrad_reduce(const float* part, float* dw, long n, int nslice, hipStream_t s);      // conv_wgrad.hip

struct WgWinoP {
    const float* x;      // [NB, I, H, W]
    const float* g;      // [NB, O, H, W]
    float* out;          // dw [O, I, 3, 3] (nslice == 1) or partials [nslice][O][I][9]
    int NB, I, O, H, W;
    int lx;              // log2 of the tiles per chunk row: chunk = TY x TX tiles = 1 x 8, 2 x 4 (W < 32) or 2 x 2, 4 x 1 (W < 16: four tiles, one k-step)
    int cty, ctx;        // chunks per image
    int nchunk, nslice;
};

namespace wgw {
constexpr int BO = 64, BI = 32;
constexpr int G_BYTES = BO * 32 * 16, X_BYTES = BI * 64 * 16, STAGE = G_BYTES + X_BYTES;     // 32 KiB + 32 KiB
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ ww_i4 make_srd(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    ww_i4 s;
    s[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    s[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    s[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s[3] = 0x00020000;
    return s;
}
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, ww_i4 srd) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd) : "memory");
}

// 1-D transforms.  Input side: B^T d (conv_wino4.hip's wino4_bt); gradient side: A e with A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float (&o)[6]) {
    const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    o[0] = 4.f * d0 - 5.f * d2 + d4;
    o[1] = a + b; o[2] = a - b;
    o[3] = c + e; o[4] = c - e;
    o[5] = 4.f * d1 - 5.f * d3 + d5;
}
__device__ __forceinline__ void a6(float e0, float e1, float e2, float e3, float (&o)[6]) {
    const float s02 = e0 + e2, s13 = e1 + e3, t = e0 + 4.f * e2, u = 2.f * (e1 + 4.f * e3);
    o[0] = e0;
    o[1] = s02 + s13; o[2] = s02 - s13;
    o[3] = t + u; o[4] = t - u;
    o[5] = e3;
}
// G^T m, G^T = [[1/4,-1/6,-1/6,1/24,1/24,0],[0,-1/6,1/6,1/12,-1/12,0],[0,-1/6,-1/6,1/6,1/6,1]]
__device__ __forceinline__ void gt3(const float (&m)[6], float (&o)[3]) {
    const float s12 = m[1] + m[2], s34 = m[3] + m[4];
    o[0] = 0.25f * m[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    o[1] = (1.f / 6.f) * (m[2] - m[1]) + (1.f / 12.f) * (m[3] - m[4]);
    o[2] = (1.f / 6.f) * (s34 - s12) + m[5];
}

template <int LX>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_kernel(const WgWinoP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TX = 1 << LX, TPC = LX >= 2 ? 8 : 4, TY = TPC >> LX, XP = TX + 2;      // tiles per chunk row / chunk / column; 16-byte pieces per input-window row
    static_assert((4 * TY + 2) * XP <= 64 && 4 * TY * TX <= 32, "a channel's windows are at most 64 / 32 pieces");
    const int tid = threadIdx.x, lane = tid & 63, cc = lane & 15, tq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int osub = wave & 3, isub = wave >> 2;
    const int i0 = blockIdx.x * BI, o0 = blockIdx.y * BO, slice = blockIdx.z;
    const int per = (p.nchunk + p.nslice - 1) / p.nslice;
    const int c_begin = slice * per, c_end = min(p.nchunk, c_begin + per);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int HW = p.H * p.W;

    // ---- LDS image of a stage: gradient window [4 channel groups][32 pieces][16 channels] x 16 B, input window [2 groups][64 pieces][16
    // channels] x 16 B (piece q = row * pieces-per-row + column piece): the 16 channels of a ds_read_b128 group sit in 16 consecutive 16-byte
    // slots (conflict-free) and a lane's pieces are 256 q bytes apart (immediate offsets from one base register).  A DMA request moves four
    // pieces of 16 channels (lane = (piece, channel)); wave w issues requests w, w + 8, w + 16, w + 24 of either window.
    auto issue = [&](int c, int buf) __attribute__((always_inline)) {
        const int cx = c % p.ctx, t = c / p.ctx, cy = t % p.cty, n = t / p.cty;
        const int gy0 = cy * 4 * TY, gx0 = cx * 4 * TX;        // window origin in the gradient; the input window starts at (gy0 - 1, gx0 - 4)
        const ww_i4 srd_g = make_srd(p.g + (long)n * p.O * HW, (unsigned)((long)p.O * HW * 4));
        const ww_i4 srd_x = make_srd(p.x + (long)n * p.I * HW, (unsigned)((long)p.I * HW * 4));
        const unsigned base = lds0 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                           // request k = wave + 8 j: channel group j, pieces 4 wave ..
            const int q = 4 * wave + tq, o = o0 + 16 * j + cc, gy = gy0 + (q >> LX), gx = gx0 + (q & (TX - 1)) * 4;
            const bool ok = (q >> LX) < 4 * TY && o < p.O && gy < p.H && gx < p.W;
            dma16(base + (wave + 8 * j) * 1024, ok ? (unsigned)((o * HW + gy * p.W + gx) * 4) : OOB, srd_g);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                           // request k = wave + 8 j: channel group j / 2, pieces 4 (wave + 8 (j & 1)) ..
            const int q = 4 * (wave + 8 * (j & 1)) + tq, row = q / XP, px = q - row * XP, i = i0 + 16 * (j >> 1) + cc;
            const int iy = gy0 - 1 + row, ix = gx0 - 4 + px * 4;
            const bool ok = row < 4 * TY + 2 && i < p.I && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            dma16(base + G_BYTES + (wave + 8 * j) * 1024, ok ? (unsigned)((i * HW + iy * p.W + ix) * 4) : OOB, srd_x);
        }
    };

    ww_f4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = ww_f4{0.f, 0.f, 0.f, 0.f};

    // this lane's channels: osub * 16 + cc of the gradient block, isub * 16 + cc of the input block
    const unsigned g_lane = (osub * 32 * 16 + cc) * 16, x_lane = G_BYTES + (isub * 64 * 16 + cc) * 16;
    if (c_begin < c_end) issue(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");       // chunk c landed; every wave is done with the other buffer
        const int buf = (c - c_begin) & 1;
        if (c + 1 < c_end) issue(c + 1, buf ^ 1);
#pragma unroll 1
        for (int s = 0; s < TPC / 4; ++s) {
            // tile j = 4 s + tq of the chunk at (jy, jx) = (j / TX, j % TX): first gradient piece 4 jy TX + jx, first input piece 4 jy XP + jx
            const int j = 4 * s + tq, qg = ((j >> LX) << (2 + LX)) + (j & (TX - 1)), qx = (j >> LX) * 4 * XP + (j & (TX - 1));
            const unsigned char* gs = lds + buf * STAGE + g_lane + qg * 256;
            const unsigned char* xs = lds + buf * STAGE + x_lane + qx * 256;
            // gradient tile: 4 rows of one piece; down the columns first: S[a][col] = sum_r A[a][r] e[r][col]
            float S[6][4];
            {
                ww_f4 e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = *(const ww_f4*)(gs + r * TX * 256);
#pragma unroll
                for (int col = 0; col < 4; ++col) {
                    float o6[6];
                    a6(e[0][col], e[1][col], e[2][col], e[3][col], o6);
#pragma unroll
                    for (int a = 0; a < 6; ++a) S[a][col] = o6[a];
                }
            }
            // input patch: rows 4 jy .. 4 jy + 5 of the window, floats 4 jx + 3 .. 4 jx + 8 of a row (pieces jx, jx + 1, jx + 2 as three
            // b128 reads: two b32 reads for the outer floats would be 4-way bank conflicts in this layout)
            float T[6][6];
            {
                float d[6][6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const ww_f4 m0 = *(const ww_f4*)(xs + (r * XP + 0) * 256), m1 = *(const ww_f4*)(xs + (r * XP + 1) * 256), m2 = *(const ww_f4*)(xs + (r * XP + 2) * 256);
                    d[r][0] = m0[3]; d[r][1] = m1[0]; d[r][2] = m1[1]; d[r][3] = m1[2]; d[r][4] = m1[3]; d[r][5] = m2[0];
                }
#pragma unroll
                for (int col = 0; col < 6; ++col) {
                    float o6[6];
                    bt6(d[0][col], d[1][col], d[2][col], d[3][col], d[4][col], d[5][col], o6);
#pragma unroll
                    for (int a = 0; a < 6; ++a) T[a][col] = o6[a];
                }
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float z[6], v[6];
                a6(S[a][0], S[a][1], S[a][2], S[a][3], z);
                bt6(T[a][0], T[a][1], T[a][2], T[a][3], T[a][4], T[a][5], v);
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[a * 6 + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[b], v[b], acc[a * 6 + b], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // C/D layout of the 16x16 forms: column (i) = lane & 15, row (o) = 4 (lane >> 4) + register
    float* dst = p.out + (p.nslice > 1 ? (long)slice * p.O * p.I * 9 : 0);
    const int i = i0 + isub * 16 + cc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + osub * 16 + 4 * tq + r;
        float h[3][6];                                          // G^T M: down the first index
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const float m[6] = {acc[b][r], acc[6 + b][r], acc[12 + b][r], acc[18 + b][r], acc[24 + b][r], acc[30 + b][r]};
            float o3[3];
            gt3(m, o3);
            h[0][b] = o3[0]; h[1][b] = o3[1]; h[2][b] = o3[2];
        }
        if (o < p.O && i < p.I) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float o3[3];
                gt3(h[ky], o3);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) dst[((long)o * p.I + i) * 9 + ky * 3 + kx] = o3[kx];
            }
        }
    }
}
}  // namespace wgw

static bool wgw_eligible(int H, int W, int OH, int OW, int kh, int kw, int stride, int pad) {
    return kh == 3 && kw == 3 && stride == 1 && pad == 1 && OH == H && OW == W && (W & 3) == 0 && W >= 4 && H >= 4;
}

static int wgw_lx(int W) { return W >= 32 ? 3 : (W >= 16 ? 2 : (W >= 8 ? 1 : 0)); }
static int wgw_ty(int lx) { return (lx >= 2 ? 8 : 4) >> lx; }

static int wgw_slices(int NB, int I, int O, int H, int W) {
    const int lx = wgw_lx(W);
    const long blocks = (long)shg_cdiv(I, wgw::BI) * shg_cdiv(O, wgw::BO);
    const long nchunk = (long)NB * shg_cdiv(H, 4 * wgw_ty(lx)) * shg_cdiv(W, 4 << lx);
    long s = (256 + blocks - 1) / blocks;                        // one workgroup per CU (2 x 64 KiB of LDS), one round
    if (s > nchunk) s = nchunk;
    if (s > 1024) s = 1024;
    return s < 1 ? 1 : (int)s;
}

// 1 when shg_conv2d_wgrad_wino_f32 serves this geometry (3x3, stride 1, pad 1, rows of whole 16-byte pieces, at least 16 columns)
extern "C" int shg_conv2d_wgrad_wino_supported(int H, int W, int OH, int OW, int kh, int kw, int stride, int pad) {
    return wgw_eligible(H, W, OH, OW, kh, kw, stride, pad) ? 1 : 0;
}

extern "C" size_t shg_conv2d_wgrad_wino_workspace_bytes(int NB, int I, int O, int H, int W) {
    const int s = wgw_slices(NB, I, O, H, W);
    return s > 1 ? (size_t)s * O * I * 9 * sizeof(float) : 0;
}

// dw [O, I, 3, 3] = weight gradient of y = conv2d(x [NB,I,H,W], w, stride 1, pad 1) given g = dL/dy [NB,O,H,W], in the Winograd domain
// (F(4x4,3x3) transposed; fp32 MFMA, fp32 accumulation; about 5e-6 relative against float64 -- the direct kernel's class).
extern "C" int shg_conv2d_wgrad_wino_f32(const float* x, const float* g, float* dw, int NB, int I, int O, int H, int W, void* workspace,
                                         size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && g && dw, "conv2d_wgrad_wino: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1, "conv2d_wgrad_wino: bad shape");
    SHG_CHECK_ARG(wgw_eligible(H, W, H, W, 3, 3, 1, 1), "conv2d_wgrad_wino: 3x3 stride-1 pad-1 layers with W %% 4 == 0, W >= 4, H >= 4");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0, "conv2d_wgrad_wino: x and g must be 16-byte aligned");
    SHG_CHECK_ARG((long)I * H * W * 4 < (1L << 31) && (long)O * H * W * 4 < (1L << 31), "conv2d_wgrad_wino: an image of x / g must stay below 2 GiB");
    WgWinoP p{};
    p.x = x; p.g = g; p.NB = NB; p.I = I; p.O = O; p.H = H; p.W = W;
    p.lx = wgw_lx(W);
    p.cty = shg_cdiv(H, 4 * wgw_ty(p.lx)); p.ctx = shg_cdiv(W, 4 << p.lx);
    p.nchunk = NB * p.cty * p.ctx;
    p.nslice = wgw_slices(NB, I, O, H, W);
    const size_t need = p.nslice > 1 ? (size_t)p.nslice * O * I * 9 * sizeof(float) : 0;
    SHG_CHECK_ARG(need == 0 || (workspace && ws_bytes >= need), "conv2d_wgrad_wino: workspace too small (shg_conv2d_wgrad_wino_workspace_bytes)");
    p.out = p.nslice > 1 ? (float*)workspace : dw;
    static ShgDeviceOnce attr_once;
    const int dev_now = shg_current_device();
    if (attr_once.pending(dev_now)) {
        if (hipFuncSetAttribute((const void*)wgw::conv_wgrad_wino_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wgw::STAGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgw::conv_wgrad_wino_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wgw::STAGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgw::conv_wgrad_wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wgw::STAGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgw::conv_wgrad_wino_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wgw::STAGE) != hipSuccess) {
            shg_set_error("conv2d_wgrad_wino: cannot reserve %d bytes of LDS", 2 * wgw::STAGE);
            return SHG_ERR_LAUNCH;
        }
        attr_once.mark(dev_now);
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(shg_cdiv(I, wgw::BI), shg_cdiv(O, wgw::BO), p.nslice);
    if (p.lx == 3) hipLaunchKernelGGL(wgw::conv_wgrad_wino_kernel<3>, grid, dim3(512), 2 * wgw::STAGE, s, p);
    else if (p.lx == 2) hipLaunchKernelGGL(wgw::conv_wgrad_wino_kernel<2>, grid, dim3(512), 2 * wgw::STAGE, s, p);
    else if (p.lx == 1) hipLaunchKernelGGL(wgw::conv_wgrad_wino_kernel<1>, grid, dim3(512), 2 * wgw::STAGE, s, p);
    else hipLaunchKernelGGL(wgw::conv_wgrad_wino_kernel<0>, grid, dim3(512), 2 * wgw::STAGE, s, p);
    SHG_CHECK_LAUNCH();
    if (p.nslice > 1) {
        shg_launch_wgrad_reduce((const float*)workspace, dw, (long)O * I * 9, p.nslice, s);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
