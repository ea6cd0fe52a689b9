/*
 * libshgan_hip.so -- C ABI of the MI355X-native SH-GAN generator-forward kernels.
 *
 * Drop-in boundary: everything the reference reaches through its native plugin
 * (`upfirdn2d_plugin.upfirdn2d`, lib/model_zoo/stylegan_utils/upfirdn2d.cpp:16,98-101) or through
 * ATen/cuDNN from the Python ops of lib/model_zoo/{stylegan,shgan}.py is exported here with plain
 * pointers and sizes.  Conventions:
 *   - all tensors are dense NCHW fp32 device buffers owned by the caller (outputs and workspaces
 *     included); no hidden allocation, no synchronisation; work is enqueued on `stream`
 *     (a hipStream_t passed as void*), like at::cuda::getCurrentCUDAStream() in upfirdn2d.cpp:91;
 *   - return value 0 = ok, <0 = error (SHG_ERR_*); shg_last_error() gives the message
 *     (the reference raises RuntimeError from TORCH_CHECK / AT_CUDA_CHECK, upfirdn2d.cpp:19-36,92);
 *   - thread-safe for distinct streams; the error string is thread-local.
 */
#ifndef SHGAN_HIP_H
#define SHGAN_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SHG_OK 0
#define SHG_ERR_ARG (-1)
#define SHG_ERR_LAUNCH (-2)
#define SHG_ERR_UNSUPPORTED (-3)

int shg_abi_version(void);
const char* shg_last_error(void);
/* gcnArchName of device `dev` into buf; returns the CU count (256 on MI355X) or <0. */
int shg_device_info(int dev, char* buf, int buflen);

/* ---- A10: upfirdn2d -- replaces upfirdn2d.cpp:16 `upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)`.
 * x [N,C,H,W], f [fh,fw], y [N,C,OH,OW] with OH/OW from shg_upfirdn2d_out_size (rule of upfirdn2d.cpp:32-33). */
int shg_upfirdn2d_out_size(int H, int W, int fh, int fw, int upx, int upy, int downx, int downy, int padx0, int padx1,
                           int pady0, int pady1, int* OH, int* OW);
int shg_upfirdn2d_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy,
                      int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream);
/* the same operator over the plugin's whole operand range (upfirdn2d.cpp:38-59: any strides, AT_DISPATCH_FLOATING_TYPES_AND_HALF):
 * dtype 0 float32 / 1 float16 / 2 float64 (accumulation float, float64: double); x_strides / y_strides = element strides (n, c, h, w);
 * f float32 [fh,fw] with element strides f_stride_y / f_stride_x.  Serves float64, float32 channels_last, float16 NCHW and views
 * without a conversion pass; the two dense network layouts keep their streaming kernels (shg_upfirdn2d_f32 / _f16). */
int shg_upfirdn2d_strided(const void* x, const float* f, void* y, int dtype, int N, int C, int H, int W, const long* x_strides,
                          const long* y_strides, int fh, int fw, long f_stride_y, long f_stride_x, int upx, int upy, int downx, int downy,
                          int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream);
/* upfirdn2d fused with the tail of an up-sampling synthesis layer (stylegan.py:295-304, comodgan.py:326-327):
 * y = lrelu_agc(FIR(x)*gain*scale[n,c] + noise*noise_strength + bias[c]) + residual.
 * noise_mode 0 none, 1 noise [OH,OW], 2 noise [N,OH,OW]; act 0 none / 1 lrelu_agc; clamp < 0 = no clamp. */
int shg_upfirdn2d_epilogue_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                               int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                               const float* scale, const float* bias, const float* noise, int noise_mode, float noise_strength,
                               int act, float alpha, float act_gain, float clamp, const float* residual, void* stream);

/* ---- A11: bias + activation -- the `x + bias -> lrelu_agc(gain)` of stylegan.py:232-238 / common/utils.py:135-143,
 * with the optional demodulation scale [N*C], noise and residual of the non-fused modconv path (stylegan.py:175-180). */
int shg_bias_act_f32(const float* x, float* y, const float* scale, const float* bias, const float* noise, int noise_mode,
                     float noise_strength, const float* residual, int N, int C, int HW, int act, float alpha, float gain,
                     float clamp, void* stream);
/* Backward of shg_bias_act_f32 without scale / noise / residual (the training path's bias + lrelu_agc): dx = g * slope(y), with
 * the forward OUTPUT y.  The bias gradient is the per-channel sum of dx. */
int shg_bias_act_backward_f32(const float* g, const float* y, float* dx, long total, int act, float alpha, float gain, float clamp,
                              void* stream);
/* stylegan_utils/fma.py:15 -- y = a*b + c, elementwise on `total` floats. */
int shg_fma_f32(const float* a, const float* b, const float* c, float* y, long total, void* stream);
/* stylegan_utils/fma.py:15-58 with NumPy broadcasting and without materialising it: y (contiguous, `shape`) = a*b + c, each operand
 * addressed through element strides (0 on a broadcast dimension); c may be NULL (y = a*b, the product of fma.py:41,44).  nd <= 6
 * (collapse neighbouring dimensions first); f64 != 0: operands are doubles. */
int shg_fma_bcast(const void* a, const void* b, const void* c, void* y, int nd, const long* shape, const long* sa, const long* sb,
                  const long* sc, int f64, void* stream);
/* `_unbroadcast(g * b, shape)` (fma.py:48-58) in one pass: dimensions ordered kept-first (nk of them), reduced-last; out (contiguous,
 * the kept sizes) = sum over the reduced index of g*b (b NULL: of g).  inner_kept != 0 when g's fastest-varying dimension is a kept
 * one (one thread per output), 0 for one workgroup per output along the reduced index.  Fixed summation order: deterministic. */
int shg_mul_reduce(const void* g, const void* b, void* out, int nd, int nk, const long* shape, const long* sg, const long* sb,
                   int inner_kept, int f64, void* stream);
/* stylegan.py:173 -- y[nc,:] = x[nc,:] * s[nc]. */
int shg_scale_channels_f32(const float* x, const float* s, float* y, int NC, int HW, void* stream);
/* out [N,K] = sum over b of part [N,B,K] in block order (the per-workgroup partial sums of shg_modtail_backward_*: deterministic). */
int shg_sum_partials_f32(const float* part, float* out, int N, int B, int K, void* stream);
/* `(weight * gain).to(float16)` of the fp16 layers and its gradient in one launch each: to_half != 0: dst (halves) = half(src (floats) * gain);
 * else dst (floats) = float(src (halves)) * gain. */
int shg_scale_cast_f32_f16(const void* src, void* dst, long n, float gain, int to_half, void* stream);
/* Phase planes of a stride-2 transposed convolution (shg_conv2d_f32 mode 2 / out_mode 1, shg_conv2d_up_poly_f32) -> image with the
 * crop / zero-extension conv2d_gradfix.py:96-128 applies: y[n,c,Y,X] = full[Y+lo][X+lo] (+ bias[c]), zero outside the
 * (2H+1) x (2W+1) result.  N*C <= 65535. */
int shg_planes_to_image_f32(const float* mid, const float* bias, float* y, int N, int C, int H, int W, int lo, int OH, int OW,
                            void* stream);

/* ---- A9/A8: convolution on the fp32 MFMA units -- replaces F.conv2d / F.conv_transpose2d reached through
 * conv2d_gradfix.py:35-43,109-116 from conv2d_resample.py:26-51.
 * Weight preparation: w [O,I,KH,KW] -> wt [OP/64][IP*KH*KW][64] (GEMM layout in 64-column blocks; OP = O rounded up to a
 *   multiple of 64, IP = I rounded up to a multiple of 32, padding zero filled),
 *   wscale [O] scratch, wsq [I*OP] (sum over taps of wt^2, for demodulation) or NULL.
 *   demod=1: w * rsqrt(mean_{I,k,k} w^2) (stylegan.py:146) * gain; demod=0: w * gain (stylegan.py:227).
 *   flip=1 gives a true convolution (w.flip([2,3]), conv2d_resample.py:32-33); one layout serves all modes. */
int shg_conv_weight_prep_f32(const float* w, float* wt, float* wscale, float* wsq, int O, int I, int KH, int KW, int OP,
                             int demod, float gain, int flip, void* stream);
/* y = act(out_scale[n,o] * conv(x * in_scale[n,i], wt) + noise*noise_strength + bias[o]) + residual.
 * mode 0: stride 1, symmetric `pad` -> [NB,O,H+2pad-kh+1,..]; mode 1: stride 2 -> [(H+2pad-kh)/2+1];
 * mode 2: transposed stride 2, pad 0 -> [2H+1, 2W+1] (conv2d_resample.py:130-137); all four sub-pixel phases in one pass.
 * Slot b uses the weight set wt + (b % wgroups)*wstride (wgroups > 1 = grouped conv, stylegan.py:187-190).
 * Any of in_scale [NB,I], out_scale [NB,O], bias [O], noise, residual [NB,O,OH,OW] may be NULL. */
int shg_conv2d_f32(const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W, int kh, int kw,
                   int mode, int pad, int wgroups, long wstride, const float* in_scale, const float* out_scale,
                   const float* bias, const float* noise, int noise_mode, float noise_strength, int act, float alpha, float gain,
                   float clamp, const float* residual, int out_mode, void* workspace, size_t ws_bytes, void* stream);
/* Split-K workspace (bytes) shg_conv2d_f32 can use for this problem; 0 = no split.  Small grids (4x4..16x16 layers)
 * are split along the input channels so that they still fill 256 CUs. */
size_t shg_conv2d_workspace_bytes(int NB, int I, int O, int H, int W, int kh, int kw, int mode, int pad, int wgroups);
/* Winograd F(2x2,3x3) form of the stride-1 3x3 'same' convolution (mode 0, pad 1 of shg_conv2d_f32; same fused operands and
 * result up to fp32 round-off, 2.25x fewer MFMA flops).  Weights: w [O,I,3,3] with the per-output-channel factor wscale [O]
 * (the `wscale` output of shg_conv_weight_prep_f32) -> wu [OP/64][ceil(I/c)][16][64][c] = G (w*wscale) G^T in the
 * register layout of the kernel, c = shg_conv_wino_chunk() input channels per K-chunk. */
int shg_conv_wino_chunk(void);
int shg_conv_weight_prep_wino_f32(const float* w, const float* wscale, float* wu, int O, int I, int OP, int flip, void* stream);
int shg_conv2d_wino_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                        const float* in_scale, const float* out_scale, const float* bias, const float* noise, int noise_mode,
                        float noise_strength, int act, float alpha, float gain, float clamp, const float* residual, void* stream);
/* the same with scratch for a split along the input channels: a workgroup owns one (tile, 64 output channels) pair, so 512-channel layers
 * at 16^2 (32^2 for the F(4x4) form below) are 32-128 workgroups for batches 4-16 and take the same time at every batch; with
 * `workspace` of shg_conv2d_wino_workspace_bytes (0: the problem fills the chip, nothing is split) the channel chunks are cut into
 * slices, partial outputs summed and the layer tail applied by a second small launch.  Same result up to the order of the fp32 sums. */
size_t shg_conv2d_wino_workspace_bytes(int NB, int I, int O, int OP, int H, int W);
int shg_conv2d_wino_ws_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                           const float* in_scale, const float* out_scale, const float* bias, const float* noise, int noise_mode,
                           float noise_strength, int act, float alpha, float gain, float clamp, const float* residual, void* workspace,
                           size_t ws_bytes, void* stream);
/* Winograd F(4x4,3x3) form of the same convolution (conv_wino4.hip; 4x fewer MFMA flops than the direct form, about 1e-5
 * relative round-off per layer): wu has shg_conv_wino4_weight_elems(OP, I) floats, [OP/64][ceil(I/8)][4][72][64] in the
 * register layout of the kernel.  shg_conv2d_wino4_supported tells whether the geometry is served (H >= 16, W >= 32,
 * W % 4 == 0, I <= 1024); otherwise call shg_conv2d_wino_f32 / shg_conv2d_f32. */
long shg_conv_wino4_weight_elems(int OP, int I);
int shg_conv_weight_prep_wino4_f32(const float* w, const float* wscale, float* wu, int O, int I, int OP, int flip, void* stream);
int shg_conv2d_wino4_supported(int NB, int I, int O, int H, int W);
int shg_conv2d_wino4_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                         const float* in_scale, const float* out_scale, const float* bias, const float* noise, int noise_mode,
                         float noise_strength, int act, float alpha, float gain, float clamp, const float* residual, void* stream);
/* ... and with scratch for the split along the input channels (see shg_conv2d_wino_ws_f32): the 512-channel layers at 32^2 */
size_t shg_conv2d_wino4_workspace_bytes(int NB, int I, int O, int OP, int H, int W);
int shg_conv2d_wino4_ws_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                            const float* in_scale, const float* out_scale, const float* bias, const float* noise, int noise_mode,
                            float noise_strength, int act, float alpha, float gain, float clamp, const float* residual, void* workspace,
                            size_t ws_bytes, void* stream);
/* mode 2 with out_mode 1 writes the four sub-pixel phases as planes [4][NB,O,H+1,W+1] (coalesced); this kernel applies the
 * 4x4 FIR of conv2d_resample.py:138 (pad 1) straight from the planes and fuses the synthesis-layer tail:
 * y [N,C,2H,2W] = lrelu_agc(FIR(mid)*gain*scale[n,c] + noise*noise_strength + bias[c]) + residual.  H, W = low-res extents. */
int shg_upfir_planar_f32(const float* mid, const float* f, float* y, int N, int C, int H, int W, int flip, float gain,
                         const float* scale, const float* bias, const float* noise, int noise_mode, float noise_strength,
                         int act, float alpha, float act_gain, float clamp, const float* residual, void* stream);
/* The same for a SEPARABLE filter (row-marching kernel): taps_host = {fx[0..3], fy[0..3]} in HOST memory; W even, W/2 a divisor of
 * 64 or W in {128, 256}; y / noise / residual 16-byte aligned. */
int shg_upfir_planar_sep_supported(int H, int W);
int shg_upfir_planar_sep_f32(const float* mid, const float* taps_host, float* y, int N, int C, int H, int W, int flip, float gain,
                             const float* scale, const float* bias, const float* noise, int noise_mode, float noise_strength,
                             int act, float alpha, float act_gain, float clamp, const float* residual, void* stream);
/* Polyphase-Winograd form of mode 2 / out_mode 1 (conv_wino_poly.hip): the `ee` phase as F(3x3,2x2), `eo`/`oe`/`oo` of
 * every 2x2 block of low-resolution pixels with 16 multiplies, one-pixel strips by single-tap contractions -- 5.8 instead of
 * 9 multiplies per low-resolution pixel, same result up to fp32 round-off.  y [4][NB,O,H+1,W+1] raw phase planes (no
 * epilogue operands: the consumer is shg_upfir_planar_f32).  wt / wscale from shg_conv_weight_prep_f32; wu_a, wu_b:
 * [OP/64][ceil(I/8)][16][64][8] floats each.  shg_conv2d_up_poly_supported tells whether the geometry is served
 * (H, W >= 16, H even, W % 4 == 0); otherwise call shg_conv2d_f32. */
int shg_conv_weight_prep_up_poly_f32(const float* w, const float* wscale, float* wu_a, float* wu_b, int O, int I, int OP, int flip,
                                     void* stream);
int shg_conv2d_up_poly_supported(int NB, int I, int O, int H, int W);
int shg_conv2d_up_poly_f32(const float* x, const float* wt, const float* wu_a, const float* wu_b, float* y, int NB, int I, int O,
                           int OP, int H, int W, const float* in_scale, void* stream);
/* ... with scratch for the split along the input channels on small grids (see shg_conv2d_wino_ws_f32): 16^2 / 32^2 inputs */
size_t shg_conv2d_up_poly_workspace_bytes(int NB, int I, int O, int OP, int H, int W);
int shg_conv2d_up_poly_ws_f32(const float* x, const float* wt, const float* wu_a, const float* wu_b, float* y, int NB, int I, int O,
                              int OP, int H, int W, const float* in_scale, void* workspace, size_t ws_bytes, void* stream);
/* Polyphase-Winograd form of the FIR-filtered stride-2 3x3 convolution (conv2d_resample.py:116-120), same reduction of the
 * multiply count.  shg_fir_down_planar_f32 applies the 4x4 pre-filter (padding 2) to x [N,C,H,W] and writes the (H+1)x(W+1)
 * result as its four polyphase planes xp [4][N*C][H/2+1][PP] (PP = a multiple of 4 >= W/2+1); shg_conv2d_down_poly_f32 then
 * computes y [NB,O,OH,OW] = act(conv3x3_stride2 + bias)*gain + residual (OH = H/2, OW = W/2; OH, OW >= 16, OW % 4 == 0). */
int shg_fir_down_planar_f32(const float* x, const float* f, float* y, int N, int C, int H, int W, int PP, int flip, float gain,
                            void* stream);
/* Row-marching form of the same pre-filter for a SEPARABLE filter f[ky][kx] = fy[ky]*fx[kx] (every filter upfirdn2d.setup_filter
 * builds from a 1-D kernel, upfirdn2d.py:61-95): taps_host = {fx[0..3], fy[0..3]} in HOST memory.  PP == 0: y [N,C,H+1,W+1]
 * (= upfirdn2d(x, f, padding=2)); PP > 0: the polyphase planes above with pitch PP.  W in {8,16,32,64,128,256,512}. */
int shg_fir_pad2_sep_supported(int H, int W, int PP);
int shg_fir_pad2_sep_f32(const float* x, const float* taps_host, float* y, int N, int C, int H, int W, int PP, int flip, float gain,
                         void* stream);
/* The x2 resampling FIRs of the training rows, separable filter (taps_host as above), row-marching kernels:
 * up == 1: y [N,C,H/2,W/2] = upfirdn2d(x, f, down=2, padding=1); up == 2: y [N,C,2H,2W] = upfirdn2d(x, f, up=2, padding=[2,1,2,1]). */
int shg_fir_resample2_sep_supported(int H, int W, int up);
int shg_fir_resample2_sep_f32(const float* x, const float* taps_host, float* y, int N, int C, int H, int W, int up, int flip, float gain,
                              void* stream);
int shg_conv_weight_prep_down_poly_f32(const float* w, const float* wscale, float* wu_a, float* wu_b, int O, int I, int OP, int flip,
                                       void* stream);
int shg_conv2d_down_poly_supported(int NB, int I, int O, int OH, int OW);
int shg_conv2d_down_poly_f32(const float* xp, const float* wu_a, const float* wu_b, float* y, int NB, int I, int O, int OP, int OH,
                             int OW, int PP, const float* in_scale, const float* bias, int act, float alpha, float gain,
                             float clamp, const float* residual, void* stream);
/* 1x1 convolution with I <= 8 input channels (encoder fromrgb, stylegan.py:640-642): y = act(W*wgain @ x + bias). */
int shg_conv1x1_thin_in_f32(const float* x, const float* w, const float* bias, float* y, int N, int I, int O, int HW, float wgain,
                            int act, float alpha, float gain, float clamp, void* stream);
/* torgb (stylegan.py:325-337) fused with the RGB skip up-sampling (comodgan.py:331-338, upfirdn2d.py:305-314):
 * y[n,o] = sum_i w[o,i]*styles[n,i]*x[n,i] + bias[o] + upsample2d(base_up[n,o], f)   (O <= 4; base_up may be NULL). */
int shg_torgb_f32(const float* x, const float* w, const float* styles, const float* bias, const float* base_up, const float* f,
                  float* y, int N, int I, int O, int H, int W, void* stream);

/* ---- A3/A2: dense (stylegan.py:87-98), normalize_2nd_moment (stylegan.py:343-344). */
int shg_dense_f32(const float* x, const float* w, const float* b, float* y, int N, int K, int O, int ldx, int ldy, float wgain,
                  float bgain, int act, float alpha, float gain, float clamp, void* stream);
/* The other two products of a dense layer under autograd (the gradients of stylegan.py:87-98; the reference reaches them through
 * torch.addmm's backward): out[N,K] = scale * a[N,M] @ b[M,K]  and  out[M,K] = scale * a[N,M]^T @ b[N,K] with, optionally,
 * colsum[m] = csum_scale * sum_n a[n,m] (weight and bias gradient in one pass). */
int shg_matmul_nn_f32(const float* a, const float* b, float* out, int N, int M, int K, int lda, int ldo, float scale, void* stream);
int shg_matmul_tn_f32(const float* a, const float* b, float* out, float* colsum, int N, int M, int K, int lda, int ldb, float scale,
                      float csum_scale, void* stream);
/* Weight side of modulated_conv2d under autograd (stylegan.py:136-138,146,150-155): wn = w * rsqrt(mean_{i,k} w^2) per output channel (with
 * the fp16 pre-normalisation by the max-norm first when `prenorm`), wsq[o,i] = sum_k wn^2, sfac[o] = wn / w; and its transpose
 * g_w = sfac * (G - wn * mean(G wn)), G = g_wn + 2 g_wsq wn.  w, wn, gwn, gw: [O,I,K]; wsq, gwsq: [O,I]; sfac: [O]. */
int shg_demod_weight_f32(const float* w, float* wn, float* wsq, float* sfac, int O, int I, int K, int prenorm, void* stream);
int shg_demod_weight_backward_f32(const float* wn, const float* sfac, const float* gwn, const float* gwsq, float* gw, int O, int I, int K,
                                  void* stream);
/* Style side of modulated_conv2d under autograd (stylegan.py:138,147,155): sn = s1 * rsqrt(mean_{n,i} s1^2), s1 = s / max_i |s| per row when
 * `prenorm` else s; d[n,o] = rsqrt(sum_i sn^2 wsq[o,i] + 1e-8) (d may be NULL); aux [N+1] = {row maxima (1 without prenorm), rsqrt(mean)}.
 * Backward: gs [N,I], gwsq [O,I] (may be NULL) from gsn [N,I] / gd [N,O] (either may be NULL); partq = scratch of ceil(O/64)*N*I floats.
 * N <= 32, N*I <= 8192 (64 KB of LDS in the backward pass). */
int shg_style_factors_f32(const float* s, const float* wsq, float* sn, float* d, float* aux, int N, int I, int O, int prenorm, void* stream);
int shg_style_factors_backward_f32(const float* sn, const float* d, const float* wsq, const float* aux, const float* gsn, const float* gd,
                                   float* gs, float* gwsq, float* partq, int N, int I, int O, int prenorm, void* stream);
int shg_normalize_2nd_moment_f32(const float* x, float* y, int N, int K, float eps, void* stream);
/* ---- A4: per-forward style side of modulated_conv2d (stylegan.py:147-155):
 * s_out = styles*pre_gain*rsqrt(mean(.^2)) when demod (else styles*pre_gain); dcoef[n,o] = rsqrt(sum_i s^2*wsq[i,o] + 1e-8). */
int shg_modconv_style_prep_f32(const float* styles, int ld, const float* wsq, float* s_out, float* dcoef, int N, int I, int O,
                               int OP, int demod, float pre_gain, void* stream);

/* Grouped forms (at most 32 groups per call): every style affine of a synthesis pass in one launch, and every
 * normalisation / demodulation-coefficient computation in a second one.  Group g of shg_dense_grouped_f32 computes
 * y = [x1 | x2] @ (w*wgain)^T + b*bgain with w [O, K1+K2] -- the concatenated input cat([w_i, x_global]) of
 * comodgan.py:245-262,316-338 is read from its two sources; K2 = 0 -> plain dense.  The descriptor arrays are host
 * memory (copied into the kernel arguments); the pointers inside are device pointers. */
typedef struct {
    const float* x1; const float* x2; const float* w; const float* b; float* y;
    int ld1, ld2, K1, K2, O, ldy;
    float wgain, bgain;
} shg_dense_group;
int shg_dense_grouped_f32(const shg_dense_group* groups, int G, int N, void* stream);
typedef struct {
    const float* styles; const float* wsq; float* s_out; float* dcoef;
    int ld, I, O, OP, demod;
    float pre_gain;
} shg_style_group;
int shg_modconv_style_prep_grouped_f32(const shg_style_group* groups, int G, int N, void* stream);

/* ---- A16-A19: Spectral Hint Unit (shgan.py:312-336).
 * rfft2(norm='forward') + row shift of [C] planes of 64x64 per sample (x + n*x_batch_stride) -> T [N,2C,64,33]. */
int shg_shu_rfft2_shift_f32(const float* x, long x_batch_stride, float* T, int N, int C, void* stream);
/* Weight gradient of y = conv2d(x, w, stride, pad) (conv2d_gradfix.py:140-146, the cuDNN backward-weight call):
 * dw[o,i,ky,kx] = sum_{n,oy,ox} g[n,o,oy,ox] * x[n,i,oy*stride-pad+ky, ox*stride-pad+kx]; 3x3 or 1x1, stride 1 or 2; deterministic
 * (K slices summed in order).  For conv_transpose2d pass dL/dy as `x` and the layer input as `g`: dw comes out as [Cin,Cout,kh,kw].
 * Input gradients are compositions of the forward entry points (conv <-> transposed conv, conv2d_gradfix.py:118-135). */
size_t shg_conv2d_wgrad_workspace_bytes(int NB, int I, int O, int OH, int OW, int kh, int kw);
int shg_conv2d_wgrad_f32(const float* x, const float* g, float* dw, int NB, int I, int O, int H, int W, int OH, int OW,
                         int kh, int kw, int stride, int pad, void* workspace, size_t ws_bytes, void* stream);
/* The same weight gradient for the stride-1 3x3 'same' layers (pad 1, OH = H, OW = W, W % 4 == 0, W >= 4) in the Winograd domain -- the
 * transpose of F(4x4,3x3): dw = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G, a quarter of the multiplications, both operands transformed
 * in the kernel, fp32 MFMA; deterministic; ~5e-6 relative against float64 (the direct form: ~2e-6).  shg_conv2d_wgrad_wino_supported
 * tells whether a geometry is served (1 / 0); x and g 16-byte aligned. */
int shg_conv2d_wgrad_wino_supported(int H, int W, int OH, int OW, int kh, int kw, int stride, int pad);
size_t shg_conv2d_wgrad_wino_workspace_bytes(int NB, int I, int O, int H, int W);
int shg_conv2d_wgrad_wino_f32(const float* x, const float* g, float* dw, int NB, int I, int O, int H, int W, void* workspace,
                              size_t ws_bytes, void* stream);
/* SHU spectral stage in one launch (shgan.py:320-321 conv0 + ReLU, :143-160 heterogeneous filter incl. the band sum):
 * S[n,o,p] = sum_k cw[k,p] * sum_i W1[o*bands+k, i] * relu(sum_j W0[i,j] T[n,j,p] + b0[i]);  T, S: [N,64,P], P % 64 == 0;
 * w0p [32][2][64] / w1p [bands*32][2][64]: weights in MFMA operand order, element [ks][mo][l] = W[mo*32 + (l & 31)][2*ks + (l >> 5)]
 * (for w1p the rows of band k are W1[o*bands+k, :]); cw [bands,P].  Then shg_shu_split_irfft2_f32 with bands = 1. */
int shg_shu_spectral_f32(const float* T, const float* w0p, const float* b0, const float* w1p, const float* cw, float* S,
                         int N, int C2, int P, int bands, void* stream);
/* band-weighted sum (bands > 1: Y [N,2C*bands,64,33], cw [bands,64,33]) -> Gaussian split -> unshift -> irfft2 at
 * r = 4,8,16,32,64: out[l] planes [C][r][r] per sample at out[l] + n*out_batch_stride[l]; accumulate=1 adds in place
 * (shgan.py:378-382).  gauss[l] = [r, r/2+1] table (shgan.py:281-310). */
int shg_shu_split_irfft2_f32(const float* Y, const float* cw, const float* const* gauss, float* const* out,
                             const long* out_batch_stride, int N, int C, int bands, int accumulate, void* stream);
/* Transpose of shg_shu_split_irfft2_f32 with bands == 1 (training rows: the gradient of shgan.py:326-336 w.r.t. the filtered
 * spectrum): g[l] = dL/d(out[l]) planes [N,C,r,r] (null = none) -> GS [N,2C,64,33].  The transpose of shg_shu_rfft2_shift_f32 is
 * shg_shu_split_irfft2_f32 itself, restricted to the 64 x 64 level with the table (1/c_k)/4096 (c_0 = c_32 = 1, else 2). */
int shg_shu_split_adjoint_f32(const float* const* g, const long* g_batch_stride, const float* const* gauss, float* GS, int N, int C,
                              void* stream);

/* ---- A24: eval composite (lib/experiments/shgan_default.py:257-262): x4 [N,4,H,W], img [N,3,H,W] -> u8 [N,3,H,W]. */
int shg_composite_u8(const float* x4, const float* img, uint8_t* out, int N, int H, int W, void* stream);

/* ---- input hand-off of the eval loop (lib/experiments/shgan_default.py:267-274): x = cat([mask-0.5, real*mask]).
 * real [N,3,H,W] in [-1,1], mask [N,H,W] in {0,1} -> x [N,4,H,W]. */
int shg_assemble_input_f32(const float* real, const float* mask, float* x, int N, int H, int W, void* stream);
/* The same from decoded uint8 pixels (ds_ffhq.py:307-347 + shgan_default.py:267-274): real [N,3,H,W] uint8, lut [256] = the float value of
 * every code as the host formatter computes it (torch: u8.float().div(255)*2-1), so the result equals the host route bit for bit. */
int shg_assemble_input_u8(const uint8_t* real, const float* mask, const float* lut, float* x, int N, int H, int W, void* stream);

/* ---- next row N2: freeform-mask rasteriser (lib/data_factory/ds_ffhq.py:145-217).  The host makes the random draws in the
 * reference's order and emits 8-word int32 records per mask (RECT / DISC / QUAD + 4 EDGE / POINT, see csrc/mask_raster.hip);
 * the device draws them exactly as Pillow's ImageDraw.line(width) / ellipse would.  records [total][8], offsets [B+1],
 * flips [B][2], disc_table [max_half+1][2*max_half+1][2]; mask [B,1,s,s] (1 keep / 0 hole), holes [B] += hole pixel counts
 * (zero it first).  s: multiple of 32, <= 512. */
int shg_mask_raster_f32(const int* records, const int* offsets, const int* flips, const int* disc_table, int max_half, float* mask,
                        int* holes, int B, int s, void* stream);

/* ---- next row N1: FID statistics (lib/evaluator/eva_fid.py:251-263).  S [DP,DP] float64 += sum_b w_b [x_b,1][x_b,1]^T on the
 * fp64 matrix cores: S[:D,:D] = sum x x^T, S[:D,D] = sum x, S[D,D] = count (tiles on / above the diagonal only).
 * feats [B,D] float32 (float64 when is_f64), weights [B] or NULL, DP >= D+1 a multiple of 32. */
int shg_fid_accumulate_f64(const void* feats, int is_f64, const float* weights, double* S, int B, int D, int DP, void* stream);

/* ---- next row N3 (training-side critic, forward only): minibatch_std_layer (stylegan.py:686-704).
 * x [N,C,H,W] -> y [N,C+F,H,W]; N % G == 0, C % F == 0; stat [N/G * F] is caller-owned scratch. */
int shg_minibatch_std_f32(const float* x, float* y, float* stat, int N, int C, int H, int W, int G, int F, void* stream);

/* ---- next row N3, fp16 route (the reference's `use_fp16` branches: stylegan.py:136-138,486,660-667, comodgan.py:40-47,305; its native
 * op is instantiated for half as well, upfirdn2d.cpp:59 AT_DISPATCH_FLOATING_TYPES_AND_HALF).  Layout of every fp16 activation:
 * channels-LAST, [N,H,W,C] IEEE halves (torch.channels_last on a [N,C,H,W] tensor) -- 8 consecutive channels are one 16-byte MFMA
 * operand.  Arithmetic: v_mfma_f32_32x32x16_f16, fp32 accumulation, one rounding to fp16.  Every tensor pointer of this section and the
 * per-channel fp32 operands (bias, out_scale / d) must be 16-byte aligned (vector loads; SHG_ERR_ARG otherwise).
 * shg_conv2d_f16_pack_weight: w [T][O][I] halves (T = k*k correlation taps, row-major (ky,kx)) -> wp in MFMA operand order
 *   [ceil(O/32) rounded up to 4][T][I/16][64][8] (shg_conv2d_f16_packed_weight_elems halves, zero beyond O): every operand of the
 *   kernel is one coalesced, unconditional 1 KiB load.
 * shg_conv2d_f16: w = that packed tensor, bias fp32 [O] or NULL, I % 32 == 0.
 *   mode 0: y [N,OH,OW,O] = conv2d(x, w, stride, pad);  mode 1: rows / columns [crop, crop+OH) x [crop, crop+OW) of
 *   conv_transpose2d(x, w, stride 2), 3x3, w[t][o][i] = torch weight[i][o][ky][kx] (conv2d_gradfix.py:109-116); entries beyond the
 *   (2H+1) x (2W+1) result are NOT written (shg_conv2d_f16_needs_clear: zero y first). */
long shg_conv2d_f16_packed_weight_elems(int T, int O, int I);
int shg_conv2d_f16_pack_weight(const void* w, void* wp, int T, int O, int I, void* stream);
/* the same operand order straight from a torch-layout weight: src [O][I][T] halves, or [I][O][T] when `transposed` (conv_transpose2d layout /
 * the channel-transposed weight of an input gradient); `flip` reverses the taps (180-degree rotation); input channels are zero-padded to a
 * multiple of 32 (wp holds shg_conv2d_f16_packed_weight_elems(T, O, roundup(I, 32)) halves). */
int shg_conv2d_f16_pack_weight_oihw(const void* src, void* wp, int T, int O, int I, int transposed, int flip, void* stream);
int shg_conv2d_f16(const void* x, const void* w, const float* bias, void* y, int N, int I, int O, int H, int W, int k, int stride, int pad,
                   int mode, int crop, int OH, int OW, void* stream);
int shg_conv2d_f16_needs_clear(int H, int W, int crop, int OH, int OW);
/* the same with the layer tail of the inference route fused: x*in_scale[n,i] at staging (both modes); mode 0 additionally
 * y = A(conv*out_scale[n,o] + noise*noise_strength + bias[o]) + residual on the half-rounded result (stylegan.py:173-181,298-304). */
int shg_conv2d_f16_fused(const void* x, const void* w, void* y, int N, int I, int O, int H, int W, int k, int stride, int pad, int mode, int crop,
                         int OH, int OW, const float* in_scale, const float* out_scale, const float* noise, int noise_mode, float noise_strength,
                         const float* bias, int act, float alpha, float gain, float clamp, const void* residual, void* stream);
/* weight gradient (replaces the cuDNN backward-weight call of conv2d_gradfix.py:140-146 for halves): dw [k*k][O][I] FP32 =
 * sum_{n,oy,ox} g[n,oy,ox,o] * x[n, oy*stride-pad+ky, ox*stride-pad+kx, i]; x [N,H,W,I], g [N,OH,OW,O] halves; I, O % 8 == 0;
 * deterministic (fp32 partial sums per pixel slice in `workspace`, reduced in a fixed order). */
size_t shg_conv2d_wgrad_f16_workspace_bytes(int N, int I, int O, int OH, int OW, int k);
int shg_conv2d_wgrad_f16(const void* x, const void* g, float* dw, int N, int I, int O, int H, int W, int OH, int OW, int k, int stride, int pad,
                         void* workspace, size_t ws_bytes, void* stream);
/* upfirdn2d on halves (same argument meaning as shg_upfirdn2d_f32; f stays fp32 [fh,fw], fp32 accumulation; C % 8 == 0). */
int shg_upfirdn2d_f16(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx, int upy, int downx, int downy,
                      int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream);
/* y = lrelu_agc(x + bias[c]) (act = 0: (x + bias) * gain) over `pixels` x C halves, and dL/dx from dL/dy and the saved OUTPUT y. */
/* Block-boundary casts `x.to(dtype)` (stylegan.py:486-495,659-663; comodgan.py:39-43,305-312) between the two activation layouts of this
 * library: float32 [N,C,H*W] (NCHW) -> float16 [N,H*W,C] (NHWC, torch.channels_last) when to_half, the reverse otherwise.  C % 8 == 0, or C <= 16
 * (thin tensors: the 4-channel network inputs). */
int shg_relayout_f32_f16(const void* src, void* dst, int N, int C, long HW, int to_half, void* stream);
int shg_bias_act_f16(const void* x, const float* bias, void* y, long pixels, int C, int act, float alpha, float gain, float clamp, void* stream);
int shg_bias_act_backward_f16(const void* g, const void* y, void* dx, long total, int act, float alpha, float gain, float clamp, void* stream);
/* modulation tail of a half layer in one pass each way (stylegan.py:173,176-181 + :298-304): y = A(t*d[n,c] + noise + bias[c]); backward from the
 * saved output: gt = gy*A'(y)*d, part [N][blocks][2][C] = per-workgroup pixel sums of gz*t and gz (caller sums over blocks: deterministic),
 * gnoise [N,HW] = channel sums of gz.  d fp32 [N,C] / noise fp32 [HW] (mode 1) or [N,HW] (mode 2) / bias fp32 [C], each optional. */
/* the same backward for float32 NCHW layers (forward = shg_bias_act_f32 with scale / noise / bias): gt = gy*A'(y)*d, part [N][blocks][2][C],
 * gnoise [cslices][N,HW] (per channel slice: the caller adds them); HW % 4 == 0, C <= 512.
 * Both take an optional second product u (a tensor like gy) / e fp32 [N,C]: gt = A'(y) * (gy*d + u*e) -- the tail's double backward
 * (R1 / path length, stylegan_default_loss.py:72-88,118-124): d/dgy = A'(y) (ggt d + t ggd) in one pass. */
int shg_modtail_backward_f32_blocks(long HW);
int shg_modtail_backward_f32_cslices(int N, int C, long HW);
int shg_modtail_backward_f32(const float* gy, const float* y, const float* t, const float* d, const float* u, const float* e, float* gt, float* part,
                             float* gnoise, int N, int C, long HW, int act, float alpha, float gain, float clamp, void* stream);
/* Kernel routes of the fp16 convolutions: bit 0 = the persistent LDS-DMA ring kernel for 3x3 stride-1 launches (csrc/conv_f16_ring.hip), bit 1 = the
 * merged-phase kernel for the stride-2 transposed form (csrc/conv_f16_upring.hip), bit 2 = the persistent kernel for 3x3 stride-2 launches
 * (csrc/conv_f16_down.hip); default 7; 0 = the gather kernel for everything.  Bits 0 / 1 give the gather kernel's bits, bit 2 the same
 * products summed k-step-major; returns the previous mask. */
int shg_conv2d_f16_set_routes(int mask);
int shg_modtail_f16(const void* t, const float* d, const float* noise, int noise_mode, const float* bias, void* y, int N, long HW, int C, int act,
                    float alpha, float gain, float clamp, void* stream);
int shg_modtail_backward_f16_blocks(long HW, int C);
int shg_modtail_backward_f16(const void* gy, const void* y, const void* t, const float* d, const void* u, const float* e, void* gt, float* part,
                             float* gnoise, int N, long HW, int C, int act, float alpha, float gain, float clamp, void* stream);

#ifdef __cplusplus
}
#endif
#endif
