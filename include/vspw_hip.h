/*
 * vspw_hip.h — C ABI of libvspw_hip.so, the hand-written HIP (gfx950 / MI355X) compute library behind the
 * ModelBuilder / SegmentationModule / Clip_PSP / ClipOCRNet surface of sssdddwww2/CVPR2021_VSPW_Implement.
 *
 * The reference has no FFI on this path: its seam is a Python nn.Module surface whose every FLOP is an ATen op
 * (SURVEY.md §8b).  Each entry point below therefore cites the reference call site(s) (file:line under
 * /root/reference) whose ATen op it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer to fp32 data unless stated otherwise;
 *   - activations are NHWC: a tensor the reference calls [N,C,H,W] is stored as [N][H][W][C]
 *     (torch memory_format=channels_last of the same logical shape); "rows" = N*H*W pixels;
 *   - convolution weights are [Cout][KH][KW][Cin] (channels_last of the reference's OIHW parameter);
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing allocates or synchronises;
 *   - every function returns VSPW_OK (0) or a negative error code and never throws;
 *   - *_workspace() functions return the scratch bytes the matching call needs (caller allocates).
 */
#ifndef VSPW_HIP_H
#define VSPW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSPW_OK 0
#define VSPW_EINVAL (-1)  /* bad argument / geometry / workspace too small */
#define VSPW_ELAUNCH (-2) /* hipLaunchKernel reported an error */

#define VSPW_ABI_VERSION 7
int vspw_abi_version(void);
/* The hipError_t behind the most recent VSPW_ELAUNCH (0 if none) - for error messages. */
int vspw_last_hip_error(void);

/* ---------------------------------------------------------------- convolution (conv_igemm.hip) ---- */
/* Geometry of one nn.Conv2d call: x [n,h,w,c] -> y [n,oh,ow,k], square stride/pad/dilation.
 * oh = (h + 2*pad - dil*(kh-1) - 1)/stride + 1 (checked). */
typedef struct vspw_conv_desc {
    int n, h, w, c;
    int oh, ow, k;
    int kh, kw, stride, pad, dil; /* pad: rows (H) on both sides */
    int pad_w;                    /* columns (W) on both sides; nn.Conv2d(padding=(pad, pad_w)) */
} vspw_conv_desc;

/* y = conv2d(x, w) (+ bias).  Replaces F.conv2d at models/resnet.py:61-66,100-106,130 (after the hyper-parameter
 * rewrite of models/models.py:737-750), models/clip_psp.py:29,35,40,74,79, models/clip_ocr.py:43,56,58,62,
 * models/ocr_modules/spatial_ocr_block.py:208-244,351, models/non_local.py:53-71, models/netwarp.py:41-54,97.
 * stat_part (optional, may be NULL): [vspw_conv2d_stats_partials(d)][2][k] per-row-tile column sums of y and y*y,
 * consumed by vspw_bn_reduce_partials_f32 (BatchNorm statistics fused into the conv epilogue). */
int vspw_conv2d_fwd(const vspw_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                    float* stat_part, void* stream);
size_t vspw_conv2d_stats_partials(const vspw_conv_desc* d);
/* Pointwise convolution whose input z = relu(scale*y_in + shift + res_in) - the BatchNorm apply + residual + ReLU of the
 * node that produced it (models/resnet.py:83-90 followed by the next Bottleneck's conv1, :75) - has NOT been
 * materialised: the GEMM evaluates it while staging its A operand (same expression tree as vspw_bn_apply: bit-identical
 * z) and writes it to z_out for the node's other readers (the skip connection, this conv's weight gradient, backward).
 * scale_shift: [2][c]; res_in may be NULL (a node without a residual branch: conv2 -> conv3 inside a Bottleneck,
 * models/resnet.py:79-84).  One pass over y_in / res_in instead of vspw_bn_apply's read-read-write followed by this
 * conv's read.  1x1, stride 1, no padding, c % 32 == 0 only (VSPW_EINVAL otherwise). */
size_t vspw_conv2d_fwd_apply_supported(const vspw_conv_desc* d); /* 1 / 0 */
int vspw_conv2d_fwd_apply(const vspw_conv_desc* d, const float* y_in, const float* res_in, const float* scale_shift,
                          float* z_out, const float* w, const float* bias, float* y, float* stat_part, void* stream);
/* Inference-side variant used by the frozen RAFT flow network (RAFT_core/update.py:6-136, extractor.py:6-190): the
 * input may be a channel slice of a wider NHWC buffer (pixel stride ldx floats >= c; torch.cat call sites such as
 * update.py:24,29,44,47 become slot writes), the output is written with row stride ldy floats >= k, and the epilogue
 * applies act(conv + bias + addend): 0 none, 1 relu, 2 sigmoid, 3 tanh (update.py:14,26-28,84-92); addend (may be
 * NULL) has y's layout.  Eval-mode conv + BatchNorm (+ residual) + ReLU of the segmentation nets also runs through
 * here, with the BN scale folded into the weights and its shift passed as the bias (ops.ConvBNActFn, inference). */
int vspw_conv2d_fwd_ex(const vspw_conv_desc* d, const float* x, long long ldx, const float* w, const float* bias,
                       const float* addend, int act, float* y, long long ldy, void* stream);
/* Direct form of a convolution with FEW OUTPUT channels (k <= 4, c % 4 == 0, c >= 64, <= 9 taps, stride 1, undilated,
 * same-size output): RAFT's FlowHead.conv2 (3x3, 256 -> 2 channels, RAFT_core/update.py:13-14) fills 2 of the 64 columns
 * of an MFMA tile - one wave per pixel instead.  Arguments as vspw_conv2d_fwd_ex (no addend).  _supported: 0 = take
 * vspw_conv2d_fwd_ex, 1 = this entry point runs it. */
int vspw_conv2d_thin_supported(const vspw_conv_desc* d, long long ldx, long long ldy);
int vspw_conv2d_thin(const vspw_conv_desc* d, const float* x, long long ldx, const float* w, const float* bias, int act,
                     float* y, long long ldy, void* stream);
/* dx = conv2d_backward_input(dy, w).  wT is the [c][kh][kw][k] copy of w made by vspw_weight_transpose. */
int vspw_conv2d_bwd_data(const vspw_conv_desc* d, const float* dy, const float* wT, float* dx, void* stream);
/* dx = conv2d_backward_input(dy, w) + addend: the gradient arriving over the skip connection (models/resnet.py:75-90:
 * `out += residual`) is added in the GEMM epilogue instead of by a separate pass.  addend has dx's shape/layout. */
int vspw_conv2d_bwd_data_acc(const vspw_conv_desc* d, const float* dy, const float* wT, const float* addend, float* dx,
                             void* stream);
/* Data gradient with the BatchNorm-backward FRONT END of the node that produced this conv's input fused into the
 * epilogue (models/resnet.py:75-90 chains conv -> bn -> relu -> conv): relu_src = that node's output z (this conv's
 * saved input), bn_y / bn_mean / bn_invstd = its pre-BN activations and batch statistics.  Stores
 * g = (z > 0) ? dx : 0 into dx and the per-tile column sums of g and g*(bn_y-mean)*invstd into stat_part
 * [vspw_conv2d_bwd_data_bn_partials(d)][2][c]; that node then runs vspw_bn_bwd_reduce_partials_f32 + vspw_bn_bwd_apply
 * (relu = 0) and needs neither the reduction pass nor the mask read.  addend may be NULL.  _partials returns 0 when
 * the geometry cannot take this path. */
size_t vspw_conv2d_bwd_data_bn_partials(const vspw_conv_desc* d);
int vspw_conv2d_bwd_data_bn(const vspw_conv_desc* d, const float* dy, const float* wT, const float* addend,
                            const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                            float* dx, float* stat_part, void* stream);
/* dw [k][kh][kw][c] = conv2d_backward_weight(dy, x); split-K over pixels, deterministic reduction. */
size_t vspw_conv2d_bwd_weight_workspace(const vspw_conv_desc* d);
int vspw_conv2d_bwd_weight(const vspw_conv_desc* d, const float* dy, const float* x, float* dw, void* ws,
                           size_t ws_bytes, void* stream);
/* BatchNorm-backward "apply" folded into the operand load of a POINTWISE convolution's two gradient GEMMs
 * (models/resnet.py:61,66 - conv1 / conv3 of a Bottleneck - followed by batchnorm.py:68-98): instead of materialising
 * dy = a*(g - mean(g) - xhat*mean(g*xhat)) (one pass reading g and y, writing dy, which both GEMMs then re-read), the
 * GEMMs stage  coef[0][k]*g + coef[1][k]*y + coef[2][k]  on the fly.  g: gradient w.r.t. the node's (masked) output,
 * y: its pre-BN activations, coef [3][k] from vspw_bn_bwd_affine_coeffs (sums = the [2][k] fp64 reductions that
 * vspw_bn_bwd_reduce_partials_f32 produces).  The data-gradient variant composes with the skip addend and with the
 * fused BN-backward front end (relu_src .. stat_part: all NULL or all set, as vspw_conv2d_bwd_data_bn).  Restrictions
 * (VSPW_EINVAL otherwise): 1x1, stride 1, no padding, Cin % 32 == 0 for the data gradient; Cout > 64 and
 * n*oh*ow % 32 == 0 for the weight gradient. */
int vspw_bn_bwd_affine_coeffs(const double* sums, double count, const float* gamma, const float* mean,
                              const float* invstd, float* coef, int c, int training, void* stream);
/* vspw_bn_bwd_reduce_partials_f32 + vspw_bn_bwd_affine_coeffs in one launch (single rank: no exchange of the sums in
 * between) - bit-identical to the two-launch form. */
int vspw_bn_bwd_reduce_partials_coeffs_f32(const float* part, int tiles, int c, double count, const float* gamma,
                                           const float* mean, const float* invstd, int training, double* sums,
                                           float* dgamma, float* dbeta, float* coef, void* stream);
int vspw_conv2d_bwd_data_aff(const vspw_conv_desc* d, const float* g, const float* y, const float* coef, const float* wT,
                             const float* addend, const float* relu_src, const float* bn_y, const float* bn_mean,
                             const float* bn_invstd, float* dx, float* stat_part, void* stream);
int vspw_conv2d_bwd_weight_aff(const vspw_conv_desc* d, const float* g, const float* y, const float* coef, const float* x,
                               float* dw, void* ws, size_t ws_bytes, void* stream);
/* 1 when both affine-operand gradients above accept the geometry (the caller otherwise applies the BatchNorm backward
 * with vspw_bn_bwd_apply and runs the plain gradients). */
size_t vspw_conv2d_bwd_aff_supported(const vspw_conv_desc* d);
/* [k][taps][c] -> [c][taps][k] */
int vspw_weight_transpose(const float* w, float* wT, int k, int taps, int c, void* stream);
/* The same transpose for many weight tensors in ONE launch.  `entries` is a DEVICE array sorted by tile0 (= the sum of
 * vspw_weight_transpose_tiles() of all preceding entries); total_tiles = grid size. */
typedef struct vspw_wt_entry {
    const float* w; /* [k][taps][c] */
    float* wT;      /* [c][taps][k] */
    long long tile0;
    int k, taps, c, reserved;
} vspw_wt_entry;
long long vspw_weight_transpose_tiles(int k, int taps, int c);
int vspw_weight_transpose_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream);
/* Summation shape of the pointwise (1x1 / plain GEMM) kernels with a reduction of K >= 2*k terms: K/k chains of k terms,
 * the finished chains parked in the output tile, instead of one k-sequential fp32 chain - the rounding error of a
 * k-blocked CPU GEMM, which is what the reference's ATen conv2d / matmul (models/resnet.py:76-86, spatial_ocr_block.py:
 * 252-274) runs on.  k: multiple of 32; 0 = one chain = the default (chains of 256 halve the error of a K >= 1024 GEMM
 * but cost 3 % of the training step and move the end-to-end parity figures by < 0.1x: the excess over the reference's
 * own fp32 noise sat in the direct 3x3 kernels, which fold their chains every 96 terms unconditionally - DESIGN.md
 * section 4).  Process-wide policy, not per-call state: set it before issuing work.
 * The chunked variants are compiled only into diagnostic builds (-DVSPW_WITH_ACCUM_CHUNK, tools/diag/build_variant.py;
 * vspw_accum_chunk_compiled() == 1): the shipped library accepts k = 0 only and returns VSPW_EINVAL otherwise. */
int vspw_set_accum_chunk(int k);
int vspw_get_accum_chunk(void);
int vspw_accum_chunk_compiled(void);
/* Diagnostics: shader cycles (out[0]) and 100 MHz wall-clock ticks (out[1]) that workgroup 0 of every forward / data-
 * gradient GEMM launch spent between its first and last instruction since the last reset - their ratio x 0.1 is the shader
 * clock in GHz the chip sustained under that load (measured 1.87 ... 2.40 GHz on the same GEMM depending on operand values
 * and on the preceding milliseconds; the 157.3 TFLOP/s fp32 MFMA peak is quoted at 2.4 GHz; bench.py reports it as
 * roofline.sustained_clock_ghz).
 * Synchronises the device.  reset != 0: zero the sums afterwards. */
int vspw_debug_nt_clock(unsigned long long* out, int reset);
/* The probe is OFF by default (an unarmed launch pays one scalar load): on != 0 arms it, 0 disarms it.  Synchronises the
 * device.  Arm it only around launches issued one after the other on one stream (one start stamp per device). */
int vspw_debug_nt_clock_enable(int on);
/* [n][c][hw] -> [n][hw][c]: the reference feeds NCHW images (train_clip2.py:45-47). */
int vspw_nchw_to_nhwc(const float* in, float* out, int n, int c, long long hw, void* stream);

/* ---------------------------------------------------------------- Winograd F(2x2,3x3) (winograd.hip) --- */
/* Stride-1 3x3 convolutions (pad == dilation; reference models/resnet.py:63-64 after models/models.py:737-750, heads
 * models/clip_psp.py:29-35,74-79, models/clip_ocr.py:44-45) as 16 batched GEMMs with 4/9 of the direct multiplications:
 *   U = vspw_wino_weights(w)            [16][rows][reduce]   (forward: rows = Cout; data gradient: rows = Cin, filter
 *                                                              rotated by 180 degrees)
 *   V = vspw_wino_input(x or dy)        [16][T][channels]    T = vspw_wino_tiles(d) (2x2 output tiles per image and
 *                                                              dilation sub-grid)
 *   M = vspw_bmm_nt(V, U, batch 16)     [16][T][rows]
 *   y = vspw_wino_output(M)             NHWC, + bias; with stat_part: per-workgroup [2][channels] partial sums for the
 *       BatchNorm that follows (vspw_wino_stat_partials(d) rows); with relu_src/bn_*: the BatchNorm-backward front end
 *       of vspw_conv2d_bwd_data_bn; addend / act (0 none, 1 ReLU): the residual add and activation of the inference
 *       path's folded conv + BatchNorm (vspw_conv2d_fwd_ex).  `d` is the convolution's descriptor in every call
 *       (h, w, dil are used). */
size_t vspw_wino_supported(const vspw_conv_desc* d);
long long vspw_wino_tiles(const vspw_conv_desc* d);
size_t vspw_wino_stat_partials(const vspw_conv_desc* d);
int vspw_wino_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream);
/* Both transforms of many weight tensors in ONE launch: entries (device array, sorted by tile0 = sum of
 * vspw_wino_weight_tiles() of the preceding entries), entry.wT -> [2][16][k*c] (forward, then data gradient). */
long long vspw_wino_weight_tiles(int k, int c);
int vspw_wino_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream);
int vspw_wino_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream);
/* vspw_wino_input of z = relu(scale*y + shift) - the BatchNorm apply + ReLU of the conv+BN+ReLU node that produces this
 * convolution's input (models/resnet.py:76-78 followed by conv2, :79) - which has NOT been materialised: V is evaluated
 * from y (same expression tree as vspw_bn_apply: bit-identical z; padding stays zero) and z is written to z_out for the
 * backward pass.  scale_shift [2][channels].  One pass over y instead of vspw_bn_apply's read-write followed by the
 * transform's read. */
int vspw_wino_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift, float* z_out, int channels,
                          float* v, void* stream);
/* vspw_wino_input + vspw_bmm_nt in one launch: the input transform is evaluated while the GEMM stages its A operand
 * (V is never written).  src = x or dY (NHWC, `channels`), u [16][rows][channels], m [16][T][rows]. */
int vspw_wino_gemm_fused(const vspw_conv_desc* d, const float* src, int channels, const float* u, int rows, float* m,
                         void* stream);
/* The same two calls for tensors that are channel slots of wider NHWC buffers (pixel strides ldx / ldy, multiples of 4):
 * the 3x3 convolutions of the frozen flow network (RAFT_core/update.py:16-17,82-87), whose operands live inside the
 * update block's concatenation buffers; y = act(A^T M A + bias), act 0 / 1 (ReLU). */
int vspw_wino_gemm_fused_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels, const float* u,
                            int rows, float* m, void* stream);
int vspw_wino_output_ex(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                        long long ldy, int act, void* stream);
int vspw_wino_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                     const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                     float* stat_part, const float* addend, int act, void* stream);
/* Row-fused form (wino_rows.hip): Y = A^T M A is separable, and the inner sum P[a][j] = sum_b M[a][b] A^T[j][b] mixes
 * only the four positions of one transform row a.  One workgroup runs the four GEMMs of a row back to back and keeps the
 * sum in registers: the GEMM writes the 8 planes P [4][2][tpad][rows] (half of M) and vspw_wino_output_rows finishes
 * Y[i][j] = sum_a A^T[i][a] P[a][j] reading half as much.  tpad = vspw_wino_rows_tpad(d, channels, rows, fused) tiles
 * per plane (T rounded up to the GEMM's tile height, which differs between the plain (fused = 0) and the fused-operand
 * (fused = 1) GEMM; 0 = geometry not supported: rows % 128, channels % 32, 32-bit spans).
 * vspw_wino_gemm_rows takes V = vspw_wino_input(..) [16][T][channels]; vspw_wino_gemm_fused_rows evaluates it from the
 * NHWC tensor src like vspw_wino_gemm_fused.  Same reference call sites as vspw_wino_output. */
long long vspw_wino_rows_tpad(const vspw_conv_desc* d, int channels, int rows, int fused);
/* 1 when the row-fused form is expected to be the faster one for this geometry (enough workgroups to fill the chip,
 * reduction short enough for the halved M traffic to matter): the dispatch rule callers use. */
int vspw_wino_rows_prefer(const vspw_conv_desc* d, int channels, int rows, int fused);
/* Tuning knob (experiments): force the GEMM tile of the row-fused form - 12 (64x128), 31 (96x128), 22 (128x128); 0 =
 * automatic.  Changes vspw_wino_rows_tpad: set it before sizing buffers.  (+100: timing build without the stores,
 * +200: without the K loop - results are then meaningless.) */
int vspw_wino_rows_config(int tile);
int vspw_wino_gemm_rows(const vspw_conv_desc* d, const float* v, int channels, const float* u, int rows, float* tp,
                        void* stream);
int vspw_wino_gemm_fused_rows(const vspw_conv_desc* d, const float* src, int channels, const float* u, int rows,
                              float* tp, void* stream);
/* ... for operands that are channel slots of wider NHWC buffers (pixel strides ldx / ldy), cf. vspw_wino_gemm_fused_ex /
 * vspw_wino_output_ex (RAFT_core/update.py:16-17,82-87). */
int vspw_wino_gemm_fused_rows_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels, const float* u,
                                 int rows, float* tp, void* stream);
int vspw_wino_output_rows_ex(const vspw_conv_desc* d, const float* tp, long long tpad, int channels, const float* bias,
                             float* y, long long ldy, int act, void* stream);
int vspw_wino_output_rows(const vspw_conv_desc* d, const float* tp, long long tpad, int channels, const float* bias,
                          float* y, const float* relu_src, const float* bn_y, const float* bn_mean,
                          const float* bn_invstd, float* stat_part, const float* addend, int act, void* stream);
/* Weight gradient in the transform domain: dM = vspw_wino_dy(dY) [16][T][Cout]; dU = vspw_bmm_tn(dM, V, batch 16)
 * [16][Cout][Cin] with V = vspw_wino_input(x); dW = vspw_wino_dw(dU) in the weight layout [Cout][3][3][Cin]. */
int vspw_wino_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream);
int vspw_wino_dw(const float* du, float* dw, int k, int c, void* stream);

/* ---------------------------------------------------------------- Winograd F(3x3,3x3) (winograd_f3.hip) --- */
/* The same convolutions (stride 1, 3x3, pad == dilation; same reference call sites as vspw_wino_*) as 25 batched GEMMs
 * over 3x3 output tiles / 5x5 input patches (interpolation points 0, 1, -1, 2, inf): 25/81 of the direct multiplications
 * (F(2x2): 36/81), and 3 divides the 60 / 30 / 15 pixel sub-grid edges of every dilation of the stride-8 stage exactly.
 *   U = vspw_wino3_weights(w)           [25][rows][reduce]   (G g G^T evaluated in fp64, rounded once)
 *   V = vspw_wino3_input(x or dy)       [25][T][channels]    T = vspw_wino3_tiles(d)
 *   M = vspw_bmm_nt(V, U, batch 25)     [25][T][rows]
 *   y = vspw_wino3_output(M)            arguments and fused epilogues as vspw_wino_output
 * Weight gradient: dM = vspw_wino3_dy(dY) [25][T][Cout]; dU = vspw_bmm_tn(dM, V, batch 25) [25][Cout][Cin];
 * dW = vspw_wino3_dw(dU) [Cout][3][3][Cin].  vspw_wino3_weights_multi: entry.wT -> [2][25][k*c], tile0 counted in
 * vspw_wino_weight_tiles(). */
size_t vspw_wino3_supported(const vspw_conv_desc* d);
long long vspw_wino3_tiles(const vspw_conv_desc* d);
size_t vspw_wino3_stat_partials(const vspw_conv_desc* d);
int vspw_wino3_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream);
int vspw_wino3_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream);
int vspw_wino3_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream);
/* ... of z = relu(scale*y + shift), not materialised (see vspw_wino_input_apply): V from y, z written to z_out. */
int vspw_wino3_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift, float* z_out, int channels,
                           float* v, void* stream);
int vspw_wino3_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                      const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                      float* stat_part, const float* addend, int act, void* stream);
int vspw_wino3_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream);
int vspw_wino3_dw(const float* du, float* dw, int k, int c, void* stream);
/* F(4x4,3x3): the same nine calls over 4x4 output tiles / 6x6 patches (points 0, 1, -1, 1/2, -2, inf), 36 planes: 36/144 of
 * the direct multiplications where 4 divides the sub-grid edge (the undilated 60x60 shapes), 36/126.6 after padding 30 ->
 * 32 / 15 -> 16 on the dilated ones.  Conv-level rounding error as F(3x3) with this point set (winograd_f3.hip). */
size_t vspw_wino4_supported(const vspw_conv_desc* d);
long long vspw_wino4_tiles(const vspw_conv_desc* d);
size_t vspw_wino4_stat_partials(const vspw_conv_desc* d);
int vspw_wino4_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream);
int vspw_wino4_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream);
int vspw_wino4_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream);
/* ... of z = relu(scale*y + shift), not materialised (see vspw_wino_input_apply): V from y, z written to z_out. */
int vspw_wino4_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift, float* z_out, int channels,
                           float* v, void* stream);
int vspw_wino4_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                      const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                      float* stat_part, const float* addend, int act, void* stream);
int vspw_wino4_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream);
int vspw_wino4_dw(const float* du, float* dw, int k, int c, void* stream);

/* F(5x5,3x3): the same nine calls over 5x5 output tiles / 7x7 patches (points 0, 1, -1, 1/2, -1/2, 2, inf), 49 planes: 49/225 of
 * the direct multiplications (1.96 per output pixel); 5 divides the 60 / 30 / 15 pixel sub-grid edges exactly.  Conv-level
 * rounding error 2.8x F(3x3)'s (4.0e-6 on a layer-3 convolution); see winograd_f3.hip and DESIGN.md section 3 for where it is
 * used. */
size_t vspw_wino5_supported(const vspw_conv_desc* d);
long long vspw_wino5_tiles(const vspw_conv_desc* d);
size_t vspw_wino5_stat_partials(const vspw_conv_desc* d);
int vspw_wino5_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream);
int vspw_wino5_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream);
int vspw_wino5_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream);
/* ... of z = relu(scale*y + shift), not materialised (see vspw_wino_input_apply): V from y, z written to z_out. */
int vspw_wino5_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift, float* z_out, int channels,
                           float* v, void* stream);
int vspw_wino5_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                      const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                      float* stat_part, const float* addend, int act, void* stream);
int vspw_wino5_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream);
int vspw_wino5_dw(const float* du, float* dw, int k, int c, void* stream);

/* ---------------------------------------------------------------- batch norm (bn.hip) ------------- */
/* Replaces SynchronizedBatchNorm2d.forward = F.batch_norm (models/sync_batchnorm/batchnorm.py:68-98) and its
 * autograd backward, fused with the ReLU / residual add / Dropout2d that follow it in models/resnet.py:40-51,75-90,
 * models/clip_psp.py:36-39,75-78 and models/clip_ocr.py:44-45,59-61. */
size_t vspw_bn_stats_workspace(long long rows, int c);
/* sums[0][c] = sum_rows x, sums[1][c] = sum_rows x*x, accumulated in fp64. */
int vspw_bn_stats(const float* x, long long rows, int c, double* sums, void* ws, size_t ws_bytes, void* stream);
/* Same sums from the per-tile fp32 partials written by vspw_conv2d_fwd(stat_part). */
int vspw_bn_reduce_partials_f32(const float* part, int tiles, int c, double* sums, void* stream);
/* Training-mode statistics -> per-channel coefficients.  count = number of rows behind `sums` (all ranks).
 * mean, invstd = 1/sqrt(var_biased + eps); scale = gamma*invstd; shift = beta - mean*scale;
 * running_mean/var updated with momentum and the unbiased variance (batchnorm.py:133-150); either may be NULL. */
int vspw_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                     float* shift, int c, void* stream);
/* The same with invstd = clamp(var_biased, eps)^-1/2: what the reference's MULTI-device SynchronizedBatchNorm computes
 * (models/sync_batchnorm/batchnorm.py:150); its single-device path is F.batch_norm's (var + eps)^-1/2. */
int vspw_bn_finalize_clamped(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                             float* shift, int c, void* stream);
/* Populations of at most 1024 rows (pyramid-pool branches, OCR object contexts; reference models/clip_psp.py:45-56,
 * models/ocr_modules/spatial_ocr_block.py:247-289 through models/sync_batchnorm/batchnorm.py:70-73): statistics taken
 * from the activations two-pass in fp64 and finalised in the same launch (what ATen's CPU batch_norm does in its double
 * accumulators).  With sums != NULL the kernel only writes the local [sum x, sum x^2] (fp64) for the cross-rank
 * exchange and the caller finalises with vspw_bn_finalize. */
int vspw_bn_small_finalize(const float* x, int rows, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                           float* shift, double* sums, int c, void* stream);
/* vspw_bn_reduce_partials_f32 + vspw_bn_finalize in one launch (single-rank training: no exchange in between). */
int vspw_bn_finalize_partials_f32(const float* part, int tiles, double count, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                  float* invstd, float* scale, float* shift, int c, void* stream);
/* Eval-mode coefficients from the running statistics. */
int vspw_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* mean, float* invstd, float* scale, float* shift, int c, void* stream);
/* Inference: conv followed by eval-mode BatchNorm is one conv with w_out[k][:] = scale[k]*w[k][:] and
 * bias_out[k] = (cbias ? cbias[k]*scale[k] : 0) + shift[k]  (then vspw_conv2d_fwd_ex adds the residual and the ReLU). */
int vspw_bn_fold_weights(const float* w, const float* cbias, const float* scale, const float* shift, float* w_out,
                         float* bias_out, int k, long long cols, void* stream);
/* z = [relu]( x*scale + shift [+ residual] ) [* chan_mask[image][c]]   (chan_mask = Dropout2d mask/(1-p)). */
int vspw_bn_apply(const float* x, const float* scale, const float* shift, const float* residual,
                  const float* chan_mask, float* z, long long rows, int c, long long rows_per_image, int relu,
                  void* stream);
size_t vspw_bn_bwd_workspace(long long rows, int c);
/* g = dz * chan_mask * (relu ? z>0 : 1);  sums[0][c] = sum g, sums[1][c] = sum g*xhat, xhat=(x-mean)*invstd. */
int vspw_bn_bwd_reduce(const float* dz, const float* z, const float* x, const float* mean, const float* invstd,
                       const float* chan_mask, long long rows, int c, long long rows_per_image, int relu,
                       double* sums, void* ws, size_t ws_bytes, void* stream);
/* Same, additionally writing dbeta = sums[0], dgamma = sums[1] as fp32 (the LOCAL parameter gradients); either may be
 * NULL. */
int vspw_bn_bwd_reduce_pg(const float* dz, const float* z, const float* x, const float* mean, const float* invstd,
                          const float* chan_mask, long long rows, int c, long long rows_per_image, int relu,
                          double* sums, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* The same two sums (+ local parameter gradients) from the per-tile fp32 partials that vspw_conv2d_bwd_data_bn wrote:
 * no pass over the activations at all. */
int vspw_bn_bwd_reduce_partials_f32(const float* part, int tiles, int c, double* sums, float* dgamma, float* dbeta,
                                    void* stream);
/* dx = gamma*invstd*(g - sums0/count - xhat*sums1/count) (training) or gamma*invstd*g (eval);
 * dres = g (optional); dgamma = sums1, dbeta = sums0 (as fp32). */
int vspw_bn_bwd_apply(const float* dz, const float* z, const float* x, const float* mean, const float* invstd,
                      const float* gamma, const double* sums, double count, const float* chan_mask, long long rows,
                      int c, long long rows_per_image, int relu, int training, float* dx, float* dres,
                      float* dgamma, float* dbeta, void* stream);

/* ---------------------------------------------------------------- pooling (pool.hip) -------------- */
/* nn.MaxPool2d(3, stride 2, pad 1) of models/resnet.py:109; idx holds the winning tap (0..8) per output. */
int vspw_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int oh, int ow,
                          void* stream);
int vspw_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, int oh,
                          int ow, void* stream);
/* nn.AdaptiveAvgPool2d(s) (models/clip_psp.py:85-87,160-166, models/models.py:947,972): y [n][s][s][c]. */
int vspw_adaptive_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int s, void* stream);
/* All pyramid scales in one pass over x (c % 4 == 0): y[k] [n][s_k][s_k][c].  scales / y are HOST arrays of nscales
 * (<= 4) entries; ws holds the per-row partial sums (vspw_pyramid_pool_fwd_workspace bytes). */
size_t vspw_pyramid_pool_fwd_workspace(const int* scales, int nscales, int n, int h, int c);
int vspw_pyramid_pool_fwd(const float* x, const int* scales, int nscales, float* const* y, int n, int h, int w, int c,
                          void* ws, size_t ws_bytes, void* stream);
/* dx (+)= adjoint; accumulate != 0 adds into dx. */
int vspw_adaptive_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int s, int accumulate,
                              void* stream);
/* Fused adjoint of the PPM pyramid: dx = sum_k adaptive_avgpool_bwd(dy[k], scales[k]) in one pass; with T > 1 each
 * dy[k] is the gradient of the temporally BLENDED [n/T][s][s][c] tensor (uniform weights) and the temporal-mean
 * adjoint (x 1/T to every frame of the clip) is folded in.  dy / scales are HOST arrays of nscales (<= 4) entries. */
int vspw_pyramid_pool_bwd(const float* const* dy, const int* scales, int nscales, float* dx, int n, int h, int w,
                          int c, int T, void* stream);
/* Temporal Context Blending of TCB-PSP (models/clip_psp.py:181-188): frames are stacked frame-major along the
 * batch ([t][b]); y[b] = (1/T) sum_t x[t*B + b] * (wts ? wts[b*T + t] : 1). `inner` = elements per image. */
int vspw_temporal_mean_fwd(const float* x, const float* wts, float* y, int T, int B, long long inner, void* stream);
int vspw_temporal_mean_bwd(const float* dy, const float* wts, float* dx, int T, int B, long long inner,
                           void* stream);

/* dwts[b*T + j] (+)= (1/T) * <dy[b], x[frame j of clip b]>: gradient of the psp_weight temporal weights
 * (models/clip_psp.py:147-152,184-186). */
int vspw_temporal_mean_wgrad(const float* dy, const float* x, float* dw, int T, int B, long long inner,
                             int accumulate, void* stream);

/* F.avg_pool2d(x, (2,2)) on NHWC activations [n][h][w][c] -> [n][h/2][w/2][c] (stride 2, no padding, floor size) and
 * its adjoint: the `downsample` switch of the non-local decoders (models/non_local_models.py:30-32,136-137).  c % 4 == 0. */
int vspw_avgpool2x2_nhwc_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream);
int vspw_avgpool2x2_nhwc_bwd(const float* dy, float* dx, int n, int h, int w, int c, void* stream);

/* ---------------------------------------------------------------- bilinear (interp.hip) ----------- */
/* F.interpolate(mode='bilinear', align_corners=False) (models/clip_psp.py:49-52, models/models.py:96,102).
 * Output rows have ldo channels and the result lands at channel offset co (writes straight into the PPM concat).
 * The source rows have ldi channels, read at channel offset ci. */
int vspw_bilinear_fwd(const float* x, float* y, int n, int ih, int iw, int oh, int ow, int c, int ldi, int ci,
                      int ldo, int co, void* stream);
int vspw_bilinear_bwd(const float* dy, float* dx, int n, int ih, int iw, int oh, int ow, int c, int ldi, int ci,
                      int ldo, int co, void* stream);
/* Strided channel-slice copy: dst[row][dco + j] = src[row][sco + j], j < c (torch.cat along channels). */
int vspw_copy_channels(const float* src, float* dst, long long rows, int c, int lds, int sco, int ldd, int dco,
                       void* stream);
/* dst[row][j] += src[row][sco + j] */
int vspw_add_channels(const float* src, float* dst, long long rows, int c, int lds, int sco, int ldd, int dco,
                      void* stream);

/* ---------------------------------------------------------------- softmax / loss (loss.hip) ------- */
/* (log_)softmax over the contiguous last dimension (class channel of NHWC): F.log_softmax(dim=1) of
 * models/clip_psp.py:198,212, softmax(dim=-1) of spatial_ocr_block.py:270.  y = softmax(alpha * x). */
int vspw_softmax_lastdim_fwd(const float* x, float* y, long long rows, int k, float alpha, int log, void* stream);
/* log: dx = dy - exp(y)*sum(dy);  prob: dx = alpha*y*(dy - sum(dy*y)) */
int vspw_softmax_lastdim_bwd(const float* dy, const float* y, float* dx, long long rows, int k, float alpha, int log,
                             void* stream);
/* softmax over the pixel dimension of [b][hw][k] (spatial_ocr_block.py:104: F.softmax(probs, dim=2)). */
int vspw_softmax_pixels_fwd(const float* x, float* y, int b, int hw, int k, float alpha, void* stream);
int vspw_softmax_pixels_bwd(const float* dy, const float* y, float* dx, int b, int hw, int k, float alpha,
                            void* stream);
/* Fused  F.interpolate(logp,(H,W),bilinear) -> NLLLoss(ignore_index) -> pixel_acc  of models/clip_psp.py:198-216,
 * models/models.py:92-107.  logp [n][h][w][k] are log-probabilities at feature resolution; label [n][H][W] is int64,
 * or fp32 when label_f32 != 0 (the drivers hand over float labels; the kernel applies the reference's label.long()).
 * out[0]=VSPW_NLL_FIXED * sum of -logp_up[label] over non-ignored pixels (each workgroup's partial sum is rounded to a
 * multiple of 1/VSPW_NLL_FIXED, so the fp64 atomic accumulation adds integers and is bit-reproducible whatever the
 * order), out[1]=#non-ignored, out[2]=#(argmax==label), out[3]=#(label>=0) (fp64).  loss = out[0]/VSPW_NLL_FIXED/out[1].
 * out must be zeroed by the caller (vspw_zero_f64). */
#define VSPW_NLL_FIXED 1048576.0
int vspw_seg_nll_fwd(const float* logp, const void* label, int label_f32, double* out, int n, int h, int w, int k,
                     int H, int W, int ignore_index, int want_acc, void* stream);
/* Gathers the bilinear adjoint of -gscale/count at the label channel into d(loss)/d(logp) [n][h][w][k]; with
 * lsm_jacobian != 0 it also applies the log-softmax Jacobian, i.e. returns d(loss)/d(logits) for
 * logp = log_softmax(logits).  gscale is a device scalar (the incoming gradient of the loss). */
int vspw_seg_nll_bwd(const float* logp, const void* label, int label_f32, const double* fwd_out, const float* gscale,
                     float* dlogits, int n, int h, int w, int k, int H, int W, int ignore_index, int lsm_jacobian,
                     void* stream);
/* Inference head: probs[n][H][W][k] = softmax_k(bilinear(logits)) (models/clip_psp.py:190-194). */
int vspw_upsample_softmax(const float* logits, float* probs, int n, int h, int w, int k, int H, int W,
                          void* stream);
int vspw_zero_f64(double* p, long long n, void* stream);

/* ---------------------------------------------------------------- misc (misc.hip) ----------------- */
/* [b][r][c] -> [b][c][r] */
int vspw_transpose_batched(const float* in, float* out, int b, int r, int c, void* stream);
/* Batched plain GEMMs of the OCR object attention / context gather (models/ocr_modules/spatial_ocr_block.py:100-109,
 * 252-274: torch.matmul / torch.bmm on [B, ., .] operands), all operands contiguous, the batch a grid dimension:
 *   vspw_bmm_nt: c[b] = a[b] @ bt[b]^T     a [B][M][K], bt [B][N][K], c [B][M][N]
 *   vspw_bmm_tn: c[b] = a[b]^T @ b_[b]     a [B][R][M], b_ [B][R][N], c [B][M][N]   (split over R, fixed-order sums) */
int vspw_bmm_nt(const float* a, const float* bt, float* c, int batch, int m, int n, int k, void* stream);
size_t vspw_bmm_tn_workspace(int batch, int r, int m, int n);
int vspw_bmm_tn(const float* a, const float* b, float* c, int batch, int r, int m, int n, void* ws, size_t ws_bytes,
                void* stream);
/* colsum[c] = sum_rows a[row][c]  (bias gradients) */
size_t vspw_colsum_workspace(long long rows, int c);
int vspw_colsum(const float* a, float* out, long long rows, int c, void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum_rows a[row][c]*b[row][c]  (gradients of per-channel blend weights) */
int vspw_colsum_prod(const float* a, const float* b, float* out, long long rows, int c, void* ws, size_t ws_bytes,
                     void* stream);
/* y = a*x + b*y elementwise */
int vspw_axpby(const float* x, float* y, long long n, float a, float b, void* stream);
/* out = w0[c]*a + w1[c]*b per channel (NetWarp blend, models/netwarp.py:201,216-217) and its adjoints. */
int vspw_chan_blend_fwd(const float* a, const float* b, const float* w0, const float* w1, float* out,
                        long long rows, int c, void* stream);
/* out[row][c] = w[c]*g[row][c] */
int vspw_chan_scale(const float* g, const float* w, float* out, long long rows, int c, void* stream);
/* flow-warp: grid_sample(bilinear, zeros, align_corners=False) with grid = 2*(xy+flow)/(dim-1)-1
 * (models/netwarp.py:12-37).  x [n][h][w][c], flow [n][h][w][2] (dx,dy), y same shape as x. */
int vspw_flowwarp_fwd(const float* x, const float* flow, float* y, int n, int h, int w, int c, void* stream);
int vspw_flowwarp_bwd(const float* dy, const float* x, const float* flow, float* dx, float* dflow, int n, int h,
                      int w, int c, void* stream);
/* The same warp with grid_sample(mode='nearest') - source coordinate rounded half-to-even, 0 outside the image: how the
 * temporal-consistency metric carries the next frame's label map onto the current one (TC_cal.py:12-38, :112). */
int vspw_flowwarp_nearest(const float* x, const float* flow, float* y, int n, int h, int w, int c, void* stream);

/* ---------------------------------------------------------------- RAFT flow network (raft.hip) ----- */
/* Forward-only kernels of the frozen RAFT that produces the flow consumed by vspw_flowwarp_fwd
 * (models/netwarp.py:170-176 -> RAFT_core/raft.py:75-127, iters=20, test_mode=True).  Convolutions use
 * vspw_conv2d_fwd_ex; the all-pairs correlation (RAFT_core/corr.py:54-62) is the same NT GEMM. */
/* nn.InstanceNorm2d (no affine, eps) coefficients of x [n][hw][c], c <= 256: scale[n][c] = 1/sqrt(var+eps),
 * shift[n][c] = -mean*scale  (RAFT_core/extractor.py:27-31,131). */
size_t vspw_instance_norm_workspace(int n, int hw, int c);
int vspw_instance_norm_coeffs(const float* x, int n, int hw, int c, float eps, float* scale, float* shift, void* ws,
                              size_t ws_bytes, void* stream);
/* y = [relu_out]( [residual +] [relu_in]( x*scale[img*coef_stride + ch] + shift[...] ) ): the norm -> relu ->
 * (x + y) -> relu chain of RAFT_core/extractor.py:44-56; coef_stride = c for instance norm, 0 for eval BatchNorm. */
int vspw_affine_act(const float* x, const float* scale, const float* shift, int coef_stride, const float* residual,
                    int relu_in, int relu_out, float* y, int n, int hw, int c, void* stream);
/* F.avg_pool2d(x, 2, stride=2) over the last two dims of [planes][h][w] (RAFT_core/corr.py:27-29). */
int vspw_avgpool2x2(const float* in, float* out, long long planes, int h, int w, void* stream);
/* CorrBlock.__call__ (RAFT_core/corr.py:31-52): for every pixel row = b*h1*w1 + i with centre (x_i, y_i) + flow[row],
 * out[row][l*81 + a*9 + c] = bilinear sample (align_corners=True, zeros) of level l's [h1>>l][w1>>l] plane of that row
 * at (cx/2^l + a - 4, cy/2^l + c - 4).  flow row stride ldf >= 2, out row stride ldo >= 324; h1, w1 >= 16. */
int vspw_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, const float* flow,
                     long long ldf, float* out, long long ldo, int b, int h1, int w1, void* stream);
/* SepConvGRU gates (RAFT_core/update.py:44-60); zr[row] = [z(c) | r(c)].  out = r*h;  h = (1-z)*h + z*q. */
int vspw_gru_rh(const float* zr, long long ldzr, const float* h, long long ldh, float* out, long long ldo,
                long long rows, int c, void* stream);
int vspw_gru_update(const float* zr, long long ldzr, const float* q, long long ldq, float* h, long long ldh,
                    long long rows, int c, void* stream);
/* RAFT.upsample_flow (RAFT_core/raft.py:57-68): flow [n][h][w] rows of (fx, fy) (stride ldf), mask [n][h][w][576]
 * (stride ldm, channel = k*64 + i*8 + j), softmax over k of mask_scale*mask; out NCHW [n][2][8h][8w]. */
int vspw_convex_upsample(const float* flow, long long ldf, const float* mask, long long ldm, float mask_scale,
                         float* out, int n, int h, int w, void* stream);

/* ---------------------------------------------------------------- input pipeline (data.hip) ------- */
/* Device side of the reference's dataset transforms (dataset2.py:852-1048 BaseDataset_longclip, :657-850
 * BaseDataset_clip, :154-490 test datasets) on decoded uint8 frames resident in HBM. */
/* One separable pass of Pillow's 8-bit resampler (Image.resize(BILINEAR), dataset2.py:1024): bounds[o] = (first,
 * count), kk[o][ksize] = Pillow's 22-bit fixed-point coefficients (host tables, device memory); axis 0 resamples
 * columns (out_h == in_h), axis 1 rows (out_w == in_w).  in/out: interleaved [h][w][channels] u8.  flip != 0 reads
 * the source mirrored along x (the reference flips the PIL image before it resizes it, dataset2.py:1015-1017). */
int vspw_resample_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize, int in_h,
                     int in_w, int out_h, int out_w, int channels, int axis, int flip, void* stream);
/* Image.resize(NEAREST) of the label map (dataset2.py:1025): out[y][x] = in[ytab[y]][xtab[x]]. */
int vspw_gather_u8(const uint8_t* in, uint8_t* out, const int32_t* xtab, const int32_t* ytab, int in_w, int out_h,
                   int out_w, int flip, void* stream);
/* flip + zero/255 pad + crop + /255 + Normalize(mean, std) + NHWC, and the label remap 0->255, v->v-1 as float
 * (dataset2.py:921-947,964-977), into one frame slot of the batch: img_out [out_h][out_w][3], lab_out
 * [out_h][out_w] (either may be NULL; lab may be NULL).  mean3 / std3 are HOST pointers to 3 floats. */
int vspw_frame_transform(const uint8_t* img, const uint8_t* lab, int h, int w, int flip, int pad_h, int pad_w,
                         int crop_y, int crop_x, int out_h, int out_w, const float* mean3, const float* std3,
                         float* img_out, float* lab_out, void* stream);

/* torch.optim.SGD(momentum, weight_decay) update applied `mult` times in a row, weight decay accumulating
 * in the gradient across the applications as in the pinned PyTorch 1.3.1 (in-place d_p.add_(wd, p)) (the reference's
 * parameter-group generators yield a parameter once per enclosing module, train_clip2.py:215-236 +
 * models/clip_psp.py:99-135).  p, g, buf are dense tensors with identical strides; first != 0 initialises buf. */
int vspw_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float wd, float momentum, int mult,
                  int first, void* stream);
/* The same update for EVERY parameter in one launch.  `entries` is a DEVICE array of n_entries records sorted by
 * chunk0 (chunk0 = number of vspw_sgd_chunk_elems()-sized chunks of all preceding tensors); total_chunks = grid size.
 * lr_table (device, may be NULL): when given, an entry with lr_slot >= 0 takes its learning rate from
 * lr_table[lr_slot] instead of entry.lr - the poly schedule of train_clip2.py:239-252 then only rewrites that small
 * array and the entry table stays constant across steps (needed by a captured hipGraph of the training step). */
typedef struct vspw_sgd_entry {
    float* p;
    const float* g;
    float* buf;
    long long n;
    long long chunk0;
    float lr, wd;
    int mult, first;
    int lr_slot, reserved;
} vspw_sgd_entry;
long long vspw_sgd_chunk_elems(void);
int vspw_sgd_multi(const vspw_sgd_entry* entries, int n_entries, long long total_chunks, float momentum,
                   const float* lr_table, void* stream);

/* ---------------------------------------------------------------- non-local affinity (nonlocal.hip) ---- */
/* out[b][i][:] = scale * sum_j (q[b][i][:] . k[b][j][:]) * v[b][j][:]   for q, k, v, out [b][n][c] (pixel rows x channels,
 * i.e. the [B,N,C] matrices NLBlockND's view/permute chains produce, zero-copy from NHWC memory), c in {32, 64, 128}.
 * Replaces  f = matmul(theta_x, phi_x); f_div_C = f / N; y = matmul(f_div_C, g_x)  of models/non_local.py:116-133
 * (mode 'dot', used by Non_local2d / Non_local3d, models/non_local_models.py:19-72,124-151) without materialising the
 * n x n affinity: y = nl(theta, phi, g; 1/N).  The same entry point evaluates the three gradients
 * (d theta = nl(dy, g, phi), d g = nl(phi, theta, dy), d phi = nl(g, dy, theta)).  ws: vspw_nl_dot_workspace() bytes
 * (per-key-chunk partial outputs, summed in chunk order: deterministic). */
size_t vspw_nl_dot_workspace(int b, int n, int c);
int vspw_nl_dot(const float* q, const float* k, const float* v, float* out, int b, int n, int c, float scale, void* ws,
                size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- flow plumbing (misc.hip) ---------- */
/* dst [n][h][w][c] = src [n][oh][ow][c] at pixels (stride*oy, stride*ox), zero elsewhere (c % 4 == 0): second half of the
 * data gradient of a strided pointwise convolution (models/resnet.py:125-131, the 1x1 stride-2 downsample) computed as a plain
 * GEMM on the output pixels (vspw_conv2d_bwd_data with a stride-1 descriptor of the OUTPUT size) + this scatter. */
int vspw_strided_scatter_nhwc(const float* src, float* dst, int n, int oh, int ow, int h, int w, int c, int stride,
                              void* stream);
/* Plane gathers on NCHW data [planes][h][w] around the flow network of the NetWarp heads:
 *   vspw_nearest_resize_fwd / bwd  F.interpolate(flow, size, mode='nearest') and its adjoint (models/netwarp.py:199,214;
 *                                  models/netwarp_ocr.py:252): src = min(floor(dst * (float)in / out), in - 1);
 *   vspw_plane_shift               out[y][x] = in[y - top][x - left], 0 outside: F.pad(mode='constant') of RAFT's
 *                                  InputPadder (RAFT_core/utils/utils.py:7-25) and, with negative offsets, its unpad crop;
 *   vspw_unnormalize_rgb           (x * std[c] + mean[c]) * post on [n][3][hw] (models/netwarp.py:186-187). */
int vspw_nearest_resize_fwd(const float* in, float* out, long long planes, int h, int w, int oh, int ow, void* stream);
int vspw_nearest_resize_bwd(const float* dout, float* din, long long planes, int h, int w, int oh, int ow, void* stream);
int vspw_plane_shift(const float* in, float* out, long long planes, int h, int w, int oh, int ow, int top, int left,
                     void* stream);
int vspw_unnormalize_rgb(const float* in, float* out, int n, long long hw, float s0, float s1, float s2, float m0, float m1,
                         float m2, float post, void* stream);

/* ---------------------------------------------------------------- peer statistics exchange (exchange.hip) --- */
/* The cross-replica sum of SynchronizedBatchNorm (models/sync_batchnorm/batchnorm.py:110-131, comm.py:59-137: the
 * master replica collects [sum, ssum, count] of every device and hands the totals back) for one process per GPU: every
 * rank owns an arena in its HBM which all peers map through hipIpc; one single-workgroup kernel per exchange pushes
 * the rank's 2*C doubles into every arena, publishes a sequence number, waits for the others' and adds the `world`
 * contributions in rank order (bit-identical totals on every rank).  See csrc/exchange.hip for the protocol.
 *   vspw_xchg_arena_bytes   size of one arena for `world` ranks (<= 16) and messages of up to slot_doubles doubles
 *   vspw_xchg_alloc         allocate + zero this rank's arena (uncached device memory), export its IPC handle
 *                           (vspw_xchg_handle_bytes() bytes of HOST memory); synchronous, start-up only
 *   vspw_xchg_open / close  map / unmap a peer's arena from its handle;  vspw_xchg_free: release the own arena
 *   vspw_xchg_allreduce_f64 in-place sum of data[0..n) over the ranks; arenas: HOST array of `world` device pointers
 *                           (own arena at [rank]); counter: device uint64 (zero at start-up, advanced by the kernel:
 *                           hipGraph replays stay in sequence); status: device int, 1 after a wait exceeded timeout_s
 *                           (the result is then NaN - a dead peer never hangs the GPU). */
size_t vspw_xchg_arena_bytes(int world, int slot_doubles);
int vspw_xchg_handle_bytes(void);
int vspw_xchg_alloc(size_t bytes, void** arena, void* handle_out);
int vspw_xchg_open(const void* handle, void** arena);
int vspw_xchg_close(void* arena);
int vspw_xchg_free(void* arena);
int vspw_xchg_allreduce_f64(double* data, int n, void* const* arenas, int world, int rank, unsigned long long* counter,
                            int slot_doubles, double timeout_s, int* status, void* stream);
/* The exchange of sums [2][c] and the vspw_bn_finalize / vspw_bn_finalize_clamped (clamp_var) that consumes the totals,
 * in one launch (batchnorm.py:110-150); count = rows behind the totals over all ranks. */
int vspw_xchg_bn_finalize(double* sums, int c, void* const* arenas, int world, int rank, unsigned long long* counter,
                          int slot_doubles, double timeout_s, int* status, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          float* mean, float* invstd, float* scale, float* shift, int clamp_var, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSPW_HIP_H */
