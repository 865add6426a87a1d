/* xrface.h -- C ABI of the MI355X (gfx950) hot-path library for the cross-resolution
 * face-recognition training/eval path.
 *
 * The reference (HyoKong/Cross-Resolution-Face-Recognition) has NO native code and no FFI: its
 * only "operator API" is the PyTorch nn.Module / ATen op protocol.  Each
 * entry point below therefore cites the ATen op call site in the reference that it replaces.
 * The Python host (cross-resolution-face-recognition_amd/xrface) binds these with ctypes and wraps
 * them in torch.autograd.Function objects living inside nn.Module subclasses that keep the
 * reference's class names, constructor signatures, forward tuples and state_dict keys.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator); the library
 *     never allocates or frees device memory and keeps no mutable global state;
 *   - activations are NHWC ("channels_last"): [N][H][W][C], C innermost;
 *   - dtype: XR_BF16 (bf16 storage, fp32 accumulate on MFMA) or XR_F32 (fp32 storage; each operand is
 *     split into three bf16 planes that together carry all 24 significand bits and the matrix products
 *     run as the six plane-pair MFMAs with fp32 accumulate -- fp32-level accuracy at 6/16 of the
 *     native fp32-MFMA cost);
 *   - stream: hipStream_t passed as void*; work is enqueued on it and never synchronised (one documented exception: xr_pack_plan,
 *     a set-up call outside the training loop's steady state, waits for its table upload);
 *   - return: 0 on success, negative XR_E_* otherwise; xr_last_error() gives a thread-local
 *     message.  No C++ exception crosses the boundary.  All entry points are re-entrant.
 */
#ifndef XRFACE_H
#define XRFACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XR_OK 0
#define XR_E_INVALID (-1)   /* bad argument / unsupported shape */
#define XR_E_LAUNCH (-2)    /* HIP launch error */
#define XR_E_NODEVICE (-3)  /* no usable gfx950 device */

#define XR_BF16 0
#define XR_F32 1
#define XR_F32X2 2  /* xr_conv_igemm / xr_conv_wgrad only: fp32 tensors, operands split into TWO bf16 planes (hi + lo), three
                     * plane-pair MFMAs per product (hi*hi, hi*lo, lo*hi; ~16 significand bits) -- half the matrix work of XR_F32;
                     * every other entry point sees such tensors as XR_F32 */

#define XR_ACT_NONE 0
#define XR_ACT_PRELU 1  /* per-channel slope */
#define XR_ACT_RELU 2
#define XR_ACT_TANH 3  /* SUPER_RESOLUTION/model/FSRnet.py:292 */

const char* xr_last_error(void);
int xr_version(void);
/* number of CUs of the current device (used by hosts to size split factors); <0 on error */
int xr_device_cus(void);
/* tuning knobs for A/B measurements in one process: 0 = igemm LDS stage buffers (1|2), 1 = wgrad LDS stage buffers */
int xr_tune(int knob, int value);
/* Deterministic reductions (debugging replica divergence, bit-exact repeat tests): on != 0 makes every kernel of this library
 * form its fp32 sums in a fixed order -- one reduction block per statistics group (hosts then use one group per image / tile and
 * fold them in order), weight-gradient slices summed by one group, loss scalars by one block.  The host side (xrface.ops) must
 * also give the convolution epilogues one partial row per tile (ep_spread >= row tiles) and skip split-K; slower, same results
 * up to summation order.  Process-wide. */
int xr_set_deterministic(int on);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  Parameters stay fp32 nn.Parameters in the reference layouts
 * (Conv2d [K][C][R][S], model/FSRnet.py:79; ConvTranspose2d [Cin][Cout][R][S], model/FSRnet.py:436;
 * Linear [K][C*H*W], model_irse.py:147).  Before a step the host packs them for the implicit-GEMM
 * kernels:   dst[a][t*Bp + b] = src[a1*sa1 + a2*sa2 + t*st + b*sb],  a = a1*A2 + a2,
 * rows padded with zeros to Kg (multiple of 64), b padded to Bp (multiple of 8).
 * nplanes = 1 (XR_BF16), 3 (XR_F32: w = p0 + p1 + p2, each bf16, together all 24 significand bits) or 2 (XR_F32X2: p0 + p1);
 * dst holds the planes back to back: [nplanes][A1*A2][Kg] bf16. */
int xr_pack_weight(const float* src, void* dst, int nplanes, int A1, int A2, int taps, int B, int Bp, int Kg,
                   int64_t sa1, int64_t sa2, int64_t st, int64_t sb, void* stream);
/* Batched refresh: a training step re-packs every convolution weight after the optimizer update (~110 small launches for
 * IR-SE-50); a plan turns them into one launch.  entries: n x 14 int64 host words per parameter
 * {src, dst, nplanes, A1, A2, taps, B, Bp, Kg, sa1, sa2, st, sb, 0} with xr_pack_weight's meaning; xr_pack_plan writes the
 * device table (n * 128 bytes at table_dev), stores the dynamic-LDS size in *smem_out and returns the grid size;
 * xr_pack_run(table_dev, n, blocks, smem) then refreshes all packs from the current parameter values.
 * xr_pack_plan uploads the table from a host staging vector and WAITS for that copy on `stream` before returning (the only
 * entry point that synchronises; it runs once per distinct set of stale packs -- a handful of times per training run -- and the
 * caller caches the table; xr_pack_run, the per-step call, does not synchronise). */
int xr_pack_plan(const int64_t* entries, int n, void* table_dev, int* smem_out, void* stream);
int xr_pack_run(const void* table_dev, int n, int blocks, int smem, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution (MFMA 32x32x16 bf16, LDS-staged NHWC tiles).
 * Replaces aten::conv2d / conv_transpose2d / linear forward and input-gradient:
 *   model/FSRnet.py:79,85,110,312,318,345,351,384,391,392,432,436,439; model_irse.py:56-60,140,147;
 *   model/resnet.py:9-16,158,170.
 * out[n,ho,wo,k] = bias[k] + sum_{r,s,c} in[n, hi, wi, c] * wpack[k][(r*S+s)*C + c]
 *   transposed == 0:  hi = ho*stride - pad + r                      (conv forward)
 *   transposed == 1:  hi = (ho + pad - r)/stride when divisible     (conv dgrad / deconv forward)
 * in: [N][H][W][C] (C % 8 == 0), out: [N][Ho][Wo][K] with row pitch ldo (elements) at channel
 * offset 0 of the caller's pointer (lets producers write into a concat buffer). */
int xr_conv_igemm(int dtype, const void* in, const void* w, const float* bias, void* out,
                  int N, int H, int W, int C, int Ho, int Wo, int K, int R, int S, int stride, int pad,
                  int transposed, int Kg, int ldo, float* splitk_ws, int splitk, const void* ep_src,
                  const float* ep_alpha, float* ep_dalpha, int ep_spread, void* ep2_out, float* ep_red, const void* ep_add, void* stream);
/* ep_src != NULL fuses a PReLU backward into the epilogue (input gradient of conv(prelu(y)), model_irse.py:59):
 * out = acc * (y > 0 ? 1 : alpha[c]) and dalpha[c] += sum acc*y*[y <= 0], with y = ep_src laid out like `out`.
 * ep_dalpha is [ep_spread][K] fp32, zero-initialised (or holding a running sum) by the caller: row tile i adds its partial
 * sums into row i % ep_spread, so the atomics of thousands of tiles do not all land on the same few cache lines; the
 * caller folds the rows with xr_reduce_groups.  ep_spread <= 1: a single [K] row.
 * ep2_out != NULL (exclusive with ep_src) fuses a PReLU FORWARD: besides `out` the kernel writes ep2_out = prelu(out, ep_alpha)
 * with the same layout -- conv -> PReLU -> conv (model_irse.py:56-60) then needs no separate activation pass.
 * ep_red != NULL (with ep_src = x, the INPUT of the BatchNorm whose output this convolution consumed; the tensor being
 * written is then dL/d(bn output)): the epilogue also accumulates the two reductions of the BatchNorm backward,
 * ep_red[0][r][c] += sum d and ep_red[1][r][c] += sum d*x (r = row tile %% ep_spread; layout [3][ep_spread][K], zeroed by the
 * caller, folded by xr_norm_bwd_coeffs(fold = ep_spread)) -- the separate reduction pass over (d, x) disappears.
 * ep_red != NULL with ep_src == NULL: x := the output itself, i.e. ep_red[0] / ep_red[1] receive the per-channel sum and sum
 * of squares of `out` -- the batch statistics of a BatchNorm that follows the convolution (xr_norm_finalize(fold = ep_spread)).
 * ep_add != NULL (no other epilogue fusion): out += ep_add (same layout) -- the input gradient arriving through an identity /
 * residual branch of the same input is summed in the dgrad epilogue instead of by a separate elementwise pass. */
/* split-K (long reductions with few output tiles, e.g. Linear(25088->512) at batch 256): splitk > 1 slices of the
 * K loop accumulate with fp32 atomics into splitk_ws [N*Ho*Wo][ldo] (zeroed by the caller); `out` is then
 * produced by xr_bias_cast.  splitk_ws == NULL / splitk <= 1: direct epilogue.
 * Transposed gathers with stride > 1 run class-wise: output pixels are grouped by (ho % stride, wo % stride)
 * and each group only visits the taps it can see (no multiply-by-zero work). */
int xr_bias_cast(int dtype, const float* ws, const float* bias, void* out, int64_t M, int K, int ld, void* stream);

/* Input gradient of a 3x3 / stride 2 / pad 1 convolution (conv2 of every stage-opening bottleneck_IR(_SE), model_irse.py:60;
 * model/resnet.py:24) as ONE dense stride-1 implicit GEMM instead of the strided transposed gather: a 2x2 window of
 * dy [N][Ho][Wo][Kp] feeds the four sub-pixel classes (a, b) of dx [N][2 Ho][2 Wo][C2] at once (class (0,0) sees 1 tap of the 3x3
 * kernel, (0,1) and (1,0) two, (1,1) four: the pack holds zeros for the other window positions, 16/9 of the algorithmic MACs at
 * the dense kernel's rate), and the epilogue scatters the 4 * C2 GEMM columns depth-to-space.  wpack: xr_pack_dgrad_s2 of the
 * convolution's [K][C][3][3] parameter, [nplanes][4 * C2][Kg], Kg >= 4 * Kp.  Epilogue fusions as xr_conv_igemm, addressed on dx:
 * ep_src + ep_alpha4 ([4 * C2]: alpha repeated per class) + ep_dalpha [ep_spread][4 * C2] (fold the 4 classes with
 * xr_reduce_groups(G = 4 * ep_spread)); ep_red [3][ep_spread][4 * C2] (= [3][4 * ep_spread][C2]); ep_add. */
int xr_pack_dgrad_s2(const float* w, void* dst, int nplanes, int K, int C, int Kp, int C2, int Kg, void* stream);
int xr_conv_dgrad_s2(int dtype, const void* dy, const void* wpack, void* dx, int N, int Ho, int Wo, int Kp, int C2, int Kg,
                     const void* ep_src, const float* ep_alpha4, float* ep_dalpha, int ep_spread, float* ep_red,
                     const void* ep_add, void* stream);

/* Weights-stationary direct convolution for the 64 -> 64 channel 3x3 / stride 1 / pad 1 bf16 layers (model/FSRnet.py:79,85: every
 * FSRNet body layer; model_irse.py:59 and model/resnet.py:9-12 stage 1).  Same result as xr_conv_igemm(XR_BF16, ..., C = K = 64,
 * R = S = 3, stride 1, pad 1, transposed) on the same weight pack (wpack = [64][576] bf16, one plane; for transposed != 0 the
 * input-gradient pack), with these fusions:
 *   in_scale / in_shift [N][64] (+ in_alpha [64], optional): the input is transformed ON LOAD, per image n and channel c,
 *     x' = prelu(x * in_scale[n][c] + in_shift[n][c], in_alpha[c]) -- the InstanceNorm apply + PReLU of the producing layer
 *     (model/FSRnet.py:81-84,92-94), zero padding applies to x'; the normalised activation is never written to HBM;
 *   out_stats [2][N][64] (zeroed by the caller): += per-image sum and sum of squares of the output as rounded to bf16 -- the
 *     statistics of the InstanceNorm that follows (xr_norm_finalize(G = N));
 *   ep_add (laid out like out): out += ep_add (residual-branch gradient, as xr_conv_igemm's ep_add); exclusive with out_stats. */
int xr_conv64_direct(const void* in, const void* wpack, const float* bias, void* out, int N, int H, int W, int transposed,
                     const float* in_scale, const float* in_shift, const float* in_alpha, float* out_stats, const void* ep_add,
                     void* stream);

/* The forward convolution (no bias) with a SECOND output out2 = prelu(out, alpha[k]) taken from the rounded value just stored:
 * conv -> PReLU -> conv of the IR units (model_irse.py:57-61) -- the activation pass disappears, as with xr_conv_igemm's ep2_out. */
int xr_conv64_direct_prelu(const void* in, const void* wpack, void* out, void* out2, const float* alpha, int N, int H, int W,
                           void* stream);

/* The same convolution (no bias, no on-load transform) whose OUTPUT is the gradient dy of y = prelu(c * red_scale[n] + red_shift[n],
 * red_alpha) -- conv2's input gradient in the FSRNet residual block (model/FSRnet.py:81-85): while the output streams out, the
 * epilogue loads the matching chunk of c = red_src (laid out like out) and accumulates, per image n and channel,
 *   red[0][n][ch] += sum dz,  red[1][n][ch] += sum dz * c,  red[2][n][ch] += sum dy * z * [z <= 0],   z = c * scale + shift,
 *   dz = dy * (z > 0 ? 1 : alpha)   (dy as rounded to bf16; red_alpha NULL: slope 1)
 * i.e. exactly the three sums of xr_affine_act_bwd_reduce(x = c, dy = out, act = PReLU, G = N): that pass disappears.
 * red [3][N][64] fp32, zeroed by the caller. */
int xr_conv64_direct_bwdred(const void* in, const void* wpack, void* out, int N, int H, int W, int transposed, const void* red_src,
                            const float* red_scale, const float* red_shift, const float* red_alpha, float* red, void* stream);

/* Two FSRNet residual blocks chained in the backward pass.  The convolution is conv1's input gradient of block i + 1; with
 * dout = conv + ep_add (ep_add = that block's residual-branch gradient) being the gradient that enters the tail
 * out = prelu(tail_c * tail_scale[n] + tail_shift[n] + tail_x, tail_alpha) of block i (model/FSRnet.py:90-98), the kernel stores
 *   out = dz = dout * (z > 0 ? 1 : alpha),  z = c * scale + shift + x           (dout rounded to bf16 first)
 * -- the gradient of the tail's pre-activation: block i's residual-branch gradient AND the input of its InstanceNorm backward --
 * and accumulates red[0] += sum dz, red[1] += sum dz * c, red[2] += sum dout * z * [z <= 0] per image and channel
 * (xr_affine_act_bwd_reduce(x = c, res = x, dy = dout)).  Block i then runs neither that reduce pass nor the residual half of its
 * apply pass.  red [3][N][64] fp32, zeroed by the caller; all tensors laid out like out. */
int xr_conv64_direct_tailred(const void* in, const void* wpack, void* out, int N, int H, int W, int transposed, const void* ep_add,
                             const void* tail_c, const void* tail_x, const float* tail_scale, const float* tail_shift,
                             const float* tail_alpha, float* red, void* stream);

/* Weight gradient (aten::convolution_backward weight part; same call sites as above).
 * slab[s][k][t*C + c] = sum_{m in slice s} dy[m][k] * gather(in)[m][t][c]   (fp32, PACKED layout [K][Kg])
 * where m runs over the N*Ho*Wo pixels of dy (row pitch ldy), split into `split` contiguous slices, and gather()
 * is the same (transposed) tap gather as xr_conv_igemm on `in` [N][H][W][C].  Every slice writes its own slab with
 * plain stores (no atomics, no zero-init): dwp must hold split*K*Kg floats.
 * RETURNS the number of slabs written (1 <= value <= split) on success, a negative XR_E_* on failure. */
int xr_conv_wgrad(int dtype, const void* in, const void* dy, float* dwp, int N, int H, int W, int C, int Ho,
                  int Wo, int K, int R, int S, int stride, int pad, int transposed, int ldy, int Kg, int split,
                  void* stream);
/* Direct weight gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 bf16 layers (same call sites as xr_conv64_direct; bf16 in /
 * dy [N][H][W][64]): the result xr_conv_wgrad(XR_BF16, ..., C = K = 64, R = S = 3, stride 1, pad 1, ldy = 64, Kg = 576) gives, as
 * per-workgroup partial slabs [s][64][576] in the same packed layout (sum them with xr_unpack_wgrad).  Every input and dy row is
 * read from HBM once (persistent workgroups walk image rows, the nine taps are shifted LDS views).  Needs W % 8 == 0, W <= 112.
 *   in_scale / in_shift [N][64] (+ in_alpha [64], optional): `in` is transformed ON LOAD as in xr_conv64_direct,
 *     x' = prelu(x * in_scale[n][c] + in_shift[n][c], in_alpha[c]) -- the weight gradient of a convolution whose forward consumed
 *     the normalised activation without ever writing it (model/FSRnet.py:81-85).
 * slabs must hold max_slabs * 64 * 576 floats.  RETURNS the number of slabs written (1 <= value <= max_slabs) or a negative XR_E_*. */
int xr_conv64_wgrad(const void* in, const void* dy, float* slabs, int N, int H, int W, int max_slabs, const float* in_scale,
                    const float* in_shift, const float* in_alpha, void* stream);
/* Direct weight gradient of the 3x3 / pad 1 / stride 1 or 2 bf16 convolutions with C % 64 == 0 and K % 64 == 0 (IR / ResNet body
 * layers, model_irse.py:56-62, model/resnet.py:24-47): the result of xr_conv_wgrad(XR_BF16, ..., R = S = 3, pad 1, ldy = K,
 * Kg = 9 * C) as `n` partial slabs [n][K][9 * C] in the same packed layout (sum them with xr_unpack_wgrad).  A workgroup owns a
 * 64 x 64 (k, c) tile and a run of output rows and stages only its 128-byte channel slices of dy [N][H/stride][W/stride][K] and
 * in [N][H][W][C], each once; stride 2 de-interleaves input columns on the way into LDS.  H, W multiples of the stride; output
 * width <= 112 (stride 1) / <= 64 (stride 2).  slabs must hold max_slabs * K * 9 * C floats.
 * RETURNS the number of slabs written (1 <= n <= max_slabs) or a negative XR_E_*. */
int xr_conv_wgrad_rows(const void* in, const void* dy, float* slabs, int N, int H, int W, int C, int K, int stride, int max_slabs,
                       void* stream);
/* Sum the `nslices` slabs and convert to the parameter layout (inverse of xr_pack_weight):
 * dst[a1*sa1 + a2*sa2 + t*st + b*sb] (+)= sum_s packed[s][a][t*Bp + b];  accumulate != 0 adds into dst. */
int xr_unpack_wgrad(const float* packed, float* dst, int A1, int A2, int taps, int B, int Bp, int Kg,
                    int64_t sa1, int64_t sa2, int64_t st, int64_t sb, int accumulate, int nslices, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Block-level entry points (SURVEY 8b "fused xr_resblock_{fwd,bwd}"): ONE call enqueues every launch of an FSRNet residual
 * block application (model/FSRnet.py:75-98: out = PReLU_out(IN2(conv2(PReLU(IN1(conv1(x))))) + x)); 64 channels, bf16 NHWC
 * tensors [N][H][W][64], W % 8 == 0 and W <= 112 for the backward (direct weight-gradient kernel).
 *   fwd: conv1 (+ statistics of c1) -> finalize -> conv2 with IN1 + PReLU on load (+ statistics of c2) -> finalize ->
 *        out = prelu_out(c2 * scale2 + shift2 + x).  Writes c1, c2, out and the eight [N][64] statistics / coefficient buffers.
 *   bwd: tail backward (reduce + coefficients + apply; with tail_red only coefficients + a two-input apply) -> conv2 input
 *        gradient with the IN1 + PReLU sums in its epilogue -> coefficients -> apply -> conv1 input gradient + residual-branch
 *        gradient (with prev_*: chained into the previous application's tail, xr_conv64_direct_tailred) -> both weight gradients
 *        (on side_stream after fork_event when given; the caller joins the streams).  dg / db / da targets are accumulated (+=).
 * Workspaces are caller-owned: ws_fwd 4*N*64 floats ZEROED; ws_bwd 12*N*64 floats, the first 6*N*64 ZEROED; prev_tail_red 3*N*64
 * ZEROED; slabs 2 * min(256, N*H) * 64 * 576 floats; dc2 / dy1 / dc1 / dres tensors laid out like x. */
typedef struct xr_resblock_desc {
  int N, H, W;
  float eps;
  const void* w1_fwd;      /* [64][576] bf16 forward packs (xr_pack_weight) */
  const void* w2_fwd;
  const void* w1_dgrad;    /* input-gradient packs (backward) */
  const void* w2_dgrad;
  const float* g1;         /* IN1 gamma / beta, PReLU slope [64] */
  const float* b1;
  const float* a1;
  const float* g2;         /* IN2 gamma / beta, output PReLU slope [64] */
  const float* b2;
  const float* ao;
  const void* x;           /* block input */
  void* c1;                /* conv1 output (saved) */
  void* c2;                /* conv2 output (saved) */
  void* out;               /* block output (fwd) */
  float* mean1;            /* [N][64] each: written by fwd, read by bwd */
  float* invstd1;
  float* scale1;
  float* shift1;
  float* mean2;
  float* invstd2;
  float* scale2;
  float* shift2;
  float* ws_fwd;
  /* ---- backward only */
  const void* dout;        /* gradient of out -- or, with tail_red, already dz = dout * prelu'(tail) */
  const float* tail_red;   /* [3][N][64] sums over (dz, c2) delivered by the next application's chained conv1 input gradient, or NULL */
  void* dx;                /* input gradient (NULL: not needed); chained: the previous application's dz */
  void* dc2;               /* workspace tensors */
  void* dy1;
  void* dc1;
  void* dres;              /* tail pre-activation gradient (unused with tail_red) */
  float* ws_bwd;
  float* dg1;              /* [64] each, accumulated; NULL: skipped */
  float* db1;
  float* da1;
  float* dg2;
  float* db2;
  float* dao;
  float* slabs;
  float* dw1;              /* [64][64][3][3] parameter-layout targets (NULL: that weight gradient is skipped) */
  float* dw2;
  int dw_accumulate;       /* 1: dw += , 0: dw = */
  int reserved;
  const void* prev_c2;     /* chaining (all or none): the previous application's c2, input x, scale2 / shift2 [N][64], output slope */
  const void* prev_x;
  const float* prev_scale2;
  const float* prev_shift2;
  const float* prev_ao;
  float* prev_tail_red;    /* [3][N][64], zeroed by the caller: becomes the previous application's tail_red */
  void* side_stream;       /* hipStream_t for the weight gradients, or NULL (main stream) */
  void* fork_event;        /* hipEvent_t recorded on the main stream before the fork */
} xr_resblock_desc;
int xr_resblock_desc_size(void);
int xr_resblock_fwd(const xr_resblock_desc* d, void* stream);
int xr_resblock_bwd(const xr_resblock_desc* d, void* stream);

/* One bottleneck_IR_SE unit with identity shortcut (model_irse.py:69-91; in_channel == depth = C >= 64, C % 64 == 0, stride 1: 18 of the
 * 24 units of IR-SE-50), bf16 NHWC tensors [N][H][W][C], training mode -- SURVEY 8b "xr_ir_block_{fwd,bwd}".
 *   fwd: BN1 coefficients from partial statistics (stats_in [2][fold_in][C] + pivot_in [fold_in][C] delivered by the previous unit's
 *        tail pass, or fold_in = 0: taken here over pg pseudo-groups into stats_own / pivot_own) incl. the running-statistics update ->
 *        bn1 = BN1(x) -> y1 = conv3x3(bn1) with p1 = prelu(y1, alpha) as second output -> y2 = conv3x3(p1) -> per-image sums of y2 ->
 *        BN2 coefficients (a2, b2c; running statistics) -> SE squeeze / excite (pooled, hidden, s, cA, cB) -> out = y2 * cA + cB + x,
 *        with the per-image statistics of out for the next unit's BN1 (stats_out [2][N][C] zeroed + pivot_out [N][C]; NULL: skipped).
 *   bwd: tail (S1 / S2 per image delivered in tail_red [2][N][C] by the next unit's BN1 backward, or one reduce pass into red_tail
 *        [3][N][C] zeroed) -> xr_bnse_bwd -> dy2 -> SE weight gradients -> conv2 input gradient with the PReLU backward in its epilogue
 *        (dal_s [da_spread][C] zeroed, folded into dalpha) -> conv1 input gradient with BN1's backward sums in its epilogue (red1
 *        [3][ep_spread][C] zeroed) -> BN1 coefficients -> dx = BN1-backward(db1t) + dout; with prev_y2 the same pass takes the previous
 *        tail's sums (prev_red2 [2][N][C] zeroed).  Weight gradients (wgrad_rows != 0: xr_conv_wgrad_rows, else xr_conv_wgrad; `split`
 *        slabs of [C][9 C] floats each in slabs1 / slabs2) and the SE ones run on side_stream after fork_event when given.
 * dg / db / dalpha / dse targets are accumulated (+=; dalpha by dalpha_accumulate); NULL targets are skipped. */
typedef struct xr_ir_block_desc {
  int N, H, W, C, Cr;
  float eps, momentum;
  int wgrad_rows, fold_in, pg, ep_spread, da_spread, split, dw_accumulate, dalpha_accumulate, reserved;
  const float* g1;         /* BN1 gamma / beta [C], running statistics (updated) */
  const float* b1;
  float* rmean1;
  float* rvar1;
  const void* w1_fwd;      /* [C][9 C] bf16 packs */
  const void* w2_fwd;
  const void* w1_dgrad;
  const void* w2_dgrad;
  const float* alpha;      /* PReLU slope [C] */
  const float* g2;         /* BN2 */
  const float* b2;
  float* rmean2;
  float* rvar2;
  const float* se1;        /* SE fc1 [Cr][C], fc2 [C][Cr] fp32 */
  const float* se2;
  const void* x;           /* unit input = shortcut */
  void* bn1;               /* BN1(x) (saved: conv1's weight gradient) */
  void* y1;
  void* p1;
  void* y2;
  void* out;
  const float* stats_in;
  const float* pivot_in;
  float* stats_own;        /* [2][pg][C] zeroed + [pg][C] */
  float* pivot_own;
  float* mean1;            /* [C] each */
  float* invstd1;
  float* scale1;
  float* shift1;
  float* sums2;            /* [2][N][C] zeroed */
  float* mean2;            /* [C] each */
  float* invstd2;
  float* a2;
  float* b2c;
  float* pooled;           /* [N][C] */
  float* hidden;           /* [N][Cr] */
  float* s;                /* [N][C] */
  float* cA;
  float* cB;
  float* stats_out;
  float* pivot_out;
  /* ---- backward only */
  const void* dout;
  const float* tail_red;
  float* red_tail;
  float* dpre2;            /* [N][C] */
  float* dhid;             /* [N][Cr] */
  float* dp;               /* [N][C] */
  float* coef2;            /* [3][N][C] */
  float* dg2;
  float* db2;
  void* dy2;               /* workspace tensors laid out like x */
  void* dy1;
  void* db1t;
  void* dx;                /* NULL: no input gradient */
  float* dal_s;
  float* dalpha;
  float* red1;
  float* coef1;            /* [3][C] */
  float* dg1;
  float* db1g;
  const void* prev_y2;
  float* prev_red2;
  float* slabs1;
  float* slabs2;
  float* dw1;              /* [C][C][3][3] parameter-layout targets (NULL: skipped) */
  float* dw2;
  float* dse1;             /* [Cr][C] / [C][Cr] (both or none) */
  float* dse2;
  void* side_stream;
  void* fork_event;
} xr_ir_block_desc;
int xr_ir_block_desc_size(void);
int xr_ir_block_fwd(const xr_ir_block_desc* d, void* stream);
int xr_ir_block_bwd(const xr_ir_block_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation / activation family.  Tensors are [G][rows][C] (G groups of `rows` pixels):
 * BatchNorm2d/1d: G = 1, rows = N*H*W (model_irse.py:56-60,141,144,148; model/resnet.py:24,27,159,167,173);
 * InstanceNorm2d: G = N, rows = H*W (model/FSRnet.py:81,87,112,115,319,347,385,434);
 * SE squeeze (AdaptiveAvgPool2d(1), model_irse.py:26): G = N, sums only. */

/* sums[0][g][c] += sum x, sums[1][g][c] += sum x^2 (fp32; caller zeroes `sums`, 2*G*C floats). */
int xr_group_stats(int dtype, const void* x, float* sums, int G, int rows, int C, void* stream);
/* The same sums taken over (x - pivot[g][c]) with pivot[g][c] = x[g][0][c] (written by the kernel, [G][C] fp32): the form every
 * normalisation statistic takes (aten::batch_norm / instance_norm compute a shifted / Welford variance; E[x^2] - mean^2 on raw fp32
 * sums cancels for |mean| >> std).  Consumed by xr_norm_finalize_pivot. */
int xr_group_stats_pivot(int dtype, const void* x, float* sums, float* pivot, int G, int rows, int C, void* stream);

/* From sums: mean/invstd (biased variance, eps), scale = gamma*invstd, shift = beta - mean*scale
 * (gamma/beta NULL -> 1/0).  If running_mean/var != NULL (BatchNorm training): running =
 * (1-momentum)*running + momentum*{mean, unbiased var}.  Outputs [G][C] fp32; gamma/beta are [C]. */
int xr_norm_finalize(const float* sums, const float* gamma, const float* beta, float* mean, float* invstd,
                     float* scale, float* shift, float* running_mean, float* running_var, int G, int rows, int C,
                     float eps, float momentum, int fold, void* stream);
/* fold > 1 (G must be 1): `sums` is [2][fold][C], partial sums of the one statistics group (per-image or pseudo-group
 * partials of a BatchNorm); they are added up inside the same launch.  fold <= 1: `sums` is [2][G][C]. */
/* pivot != NULL: sums were taken relative to pivot ([G][C], or [fold][C] for partials of one group -- each partial relative to
 * its own pivot and over rows / fold rows): mean = pivot + S1 / n, var = S2 / n - (S1 / n)^2.  pivot == NULL: xr_norm_finalize. */
int xr_norm_finalize_pivot(const float* sums, const float* pivot, const float* gamma, const float* beta, float* mean, float* invstd,
                           float* scale, float* shift, float* running_mean, float* running_var, int G, int rows, int C, float eps,
                           float momentum, int fold, void* stream);

/* Eval-mode BatchNorm: scale/shift [C] from running statistics. */
int xr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float* scale, float* shift, int C, float eps, void* stream);

/* y = act(x*scale[g][c] + shift[g][c] + res)      (scale/shift/res optional; alpha [C] for PReLU)
 * aten::batch_norm/instance_norm apply + prelu/relu + add (model/FSRnet.py:90-98,119-135;
 * model_irse.py:62-66; model/resnet.py:32-47). */
int xr_affine_act(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                  const float* alpha, int act, void* y, int G, int rows, int C, int coef_per_group, void* stream);
/* coef_per_group: scale/shift are [G][C] (1) or shared [C] (0) */

/* The same pass that also takes the statistics of its OUTPUT: stats[0][g][c] += sum y, stats[1][g][c] += sum y*y over the rows
 * of group g (y as stored, i.e. rounded to bf16; [2][G][C] fp32 zeroed by the caller).  When y is the block output that the next
 * block's training-mode BatchNorm normalises (model_irse.py:56-66 -> :76-91 of the next unit) that norm's xr_group_stats pass
 * disappears: xr_norm_finalize(fold = G) folds the per-group sums. */
int xr_affine_act_stats(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                        const float* alpha, int act, void* y, float* stats, int G, int rows, int C, int coef_per_group,
                        void* stream);
/* ... with the statistics taken relative to pivot[g][c] = y[g][0][c] (written; [G][C]): see xr_group_stats_pivot. */
int xr_affine_act_stats_pivot(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                              const float* alpha, int act, void* y, float* stats, float* pivot, int G, int rows, int C,
                              int coef_per_group, void* stream);

/* Backward, pass 1:  z = x*scale+shift+res ; dz = dy*act'(z)
 *   red[0][g][c] += sum dz ; red[1][g][c] += sum dz*x ; red[2][g][c] += sum dy*z*[z<=0]  (PReLU dalpha)
 * (3*G*C floats, caller zeroes). */
int xr_affine_act_bwd_reduce(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                             const float* alpha, int act, const void* dy, float* red, int G, int rows, int C,
                             int coef_per_group, void* stream);

/* Training-norm backward coefficients from pass-1 sums:
 *   dx = A*dz + B*x + C0 with A = gamma*invstd, B = -gamma*invstd^2*m2, C0 = -A*m1 - B*mean,
 *   m1 = mean(dz), m2 = mean(dz*xhat);  dgamma[c] += sum_g sum dz*xhat ; dbeta[c] += sum_g sum dz;
 *   dalpha[c] += sum_g red[2].  coef = [3][G][C] fp32.  dgamma/dbeta/dalpha optional (accumulated). */
int xr_norm_bwd_coeffs(const float* red, const float* gamma, const float* mean, const float* invstd, float* coef,
                       float* dgamma, float* dbeta, float* dalpha, int G, int rows, int C, int fold, void* stream);
/* fold > 1 (G must be 1): `red` is [3][fold][C] partial reductions of the one statistics group, added up in the same launch. */

/* Backward, pass 2: dx = A*dz + B*x + C0  (coef NULL -> dx = dz*scale, or dz when scale NULL);
 * dres (optional) = dz. */
int xr_affine_act_bwd_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                            const float* alpha, int act, const void* dy, const float* coef, void* dx, void* dres,
                            int G, int rows, int C, int coef_per_group, const void* dx_add, void* stream);
/* dx_add (optional, laid out like x): added to dx -- the gradient arriving through an identity branch of the same
 * input (block shortcut), so autograd's separate summation pass disappears. */

/* The same pass for a BatchNorm (G = 1 over [N][H][W][C]) whose input ALSO feeds a strided identity branch (MaxPool2d(1, s) shortcut of a
 * stage-opening unit, model_irse.py:53,62-66; or the sub-sampled input of its 1x1 shortcut convolution): dx_add_sub is that branch's
 * gradient in COMPACT form [N][H / s][W / s][C] and is added at the pixels with h % s == 0 and w % s == 0 -- the zero-filled full-size
 * tensor of xr_subsample_bwd (written once, read once) never exists. */
int xr_affine_act_bwd_apply_sub(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                const float* alpha, int act, const void* dy, const float* coef, void* dx, int N, int H, int W,
                                int C, const void* dx_add_sub, int sub_stride, const void* y2, float* red2, void* stream);
/* y2 / red2 != NULL: also the second reduction of xr_affine_act_bwd_apply_red (one group per image). */

/* Pass 2 of a BatchNorm that OPENS a residual unit, chained with pass 1 of the unit before it (model_irse.py:76-91): the G
 * groups are the images (rows = H*W), scale / shift / coef are per channel ([C], [3][C]: one statistics group spanning the
 * batch), and while dx -- which is the gradient dout entering the previous unit's tail out = SE(BN(y2)) + shortcut -- streams
 * out, the kernel accumulates
 *   red2[0][g][c] += sum dx,   red2[1][g][c] += sum dx * y2        (dx as stored; [2][G][C] fp32 zeroed by the caller)
 * = S1, S2 of xr_bnse_bwd, so that tail's xr_affine_act_bwd_reduce pass over (dout, y2) disappears.  y2 is laid out like x. */
int xr_affine_act_bwd_apply_red(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                const float* alpha, int act, const void* dy, const float* coef, void* dx, void* dres,
                                int G, int rows, int C, const void* dx_add, const void* y2, float* red2, void* stream);
/* out[v][c] (+)= sum_g red[v][g][c], v < NV.  BatchNorm statistics / backward sums are taken per image
 * (G = N: at most a few blocks contend on one atomic address) and folded over the batch here. */
int xr_reduce_groups(const float* red, float* out, int NV, int G, int C, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SE excitation (model_irse.py:40-46): s[n][c] = sigmoid(W2 relu(W1 (pooled_sum[n]/HW))), Cr = C/16.
 * bwd: given ds[n][c] -> dpre2[n][c], dhid[n][Cr], dpooled[n][c] (already divided by HW). */
int xr_se_excite_fwd(const float* pooled_sum, const float* w1, const float* w2, float* hidden, float* s,
                     int N, int C, int Cr, float inv_hw, void* stream);
int xr_se_excite_bwd(const float* w1, const float* w2, const float* hidden, const float* s, const float* ds,
                     float* dpre2, float* dhid, float* dpooled, int N, int C, int Cr, float inv_hw, void* stream);
/* Fused BatchNorm -> SE -> +shortcut tail of bottleneck_IR_SE (model_irse.py:76-91).  With r = a*y + b the squeeze is
 * a*mean(y) + b, so r is never materialised: out = y*coefA[n][c] + coefB[n][c] + shortcut (xr_affine_act).
 * fwd: sum_y [N][C] per-image sums of y (xr_group_stats); a/b [C] the BatchNorm scale/shift;
 *      outputs pooled_r_sum, hidden [N][Cr], s, coefA = a*s, coefB = b*s (all [N][C] unless noted).
 * bwd: S1 = sum_hw dout, S2 = sum_hw dout*y per image (xr_affine_act_bwd_reduce);  outputs dpre2 / dhid (for the SE
 *      weight gradients via xr_small_atb), dp, and coef [3][N][C] such that dy = coef0*dout + coef1*y + coef2
 *      (xr_affine_act_bwd_apply); dgamma/dbeta (optional) are accumulated.  train = 0: frozen statistics. */
int xr_bnse_fwd(const float* sum_y, const float* a, const float* b, const float* w1, const float* w2,
                float* pooled_r_sum, float* hidden, float* s, float* coefA, float* coefB, int N, int C, int Cr, int HW,
                void* stream);
int xr_bnse_bwd(const float* S1, const float* S2, const float* sum_y, const float* a, const float* b,
                const float* w1, const float* w2, const float* hidden, const float* s, const float* gamma,
                const float* mean, const float* invstd, float* dpre2, float* dhid, float* dp, float* coef,
                float* dgamma, float* dbeta, int N, int C, int Cr, int HW, int train, void* stream);
/* out[i][j] (+)= scale * sum_n A[n][i]*B[n][j]  -- dW1 = dhid^T pooled, dW2 = dpre2^T hidden */
int xr_small_atb(const float* A, const float* B, float* out, int N, int I, int J, float scale, int accumulate,
                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * Resampling glue. */
/* MaxPool2d(1, stride) == strided sub-sampling (model_irse.py:53,73); bwd scatters into zeros. */
int xr_subsample(int dtype, const void* x, void* y, int N, int H, int W, int C, int stride, void* stream);
int xr_subsample_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int stride, void* stream);
/* F.max_pool2d(x,2,2) (model/FSRnet.py:202); bwd routes to the (first) arg-max, recomputed from x. */
int xr_maxpool2(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
int xr_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
/* out = up1 + nearest_up2(low) (model/FSRnet.py:210-211); bwd: dlow = 2x2 sum of dy (dup1 = dy). */
int xr_upadd2(int dtype, const void* up1, const void* low, void* y, int N, int H, int W, int C, void* stream);
int xr_upadd2_bwd(int dtype, const void* dy, void* dlow, int N, int H, int W, int C, void* stream);
/* nn.ReflectionPad2d(p) (SUPER_RESOLUTION/model/FSRnet.py:255-292): y [N][H+2p][W+2p][C]; bwd folds the mirrored
 * border gradients back (each input pixel gathers its <= 4 pre-images, no atomics). */
int xr_reflect_pad(int dtype, const void* x, void* y, int N, int H, int W, int C, int pad, void* stream);
int xr_reflect_pad_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int pad, void* stream);
/* channel-slice copy: dst[m][dst_off + c] = src[m][src_off + c], c < C  (torch.cat, model/FSRnet.py:505,534) */
int xr_copy_channels(int dtype, const void* src, int lds_, int src_off, void* dst, int ldd, int dst_off, int64_t M,
                     int C, void* stream);
/* NCHW fp32 -> NHWC dtype with channel padding to Cp (module entry), and back (module exit). */
int xr_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, void* stream);
int xr_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, void* stream);
/* dtype conversion / elementwise add on flat buffers. */
int xr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
int xr_add(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream);
int xr_sub(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream);

/* Dropout(p) (model_irse.py:145): y = x*keep/(1-p); keep from `mask` (uint8, optional) or from the
 * counter-based generator hash(seed, element index) -- the same function serves backward.  `tick` (optional device
 * uint64): mixed into the seed inside the kernel, so a step captured in a HIP graph draws a fresh mask on every replay
 * (the graph increments the counter; a host-side seed would be frozen into the captured launch). */
int xr_dropout(int dtype, const void* x, const uint8_t* mask, void* y, int64_t n, float p, uint64_t seed,
               const void* tick, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses: value ACCUMULATED with atomics into loss[0] (fp32, caller zeroes; NULL = gradient-only launch);
 * gradient wrt the prediction written to dpred (NULL = value-only launch), multiplied by `gscale` and, when
 * gscale_dev != NULL, by the upstream gradient scalar read from that device pointer (no host sync).
 * xr_loss_mse: scale*mean((a-b)^2): MSELossFunc (scale 97, loss/loss.py:14), nn.MSELoss (scale 1).
 *   da = gscale*2*scale/n*(a-b), db = -da (either may be NULL). */
int xr_loss_mse(int dtype, const void* a, const void* b, float scale, float gscale, float* loss, void* da, void* db,
                int64_t n, int64_t n_valid, const float* gscale_dev, void* stream);
/* n = elements walked (incl. zero channel padding), n_valid = the mean's divisor */
/* MSELoss_Landmark (loss/loss.py:28-31): scale*mean((sum_c pred[n][c][p] - target[n][p])^2);
 * pred/dpred NCHW fp32 [N][C][HW] (the 97-channel head leaves the library as plain NCHW), target fp32 [N][HW]. */
int xr_loss_landmark(const float* pred, const float* target, float scale, float gscale, float* loss, float* dpred,
                     int N, int C, int HW, const float* gscale_dev, void* stream);
/* CrossEntropyLoss2d (loss/loss.py:61-62): mean over pixels of -log_softmax(pred[n,:,p])[target[n,p]], NCHW fp32. */
int xr_loss_ce_nchw(const float* pred, const int64_t* target, float gscale, float* loss, float* dpred, int N, int C,
                    int HW, const float* gscale_dev, void* stream);
/* nn.CrossEntropyLoss on rows (main.py:132, train_teacher_model.py:190): pred [M][ld], C valid columns. */
int xr_loss_softmax_ce(int dtype, const void* pred, const int64_t* target, float gscale, float* loss, void* dpred,
                       int64_t M, int C, int ld, const float* gscale_dev, void* stream);
/* Build-defined ArcFace margin (absent from the reference): in-place on cos logits [M][C]:
 * logits = s*(phi(cos) at target else cos); dlogit scaling handled by xr_arcface_bwd. */
int xr_arcface_margin(float* cos_logits, const int64_t* target, float* dphi_dcos, int64_t M, int C, float s,
                      float m, void* stream);
/* Build-defined MMD (the reference imports an undefined `MMD`, Face_Hallucination_sub_Net.py:25): biased
 * multi-bandwidth Gaussian-kernel MMD^2 between z[0:N] and z[N:2N] (fp32 [2N][D]).  fwd accumulates the value into
 * loss[0] and stores w [2N][2N] = dL/d|z_i - z_j|^2; bwd: dz = gscale * 2 * sum_j (w_ij + w_ji)(z_i - z_j). */
int xr_mmd_fwd(const float* z, float* w, float* loss, int N, int D, const float* sigmas, int nsig, void* stream);
int xr_mmd_bwd(const float* z, const float* w, float* dz, int N, int D, const float* gscale_dev, void* stream);
/* row-wise L2 normalisation y = x/||x|| and its backward (l2_norm, model_irse.py:16-20). */
int xr_l2norm_rows(const float* x, float* y, float* inv_norm, int64_t M, int C, void* stream);
int xr_l2norm_rows_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int64_t M, int C,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-tensor optimizers over flat fp32 buffers (torch.optim semantics):
 * SGD(momentum, weight_decay) DISTILLATION/train_HRN.py:75-84; RMSprop(alpha, eps, wd)
 * Face_Hallucination_sub_Net.py:120-124, distill_main.py:222-225; Adam(betas, eps, wd)
 * SUPER_RESOLUTION/train_FHN.py:115-121.  `wd_mask` (optional, uint8 per element): bit 0 = apply weight decay (NULL: decay
 * everywhere), bit 1 = skip the element entirely -- a parameter that received no gradient this step is left untouched,
 * exactly as a stock torch.optim optimizer skips parameters whose .grad is None (bn_end, residual_next,
 * ... must not start moving under weight decay). */
int xr_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float wd,
                const uint8_t* wd_mask, int first_step, void* stream);
int xr_rmsprop_step(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float wd,
                    const uint8_t* wd_mask, void* stream);
int xr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, int step, const void* tick, int64_t tick_ref, const uint8_t* wd_mask, void* stream);
/* `tick` (optional device uint64) / `tick_ref`: inside a captured HIP graph the effective step is
 * step + (*tick - tick_ref), so the bias corrections advance with the replays (see xr_dropout). */

/* ---------------------------------------------------------------------------------------------
 * Verification (utils/utils.py:14-87, distill_main.py:121-136).
 * dist[i] = sum_d (e1[i][d]-e2[i][d])^2 (fp32). */
int xr_pairdist_l2(const float* e1, const float* e2, float* dist, int64_t P, int D, void* stream);
/* hist[fold][label][j] += 1 with j = #{t : thresholds[t] <= dist[i]} in [0,T] (predict_same at t iff j <= t);
 * thresholds ascending fp32 [T]; fold_id int32 [P] (values < F); hist int64 [F][2][T+1], caller zeroes. */
int xr_roc_hist(const float* dist, const uint8_t* issame, const int32_t* fold_id, const float* thresholds,
                unsigned long long* hist, int64_t P, int T, int F, void* stream);
/* K-fold sweep over that histogram on the device (utils/utils.py:51-83; one workgroup): hist is overwritten with its prefix
 * sums over j; per fold f the threshold index maximising the accuracy of the OTHER folds (first maximum, numpy.argmax) ->
 * best_idx[f], the fold's own accuracy at it -> acc[f] (fp64); mean over folds of tpr / fpr at every threshold -> mean_tpr[T],
 * mean_fpr[T] (fp64; bit-identical to the numpy evaluation of the same integer counts).  2 <= F <= 256 (the reference default is 50). */
int xr_roc_sweep(unsigned long long* hist, int T, int F, double* mean_tpr, double* mean_fpr, double* acc, int* best_idx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loader-side tensor synthesis on the device (SUPER_RESOLUTION/FHN_loader.py:65-66,119-137,
 * helen_loader.py:124-143).
 * xr_lr_synth: hr uint8 [N][H][W][3] (the PIL crop) -> Image.resize((low[n], low[n])).resize((W, H), BICUBIC), Pillow's 8-bit
 * resampler restated (bit-identical to PIL), as uint8 [N][H][W][3] (lr_u8, optional) and / or after ToTensor + Normalize(0.5, 0.5)
 * as float32 [N][3][H][W] (lr_norm, optional).  low int32 [N] on the device, values in [max(H, W) / 16, max_low] (clamped);
 * max_low <= min(H, W); one workgroup per image, everything in LDS (H * W * 3 + intermediates <= 160 KiB). */
int xr_lr_synth(const uint8_t* hr, const int32_t* low, int max_low, uint8_t* lr_u8, float* lr_norm, int N, int H, int W, void* stream);
/* xr_heatmap: hm[n][y][x] = sum_i exp(-((x - l[n][i][0])^2 + (y - l[n][i][1])^2) / (2 sigma^2)), landmarks fp64 [N][L][2] = (x, y);
 * bumps in fp64, float32 running sum in landmark order (generate_hm / gaussian_k of the reference loaders). */
int xr_heatmap(const double* landmarks, float* hm, int N, int L, int H, int W, double sigma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XRFACE_H */
