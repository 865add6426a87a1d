// xr_blocks.cpp -- block-level entry points of the C ABI (xr_resblock_{fwd,bwd}): ONE call enqueues every launch of an
// FSRNet residual block application (model/FSRnet.py:75-98: out = PReLU_out(IN2(conv2(PReLU(IN1(conv1(x))))) + x)) on the direct
// 64-channel kernels -- forward 5 launches, backward 8-12 including both weight gradients (optionally forked onto a side stream).
// Host-only translation unit: it sequences the kernel-level entry points of this library, exactly as xrface.ops did from Python
// (one ctypes call and ~3 tensor allocations per launch: the host enqueued 18 ms of a 37 ms FHN step).
#include "xr_common.h"

#define XR_TRY(call)                 \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ < 0) return rc_;         \
  } while (0)

extern "C" int xr_resblock_desc_size(void) { return (int)sizeof(xr_resblock_desc); }

static int check_desc(const xr_resblock_desc* d, const char* who) {
  XR_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  XR_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0, "%s: non-positive dimension", who);
  XR_CHECK_ARG((long long)d->N * d->H * d->W * 128 < (1ll << 31), "%s: tensor larger than 2 GiB", who);
  XR_CHECK_ARG(d->x && d->c1 && d->c2 && d->g1 && d->b1 && d->a1 && d->g2 && d->b2 && d->ao, "%s: null tensor / parameter", who);
  XR_CHECK_ARG(d->mean1 && d->invstd1 && d->scale1 && d->shift1 && d->mean2 && d->invstd2 && d->scale2 && d->shift2,
               "%s: null statistics buffer", who);
  return XR_OK;
}

extern "C" int xr_resblock_fwd(const xr_resblock_desc* d, void* stream) {
  XR_TRY(check_desc(d, "xr_resblock_fwd"));
  XR_CHECK_ARG(d->w1_fwd && d->w2_fwd && d->out && d->ws_fwd, "xr_resblock_fwd: null pack / output / workspace");
  const int N = d->N, H = d->H, W = d->W, HW = H * W;
  float* st1 = d->ws_fwd;                    // [2][N][64] sum, sum of squares of c1 (zeroed by the caller)
  float* st2 = d->ws_fwd + (size_t)2 * N * 64;
  // c1 = conv1(x), per-image statistics of c1 in the epilogue
  XR_TRY(xr_conv64_direct(d->x, d->w1_fwd, nullptr, d->c1, N, H, W, 0, nullptr, nullptr, nullptr, st1, nullptr, stream));
  XR_TRY(xr_norm_finalize(st1, d->g1, d->b1, d->mean1, d->invstd1, d->scale1, d->shift1, nullptr, nullptr, N, HW, 64, d->eps, 0.f, 1,
                          stream));
  // c2 = conv2(prelu(IN1(c1))): InstanceNorm + PReLU applied on load, statistics of c2 in the epilogue
  XR_TRY(xr_conv64_direct(d->c1, d->w2_fwd, nullptr, d->c2, N, H, W, 0, d->scale1, d->shift1, d->a1, st2, nullptr, stream));
  XR_TRY(xr_norm_finalize(st2, d->g2, d->b2, d->mean2, d->invstd2, d->scale2, d->shift2, nullptr, nullptr, N, HW, 64, d->eps, 0.f, 1,
                          stream));
  // out = prelu_out(IN2(c2) + x)
  XR_TRY(xr_affine_act(XR_BF16, d->c2, d->scale2, d->shift2, d->x, d->ao, XR_ACT_PRELU, d->out, N, HW, 64, 1, stream));
  return XR_OK;
}

extern "C" int xr_resblock_bwd(const xr_resblock_desc* d, void* stream) {
  XR_TRY(check_desc(d, "xr_resblock_bwd"));
  XR_CHECK_ARG(d->dout && d->dc2 && d->dy1 && d->dc1 && d->ws_bwd && d->w2_dgrad, "xr_resblock_bwd: null gradient / workspace / pack");
  XR_CHECK_ARG(d->W % 8 == 0 && d->W <= 112, "xr_resblock_bwd: the direct weight-gradient kernel needs W %% 8 == 0, W <= 112");
  XR_CHECK_ARG(d->tail_red || d->dres, "xr_resblock_bwd: the tail's pre-activation gradient needs a buffer (dres) unless it is dout");
  XR_CHECK_ARG(!d->prev_c2 || (d->prev_x && d->prev_scale2 && d->prev_shift2 && d->prev_ao && d->prev_tail_red && d->dx),
               "xr_resblock_bwd: chaining into the previous application needs its c2 / x / scale2 / shift2 / alpha, a sums buffer and dx");
  XR_CHECK_ARG((d->dx == nullptr && d->prev_c2 == nullptr) || d->w1_dgrad, "xr_resblock_bwd: input gradient needs conv1's dgrad pack");
  XR_CHECK_ARG((d->dw1 == nullptr && d->dw2 == nullptr) || d->slabs, "xr_resblock_bwd: weight gradients need the slab workspace");
  const int N = d->N, H = d->H, W = d->W, HW = H * W;
  const size_t NC = (size_t)N * 64;
  float* red = d->ws_bwd;              // [3][N][64] tail sums           (zeroed by the caller)
  float* red1 = d->ws_bwd + 3 * NC;    // [3][N][64] IN1 + PReLU sums    (zeroed by the caller)
  float* coef = d->ws_bwd + 6 * NC;    // [3][N][64]
  float* coef1 = d->ws_bwd + 9 * NC;   // [3][N][64]
  const void* dz;
  // ---- tail: out = prelu_out(IN2(c2) + x)
  if (d->tail_red != nullptr) {
    // dout already IS dz = dout * prelu'(tail) and its three sums came out of the epilogue of the next application's conv1
    // input gradient (xr_conv64_direct_tailred): coefficients + a two-input apply pass
    XR_TRY(xr_norm_bwd_coeffs(d->tail_red, d->g2, d->mean2, d->invstd2, coef, d->dg2, d->db2, d->dao, N, HW, 64, 1, stream));
    XR_TRY(xr_affine_act_bwd_apply(XR_BF16, d->c2, nullptr, nullptr, nullptr, nullptr, XR_ACT_NONE, d->dout, coef, d->dc2, nullptr, N, HW, 64,
                                   1, nullptr, stream));
    dz = d->dout;
  } else {
    XR_TRY(xr_affine_act_bwd_reduce(XR_BF16, d->c2, d->scale2, d->shift2, d->x, d->ao, XR_ACT_PRELU, d->dout, red, N, HW, 64, 1, stream));
    XR_TRY(xr_norm_bwd_coeffs(red, d->g2, d->mean2, d->invstd2, coef, d->dg2, d->db2, d->dao, N, HW, 64, 1, stream));
    XR_TRY(xr_affine_act_bwd_apply(XR_BF16, d->c2, d->scale2, d->shift2, d->x, d->ao, XR_ACT_PRELU, d->dout, coef, d->dc2, d->dres, N, HW,
                                   64, 1, nullptr, stream));
    dz = d->dres;
  }
  // ---- conv2's input gradient with the sums of IN1 + PReLU's backward in its epilogue, then IN1's apply pass
  XR_TRY(xr_conv64_direct_bwdred(d->dc2, d->w2_dgrad, d->dy1, N, H, W, 1, d->c1, d->scale1, d->shift1, d->a1, red1, stream));
  XR_TRY(xr_norm_bwd_coeffs(red1, d->g1, d->mean1, d->invstd1, coef1, d->dg1, d->db1, d->da1, N, HW, 64, 1, stream));
  XR_TRY(xr_affine_act_bwd_apply(XR_BF16, d->c1, d->scale1, d->shift1, nullptr, d->a1, XR_ACT_PRELU, d->dy1, coef1, d->dc1, nullptr, N, HW, 64,
                                 1, nullptr, stream));
  // ---- conv1's input gradient + the residual-branch gradient dz
  if (d->prev_c2 != nullptr) {
    // chained: store the PREVIOUS application's tail pre-activation gradient and its three sums instead of dout
    XR_TRY(xr_conv64_direct_tailred(d->dc1, d->w1_dgrad, d->dx, N, H, W, 1, dz, d->prev_c2, d->prev_x, d->prev_scale2, d->prev_shift2,
                                    d->prev_ao, d->prev_tail_red, stream));
  } else if (d->dx != nullptr) {
    XR_TRY(xr_conv64_direct(d->dc1, d->w1_dgrad, nullptr, d->dx, N, H, W, 1, nullptr, nullptr, nullptr, nullptr, dz, stream));
  }
  // ---- weight gradients: nothing inside the backward pass consumes them, so they may run on a side stream
  if (d->dw1 != nullptr || d->dw2 != nullptr) {
    void* ws = stream;
    if (d->side_stream != nullptr) {
      XR_CHECK_ARG(d->fork_event != nullptr, "xr_resblock_bwd: side_stream needs fork_event");
      if (hipEventRecord((hipEvent_t)d->fork_event, (hipStream_t)stream) != hipSuccess ||
          hipStreamWaitEvent((hipStream_t)d->side_stream, (hipEvent_t)d->fork_event, 0) != hipSuccess) {
        xr_set_error("xr_resblock_bwd: forking onto the side stream failed");
        return XR_E_LAUNCH;
      }
      ws = d->side_stream;
    }
    int split = N * H < 256 ? N * H : 256;
    float* slabs2 = d->slabs;
    float* slabs1 = d->slabs + (size_t)split * 64 * 576;
    if (d->dw2 != nullptr) {   // conv2's input y1 = prelu(IN1(c1)) is rebuilt on load (the forward never wrote it either)
      const int ns = xr_conv64_wgrad(d->c1, d->dc2, slabs2, N, H, W, split, d->scale1, d->shift1, d->a1, ws);
      if (ns < 0) return ns;
      XR_TRY(xr_unpack_wgrad(slabs2, d->dw2, 64, 1, 9, 64, 64, 576, 576, 0, 1, 9, d->dw_accumulate, ns, ws));
    }
    if (d->dw1 != nullptr) {
      const int ns = xr_conv64_wgrad(d->x, d->dc1, slabs1, N, H, W, split, nullptr, nullptr, nullptr, ws);
      if (ns < 0) return ns;
      XR_TRY(xr_unpack_wgrad(slabs1, d->dw1, 64, 1, 9, 64, 64, 576, 576, 0, 1, 9, d->dw_accumulate, ns, ws));
    }
  }
  return XR_OK;
}

// ================================================================================================================================
// xr_ir_block_{fwd,bwd}: one bottleneck_IR_SE unit with identity shortcut (model_irse.py:69-91; in_channel == depth = C, stride 1:
// 18 of the 24 units of IR-SE-50), bf16, training mode:
//   b1 = BN1(x) -> y1 = conv3x3(b1), p1 = prelu(y1) (second output of the epilogue) -> y2 = conv3x3(p1)
//   -> out = SE(BN2(y2)) + x with BN2's output never written (out = y2 * cA[n][c] + cB[n][c] + x)
// chained with its neighbours exactly as the op-level path of xrface.ops does: BN1's batch statistics arrive as per-image partial
// sums taken by the PREVIOUS unit's tail pass (stats_in / pivot_in), this unit's tail takes those of `out` for the next one
// (stats_out / pivot_out); in the backward pass the tail's two per-image sums arrive from the NEXT unit's BN1 backward (tail_red)
// and this unit's BN1 backward takes those of the previous tail while it writes dx (prev_y2 / prev_red2).
extern "C" int xr_ir_block_desc_size(void) { return (int)sizeof(xr_ir_block_desc); }

static int check_ir(const xr_ir_block_desc* d, const char* who) {
  XR_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  XR_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C >= 64 && d->C % 64 == 0 && d->Cr > 0, "%s: bad geometry", who);
  XR_CHECK_ARG(d->x && d->bn1 && d->y1 && d->p1 && d->y2, "%s: null activation tensor", who);
  XR_CHECK_ARG(d->g1 && d->b1 && d->alpha && d->g2 && d->b2 && d->se1 && d->se2, "%s: null parameter", who);
  XR_CHECK_ARG(d->mean1 && d->invstd1 && d->scale1 && d->shift1 && d->sums2 && d->mean2 && d->invstd2 && d->a2 && d->b2c && d->pooled &&
                   d->hidden && d->s && d->cA && d->cB,
               "%s: null statistics / coefficient buffer", who);
  return XR_OK;
}

extern "C" int xr_ir_block_fwd(const xr_ir_block_desc* d, void* stream) {
  XR_TRY(check_ir(d, "xr_ir_block_fwd"));
  XR_CHECK_ARG(d->w1_fwd && d->w2_fwd && d->out, "xr_ir_block_fwd: null pack / output");
  XR_CHECK_ARG((d->fold_in > 0 && d->stats_in) || (d->fold_in == 0 && d->pg > 0 && d->stats_own && d->pivot_own && d->N % d->pg == 0),
               "xr_ir_block_fwd: BN1 needs delivered partial statistics or buffers for its own");
  XR_CHECK_ARG(d->stats_out == nullptr || d->pivot_out != nullptr, "xr_ir_block_fwd: stats_out needs pivot_out");
  const int N = d->N, H = d->H, W = d->W, C = d->C, HW = H * W, Kg = 9 * C;
  // ---- BN1: statistics (delivered by the previous tail, or taken here over pg pseudo-groups), coefficients, apply
  if (d->fold_in > 0) {
    XR_TRY(xr_norm_finalize_pivot(d->stats_in, d->pivot_in, d->g1, d->b1, d->mean1, d->invstd1, d->scale1, d->shift1, d->rmean1, d->rvar1, 1,
                                  N * HW, C, d->eps, d->momentum, d->fold_in, stream));
  } else {
    XR_TRY(xr_group_stats_pivot(XR_BF16, d->x, d->stats_own, d->pivot_own, d->pg, (N / d->pg) * HW, C, stream));
    XR_TRY(xr_norm_finalize_pivot(d->stats_own, d->pivot_own, d->g1, d->b1, d->mean1, d->invstd1, d->scale1, d->shift1, d->rmean1, d->rvar1,
                                  1, N * HW, C, d->eps, d->momentum, d->pg, stream));
  }
  XR_TRY(xr_affine_act(XR_BF16, d->x, d->scale1, d->shift1, nullptr, nullptr, XR_ACT_NONE, d->bn1, 1, N * HW, C, 1, stream));
  // ---- conv1 with the PReLU second output, conv2
  XR_TRY(xr_conv_igemm(XR_BF16, d->bn1, d->w1_fwd, nullptr, d->y1, N, H, W, C, H, W, C, 3, 3, 1, 1, 0, Kg, C, nullptr, 0, nullptr, d->alpha,
                       nullptr, 1, d->p1, nullptr, nullptr, stream));
  XR_TRY(xr_conv_igemm(XR_BF16, d->p1, d->w2_fwd, nullptr, d->y2, N, H, W, C, H, W, C, 3, 3, 1, 1, 0, Kg, C, nullptr, 0, nullptr, nullptr,
                       nullptr, 1, nullptr, nullptr, nullptr, stream));
  // ---- tail: per-image sums of y2 serve BN2's batch statistics and the SE squeeze; one elementwise pass writes out
  XR_TRY(xr_group_stats(XR_BF16, d->y2, d->sums2, N, HW, C, stream));
  XR_TRY(xr_norm_finalize(d->sums2, d->g2, d->b2, d->mean2, d->invstd2, d->a2, d->b2c, d->rmean2, d->rvar2, 1, N * HW, C, d->eps,
                          d->momentum, N, stream));
  XR_TRY(xr_bnse_fwd(d->sums2, d->a2, d->b2c, d->se1, d->se2, d->pooled, d->hidden, d->s, d->cA, d->cB, N, C, d->Cr, HW, stream));
  if (d->stats_out != nullptr) {
    XR_TRY(xr_affine_act_stats_pivot(XR_BF16, d->y2, d->cA, d->cB, d->x, nullptr, XR_ACT_NONE, d->out, d->stats_out, d->pivot_out, N, HW, C,
                                     1, stream));
  } else {
    XR_TRY(xr_affine_act(XR_BF16, d->y2, d->cA, d->cB, d->x, nullptr, XR_ACT_NONE, d->out, N, HW, C, 1, stream));
  }
  return XR_OK;
}

static int fork_side(const xr_ir_block_desc* d, void* stream) {
  if (hipEventRecord((hipEvent_t)d->fork_event, (hipStream_t)stream) != hipSuccess ||
      hipStreamWaitEvent((hipStream_t)d->side_stream, (hipEvent_t)d->fork_event, 0) != hipSuccess) {
    xr_set_error("xr_ir_block_bwd: forking onto the side stream failed");
    return XR_E_LAUNCH;
  }
  return XR_OK;
}

static int ir_wgrad(const xr_ir_block_desc* d, const void* in, const void* dy, float* slabs, float* dw, void* st) {
  const int N = d->N, H = d->H, W = d->W, C = d->C, Kg = 9 * C;
  int ns;
  if (d->wgrad_rows) ns = xr_conv_wgrad_rows(in, dy, slabs, N, H, W, C, C, 1, d->split, st);
  else ns = xr_conv_wgrad(XR_BF16, in, dy, slabs, N, H, W, C, H, W, C, 3, 3, 1, 1, 0, C, Kg, d->split, st);
  if (ns < 0) return ns;
  return xr_unpack_wgrad(slabs, dw, C, 1, 9, C, C, Kg, (int64_t)C * 9, 0, 1, 9, d->dw_accumulate, ns, st);
}

extern "C" int xr_ir_block_bwd(const xr_ir_block_desc* d, void* stream) {
  XR_TRY(check_ir(d, "xr_ir_block_bwd"));
  XR_CHECK_ARG(d->dout && d->dy2 && d->dy1 && d->db1t && d->w1_dgrad && d->w2_dgrad, "xr_ir_block_bwd: null gradient tensor / pack");
  XR_CHECK_ARG(d->tail_red || d->red_tail, "xr_ir_block_bwd: the tail's sums need a buffer unless they are delivered");
  XR_CHECK_ARG(d->dpre2 && d->dhid && d->dp && d->coef2 && d->dal_s && d->red1 && d->coef1 && d->ep_spread > 0 && d->da_spread > 0,
               "xr_ir_block_bwd: null workspace");
  XR_CHECK_ARG(d->prev_y2 == nullptr || (d->prev_red2 && d->dx), "xr_ir_block_bwd: chaining into the previous tail needs prev_red2 and dx");
  XR_CHECK_ARG((d->dw1 == nullptr && d->dw2 == nullptr) || (d->slabs1 && d->slabs2 && d->split > 0), "xr_ir_block_bwd: weight gradients need slabs");
  const bool side = d->side_stream != nullptr;
  XR_CHECK_ARG(!side || d->fork_event, "xr_ir_block_bwd: side_stream needs fork_event");
  const int N = d->N, H = d->H, W = d->W, C = d->C, Cr = d->Cr, HW = H * W, Kg = 9 * C;
  void* ss = side ? d->side_stream : stream;
  // ---- tail: out = y2 * cA + cB + x.  S1 = sum dout, S2 = sum dout * y2 per image: delivered, or one reduce pass
  const float *S1, *S2;
  if (d->tail_red != nullptr) {
    S1 = d->tail_red; S2 = d->tail_red + (size_t)N * C;
  } else {
    XR_TRY(xr_affine_act_bwd_reduce(XR_BF16, d->y2, nullptr, nullptr, nullptr, nullptr, XR_ACT_NONE, d->dout, d->red_tail, N, HW, C, 1, stream));
    S1 = d->red_tail; S2 = d->red_tail + (size_t)N * C;
  }
  XR_TRY(xr_bnse_bwd(S1, S2, d->sums2, d->a2, d->b2c, d->se1, d->se2, d->hidden, d->s, d->g2, d->mean2, d->invstd2, d->dpre2, d->dhid, d->dp,
                     d->coef2, d->dg2, d->db2, N, C, Cr, HW, 1, stream));
  XR_TRY(xr_affine_act_bwd_apply(XR_BF16, d->y2, nullptr, nullptr, nullptr, nullptr, XR_ACT_NONE, d->dout, d->coef2, d->dy2, nullptr, N, HW, C, 1,
                                 nullptr, stream));
  if (d->dse1 != nullptr && d->dse2 != nullptr) {   // SE weight gradients: off the critical path
    if (side) XR_TRY(fork_side(d, stream));
    XR_TRY(xr_small_atb(d->dhid, d->pooled, d->dse1, N, Cr, C, 1.0f / (float)HW, 1, ss));
    XR_TRY(xr_small_atb(d->dpre2, d->hidden, d->dse2, N, C, Cr, 1.0f, 1, ss));
  }
  // ---- conv2: input gradient with the PReLU backward in its epilogue (dy1 = acc * prelu'(y1), dalpha partials), weight gradient
  XR_TRY(xr_conv_igemm(XR_BF16, d->dy2, d->w2_dgrad, nullptr, d->dy1, N, H, W, C, H, W, C, 3, 3, 1, 1, 1, Kg, C, nullptr, 0, d->y1, d->alpha,
                       d->dal_s, d->da_spread, nullptr, nullptr, nullptr, stream));
  if (d->dalpha != nullptr) XR_TRY(xr_reduce_groups(d->dal_s, d->dalpha, 1, d->da_spread, C, d->dalpha_accumulate, stream));
  if (d->dw2 != nullptr) {
    if (side) XR_TRY(fork_side(d, stream));
    XR_TRY(ir_wgrad(d, d->p1, d->dy2, d->slabs2, d->dw2, ss));
  }
  // ---- conv1: input gradient with BN1's backward sums (sum d, sum d * x per channel) in its epilogue, weight gradient
  XR_TRY(xr_conv_igemm(XR_BF16, d->dy1, d->w1_dgrad, nullptr, d->db1t, N, H, W, C, H, W, C, 3, 3, 1, 1, 1, Kg, C, nullptr, 0, d->x, nullptr, nullptr,
                       d->ep_spread, nullptr, d->red1, nullptr, stream));
  if (d->dw1 != nullptr) {
    if (side) XR_TRY(fork_side(d, stream));
    XR_TRY(ir_wgrad(d, d->bn1, d->dy1, d->slabs1, d->dw1, ss));
  }
  // ---- BN1 backward: coefficients from the folded partial sums, then dx = A * d + B * x + C0 + dout (the shortcut's gradient);
  // chained, the same pass takes the previous tail's per-image sums (sum dx, sum dx * y2_prev)
  XR_TRY(xr_norm_bwd_coeffs(d->red1, d->g1, d->mean1, d->invstd1, d->coef1, d->dg1, d->db1g, nullptr, 1, N * HW, C, d->ep_spread, stream));
  if (d->dx != nullptr) {
    if (d->prev_y2 != nullptr) {
      XR_TRY(xr_affine_act_bwd_apply_red(XR_BF16, d->x, d->scale1, d->shift1, nullptr, nullptr, XR_ACT_NONE, d->db1t, d->coef1, d->dx, nullptr, N,
                                         HW, C, d->dout, d->prev_y2, d->prev_red2, stream));
    } else {
      XR_TRY(xr_affine_act_bwd_apply(XR_BF16, d->x, d->scale1, d->shift1, nullptr, nullptr, XR_ACT_NONE, d->db1t, d->coef1, d->dx, nullptr, 1,
                                     N * HW, C, 1, d->dout, stream));
    }
  }
  return XR_OK;
}
