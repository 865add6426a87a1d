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
