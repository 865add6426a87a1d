// xr_misc.hip -- resampling glue, layout conversion, dropout, losses, fused optimizers, verification.
// All HBM-bound: 16-B-per-lane coalesced NHWC chunk accesses, wave-shuffle reductions, one atomic per block.
//
// Replaces (reference call sites, /root/reference): MaxPool2d(1,stride) model_irse.py:53,73; F.max_pool2d
// model/FSRnet.py:202; F.interpolate model/FSRnet.py:210; torch.cat model/FSRnet.py:505,534; Dropout
// model_irse.py:145; losses loss/loss.py:7-62; torch.optim.{SGD,RMSprop,Adam} call sites
// DISTILLATION/train_HRN.py:75-84, Face_Hallucination_sub_Net.py:120-124, SUPER_RESOLUTION/train_FHN.py:115-121;
// pair distance + threshold sweep utils/utils.py:14-87.
#include "xr_common.h"

namespace {

constexpr int NT = 256;

static inline int grid_for(int64_t n, int per_block = NT, int cap = 8192) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// deterministic mode: the loss scalar is the sum of ONE block (the blocks otherwise meet in it by fp32 atomics)
static inline int det_grid(int g) { return XR_DET() ? 1 : g; }

#define GRID_STRIDE(i, n) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ resampling
// All kernels iterate over 8-channel chunks of the OUTPUT tensor.
template <typename T>
__global__ void subsample_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int s, int bwd) {
  // fwd: y[n,ho,wo,:] = x[n,ho*s,wo*s,:]   bwd (x := dy, y := dx): dx[n,h,w,:] = (h%s==0&&w%s==0) ? dy[n,h/s,w/s,:] : 0
  const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;  // MaxPool2d(1, s): floor((H-1)/s)+1
  const int cpr = C / 8;
  const int64_t total = bwd ? (int64_t)N * H * W * cpr : (int64_t)N * Ho * Wo * cpr;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    float v[8];
    if (!bwd) {
      const int wo = (int)(pix % Wo); pix /= Wo;
      const int ho = (int)(pix % Ho);
      const int n = (int)(pix / Ho);
      ld8(x + (((size_t)n * H + ho * s) * W + wo * s) * C + ch * 8, v);
    } else {
      const int w = (int)(pix % W); pix /= W;
      const int h = (int)(pix % H);
      const int n = (int)(pix / H);
      if (h % s == 0 && w % s == 0) {
        ld8(x + (((size_t)n * Ho + h / s) * Wo + w / s) * C + ch * 8, v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
    }
    st8(y + (size_t)i * 8, v);
  }
}

template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cpr = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * cpr;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const T* b = x + (((size_t)n * H + ho * 2) * W + wo * 2) * C + ch * 8;
    float a[8], v[8];
    ld8(b, a);
    ld8(b + C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v[e] > a[e] ? v[e] : a[e];
    ld8(b + (size_t)W * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v[e] > a[e] ? v[e] : a[e];
    ld8(b + (size_t)W * C + C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v[e] > a[e] ? v[e] : a[e];
    st8(y + (size_t)i * 8, a);
  }
}

template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                    int C) {
  // one thread per pooled window chunk; writes the 4 input positions (first arg-max in row-major window order gets dy)
  const int Ho = H / 2, Wo = W / 2, cpr = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * cpr;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const size_t o00 = (((size_t)n * H + ho * 2) * W + wo * 2) * C + ch * 8;
    const size_t offs[4] = {o00, o00 + C, o00 + (size_t)W * C, o00 + (size_t)W * C + C};
    float v[4][8], g[8], o[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld8(x + offs[k], v[k]);
    ld8(dy + (size_t)i * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int best = 0;
      float bv = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > bv) { bv = v[k][e]; best = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] = (k == best) ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) st8(dx + offs[k], o[k]);
  }
}

template <typename T>
__global__ void upadd2_kernel(const T* __restrict__ up1, const T* __restrict__ low, T* __restrict__ y, int N, int H, int W,
                              int C) {
  // y[n,h,w,:] = up1[n,h,w,:] + low[n,h/2,w/2,:]   (H, W = output size)
  const int cpr = C / 8, Hl = H / 2, Wl = W / 2;
  const int64_t total = (int64_t)N * H * W * cpr;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    float a[8], b[8];
    ld8(up1 + (size_t)i * 8, a);
    ld8(low + (((size_t)n * Hl + h / 2) * Wl + w / 2) * C + ch * 8, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    st8(y + (size_t)i * 8, a);
  }
}

template <typename T>
__global__ void upadd2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dlow, int N, int H, int W, int C) {
  // dlow[n,hl,wl,:] = sum of the 2x2 dy block (H, W = dy size)
  const int cpr = C / 8, Hl = H / 2, Wl = W / 2;
  const int64_t total = (int64_t)N * Hl * Wl * cpr;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int wl = (int)(pix % Wl); pix /= Wl;
    const int hl = (int)(pix % Hl);
    const int n = (int)(pix / Hl);
    const T* b = dy + (((size_t)n * H + hl * 2) * W + wl * 2) * C + ch * 8;
    float a[8], v[8];
    ld8(b, a);
    ld8(b + C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += v[e];
    ld8(b + (size_t)W * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += v[e];
    ld8(b + (size_t)W * C + C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += v[e];
    st8(dlow + (size_t)i * 8, a);
  }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {  // index i in [-p, n+p) mirrored into [0, n), p < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}
template <typename T>
__global__ void reflect_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int p) {
  const int Ho = H + 2 * p, Wo = W + 2 * p, cpr = C / 8;
  GRID_STRIDE(i, (int64_t)N * Ho * Wo * cpr) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float v[8];
    ld8(x + (((size_t)n * H + reflect_idx(ho - p, H)) * W + reflect_idx(wo - p, W)) * C + ch * 8, v);
    st8(y + (size_t)i * 8, v);
  }
}
template <typename T>
__global__ void reflect_pad_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int p) {
  // one thread per INPUT pixel chunk: its own image at (h+p, w+p) plus the mirrored pre-images in each dimension
  const int Ho = H + 2 * p, Wo = W + 2 * p, cpr = C / 8;
  GRID_STRIDE(i, (int64_t)N * H * W * cpr) {
    const int ch = (int)(i % cpr);
    int64_t pix = i / cpr;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + p;
    if (h >= 1 && h <= p) hs[nh++] = p - h;
    if (h <= H - 2 && h >= H - 1 - p) hs[nh++] = 2 * (H - 1) - h + p;
    ws[nw++] = w + p;
    if (w >= 1 && w <= p) ws[nw++] = p - w;
    if (w <= W - 2 && w >= W - 1 - p) ws[nw++] = 2 * (W - 1) - w + p;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int a_ = 0; a_ < nh; ++a_)
      for (int b_ = 0; b_ < nw; ++b_) {
        float v[8];
        ld8(dy + (((size_t)n * Ho + hs[a_]) * Wo + ws[b_]) * C + ch * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += v[e];
      }
    st8(dx + (size_t)i * 8, a);
  }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds_, int so, T* __restrict__ dst, int ldd, int doff,
                                     int64_t M, int C) {
  const int cpr = C / 8;
  GRID_STRIDE(i, M * cpr) {
    const int ch = (int)(i % cpr);
    const int64_t m = i / cpr;
    float v[8];
    ld8(src + (size_t)m * lds_ + so + ch * 8, v);
    st8(dst + (size_t)m * ldd + doff + ch * 8, v);
  }
}

// NCHW fp32 <-> NHWC(T, channel pitch Cp); LDS-tiled transpose over (C, HW) per image would be overkill for C<=512:
// one thread per (n, pixel, 8-chunk): reads are strided by HW (served by L2), writes are coalesced.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int HW, int Cp) {
  const int cpr = Cp / 8;
  GRID_STRIDE(i, (int64_t)N * HW * cpr) {
    const int ch = (int)(i % cpr);
    const int64_t r = i / cpr;
    const int pix = (int)(r % HW);
    const int n = (int)(r / HW);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ch * 8 + e;
      v[e] = c < C ? src[((size_t)n * C + c) * HW + pix] : 0.f;
    }
    st8(dst + (size_t)i * 8, v);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cp) {
  GRID_STRIDE(i, (int64_t)N * C * HW) {
    const int pix = (int)(i % HW);
    const int64_t r = i / HW;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    dst[i] = XrT<T>::ld(src + ((size_t)n * HW + pix) * Cp + c);
  }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  GRID_STRIDE(i, n) XrT<D>::st(d + i, XrT<S>::ld(s + i));
}
template <typename T>
__global__ void addsub_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n8, int64_t n,
                              float sign) {
  GRID_STRIDE(i, n8) {
    float u[8], v[8];
    ld8(a + i * 8, u);
    ld8(b + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += sign * v[e];
    st8(y + i * 8, u);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) XrT<T>::st(y + i, XrT<T>::ld(a + i) + sign * XrT<T>::ld(b + i));
  }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, T* __restrict__ y, int64_t n, float p,
                               uint64_t seed, const uint64_t* __restrict__ tick) {
  if (tick != nullptr) seed = mix64(seed + *tick * 0x9E3779B97F4A7C15ull);  // replayed HIP graph: the step counter lives on the device
  const float sc = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
  GRID_STRIDE(i, n) {
    bool keep;
    if (mask) keep = mask[i] != 0;
    else keep = (uint32_t)(mix64(seed ^ (uint64_t)i * 0xD1B54A32D192ED03ull) >> 32) >= thr;
    XrT<T>::st(y + i, keep ? XrT<T>::ld(x + i) * sc : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ losses
__device__ __forceinline__ void block_atomic_sum(float v, float* dst) {
  if (dst == nullptr) return;  // gradient-only launch (uniform across the block)
  __shared__ float part[NT / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < NT / 64; ++w) s += part[w];
    atomicAdd(dst, s);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void loss_mse_kernel(const T* __restrict__ a, const T* __restrict__ b, float lscale, float gcoef,
                                                      float* __restrict__ loss, T* __restrict__ da, T* __restrict__ db,
                                                      int64_t n, const float* __restrict__ gdev) {
  if (gdev) gcoef *= *gdev;
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float d = XrT<T>::ld(a + i) - XrT<T>::ld(b + i);
    s += d * d;
    if (da) XrT<T>::st(da + i, gcoef * d);
    if (db) XrT<T>::st(db + i, -gcoef * d);
  }
  block_atomic_sum(s * lscale, loss);
}

// pred NCHW fp32 [N][C][HW]; target [N][HW]: scale*mean((sum_c pred - target)^2)  (loss/loss.py:28-31)
__global__ __launch_bounds__(NT) void loss_landmark_nchw_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                float lscale, float gcoef, float* __restrict__ loss,
                                                                float* __restrict__ dpred, int64_t NP, int C, int HW,
                                                                const float* __restrict__ gdev) {
  if (gdev) gcoef *= *gdev;
  float tot = 0.f;
  GRID_STRIDE(i, NP) {
    const int64_t n = i / HW;
    const int pix = (int)(i - n * HW);
    const float* b = pred + (size_t)n * C * HW + pix;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += b[(size_t)c * HW];
    const float d = s - target[i];
    tot += d * d;
    if (dpred) {
      float* g = dpred + (size_t)n * C * HW + pix;
      for (int c = 0; c < C; ++c) g[(size_t)c * HW] = gcoef * d;
    }
  }
  block_atomic_sum(tot * lscale, loss);
}

// logits NCHW fp32 [N][C][HW], target int64 [N][HW]: mean NLL of log_softmax over C (loss/loss.py:61-62)
__global__ __launch_bounds__(NT) void loss_ce_nchw_kernel(const float* __restrict__ pred, const int64_t* __restrict__ target,
                                                          float gcoef, float lscale, float* __restrict__ loss,
                                                          float* __restrict__ dpred, int64_t NP, int C, int HW,
                                                          const float* __restrict__ gdev) {
  if (gdev) gcoef *= *gdev;
  float tot = 0.f;
  GRID_STRIDE(i, NP) {
    const int64_t n = i / HW;
    const int pix = (int)(i - n * HW);
    const float* b = pred + (size_t)n * C * HW + pix;
    float mx = -3.0e38f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, b[(size_t)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(b[(size_t)c * HW] - mx);
    const float lse = mx + __logf(se);
    const int tg = (int)target[i];
    tot += lse - b[(size_t)tg * HW];
    if (dpred) {
      float* g = dpred + (size_t)n * C * HW + pix;
      for (int c = 0; c < C; ++c) g[(size_t)c * HW] = gcoef * (__expf(b[(size_t)c * HW] - lse) - (c == tg ? 1.f : 0.f));
    }
  }
  block_atomic_sum(tot * lscale, loss);
}

template <typename T>
__global__ __launch_bounds__(NT) void loss_softmax_ce_kernel(const T* __restrict__ pred, const int64_t* __restrict__ target,
                                                             float gcoef, float lscale, float* __restrict__ loss,
                                                             T* __restrict__ dpred, int64_t M, int C, int ld,
                                                             const float* __restrict__ gdev) {
  if (gdev) gcoef *= *gdev;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)NT + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * NT) >> 6;
  float tot = 0.f;
  for (int64_t m = wave; m < M; m += nwaves) {
    float mx = -3.0e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, XrT<T>::ld(pred + m * ld + c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(XrT<T>::ld(pred + m * ld + c) - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const int tg = (int)target[m];
    if (lane == 0) tot += lse - XrT<T>::ld(pred + m * ld + tg);
    if (dpred)
      for (int c = lane; c < ld; c += 64) {
        float g = 0.f;
        if (c < C) g = gcoef * (__expf(XrT<T>::ld(pred + m * ld + c) - lse) - (c == tg ? 1.f : 0.f));
        XrT<T>::st(dpred + m * ld + c, g);
      }
  }
  block_atomic_sum(tot * lscale, loss);
}

__global__ void arcface_margin_kernel(float* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ dphi,
                                      int64_t M, int C, float s, float cm, float sm, float th, float mm) {
  // logits hold cos(theta); scale all by s, apply the additive angular margin at the target column
  GRID_STRIDE(i, M * (int64_t)C) {
    const int64_t m = i / C;
    const int c = (int)(i - m * C);
    float cs = fminf(fmaxf(logits[i], -1.f), 1.f);
    float out = cs;
    if (c == (int)target[m]) {
      const float sn = sqrtf(fmaxf(1.f - cs * cs, 0.f));
      float d;
      if (cs > th) {
        out = cs * cm - sn * sm;
        d = cm + (sn > 1e-6f ? cs / sn * sm : 0.f);
      } else {
        out = cs - mm;
        d = 1.f;
      }
      dphi[m] = d;
    }
    logits[i] = s * out;
  }
}

__global__ __launch_bounds__(NT) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv_norm, int64_t M, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)NT + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t m = wave; m < M; m += nwaves) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = x[m * C + c]; s += v * v; }
    s = wave_sum(s);
    const float inv = rsqrtf(s);
    if (lane == 0) inv_norm[m] = inv;
    for (int c = lane; c < C; c += 64) y[m * C + c] = x[m * C + c] * inv;
  }
}
__global__ __launch_bounds__(NT) void l2norm_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv_norm,
                                                             const float* __restrict__ dy, float* __restrict__ dx, int64_t M,
                                                             int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)NT + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t m = wave; m < M; m += nwaves) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += y[m * C + c] * dy[m * C + c];
    s = wave_sum(s);
    const float inv = inv_norm[m];
    for (int c = lane; c < C; c += 64) dx[m * C + c] = (dy[m * C + c] - y[m * C + c] * s) * inv;
  }
}

// ------------------------------------------------------------------------------------------------ MMD (build-defined)
// biased multi-bandwidth Gaussian-kernel MMD^2 between a = z[0:N] and b = z[N:2N] (D floats per row):
//   L = mean K(a,a) + mean K(b,b) - 2 mean K(a,b),  K(x,y) = sum_s exp(-|x-y|^2 / (2 sigma_s^2)).
// One wave per (i, j) pair: d2 by a wave reduction; w[i][j] = dL/d(d2_ij) is kept for the backward pass.
__global__ __launch_bounds__(NT) void mmd_fwd_kernel(const float* __restrict__ z, float* __restrict__ w, float* __restrict__ loss,
                                                     int N, int D, const float* __restrict__ sig, int ns) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (blockIdx.x * (int64_t)NT + threadIdx.x) >> 6;
  const int M = 2 * N;
  float contrib = 0.f;
  if (pair < (int64_t)M * M) {
    const int i = (int)(pair / M), j = (int)(pair - (int64_t)i * M);
    float d2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float t = z[(size_t)i * D + d] - z[(size_t)j * D + d];
      d2 += t * t;
    }
    d2 = wave_sum(d2);
    if (lane == 0) {
      const float sgn = ((i < N) == (j < N)) ? 1.f : -1.f;
      const float coef = sgn / ((float)N * (float)N);
      float k = 0.f, dk = 0.f;
      for (int s = 0; s < ns; ++s) {
        const float g = 1.f / (2.f * sig[s] * sig[s]);
        const float e = __expf(-d2 * g);
        k += e;
        dk -= g * e;
      }
      w[pair] = coef * dk;
      contrib = coef * k;
    }
  }
  block_atomic_sum(contrib, loss);
}
// dz[i] = gscale * 2 * sum_j (w[i][j] + w[j][i]) * (z[i] - z[j]); one block per row i
__global__ __launch_bounds__(NT) void mmd_bwd_kernel(const float* __restrict__ z, const float* __restrict__ w, float* __restrict__ dz,
                                                     int M, int D, const float* __restrict__ gdev) {
  const int i = blockIdx.x;
  const float gs = gdev ? *gdev : 1.f;
  for (int d = threadIdx.x; d < D; d += NT) {
    const float zi = z[(size_t)i * D + d];
    float a = 0.f;
    for (int j = 0; j < M; ++j) a += (w[(size_t)i * M + j] + w[(size_t)j * M + i]) * (zi - z[(size_t)j * D + d]);
    dz[(size_t)i * D + d] = 2.f * gs * a;
  }
}

// ------------------------------------------------------------------------------------------------ optimizers
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, int64_t n, float lr,
                           float momentum, float wd, const uint8_t* __restrict__ wd_mask, int first) {
  GRID_STRIDE(i, n) {
    const int mk = wd_mask ? wd_mask[i] : 1;
    if (mk & 2) continue;   // parameter received no gradient this step: skipped, like a stock optimizer skips .grad None
    float gi = g[i];
    if (wd != 0.f && (mk & 1)) gi += wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? gi : momentum * mom[i] + gi;
      mom[i] = b;
      gi = b;
    }
    p[i] -= lr * gi;
  }
}
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, int64_t n, float lr,
                               float alpha, float eps, float wd, const uint8_t* __restrict__ wd_mask) {
  GRID_STRIDE(i, n) {
    const int mk = wd_mask ? wd_mask[i] : 1;
    if (mk & 2) continue;
    float gi = g[i];
    if (wd != 0.f && (mk & 1)) gi += wd * p[i];
    const float v = alpha * sq[i] + (1.f - alpha) * gi * gi;
    sq[i] = v;
    p[i] -= lr * gi / (sqrtf(v) + eps);
  }
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, int step,
                            const uint64_t* __restrict__ tick, uint64_t tick_ref, const uint8_t* __restrict__ wd_mask) {
  if (tick != nullptr) {  // replayed HIP graph: the bias corrections follow the device-side step counter
    const float st = (float)(step + (int)(*tick - tick_ref));
    bc1 = 1.f - powf(b1, st);
    bc2 = 1.f - powf(b2, st);
  }
  GRID_STRIDE(i, n) {
    const int mk = wd_mask ? wd_mask[i] : 1;
    if (mk & 2) continue;
    float gi = g[i];
    if (wd != 0.f && (mk & 1)) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * mi / denom;
  }
}

// ------------------------------------------------------------------------------------------------ verification
__global__ __launch_bounds__(NT) void pairdist_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                      float* __restrict__ dist, int64_t P, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)NT + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = wave; i < P; i += nwaves) {
    const float* a = e1 + i * D;
    const float* b = e2 + i * D;
    float s = 0.f;
    if ((D & 3) == 0) {
      for (int d = lane * 4; d < D; d += 256) {
        const float4 u = *reinterpret_cast<const float4*>(a + d);
        const float4 v = *reinterpret_cast<const float4*>(b + d);
        const float x0 = u.x - v.x, x1 = u.y - v.y, x2 = u.z - v.z, x3 = u.w - v.w;
        s += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      }
    } else {
      for (int d = lane; d < D; d += 64) { const float x = a[d] - b[d]; s += x * x; }
    }
    s = wave_sum(s);
    if (lane == 0) dist[i] = s;
  }
}

__global__ __launch_bounds__(NT) void roc_hist_kernel(const float* __restrict__ dist, const uint8_t* __restrict__ issame,
                                                      const int32_t* __restrict__ fold_id, const float* __restrict__ thr,
                                                      unsigned long long* __restrict__ hist, int64_t P, int T, int F) {
  extern __shared__ float sthr[];
  for (int i = threadIdx.x; i < T; i += NT) sthr[i] = thr[i];
  __syncthreads();
  GRID_STRIDE(i, P) {
    const float d = dist[i];
    // j = number of thresholds <= d  (upper bound); predict_same(t) = d < thr[t]  <=>  t >= j
    int lo = 0, hi = T;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sthr[mid] <= d) lo = mid + 1; else hi = mid;
    }
    const int f = fold_id ? fold_id[i] : 0;
    const int lab = issame[i] ? 1 : 0;
    atomicAdd(hist + ((size_t)f * 2 + lab) * (T + 1) + lo, 1ull);
  }
}


// K-fold ROC sweep over the histogram of roc_hist_kernel (utils/utils.py:51-83), ONE workgroup: hist[f][lab][j] is turned into
// prefix sums in place (cum[f][lab][t] = #{pairs of the fold's test split predicted same at threshold t}), then per fold the
// threshold maximising the TRAIN-split accuracy (first maximum, as numpy.argmax; the accuracy's denominator is the same for
// every threshold, so the integer numerator tp + tn decides -- exact), the test accuracy at it, and tpr / fpr at every
// threshold; the fold means are summed in fold order and divided by F like numpy.mean(axis=0).  All rates are IEEE double
// divisions of the same integers the host formula divides: bit-identical to the numpy evaluation.
constexpr int RT = 1024;
constexpr int ROC_MAX_FOLDS = 256;   // the reference's default is nrof_folds = 50 (utils/utils.py:26)
__global__ __launch_bounds__(RT) void roc_sweep_kernel(unsigned long long* __restrict__ hist, int T, int F, double* __restrict__ mean_tpr,
                                                       double* __restrict__ mean_fpr, double* __restrict__ acc,
                                                       int* __restrict__ best_idx) {
  __shared__ long long s_part[RT];
  __shared__ long long s_tot[2 * ROC_MAX_FOLDS];     // [f][lab]
  __shared__ long long s_bv[RT / 64];
  __shared__ int s_bi[RT / 64];
  __shared__ int s_best;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nb = T + 1, seg = (nb + RT - 1) / RT;
  const int j0 = t * seg, j1 = (j0 + seg < nb) ? j0 + seg : nb;
  // ---- in-place inclusive prefix sums of the 2 F rows
  for (int row = 0; row < 2 * F; ++row) {
    unsigned long long* h = hist + (size_t)row * nb;
    long long sum = 0;
    for (int j = j0; j < j1; ++j) sum += (long long)h[j];
    s_part[t] = sum;
    __syncthreads();
    for (int off = 1; off < RT; off <<= 1) {       // Hillis-Steele inclusive scan of the segment sums
      const long long v = t >= off ? s_part[t - off] : 0;
      __syncthreads();
      s_part[t] += v;
      __syncthreads();
    }
    long long run = s_part[t] - sum;               // exclusive base of this thread's segment
    for (int j = j0; j < j1; ++j) {
      run += (long long)h[j];
      h[j] = (unsigned long long)run;
    }
    if (t == RT - 1) s_tot[row] = s_part[t];
    __syncthreads();
  }
  long long pos_all = 0, neg_all = 0;
  for (int f = 0; f < F; ++f) { neg_all += s_tot[2 * f]; pos_all += s_tot[2 * f + 1]; }
  // ---- per fold: arg max over thresholds of the train-split (tp + tn); ties -> smallest index
  for (int f = 0; f < F; ++f) {
    const long long neg_tr = neg_all - s_tot[2 * f];
    long long bv = -1;
    int bi = 0x7fffffff;
    for (int th = t; th < T; th += RT) {
      long long tp_all = 0, fp_all = 0;
      for (int g = 0; g < F; ++g) {
        fp_all += (long long)hist[(size_t)(2 * g) * nb + th];
        tp_all += (long long)hist[(size_t)(2 * g + 1) * nb + th];
      }
      const long long tp_tr = tp_all - (long long)hist[(size_t)(2 * f + 1) * nb + th];
      const long long fp_tr = fp_all - (long long)hist[(size_t)(2 * f) * nb + th];
      const long long v = tp_tr + (neg_tr - fp_tr);
      if (v > bv) { bv = v; bi = th; }             // th ascending per thread: keeps the first maximum
    }
    for (int o = 32; o > 0; o >>= 1) {
      const long long ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_bv[wv] = bv; s_bi[wv] = bi; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < RT / 64; ++w)
        if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < bi)) { bv = s_bv[w]; bi = s_bi[w]; }
      s_best = bi;
      best_idx[f] = bi;
      const long long tp = (long long)hist[(size_t)(2 * f + 1) * nb + bi], fp = (long long)hist[(size_t)(2 * f) * nb + bi];
      const long long tn = s_tot[2 * f] - fp, n_te = s_tot[2 * f] + s_tot[2 * f + 1];
      acc[f] = (double)(tp + tn) / (double)n_te;   // n_te == 0 -> nan, as the host formula
    }
    __syncthreads();
  }
  // ---- tpr / fpr at every threshold, mean over folds (fold order)
  for (int th = t; th < T; th += RT) {
    double st = 0.0, sf = 0.0;
    for (int f = 0; f < F; ++f) {
      const long long tp = (long long)hist[(size_t)(2 * f + 1) * nb + th], fp = (long long)hist[(size_t)(2 * f) * nb + th];
      const long long pos = s_tot[2 * f + 1], neg = s_tot[2 * f];
      st += pos == 0 ? 0.0 : (double)tp / (double)pos;
      sf += neg == 0 ? 0.0 : (double)fp / (double)neg;
    }
    mean_tpr[th] = st / (double)F;
    mean_fpr[th] = sf / (double)F;
  }
}

}  // namespace

#define XR_DISPATCH(dtype, ...)                 \
  do {                                          \
    if ((dtype) == XR_BF16) {                   \
      using T = bf16_t;                         \
      __VA_ARGS__;                              \
    } else {                                    \
      using T = float;                          \
      __VA_ARGS__;                              \
    }                                           \
  } while (0)

static int chk_img(const char* name, int dtype, int N, int H, int W, int C) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "%s: bad dtype %d", name, dtype);
  XR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "%s: bad dims N=%d H=%d W=%d C=%d", name, N, H, W, C);
  return XR_OK;
}

extern "C" int xr_subsample(int dtype, const void* x, void* y, int N, int H, int W, int C, int stride, void* stream) {
  if (int e = chk_img("xr_subsample", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(x && y && stride >= 1, "xr_subsample: bad arguments");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(subsample_kernel<T>, dim3(grid_for((int64_t)N * Ho * Wo * (C / 8))), dim3(NT), 0, st,
                                        (const T*)x, (T*)y, N, H, W, C, stride, 0));
  XR_CHECK_LAUNCH("xr_subsample");
  return XR_OK;
}
extern "C" int xr_subsample_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int stride, void* stream) {
  if (int e = chk_img("xr_subsample_bwd", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(dy && dx && stride >= 1, "xr_subsample_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(subsample_kernel<T>, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(NT), 0, st,
                                        (const T*)dy, (T*)dx, N, H, W, C, stride, 1));
  XR_CHECK_LAUNCH("xr_subsample_bwd");
  return XR_OK;
}
extern "C" int xr_maxpool2(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (int e = chk_img("xr_maxpool2", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(x && y && H % 2 == 0 && W % 2 == 0, "xr_maxpool2: needs even H, W");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(maxpool2_kernel<T>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(NT), 0,
                                        st, (const T*)x, (T*)y, N, H, W, C));
  XR_CHECK_LAUNCH("xr_maxpool2");
  return XR_OK;
}
extern "C" int xr_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int e = chk_img("xr_maxpool2_bwd", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(x && dy && dx && H % 2 == 0 && W % 2 == 0, "xr_maxpool2_bwd: needs even H, W");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(maxpool2_bwd_kernel<T>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(NT),
                                        0, st, (const T*)x, (const T*)dy, (T*)dx, N, H, W, C));
  XR_CHECK_LAUNCH("xr_maxpool2_bwd");
  return XR_OK;
}
extern "C" int xr_upadd2(int dtype, const void* up1, const void* low, void* y, int N, int H, int W, int C, void* stream) {
  if (int e = chk_img("xr_upadd2", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(up1 && low && y && H % 2 == 0 && W % 2 == 0, "xr_upadd2: needs even output H, W");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(upadd2_kernel<T>, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(NT), 0, st,
                                        (const T*)up1, (const T*)low, (T*)y, N, H, W, C));
  XR_CHECK_LAUNCH("xr_upadd2");
  return XR_OK;
}
extern "C" int xr_upadd2_bwd(int dtype, const void* dy, void* dlow, int N, int H, int W, int C, void* stream) {
  if (int e = chk_img("xr_upadd2_bwd", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(dy && dlow && H % 2 == 0 && W % 2 == 0, "xr_upadd2_bwd: needs even H, W");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(upadd2_bwd_kernel<T>, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(NT), 0,
                                        st, (const T*)dy, (T*)dlow, N, H, W, C));
  XR_CHECK_LAUNCH("xr_upadd2_bwd");
  return XR_OK;
}
extern "C" int xr_reflect_pad(int dtype, const void* x, void* y, int N, int H, int W, int C, int pad, void* stream) {
  if (int e = chk_img("xr_reflect_pad", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(x && y && pad >= 1 && pad < H && pad < W, "xr_reflect_pad: pad must be in [1, min(H,W))");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(reflect_pad_kernel<T>, dim3(grid_for((int64_t)N * (H + 2 * pad) * (W + 2 * pad) * (C / 8))),
                                        dim3(NT), 0, st, (const T*)x, (T*)y, N, H, W, C, pad));
  XR_CHECK_LAUNCH("xr_reflect_pad");
  return XR_OK;
}
extern "C" int xr_reflect_pad_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int pad, void* stream) {
  if (int e = chk_img("xr_reflect_pad_bwd", dtype, N, H, W, C)) return e;
  XR_CHECK_ARG(dy && dx && pad >= 1 && pad < H && pad < W, "xr_reflect_pad_bwd: pad must be in [1, min(H,W))");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(reflect_pad_bwd_kernel<T>, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(NT), 0, st,
                                        (const T*)dy, (T*)dx, N, H, W, C, pad));
  XR_CHECK_LAUNCH("xr_reflect_pad_bwd");
  return XR_OK;
}
extern "C" int xr_copy_channels(int dtype, const void* src, int lds_, int src_off, void* dst, int ldd, int dst_off, int64_t M,
                                int C, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_copy_channels: bad dtype");
  XR_CHECK_ARG(src && dst && M > 0 && C > 0 && C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0 && src_off % 8 == 0 && dst_off % 8 == 0 &&
                   src_off + C <= lds_ && dst_off + C <= ldd,
               "xr_copy_channels: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(grid_for(M * (C / 8))), dim3(NT), 0, st, (const T*)src, lds_,
                                        src_off, (T*)dst, ldd, dst_off, M, C));
  XR_CHECK_LAUNCH("xr_copy_channels");
  return XR_OK;
}
extern "C" int xr_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_nchw_to_nhwc: bad dtype");
  XR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 8 == 0, "xr_nchw_to_nhwc: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for((int64_t)N * H * W * (Cp / 8))), dim3(NT), 0, st, src,
                                        (T*)dst, N, C, H * W, Cp));
  XR_CHECK_LAUNCH("xr_nchw_to_nhwc");
  return XR_OK;
}
extern "C" int xr_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_nhwc_to_nchw: bad dtype");
  XR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "xr_nhwc_to_nchw: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for((int64_t)N * C * H * W)), dim3(NT), 0, st,
                                        (const T*)src, dst, N, C, H * W, Cp));
  XR_CHECK_LAUNCH("xr_nhwc_to_nchw");
  return XR_OK;
}
extern "C" int xr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream) {
  XR_CHECK_ARG(src && dst && n > 0, "xr_cast: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(n);
  if (src_dtype == XR_F32 && dst_dtype == XR_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(NT), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == XR_BF16 && dst_dtype == XR_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(NT), 0, st, (const bf16_t*)src, (float*)dst, n);
  else {
    xr_set_error("xr_cast: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    return XR_E_INVALID;
  }
  XR_CHECK_LAUNCH("xr_cast");
  return XR_OK;
}
static int addsub(int dtype, const void* a, const void* b, void* y, int64_t n, float sign, void* stream, const char* name) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "%s: bad dtype", name);
  XR_CHECK_ARG(a && b && y && n > 0, "%s: bad arguments", name);
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(addsub_kernel<T>, dim3(grid_for(n / 8 + 1)), dim3(NT), 0, st, (const T*)a, (const T*)b,
                                        (T*)y, n / 8, n, sign));
  XR_CHECK_LAUNCH(name);
  return XR_OK;
}
extern "C" int xr_add(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream) {
  return addsub(dtype, a, b, y, n, 1.f, stream, "xr_add");
}
extern "C" int xr_sub(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream) {
  return addsub(dtype, a, b, y, n, -1.f, stream, "xr_sub");
}
extern "C" int xr_dropout(int dtype, const void* x, const uint8_t* mask, void* y, int64_t n, float p, uint64_t seed,
                          const void* tick, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_dropout: bad dtype");
  XR_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f, "xr_dropout: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid_for(n)), dim3(NT), 0, st, (const T*)x, mask, (T*)y, n, p,
                                        (uint64_t)seed, (const uint64_t*)tick));
  XR_CHECK_LAUNCH("xr_dropout");
  return XR_OK;
}

extern "C" int xr_loss_mse(int dtype, const void* a, const void* b, float scale, float gscale, float* loss, void* da, void* db,
                           int64_t n, int64_t n_valid, const float* gscale_dev, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_loss_mse: bad dtype");
  XR_CHECK_ARG(a && b && (loss || da || db) && n > 0 && n_valid > 0, "xr_loss_mse: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float ls = scale / (float)n_valid, gc = gscale * 2.f * scale / (float)n_valid;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(loss_mse_kernel<T>, dim3(loss ? det_grid(grid_for(n, NT * 4, 2048)) : grid_for(n, NT * 4, 2048)), dim3(NT), 0, st, (const T*)a,
                                        (const T*)b, ls, gc, loss, (T*)da, (T*)db, n, gscale_dev));
  XR_CHECK_LAUNCH("xr_loss_mse");
  return XR_OK;
}
extern "C" int xr_loss_landmark(const float* pred, const float* target, float scale, float gscale, float* loss, float* dpred,
                                int N, int C, int HW, const float* gscale_dev, void* stream) {
  XR_CHECK_ARG(pred && target && (loss || dpred) && N > 0 && C > 0 && HW > 0, "xr_loss_landmark: bad arguments");
  const int64_t NP = (int64_t)N * HW;
  const float ls = scale / (float)NP, gc = gscale * 2.f * scale / (float)NP;
  hipLaunchKernelGGL(loss_landmark_nchw_kernel, dim3(loss ? det_grid(grid_for(NP, NT, 2048)) : grid_for(NP, NT, 2048)), dim3(NT), 0, (hipStream_t)stream, pred, target, ls,
                     gc, loss, dpred, NP, C, HW, gscale_dev);
  XR_CHECK_LAUNCH("xr_loss_landmark");
  return XR_OK;
}
extern "C" int xr_loss_ce_nchw(const float* pred, const int64_t* target, float gscale, float* loss, float* dpred, int N, int C,
                               int HW, const float* gscale_dev, void* stream) {
  XR_CHECK_ARG(pred && target && (loss || dpred) && N > 0 && C > 0 && HW > 0, "xr_loss_ce_nchw: bad arguments");
  const int64_t NP = (int64_t)N * HW;
  hipLaunchKernelGGL(loss_ce_nchw_kernel, dim3(loss ? det_grid(grid_for(NP, NT, 2048)) : grid_for(NP, NT, 2048)), dim3(NT), 0, (hipStream_t)stream, pred, target,
                     gscale / (float)NP, 1.f / (float)NP, loss, dpred, NP, C, HW, gscale_dev);
  XR_CHECK_LAUNCH("xr_loss_ce_nchw");
  return XR_OK;
}
extern "C" int xr_loss_softmax_ce(int dtype, const void* pred, const int64_t* target, float gscale, float* loss, void* dpred,
                                  int64_t M, int C, int ld, const float* gscale_dev, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "xr_loss_softmax_ce: bad dtype");
  XR_CHECK_ARG(pred && target && (loss || dpred) && M > 0 && C > 0 && ld >= C, "xr_loss_softmax_ce: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  XR_DISPATCH(dtype, hipLaunchKernelGGL(loss_softmax_ce_kernel<T>, dim3(loss ? det_grid(grid_for(M, 4, 2048)) : grid_for(M, 4, 2048)), dim3(NT), 0, st, (const T*)pred,
                                        target, gscale / (float)M, 1.f / (float)M, loss, (T*)dpred, M, C, ld, gscale_dev));
  XR_CHECK_LAUNCH("xr_loss_softmax_ce");
  return XR_OK;
}
extern "C" int xr_arcface_margin(float* cos_logits, const int64_t* target, float* dphi_dcos, int64_t M, int C, float s, float m,
                                 void* stream) {
  XR_CHECK_ARG(cos_logits && target && dphi_dcos && M > 0 && C > 0, "xr_arcface_margin: bad arguments");
  const float cm = cosf(m), sm = sinf(m), th = cosf(3.14159265358979323846f - m), mm = sinf(3.14159265358979323846f - m) * m;
  hipLaunchKernelGGL(arcface_margin_kernel, dim3(grid_for(M * C)), dim3(NT), 0, (hipStream_t)stream, cos_logits, target,
                     dphi_dcos, M, C, s, cm, sm, th, mm);
  XR_CHECK_LAUNCH("xr_arcface_margin");
  return XR_OK;
}
extern "C" int xr_l2norm_rows(const float* x, float* y, float* inv_norm, int64_t M, int C, void* stream) {
  XR_CHECK_ARG(x && y && inv_norm && M > 0 && C > 0, "xr_l2norm_rows: bad arguments");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid_for(M, 4, 4096)), dim3(NT), 0, (hipStream_t)stream, x, y, inv_norm, M, C);
  XR_CHECK_LAUNCH("xr_l2norm_rows");
  return XR_OK;
}
extern "C" int xr_l2norm_rows_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int64_t M, int C,
                                  void* stream) {
  XR_CHECK_ARG(y && inv_norm && dy && dx && M > 0 && C > 0, "xr_l2norm_rows_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(grid_for(M, 4, 4096)), dim3(NT), 0, (hipStream_t)stream, y, inv_norm, dy, dx, M,
                     C);
  XR_CHECK_LAUNCH("xr_l2norm_rows_bwd");
  return XR_OK;
}

extern "C" int xr_mmd_fwd(const float* z, float* w, float* loss, int N, int D, const float* sigmas, int nsig, void* stream) {
  XR_CHECK_ARG(z && w && loss && sigmas && N > 0 && D > 0 && nsig > 0, "xr_mmd_fwd: bad arguments");
  const int64_t pairs = (int64_t)4 * N * N;
  hipLaunchKernelGGL(mmd_fwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(NT), 0, (hipStream_t)stream, z, w, loss, N, D, sigmas,
                     nsig);
  XR_CHECK_LAUNCH("xr_mmd_fwd");
  return XR_OK;
}
extern "C" int xr_mmd_bwd(const float* z, const float* w, float* dz, int N, int D, const float* gscale_dev, void* stream) {
  XR_CHECK_ARG(z && w && dz && N > 0 && D > 0, "xr_mmd_bwd: bad arguments");
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(2 * N), dim3(NT), 0, (hipStream_t)stream, z, w, dz, 2 * N, D, gscale_dev);
  XR_CHECK_LAUNCH("xr_mmd_bwd");
  return XR_OK;
}

extern "C" int xr_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float wd,
                           const uint8_t* wd_mask, int first_step, void* stream) {
  XR_CHECK_ARG(p && g && n > 0 && (momentum == 0.f || mom), "xr_sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, NT * 4)), dim3(NT), 0, (hipStream_t)stream, p, g, mom, n, lr, momentum, wd,
                     wd_mask, first_step);
  XR_CHECK_LAUNCH("xr_sgd_step");
  return XR_OK;
}
extern "C" int xr_rmsprop_step(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float wd,
                               const uint8_t* wd_mask, void* stream) {
  XR_CHECK_ARG(p && g && sq && n > 0, "xr_rmsprop_step: bad arguments");
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n, NT * 4)), dim3(NT), 0, (hipStream_t)stream, p, g, sq, n, lr, alpha, eps, wd,
                     wd_mask);
  XR_CHECK_LAUNCH("xr_rmsprop_step");
  return XR_OK;
}
extern "C" int xr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                            float wd, int step, const void* tick, int64_t tick_ref, const uint8_t* wd_mask, void* stream) {
  XR_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "xr_adam_step: bad arguments");
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, NT * 4)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, wd,
                     bc1, bc2, step, (const uint64_t*)tick, (uint64_t)tick_ref, wd_mask);
  XR_CHECK_LAUNCH("xr_adam_step");
  return XR_OK;
}

extern "C" int xr_pairdist_l2(const float* e1, const float* e2, float* dist, int64_t P, int D, void* stream) {
  XR_CHECK_ARG(e1 && e2 && dist && P > 0 && D > 0, "xr_pairdist_l2: bad arguments");
  hipLaunchKernelGGL(pairdist_kernel, dim3(grid_for(P, 4, 8192)), dim3(NT), 0, (hipStream_t)stream, e1, e2, dist, P, D);
  XR_CHECK_LAUNCH("xr_pairdist_l2");
  return XR_OK;
}
extern "C" int xr_roc_hist(const float* dist, const uint8_t* issame, const int32_t* fold_id, const float* thresholds,
                           unsigned long long* hist, int64_t P, int T, int F, void* stream) {
  XR_CHECK_ARG(dist && issame && thresholds && hist && P > 0 && T > 0 && T <= 16000 && F > 0, "xr_roc_hist: bad arguments");
  hipLaunchKernelGGL(roc_hist_kernel, dim3(grid_for(P, NT, 2048)), dim3(NT), (size_t)T * sizeof(float), (hipStream_t)stream, dist,
                     issame, fold_id, thresholds, hist, P, T, F);
  XR_CHECK_LAUNCH("xr_roc_hist");
  return XR_OK;
}
extern "C" int xr_roc_sweep(unsigned long long* hist, int T, int F, double* mean_tpr, double* mean_fpr, double* acc, int* best_idx,
                            void* stream) {
  XR_CHECK_ARG(hist && mean_tpr && mean_fpr && acc && best_idx, "xr_roc_sweep: null pointer");
  XR_CHECK_ARG(T > 0 && T <= 16000 && F >= 2 && F <= ROC_MAX_FOLDS, "xr_roc_sweep: needs 0 < T <= 16000 thresholds and 2 <= F <= 256 folds");
  hipLaunchKernelGGL(roc_sweep_kernel, dim3(1), dim3(RT), 0, (hipStream_t)stream, hist, T, F, mean_tpr, mean_fpr, acc, best_idx);
  XR_CHECK_LAUNCH("xr_roc_sweep");
  return XR_OK;
}
