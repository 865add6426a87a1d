// xr_data.hip -- the two tensor-producing steps either side of the FHN hot path that the reference runs on the CPU inside its
// DataLoader workers, on the device:
//   * low-resolution input synthesis, SUPER_RESOLUTION/FHN_loader.py:65-66:
//         lr_img = sr_img.resize((int(128 / scale),) * 2).resize((112, 112), Image.BICUBIC)      (Image.resize defaults to BICUBIC)
//     i.e. Pillow's 8-bit resampler twice (shrink with antialiasing, then enlarge), followed by ToTensor + Normalize(0.5, 0.5)
//     (:30-35).  Pillow's algorithm (src/libImaging/Resample.c) is restated exactly: per-output-pixel tap windows, the a = -0.5
//     cubic with support 2 * max(scale, 1), taps normalised in double and rounded to 22-bit fixed point, horizontal pass first,
//     every pass rounded to uint8 -- integer arithmetic from there on, so the result is BIT-IDENTICAL to PIL (oracle:
//     oracle/cpu_ref.py:pil_resize_bicubic, pinned against PIL 12.2.0 by tests/golden/loader.npz);
//   * landmark heat-maps, FHN_loader.py:119-137 / helen_loader.py:124-143: hm = sum_i exp(-((x - x_i)^2 + (y - y_i)^2) / (2 s^2)),
//     each bump evaluated in double and added to a float32 running sum in landmark order, as the reference's in-place += does.
// One workgroup per image: a 112 x 112 x 3 crop (37 KB) and all four intermediate images live in LDS; HBM sees one read of the
// crop and one write of the result.
#include "xr_common.h"

namespace {

constexpr int DT = 256;
constexpr int PBITS = 32 - 8 - 2;   // Pillow's PRECISION_BITS for 8-bit channels

struct Taps {        // one resampling pass: out_size windows of <= ksize taps
  int* kk;           // [out][ksize] fixed-point taps
  int* bnd;          // [out][2]: first input index, tap count
  int ksize;
};

__device__ __forceinline__ double pil_bicubic(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

__device__ __forceinline__ int pil_ksize(int in_size, int out_size) {
#pragma clang fp contract(off)
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(2.0 * fs) * 2 + 1;
}

// Resample.c:precompute_coeffs + normalize_coeffs_8bpc, one output index per thread (no fused multiply-adds: the doubles must
// round exactly as the C code's do)
__device__ void build_taps(int in_size, int out_size, const Taps& tp) {
#pragma clang fp contract(off)
  double scale = (double)in_size / out_size, fs = scale;
  if (fs < 1.0) fs = 1.0;
  const double support = 2.0 * fs, ss = 1.0 / fs;
  for (int xx = threadIdx.x; xx < out_size; xx += DT) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pil_bicubic((x + xmin - center + 0.5) * ss);
    int* k = tp.kk + xx * tp.ksize;
    for (int x = 0; x < xmax; ++x) {
      double v = pil_bicubic((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) v = v / ww;
      k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PBITS)) : (int)(0.5 + v * (double)(1 << PBITS));
    }
    for (int x = xmax; x < tp.ksize; ++x) k[x] = 0;
    tp.bnd[2 * xx] = xmin;
    tp.bnd[2 * xx + 1] = xmax;
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PBITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: src [rows][in_w][3] -> dst [rows][out_w][3]
__device__ void pass_h(const uint8_t* src, uint8_t* dst, int rows, int in_w, int out_w, const Taps& tp) {
  const int total = rows * out_w * 3;
  for (int e = threadIdx.x; e < total; e += DT) {
    const int c = e % 3, xx = (e / 3) % out_w, y = e / (3 * out_w);
    const int x0 = tp.bnd[2 * xx], n = tp.bnd[2 * xx + 1];
    const int* k = tp.kk + xx * tp.ksize;
    const uint8_t* s = src + (y * in_w + x0) * 3 + c;
    int acc = 1 << (PBITS - 1);
    for (int x = 0; x < n; ++x) acc += (int)s[3 * x] * k[x];
    dst[e] = (uint8_t)clip8(acc);
  }
}

// vertical pass: src [in_h][w][3] -> dst [out_h][w][3]
__device__ void pass_v(const uint8_t* src, uint8_t* dst, int in_h, int out_h, int w, const Taps& tp) {
  const int row = w * 3, total = out_h * row;
  for (int e = threadIdx.x; e < total; e += DT) {
    const int yy = e / row, r = e - yy * row;
    const int y0 = tp.bnd[2 * yy], n = tp.bnd[2 * yy + 1];
    const int* k = tp.kk + yy * tp.ksize;
    const uint8_t* s = src + y0 * row + r;
    int acc = 1 << (PBITS - 1);
    for (int y = 0; y < n; ++y) acc += (int)s[y * row] * k[y];
    dst[e] = (uint8_t)clip8(acc);
  }
}

struct LRP {
  const uint8_t* hr;     // [N][H][W][3]
  const int32_t* low;    // [N] low-resolution edge per image
  uint8_t* lr_u8;        // [N][H][W][3] or null
  float* lr_norm;        // [N][3][H][W] or null
  int N, H, W, max_low, max_ks;
};

__global__ __launch_bounds__(DT) void lr_synth_kernel(LRP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = blockIdx.x, H = p.H, W = p.W, t = threadIdx.x;
  int S = p.low[n];
  if (S < 1) S = 1;
  if (S > p.max_low) S = p.max_low;   // never index past the LDS plan the host sized for max_low ...
  {
    const int big = H > W ? H : W;      // ... nor past the tap tables, which hold shrink factors up to 16
    if (S * 16 < big) S = (big + 15) / 16;
  }
  const int img = H * W * 3;
  // LDS plan (sized by the host for max_low / max_ks): A input + final image, B = H x S, C = S x S, D = S x W, then the tap tables
  uint8_t* A = smem;
  uint8_t* B = A + ((img + 15) & ~15);
  uint8_t* C = B + ((H * p.max_low * 3 + 15) & ~15);
  uint8_t* D = C + ((p.max_low * p.max_low * 3 + 15) & ~15);
  int* tab = reinterpret_cast<int*>(D + ((p.max_low * W * 3 + 15) & ~15));
  Taps dx{tab, tab + p.max_low * p.max_ks, pil_ksize(W, S)};
  tab += p.max_low * (p.max_ks + 2);
  Taps dy{tab, tab + p.max_low * p.max_ks, pil_ksize(H, S)};
  tab += p.max_low * (p.max_ks + 2);
  Taps ux{tab, tab + W * 5, 5};
  tab += W * 7;
  Taps uy{tab, tab + H * 5, 5};

  const uint8_t* src = p.hr + (size_t)n * img;
  if ((img & 3) == 0 && ((size_t)src & 3) == 0) {
    for (int e = t; e < img / 4; e += DT) reinterpret_cast<unsigned*>(A)[e] = reinterpret_cast<const unsigned*>(src)[e];
  } else {
    for (int e = t; e < img; e += DT) A[e] = src[e];
  }
  build_taps(W, S, dx);
  build_taps(H, S, dy);
  build_taps(S, W, ux);
  build_taps(S, H, uy);
  __syncthreads();
  const uint8_t* cur = A;
  int ch = H, cw = W;
  if (S != W) { pass_h(cur, B, ch, cw, S, dx); cur = B; cw = S; __syncthreads(); }
  if (S != H) { pass_v(cur, C, ch, S, cw, dy); cur = C; ch = S; __syncthreads(); }
  if (cw != W) { pass_h(cur, D, ch, cw, W, ux); cur = D; cw = W; __syncthreads(); }
  if (ch != H) { pass_v(cur, A, ch, H, cw, uy); cur = A; ch = H; __syncthreads(); }   // A (the input) is dead by now
  if (p.lr_u8 != nullptr) {
    uint8_t* dst = p.lr_u8 + (size_t)n * img;
    for (int e = t; e < img; e += DT) dst[e] = cur[e];
  }
  if (p.lr_norm != nullptr) {
    // ToTensor (uint8 HWC -> float32 CHW, / 255) + Normalize(0.5, 0.5): the same three float32 operations, correctly rounded
    float* dst = p.lr_norm + (size_t)n * img;
    const int hw = H * W;
    for (int e = t; e < img; e += DT) {
      const int c = e / hw, px = e - c * hw;
      const float v = __fdiv_rn((float)cur[px * 3 + c], 255.0f);
      dst[e] = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
    }
  }
}

__global__ __launch_bounds__(DT) void heatmap_kernel(const double* __restrict__ lmk, float* __restrict__ hm, int N, int L, int H, int W,
                                                     double two_s2) {
#pragma clang fp contract(off)
  extern __shared__ double s_l[];   // this image's landmarks
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * L; i += DT) s_l[i] = lmk[(size_t)n * 2 * L + i];
  __syncthreads();
  for (int px = blockIdx.x * DT + threadIdx.x; px < H * W; px += gridDim.x * DT) {
    const double y = (double)(px / W), x = (double)(px % W);
    float acc = 0.f;
    for (int i = 0; i < L; ++i) {
      const double dx = x - s_l[2 * i], dy = y - s_l[2 * i + 1];
      const double g = exp(-(dx * dx + dy * dy) / two_s2);
      acc = (float)((double)acc + g);
    }
    hm[(size_t)n * H * W + px] = acc;
  }
}

}  // namespace

extern "C" int xr_lr_synth(const uint8_t* hr, const int32_t* low, int max_low, uint8_t* lr_u8, float* lr_norm, int N, int H, int W,
                           void* stream) {
  XR_CHECK_ARG(hr && low && (lr_u8 || lr_norm), "xr_lr_synth: null pointer");
  XR_CHECK_ARG(N > 0 && H > 0 && W > 0 && max_low > 0, "xr_lr_synth: non-positive dimension");
  XR_CHECK_ARG(max_low <= H && max_low <= W, "xr_lr_synth: the low-resolution edge %d exceeds the image (%d x %d)", max_low, H, W);
  // tap tables are sized for shrink factors up to 16 (2 * ceil(2 * 16) + 1 taps); low[n] outside [max(H, W) / 16, max_low] is
  // clamped into that range by the kernel
  const int big = H > W ? H : W;
  const int max_ks = 2 * 2 * 16 + 1;
  XR_CHECK_ARG(max_low * 16 >= big, "xr_lr_synth: shrink factor above 16 (max_low %d for a %d x %d image)", max_low, H, W);
  auto r16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t smem = r16((size_t)H * W * 3) + r16((size_t)H * max_low * 3) + r16((size_t)max_low * max_low * 3) +
                      r16((size_t)max_low * W * 3) + sizeof(int) * ((size_t)2 * max_low * (max_ks + 2) + 7 * (size_t)(W + H));
  XR_CHECK_ARG(smem <= 160 * 1024, "xr_lr_synth: image %d x %d (low %d) needs %zu B of LDS (> 160 KiB)", H, W, max_low, smem);
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(lr_synth_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (err != hipSuccess) {
    xr_set_error("xr_lr_synth: hipFuncSetAttribute failed: %s", hipGetErrorString(err));
    return XR_E_LAUNCH;
  }
  LRP p{hr, low, lr_u8, lr_norm, N, H, W, max_low, max_ks};
  hipLaunchKernelGGL(lr_synth_kernel, dim3((unsigned)N), dim3(DT), smem, (hipStream_t)stream, p);
  XR_CHECK_LAUNCH("xr_lr_synth");
  return XR_OK;
}

extern "C" int xr_heatmap(const double* landmarks, float* hm, int N, int L, int H, int W, double sigma, void* stream) {
  XR_CHECK_ARG(landmarks && hm && N > 0 && L > 0 && H > 0 && W > 0 && sigma > 0.0, "xr_heatmap: bad arguments");
  XR_CHECK_ARG(L <= 2048 && N <= 65535, "xr_heatmap: at most 2048 landmarks and 65535 images per call");
  const int bx = (H * W + DT - 1) / DT;
  hipLaunchKernelGGL(heatmap_kernel, dim3((unsigned)(bx < 64 ? bx : 64), (unsigned)N), dim3(DT), (size_t)2 * L * sizeof(double),
                     (hipStream_t)stream, landmarks, hm, N, L, H, W, 2 * sigma * sigma);
  XR_CHECK_LAUNCH("xr_heatmap");
  return XR_OK;
}
