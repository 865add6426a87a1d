// xr_wgrad64.hip -- direct weight gradient of the 64 -> 64 channel, 3x3, stride-1, pad-1 bf16 layers (every FSRNet body layer,
// model/FSRnet.py:79,85; IR / ResNet stage 1).  dW[k][tap][c] = sum over pixels of dY[pixel][k] * X[pixel + tap][c].
//
// Why not the sliced implicit GEMM of xr_conv.hip (wgrad_kernel<64, 256>): with Kg = 576 its 256-wide column tiles waste a
// quarter of the MFMAs, every tile re-gathers dY, the im2col gather pulls X nine times through L2, and at 52 FLOP per byte of
// L2 -> LDS traffic the kernel sits at ~300 TFLOP/s.  Here
//   * one persistent 12-wave workgroup per CU walks a contiguous run of image rows; an input row is staged ONCE into a ring of
//     four LDS row slots (global -> registers -> LDS) and serves the three output rows and nine taps that touch it; a dY row is
//     staged once into a ring of two.  HBM traffic = one read of X and one of dY: 288 FLOP per byte;
//   * the whole 64 x 576 gradient stays in registers for the workgroup's lifetime: wave = (tap row dh, 32-row half of K,
//     32-column half of C) owns the three taps (dh, -1 / 0 / +1) of its quadrant = 48 accumulator registers; the three waves
//     of a SIMD are the three tap rows of one quadrant, so a row at the top / bottom edge of an image (one tap row has
//     nothing to add there and is skipped, never multiplied by zeros) idles each SIMD equally;
//   * the reduction runs over pixels, the slow axis of both NHWC operands: both LDS images are pixel-major (128 B per pixel)
//     and fragments come from ds_read_b64_tr_b16 (the transposing LDS read of gfx950) -- no transposing pass anywhere; the
//     16-B chunk is XOR-swizzled by bit 1 of the pixel row so that the four pixel rows a half-wave touches fall on the four
//     quarters of the 256-B bank row for EVERY tap shift (any four consecutive rows have distinct (row & 1, row >> 1 & 1));
//   * staging goes through registers because it can TRANSFORM (XF): x' = prelu(x * scale[n][c] + shift[n][c], alpha[c]), the
//     InstanceNorm apply + PReLU of the producing layer (model/FSRnet.py:81-84) -- the normalised activation y1 that conv2's
//     weight gradient needs is never written to HBM (the forward never wrote it either, xr_conv64.hip);
//   * one barrier per image row (21 MFMAs of 32 cycles per wave between barriers, three waves per SIMD to cover the LDS
//     reads and the staging); the loads of row r + 2 are issued before the MFMAs of row r and written to LDS after them;
//   * every workgroup writes its 64 x 576 fp32 partial as one slab in the packed [K][Kg] layout of xr_conv_wgrad (plain
//     128-B row stores); xr_unpack_wgrad sums the slabs: <= 256 slabs of 147 KB instead of an atomics tail.
#include "xr_common.h"
#include <type_traits>

namespace {

constexpr int NT = 768;           // 12 waves: 3 tap rows x 4 quadrants
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define XRW_OOR 0x80000000u

struct WG64P {
  const bf16_t* x;        // [N][H][W][64]
  const bf16_t* dy;       // [N][H][W][64]
  float* slabs;           // [grid][64][576]
  const float* n_scale;   // [N][64] per-image affine applied to x on load (XF), or null
  const float* n_shift;
  const float* n_alpha;   // [64] PReLU slope after the affine, or null
  int N, H, W, rows_total, rows_per_wg;
  unsigned io_bytes;      // extent of x / dy
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) << 2; }

// fragment for v_mfma_f32_32x32x16_bf16 from a pixel-major swizzled image: 16 pixels x 32 channels, lane = (channel lane & 31,
// pixels 8 * (lane >> 5) .. + 7); p = address of the lane's first 8-byte packet, the second one sits 4 pixel rows further
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * 128));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// per-lane byte offset of the first packet: pixel row `row0` + the lane's row, channel column col0 + the lane's column
__device__ __forceinline__ int frag_off(int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, pp = i & 3, cgrp = g & 1, hh = g >> 1;
  const int row = row0 + 8 * hh + q;
  const int cb = (col0 + 16 * cgrp + 4 * pp) * 2;
  return row * 128 + ((((cb >> 4) ^ swz(row)) << 4) | (cb & 15));
}

// NCH: 16-pixel chunks per image row (W <= 16 * NCH); XF: transform x on load
template <int NCH, bool XF>
__global__ __launch_bounds__(NT, 3) void wgrad64_kernel(WG64P p) {
  constexpr int WP = 16 * NCH;
  constexpr int XS = (WP + 2) * 128, YS = WP * 128;     // slot strides
  constexpr int NXS = 4, NYS = 2;
  constexpr int YBASE = NXS * XS;
  constexpr int SMEM = NXS * XS + NYS * YS;
  constexpr int NSTG = (2 * WP * 8 + NT - 1) / NT;      // 16-B chunks a thread stages per step (X row + dY row)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int dh = wave / 4 - 1, quad = wave & 3, kh = quad >> 1, ch = quad & 1;

  const int r0 = blockIdx.x * p.rows_per_wg;
  int r1 = r0 + p.rows_per_wg;
  if (r1 > p.rows_total) r1 = p.rows_total;
  if (r0 >= r1) return;

  // ---- zero the LDS once: the border pixel rows 0 and W + 1 of the X slots and the padding rows beyond W stay zero
  for (int o = t * 16; o < SMEM; o += NT * 16) *reinterpret_cast<v4u_t*>(smem + o) = v4u_t{0u, 0u, 0u, 0u};

  // ---- staging constants: chunk q = t + NT * i of the step's (X row, dY row) pair; the channel chunk cc is fixed per thread.
  // W % 8 == 0 makes the X / dY boundary (W * 8 chunks) a multiple of 64: a wave's chunk i is all-X, all-dY or nothing, so the
  // descriptor, the row base and the LDS slot are scalar selects (no per-lane branch around a memory operation)
  const int cc = t & 7;
  const int XW8 = p.W * 8;
  const int row_bytes = p.W * 128;
  int gof[NSTG], lof[NSTG];     // global byte offset inside the row (X or dY), LDS byte offset inside the slot
  int kind[NSTG];               // wave-uniform: 0 X chunk, 1 dY chunk, 2 nothing
  __amdgpu_buffer_rsrc_t rs[NSTG];
#pragma unroll
  for (int i = 0; i < NSTG; ++i) {
    const int qw = wave * 64 + NT * i;          // first chunk of this wave's group i
    kind[i] = qw < XW8 ? 0 : (qw < 2 * XW8 ? 1 : 2);
    const int q = t + NT * i - (kind[i] == 1 ? XW8 : 0);
    const int row = (q >> 3) + (kind[i] == 0 ? 1 : 0);
    gof[i] = q * 16;
    lof[i] = row * 128 + ((cc ^ swz(row)) << 4);
    rs[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kind[i] == 0 ? p.x : p.dy), 0, p.io_bytes, 0x00020000);
  }
  float sc[8], sh[8], al[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = 1.f, sh[e] = 0.f, al[e] = 1.f;
  if (XF && p.n_alpha != nullptr) ld8(p.n_alpha + cc * 8, al);
  int n_coef = -1;

  v4u_t st[NSTG];
  // xr / yr: flattened image rows (n * H + h) to stage, or anything outside [0, rows_total) for "nothing" (zeros land in LDS)
  auto stage_load = [&](int xr, int yr) {
    const bool xok = (unsigned)xr < (unsigned)p.rows_total, yok = (unsigned)yr < (unsigned)p.rows_total;
    const unsigned xb = (unsigned)(xr * row_bytes), yb = (unsigned)(yr * row_bytes);
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      const bool ok = kind[i] == 0 ? xok : (kind[i] == 1 && yok);
      const unsigned voff = ok ? (kind[i] == 0 ? xb : yb) + (unsigned)gof[i] : XRW_OOR;
      st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[i], voff, 0, 0);
    }
    if constexpr (XF) {
      if (xok) {
        const int n = xr / p.H;
        if (n != n_coef) {   // wave-uniform: coefficients of the image this X row belongs to
          ld8(p.n_scale + (size_t)n * 64 + cc * 8, sc);
          ld8(p.n_shift + (size_t)n * 64 + cc * 8, sh);
          n_coef = n;
        }
      }
    }
  };
  auto stage_write = [&](int xr, int yr) {
    unsigned char* xs = smem + ((xr + 1) & (NXS - 1)) * XS;
    unsigned char* ys = smem + YBASE + (yr & (NYS - 1)) * YS;
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      v4u_t v = st[i];
      if constexpr (XF) {
        if (kind[i] == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = __uint_as_float(v[q] << 16), b = __uint_as_float(v[q] & 0xFFFF0000u);
            a = a * sc[2 * q] + sh[2 * q];
            b = b * sc[2 * q + 1] + sh[2 * q + 1];
            a = a > 0.f ? a : a * al[2 * q];
            b = b > 0.f ? b : b * al[2 * q + 1];
            v[q] = pack2bf(a, b);
          }
        }
      }
      if (kind[i] != 2) *reinterpret_cast<v4u_t*>((kind[i] == 0 ? xs : ys) + lof[i]) = v;
    }
  };

  // ---- fragment addressing (per lane): A = dY^T (rows = output channels kh * 32 ..), B = X shifted by the tap column
  const int offA = frag_off(0, kh * 32, lane);
  int offB[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) offB[s] = frag_off(s, ch * 32, lane);   // X pixel w + (s - 1) lives in slot row w + s

  f32x16_t acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;

  __syncthreads();   // LDS zeroed
  // prologue: X rows r0 - 1, r0, r0 + 1 and dY row r0
  stage_load(r0 - 1, -1);
  stage_write(r0 - 1, -1);
  stage_load(r0, -1);
  stage_write(r0, -1);
  stage_load(r0 + 1, r0);
  stage_write(r0 + 1, r0);
  __syncthreads();

  int h = r0 % p.H;
  for (int r = r0; r < r1; ++r) {
    const bool more = r + 1 < r1;
    // X row r + 2 / dY row r + 1 for the next step; the loads fly under this step's MFMAs
    stage_load(more ? r + 2 : -1, more ? r + 1 : -1);
    if ((unsigned)(h + dh) < (unsigned)p.H) {
      const unsigned char* ys = smem + YBASE + (r & (NYS - 1)) * YS + offA;
      const unsigned char* xs = smem + ((r + dh + 1) & (NXS - 1)) * XS;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bf16x8_t a = tr_frag(ys + c * 2048);
        const bf16x8_t b0 = tr_frag(xs + offB[0] + c * 2048);
        const bf16x8_t b1 = tr_frag(xs + offB[1] + c * 2048);
        const bf16x8_t b2 = tr_frag(xs + offB[2] + c * 2048);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc[2], 0, 0, 0);
      }
    }
    if (more) stage_write(r + 2, r + 1);
    __syncthreads();
    h = h + 1 == p.H ? 0 : h + 1;
  }

  // ---- slab [64][576]: row = output channel, column = tap * 64 + input channel
  float* slab = p.slabs + (size_t)blockIdx.x * 64 * 576;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int col = ((dh + 1) * 3 + s) * 64 + ch * 32 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = kh * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      slab[row * 576 + col] = acc[s][e];
    }
  }
}

template <int NCH, bool XF>
int launch_wgrad64(WG64P& p, int grid, hipStream_t st) {
  constexpr int WP = 16 * NCH;
  constexpr int SMEM = 4 * (WP + 2) * 128 + 2 * WP * 128;
  auto kern = wgrad64_kernel<NCH, XF>;
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [&] {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  });
  if (err != hipSuccess) {
    xr_set_error("xr_conv64_wgrad: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(err));
    return XR_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), SMEM, st, p);
  XR_CHECK_LAUNCH("xr_conv64_wgrad");
  return grid;
}

int cu_count() {
  static std::once_flag once;
  static int cus = 256;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  });
  return cus;
}

}  // namespace

bool xr_wgrad64_eligible(int N, int H, int W) {
  return W >= 8 && W <= 112 && W % 8 == 0 && (long long)N * H * W * 128 < (1ll << 31);
}

extern "C" int xr_conv64_wgrad(const void* in, const void* dy, float* slabs, int N, int H, int W, int max_slabs,
                               const float* in_scale, const float* in_shift, const float* in_alpha, void* stream) {
  XR_CHECK_ARG(in && dy && slabs && N > 0 && H > 0 && W > 0 && max_slabs > 0, "xr_conv64_wgrad: null pointer / non-positive dimension");
  XR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "xr_conv64_wgrad: in_scale and in_shift come together");
  XR_CHECK_ARG(in_alpha == nullptr || in_scale != nullptr, "xr_conv64_wgrad: in_alpha needs in_scale / in_shift");
  XR_CHECK_ARG(xr_wgrad64_eligible(N, H, W), "xr_conv64_wgrad: needs W in {8, 16, ..., 112} and a tensor below 2 GiB (use xr_conv_wgrad)");
  WG64P p{};
  p.x = (const bf16_t*)in; p.dy = (const bf16_t*)dy; p.slabs = slabs;
  p.n_scale = in_scale; p.n_shift = in_shift; p.n_alpha = in_alpha;
  p.N = N; p.H = H; p.W = W;
  p.rows_total = N * H;
  p.io_bytes = (unsigned)((long long)N * H * W * 128);
  int grid = cu_count() - (int)g_tune[17];   // knob 17: see xr_conv_wgrad_rows
  if (grid < 8) grid = 8;
  if (grid > max_slabs) grid = max_slabs;
  if (grid > p.rows_total) grid = p.rows_total;
  p.rows_per_wg = cdiv(p.rows_total, grid);
  grid = cdiv(p.rows_total, p.rows_per_wg);
  hipStream_t st = (hipStream_t)stream;
  const bool xf = in_scale != nullptr;
  const int nch = cdiv(W, 16);
  if (nch <= 2) return xf ? launch_wgrad64<2, true>(p, grid, st) : launch_wgrad64<2, false>(p, grid, st);
  if (nch <= 4) return xf ? launch_wgrad64<4, true>(p, grid, st) : launch_wgrad64<4, false>(p, grid, st);
  return xf ? launch_wgrad64<7, true>(p, grid, st) : launch_wgrad64<7, false>(p, grid, st);
}
