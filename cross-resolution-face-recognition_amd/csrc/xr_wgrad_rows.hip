// xr_wgrad_rows.hip -- direct weight gradient of the 3x3, pad-1, stride-1 / stride-2 bf16 convolutions with C % 64 == 0 and
// K % 64 == 0 (every body layer of IR-50 / IR-SE-50 / ResNet-34, model_irse.py:56-62, model/resnet.py:24-47):
//     dW[k][tap][c] = sum over output pixels p of dY[p][k] * X[stride * p + tap - 1][c].
// The generalisation of xr_wgrad64.hip (same arithmetic core: pixel-major swizzled LDS images, ds_read_b64_tr_b16 fragments,
// 12 waves = 3 tap rows x 4 quadrants of a 64 x 64 (k, c) tile, the three taps of a tap row in 48 accumulator registers):
//   * a workgroup owns ONE 64 x 64 tile of (output channel, input channel) and a contiguous run of output rows; it reads only
//     the 128-byte channel slices of dY and X that its tile needs (a 64-channel slice of a 256-channel pixel is one cache line),
//     each once: 288 FLOP per byte staged into LDS, against 64 for the 128 x 128 tiles of the sliced implicit GEMM
//     (wgrad_kernel, xr_conv.hip), which also pulls X nine times through the im2col gather;
//   * a step covers RS output rows (RS x NCH x 3 MFMAs per wave between two barriers; the 14 x 14 layers take a whole image per
//     step); input rows live in a ring of 2 * S * RS + 3 - S row slots, dY rows in a ring of 2 * RS, the loads of step t + 1 fly
//     under the MFMAs of step t;
//   * stride 2 (the first 3x3 convolution of a down-sampling block runs at the INPUT resolution, model_irse.py:60): an input row is
//     de-interleaved into its even and odd columns while it is staged, so that every tap is again a contiguous shifted view
//     (tap column 0 / 2 -> odd plane at output column - 1 / + 0, tap column 1 -> even plane); tap rows read input rows
//     2 * ho - 1 .. 2 * ho + 1;
//   * top / bottom image edges: a tap row that falls outside the image is skipped for that output row (wave-uniform), never
//     multiplied by zeros; left / right edges are zero border pixels of the LDS planes;
//   * output: one fp32 slab [K][9 * C] per pixel split in the packed layout of xr_conv_wgrad, each workgroup filling its
//     64 x (9 x 64) sub-block; xr_unpack_wgrad sums the splits.
#include "xr_common.h"
#include <type_traits>

namespace {

constexpr int NT = 768;
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define XRR_OOR 0x80000000u

struct WGRP {
  const bf16_t* x;        // [N][H][W][C]
  const bf16_t* dy;       // [N][Ho][Wo][K]
  float* slabs;           // [nsplit][K][9 * C]
  int N, H, W, C, Ho, Wo, K;
  int rows_total;         // N * Ho output rows
  int rows_per_wg;        // multiple of RS
  int tiles_c, tiles;     // C / 64, (K / 64) * (C / 64)
  unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) << 2; }

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * 128));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ int frag_off(int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, pp = i & 3, cgrp = g & 1, hh = g >> 1;
  const int row = row0 + 8 * hh + q;
  const int cb = (col0 + 16 * cgrp + 4 * pp) * 2;
  return row * 128 + ((((cb >> 4) ^ swz(row)) << 4) | (cb & 15));
}

constexpr int r64(int v) { return (v + 63) / 64 * 64; }

// NCH: 16-pixel chunks per OUTPUT row (Wo <= 16 * NCH); RS: output rows per step; S: stride
template <int NCH, int RS, int S>
__global__ __launch_bounds__(NT, 3) void wgrad_rows_kernel(WGRP p) {
  constexpr int WP = 16 * NCH;
  constexpr int PLANE = (WP + 2) * 128;             // one column-parity plane of an input row: border pixel + WP + 1 pixels
  constexpr int XS = S * PLANE, YS = WP * 128;      // slot strides
  constexpr int NXR = 2 * S * RS + 3 - S, NYR = 2 * RS;
  constexpr int YBASE = NXR * XS;
  constexpr int SMEM = NXR * XS + NYR * YS;
  constexpr int SRS = S * RS;                       // new input rows per step
  constexpr int C0 = SRS - S + 3;                   // input rows one step needs
  constexpr int NP = (C0 + SRS - 1) / SRS;          // prologue stages
  constexpr int NSTG = (r64(SRS * S * WP * 8) + r64(RS * WP * 8) + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int dh = wave / 4, quad = wave & 3, kh = quad >> 1, ch = quad & 1;     // tap row 0..2: input row S * ho - 1 + dh

  const int tile = blockIdx.x % p.tiles, split = blockIdx.x / p.tiles;
  const int k0 = (tile / p.tiles_c) * 64, c0 = (tile % p.tiles_c) * 64;
  const int r0 = split * p.rows_per_wg;
  int r1 = r0 + p.rows_per_wg;
  if (r1 > p.rows_total) r1 = p.rows_total;
  if (r0 >= r1) return;

  for (int o = t * 16; o < SMEM; o += NT * 16) *reinterpret_cast<v4u_t*>(smem + o) = v4u_t{0u, 0u, 0u, 0u};

  // ---- staging: chunk q = t + NT * i of a stage = S * RS input rows (X chunks first, padded to a multiple of 64 so that a wave's
  // group is all-X or all-dY: descriptor and slot base are scalar selects) followed by RS dY rows
  const int cc = t & 7;
  const int nxq = SRS * p.W * 8, nyq = RS * p.Wo * 8;
  const int XQ = r64(nxq), YQ = r64(nyq);
  const int xrow_bytes = p.W * p.C * 2, yrow_bytes = p.Wo * p.K * 2;
  int gof[NSTG], lof[NSTG], jrow[NSTG];
  int kind[NSTG];               // wave-uniform: 0 X, 1 dY, 2 nothing
  bool live[NSTG];              // per lane: the chunk exists
  __amdgpu_buffer_rsrc_t rs[NSTG];
#pragma unroll
  for (int i = 0; i < NSTG; ++i) {
    const int qw = wave * 64 + NT * i;
    kind[i] = qw < XQ ? 0 : (qw < XQ + YQ ? 1 : 2);
    const int q = t + NT * i - (kind[i] == 1 ? XQ : 0);
    const int pix = q >> 3;
    if (kind[i] == 0) {
      live[i] = q < nxq;
      const int jr = pix / p.W, col = pix - jr * p.W;
      const int plane = col % S, prow = col / S + 1;
      jrow[i] = jr;
      gof[i] = pix * p.C * 2 + c0 * 2 + cc * 16;
      lof[i] = plane * PLANE + prow * 128 + ((cc ^ swz(prow)) << 4);
    } else {
      live[i] = kind[i] == 1 && q < nyq;
      const int jr = pix / p.Wo, col = pix - jr * p.Wo;
      jrow[i] = jr;
      gof[i] = pix * p.K * 2 + k0 * 2 + cc * 16;
      lof[i] = col * 128 + ((cc ^ swz(col)) << 4);
    }
    rs[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kind[i] == 0 ? p.x : p.dy), 0, kind[i] == 0 ? p.x_bytes : p.y_bytes,
                                              0x00020000);
  }
  const int ri_base = S * r0 - 1;            // flattened input row of local input row 0
  const int in_rows = p.N * p.H;

  // two register sets: the loads of step t + 2 are issued while step t is multiplied and step t + 1's set waits to be written
  v4u_t st0[NSTG], st1[NSTG];
  bool wr0[NSTG], wr1[NSTG];
  // li0: local index of the first of the S * RS input rows to stage (may be negative in the prologue: those rows are skipped);
  // lo0: local index of the first of the RS dY rows, or a negative number for none
  auto stage_load = [&](v4u_t (&st)[NSTG], bool (&wr)[NSTG], int li0, int lo0) {
    const long long xbase = (long long)(ri_base + li0) * xrow_bytes, ybase = (long long)(r0 + lo0) * yrow_bytes;
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      bool ok = live[i];
      unsigned voff = XRR_OOR;
      if (kind[i] == 0) {
        const int li = li0 + jrow[i], ri = ri_base + li;
        ok = ok && li >= 0 && ri >= 0 && ri < in_rows;
        wr[i] = live[i] && li >= 0;
        if (ok) voff = (unsigned)(xbase + gof[i]);
      } else {
        const int ro = r0 + lo0 + jrow[i];
        ok = ok && lo0 >= 0 && ro < p.rows_total;
        wr[i] = live[i] && lo0 >= 0;
        if (ok) voff = (unsigned)(ybase + gof[i]);
      }
      st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[i], voff, 0, 0);
    }
  };
  // ypar: which half of the dY ring the RS rows go to
  auto stage_write = [&](v4u_t (&st)[NSTG], bool (&wr)[NSTG], int li0, int ypar) {
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      if (kind[i] == 2) continue;
      unsigned char* dst;
      if (kind[i] == 0) {
        int li = li0 + jrow[i];
        if (li < 0) li = 0;
        dst = smem + (li % NXR) * XS + lof[i];
      } else {
        dst = smem + YBASE + (ypar * RS + jrow[i]) * YS + lof[i];
      }
      if (wr[i]) *reinterpret_cast<v4u_t*>(dst) = st[i];
    }
  };

  const int offA = frag_off(0, kh * 32, lane);
  int offB[3];
  if constexpr (S == 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) offB[d] = frag_off(d, ch * 32, lane);        // input column wo + d - 1 lives in plane row wo + d
  } else {
    offB[0] = PLANE + frag_off(0, ch * 32, lane);                            // column 2 wo - 1: odd plane, index wo - 1 -> row wo
    offB[1] = frag_off(1, ch * 32, lane);                                    // column 2 wo    : even plane, index wo -> row wo + 1
    offB[2] = PLANE + frag_off(1, ch * 32, lane);                            // column 2 wo + 1: odd plane, index wo -> row wo + 1
  }

  f32x16_t acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  __syncthreads();   // LDS zeroed
  // prologue: local input rows 0 .. C0 - 1 in NP stages of S * RS rows ending at C0 - 1; dY rows 0 .. RS - 1 with the last one
#pragma unroll
  for (int k = NP - 1; k >= 0; --k) {
    stage_load(st0, wr0, C0 - (k + 1) * SRS, k == 0 ? 0 : -1);
    stage_write(st0, wr0, C0 - (k + 1) * SRS, 0);
  }
  __syncthreads();

  int h = r0 % p.Ho;
  int xb = 0;                                   // ring slot of local input row S * it * RS
  // one step: issue the loads of step it + 2 into (stL, wrL), multiply step it, write step it + 1's rows -- loaded one step ago
  // into (stW, wrW) -- to LDS.  With one step of lookahead (12-42 MFMAs per wave) every step waited a full memory round trip.
  auto step = [&](int r, int it, v4u_t (&stL)[NSTG], bool (&wrL)[NSTG], v4u_t (&stW)[NSTG], bool (&wrW)[NSTG]) {
    const bool more = r + RS < r1, more2 = r + 2 * RS < r1;
    const int li_next = S * it * RS + C0;       // first input row the next step needs beyond this step's
    if (more2) stage_load(stL, wrL, li_next + SRS, (it + 2) * RS);
#pragma unroll
    for (int j = 0; j < RS; ++j) {
      int ho = h + j;
      ho = ho >= p.Ho ? ho % p.Ho : ho;
      const int hi = S * ho - 1 + dh;
      if (r + j < r1 && hi >= 0 && hi < p.H) {
        int xi = xb + S * j + dh;
        if (xi >= NXR) xi -= NXR;
        const unsigned char* ys = smem + YBASE + ((it & 1) * RS + j) * YS + offA;
        const unsigned char* xs = smem + xi * XS;
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
    }
    if (more) stage_write(stW, wrW, li_next, (it + 1) & 1);
    __syncthreads();
    h += RS;
    if (h >= p.Ho) h %= p.Ho;
    xb += SRS;
    if (xb >= NXR) xb -= NXR;
  };
  if (r0 + RS < r1) stage_load(st0, wr0, C0, RS);     // step 1's rows
  int it = 0;
  for (int r = r0; r < r1; r += 2 * RS, it += 2) {
    step(r, it, st1, wr1, st0, wr0);
    if (r + RS < r1) step(r + RS, it + 1, st0, wr0, st1, wr1);
  }

  const int Kg = 9 * p.C;
  float* slab = p.slabs + (size_t)split * p.K * Kg;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int col = (dh * 3 + d) * p.C + c0 + ch * 32 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = k0 + kh * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      slab[(size_t)row * Kg + col] = acc[d][e];
    }
  }
}

template <int NCH, int RS, int S>
int launch_rows(WGRP& p, int max_slabs, int ncu, hipStream_t st) {
  constexpr int WP = 16 * NCH;
  constexpr int SMEM = (2 * S * RS + 3 - S) * S * (WP + 2) * 128 + 2 * RS * WP * 128;
  static_assert(SMEM <= 160 * 1024, "LDS plan exceeds a CU");
  auto kern = wgrad_rows_kernel<NCH, RS, S>;
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [&] {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  });
  if (err != hipSuccess) {
    xr_set_error("xr_conv_wgrad_rows: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(err));
    return XR_E_LAUNCH;
  }
  const int steps = cdiv(p.rows_total, RS);
  int nsplit = ncu / p.tiles;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > max_slabs) nsplit = max_slabs;
  if (nsplit > steps) nsplit = steps;
  p.rows_per_wg = cdiv(steps, nsplit) * RS;
  nsplit = cdiv(p.rows_total, p.rows_per_wg);
  hipLaunchKernelGGL(kern, dim3((unsigned)(nsplit * p.tiles)), dim3(NT), SMEM, st, p);
  XR_CHECK_LAUNCH("xr_conv_wgrad_rows");
  return nsplit;
}

int cu_count_rows() {
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

extern "C" int xr_conv_wgrad_rows(const void* in, const void* dy, float* slabs, int N, int H, int W, int C, int K, int stride,
                                  int max_slabs, void* stream) {
  XR_CHECK_ARG(in && dy && slabs && N > 0 && H > 0 && W > 0 && max_slabs > 0, "xr_conv_wgrad_rows: null pointer / non-positive dimension");
  XR_CHECK_ARG(C > 0 && K > 0 && C % 64 == 0 && K % 64 == 0, "xr_conv_wgrad_rows: C = %d and K = %d must be multiples of 64", C, K);
  XR_CHECK_ARG(stride == 1 || stride == 2, "xr_conv_wgrad_rows: stride 1 or 2");
  XR_CHECK_ARG(H % stride == 0 && W % stride == 0, "xr_conv_wgrad_rows: H and W must be multiples of the stride");
  const int Ho = H / stride, Wo = W / stride;
  XR_CHECK_ARG(Wo <= 112 && !(stride == 2 && Wo > 64), "xr_conv_wgrad_rows: output width %d not supported (use xr_conv_wgrad)", Wo);
  const long long xb = (long long)N * H * W * C * 2, yb = (long long)N * Ho * Wo * K * 2;
  XR_CHECK_ARG(xb < (1ll << 31) && yb < (1ll << 31), "xr_conv_wgrad_rows: tensor larger than 2 GiB (use xr_conv_wgrad)");
  WGRP p{};
  p.x = (const bf16_t*)in; p.dy = (const bf16_t*)dy; p.slabs = slabs;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K;
  p.rows_total = N * Ho;
  p.tiles_c = C / 64;
  p.tiles = (K / 64) * (C / 64);
  p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
  hipStream_t st = (hipStream_t)stream;
  // knob 17: compute units the persistent side-stream kernels leave free (their workgroups hold a CU for the whole launch; a short
  // kernel of the backward chain that arrives meanwhile otherwise waits for one of them to retire)
  int ncu = cu_count_rows() - (int)g_tune[17];
  if (ncu < 8) ncu = 8;
  if (stride == 1) {
    // rows per step: measured per plan (tools/wgrad_rows_bench.py): 14 / 7 / 3 rows beat 16 / 8 / 4 on the narrow plans, two rows per
    // step beat one on the 112-wide plan (966 vs 777 TFLOP/s at 64 -> 64, batch 256: half the barriers per MFMA)
    if (Wo <= 16) return launch_rows<1, 14, 1>(p, max_slabs, ncu, st);
    if (Wo <= 32) return launch_rows<2, 7, 1>(p, max_slabs, ncu, st);
    if (Wo <= 64) return launch_rows<4, 3, 1>(p, max_slabs, ncu, st);
    return launch_rows<7, 2, 1>(p, max_slabs, ncu, st);
  }
  if (Wo <= 16) return launch_rows<1, 6, 2>(p, max_slabs, ncu, st);
  if (Wo <= 32) return launch_rows<2, 3, 2>(p, max_slabs, ncu, st);
  return launch_rows<4, 1, 2>(p, max_slabs, ncu, st);
}
