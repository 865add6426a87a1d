// xr_conv64.hip -- weights-stationary direct 3x3 convolution for the 64 -> 64 channel, stride-1, pad-1 bf16 layers
// (every FSRNet body layer: model/FSRnet.py:79,85 at 112x112 and 28x28; IR / ResNet stage 1: model_irse.py:59,
// model/resnet.py:9-12 at 112x112 / 56x56).  Forward and input gradient (the same kernel with mirrored tap offsets).
//
// Why not the implicit GEMM of xr_conv.hip: with C = K = 64 a K-loop has only nine stages, the im2col gather re-reads the
// input nine times through the texture path and the 128x64 tiles are instruction-issue bound (~13 VALU per MFMA).  Here
//   * the 64 x 576 weight panel lives in REGISTERS as MFMA A-operand fragments for the whole kernel (persistent workgroups);
//   * a workgroup walks a contiguous run of 16x16-pixel output tiles; the 18x18-pixel input tile with halo is staged ONCE
//     into LDS (global -> registers -> LDS, double buffered: the loads of tile t+1 are in flight under the MFMAs of tile t)
//     and all nine taps are read from LDS with ds_read_b128;
//   * staging goes through registers because it can TRANSFORM: y = prelu(x * scale[n][c] + shift[n][c], alpha[c]) -- the
//     InstanceNorm apply + PReLU of the producing layer (model/FSRnet.py:81-84) is folded into the consumer's load, once
//     per input element (an im2col gather would pay it nine times), so the normalised activation never exists in HBM;
//   * MFMA is v_mfma_f32_32x32x16_bf16 with the 32 "rows" = 2 image rows x 16 columns: the LDS image is pixel-major
//     (128 B per pixel) with the 16-B chunk XOR-swizzled by (halo column >> 1) & 7 -- conflict-free ds_read_b128 for every
//     tap shift (the 16-lane groups of a b128 read then see each 16-B slot of the 256-B bank row once);
//   * accumulators are kept transposed (D[channel][pixel]): a lane owns 4 consecutive channels of a pixel, the epilogue
//     writes 8-byte packets into an LDS image and streams full 128-B pixel rows out;
//   * epilogue fusions: bias; per-image sum / sum of squares of the (rounded) output for the InstanceNorm that follows
//     (accumulated in registers across the tiles of one image, one atomic per channel and image change); residual-gradient
//     sum (out += ep_add).
#include "xr_common.h"

extern int g_tune[16];

namespace {

constexpr int NT = 256;
constexpr int TS = 16;                    // output tile edge
constexpr int HS = TS + 2;                // halo tile edge
constexpr int HPIX = HS * HS;             // 324 halo pixels
constexpr int INBUF = HPIX * 128;         // one input stage (bytes)
constexpr int NCH = (HPIX * 8 + NT - 1) / NT;  // 16-B chunks a thread stages per tile (11)
constexpr int OPITCH = 144;               // epilogue image pitch (bytes): 16-B aligned, rows 4 banks apart
static_assert(TS * TS * OPITCH <= INBUF, "the output image reuses an input stage");

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define XR64_OOR 0x80000000u

struct DC64P {
  const bf16_t* in;
  const bf16_t* w;        // [64][576] bf16: row = GEMM output channel, column = tap * 64 + reduction channel
  const float* bias;      // [64] or null
  bf16_t* out;
  const bf16_t* ep_add;   // laid out like out, or null
  const float* n_scale;   // [N][64] per-image affine applied to the input on load, or null
  const float* n_shift;
  const float* n_alpha;   // [64] PReLU slope applied after the affine, or null (no activation)
  float* stats;           // [2][N][64]: sum, sum of squares of the output per image and channel, or null
  int N, H, W, tiles_x, tiles_img, ntiles, tpw;
  unsigned in_bytes;
  int dbg;   // tuning knob 14 (bit 0: skip staging after the first tile, bit 1: skip the accumulator -> LDS epilogue, bit 2: skip the
             // store phase, bit 3: skip the MFMA stages) -- timing experiments only, results are wrong
};

// CBW = 32-channel blocks per wave: 2 -> a wave owns 2 pixel blocks x all 64 channels (288 weight VGPRs, 72 fragment reads
// per tile), 1 -> 4 pixel blocks x 32 channels (144 weight VGPRs, 144 fragment reads per tile)
template <bool TR, bool NORM, int CBW>
__global__ __launch_bounds__(NT, 1) void dconv64_kernel(DC64P p) {
  constexpr int WN = 2 / CBW, WM = 4 / WN, NPB = 8 / WM;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lp = lane & 31, kg = lane >> 5;

  int tile = blockIdx.x * p.tpw;
  int tile_end = tile + p.tpw;
  if (tile_end > p.ntiles) tile_end = p.ntiles;
  if (tile >= tile_end) return;

  // ---- weight panel -> registers (MFMA A operand: row = channel lp of the block, 8 reduction elements 8*kg..+7 of the step)
  bf16x8_t wf[9][4][CBW];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < CBW; ++j) {
        const int row = (wn * CBW + j) * 32 + lp;
        wf[tp][ks][j] = *reinterpret_cast<const bf16x8_t*>(p.w + (size_t)row * 576 + tp * 64 + ks * 16 + kg * 8);
      }

  // ---- staging geometry of this thread: chunk column cc is fixed (NT % 8 == 0), halo pixels hp = (t >> 3) + 32 i
  const int cc = t & 7;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in), 0, p.in_bytes, 0x00020000);
  float sc[8], sh[8], al[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = 1.f, sh[e] = 0.f, al[e] = 1.f;
  if (NORM && p.n_alpha != nullptr) ld8(p.n_alpha + cc * 8, al);
  int n_staged = -1;

  v4u_t st[NCH];
  unsigned stv = 0;   // bit i: chunk i lies inside the image (zero padding otherwise -- also AFTER the affine)
  auto tile_coords = [&](int tl, int& n, int& y0, int& x0) {
    n = tl / p.tiles_img;
    const int r = tl - n * p.tiles_img;
    const int ty = r / p.tiles_x;
    y0 = ty * TS;
    x0 = (r - ty * p.tiles_x) * TS;
  };
  auto issue_loads = [&](int tl) {
    int n, y0, x0;
    tile_coords(tl, n, y0, x0);
    if (NORM && n != n_staged) {
      ld8(p.n_scale + (size_t)n * 64 + cc * 8, sc);
      ld8(p.n_shift + (size_t)n * 64 + cc * 8, sh);
      n_staged = n;
    }
    stv = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int hp = (t >> 3) + 32 * i;
      const int hy = (hp * 57) >> 10, hx = hp - hy * HS;   // hp / 18 for hp < 324 (+ the tail rows >= 324: masked below)
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = hp < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * 64 + cc * 8) * 2) : XR64_OOR;
      st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
      if (ok) stv |= 1u << i;
    }
  };
  auto write_stage = [&](int buf) {
    unsigned char* dst = smem + buf * INBUF;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int hp = (t >> 3) + 32 * i;
      if (hp >= HPIX) continue;
      const int hy = (hp * 57) >> 10, hx = hp - hy * HS;
      v4u_t v = st[i];
      if constexpr (NORM) {
        const bool ok = (stv >> i) & 1u;
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = __uint_as_float(v[q] << 16), b = __uint_as_float(v[q] & 0xFFFF0000u);
          a = a * sc[2 * q] + sh[2 * q];
          b = b * sc[2 * q + 1] + sh[2 * q + 1];
          a = a > 0.f ? a : a * al[2 * q];
          b = b > 0.f ? b : b * al[2 * q + 1];
          o[q] = ok ? pack2bf(a, b) : 0u;
        }
        v = v4u_t{o[0], o[1], o[2], o[3]};
      }
      *reinterpret_cast<v4u_t*>(dst + hp * 128 + ((cc ^ ((hx >> 1) & 7)) << 4)) = v;
    }
  };

  // ---- fragment read addressing: lane -> pixel (row lp >> 4 of the block's two image rows, column lp & 15)
  const int lrow = lp >> 4, lcol = lp & 15;
  unsigned xoff[3][4];   // [tap column s'][k-step]: byte offset of the lane's chunk inside its pixel row, plus s' pixels
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      xoff[s][ks] = (unsigned)((lrow * HS + lcol + s) * 128 + (((2 * ks + kg) ^ (((lcol + s) >> 1) & 7)) << 4));

  f32x16_t acc[NPB][CBW];
  float bs[8], bss[8];   // per-image statistics of this thread's channel chunk (store phase: chunk column cc)
#pragma unroll
  for (int e = 0; e < 8; ++e) bs[e] = 0.f, bss[e] = 0.f;
  int n_stats = -1;
  auto flush_stats = [&]() {
    // fold the 32 threads that share a chunk column (lanes 8 apart, then the 4 waves through global atomics: rare)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = bs[e], b = bss[e];
      a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 8, 64); b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (lane < 8 && n_stats >= 0) {
        atomicAdd(p.stats + (size_t)n_stats * 64 + cc * 8 + e, a);
        atomicAdd(p.stats + ((size_t)p.N + n_stats) * 64 + cc * 8 + e, b);
      }
      bs[e] = 0.f;
      bss[e] = 0.f;
    }
  };

  // prologue: first tile into stage 0
  issue_loads(tile);
  write_stage(0);
  __syncthreads();

  for (int it = 0; tile < tile_end; ++tile, ++it) {
    const int buf = it & 1;
    const bool more = tile + 1 < tile_end;
    if (more && !(p.dbg & 1)) issue_loads(tile + 1);   // in flight under the MFMAs below

    // ---- 9 taps x 4 k-steps: NPB pixel fragments from LDS against the register-resident weight fragments
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
      for (int j = 0; j < CBW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const unsigned char* src = smem + buf * INBUF;
    // software pipeline over the 36 (tap, k-step) stages: the fragments of stage g+1 are read while the MFMAs of stage g
    // run; sched_barriers keep hipcc from hoisting all 36 stages' reads to the top (it would: 1 wave / SIMD "has" 512 VGPRs)
    bf16x8_t xf[2][NPB];
    auto read_stage = [&](int g, bf16x8_t (&dst)[NPB]) {
      const int tp = g >> 2, ks = g & 3;
      const int r = TR ? 2 - tp / 3 : tp / 3, s = TR ? 2 - tp % 3 : tp % 3;
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int prow = (wm * NPB + i) * 2 + r;   // halo row of the block's first image row for this tap
        dst[i] = *reinterpret_cast<const bf16x8_t*>(src + prow * HS * 128 + xoff[s][ks]);
      }
    };
    read_stage(0, xf[0]);
    if (!(p.dbg & 8))
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      if (g + 1 < 36) read_stage(g + 1, xf[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int j = 0; j < CBW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g >> 2][g & 3][j], xf[g & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every wave is done reading stage `buf`

    // ---- accumulators (+bias) -> [256 pixels][64 channels] bf16 image in stage `buf`
    unsigned char* img = smem + buf * INBUF;
    if (!(p.dbg & 2))
#pragma unroll
    for (int j = 0; j < CBW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch0 = (wn * CBW + j) * 32 + 8 * q + 4 * kg;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch0);
          bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
          const int prow = (wm * NPB + i) * 32 + lp;
          uint2 pk;
          pk.x = pack2bf(acc[i][j][4 * q] + bv[0], acc[i][j][4 * q + 1] + bv[1]);
          pk.y = pack2bf(acc[i][j][4 * q + 2] + bv[2], acc[i][j][4 * q + 3] + bv[3]);
          *reinterpret_cast<uint2*>(img + prow * OPITCH + ch0 * 2) = pk;
        }
      }
    // next tile: registers -> the other stage (its last readers finished before the barrier above)
    if (more && !(p.dbg & 1)) write_stage(buf ^ 1);
    __syncthreads();

    // ---- stream the image out: thread = (chunk column cc, pixel rows (t >> 3) + 32 i)
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    if (p.stats != nullptr && n != n_stats) {
      if (n_stats >= 0) flush_stats();
      n_stats = n;
    }
    if (!(p.dbg & 4))
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pr = (t >> 3) + 32 * i;
      const int gy = y0 + (pr >> 4), gx = x0 + (pr & 15);
      if (gy >= p.H || gx >= p.W) continue;
      const size_t go = (((size_t)n * p.H + gy) * p.W + gx) * 64 + cc * 8;
      const uint4 u = *reinterpret_cast<const uint4*>(img + pr * OPITCH + cc * 16);
      if (p.ep_add != nullptr) {
        float d[8], av[8];
        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[2 * q] = __uint_as_float(w4[q] << 16);
          d[2 * q + 1] = __uint_as_float(w4[q] & 0xFFFF0000u);
        }
        ld8(p.ep_add + go, av);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] += av[e];
        st8(p.out + go, d);
      } else {
        *reinterpret_cast<uint4*>(p.out + go) = u;
        if (p.stats != nullptr) {
          const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = __uint_as_float(w4[q] << 16), b = __uint_as_float(w4[q] & 0xFFFF0000u);
            bs[2 * q] += a; bss[2 * q] += a * a;
            bs[2 * q + 1] += b; bss[2 * q + 1] += b * b;
          }
        }
      }
    }
  }
  if (p.stats != nullptr) flush_stats();
}

template <bool TR, bool NORM, int CBW>
int launch_dconv64(DC64P& p, int grid, hipStream_t st) {
  auto kern = dconv64_kernel<TR, NORM, CBW>;
  constexpr int smem = 2 * INBUF;
  static bool attr_done = false;   // idempotent attribute; a racing second call only repeats it
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      xr_set_error("xr_conv64_direct: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
      return XR_E_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv64_direct");
  return XR_OK;
}

}  // namespace

extern "C" int xr_conv64_direct(const void* in, const void* wpack, const float* bias, void* out, int N, int H, int W,
                                int transposed, const float* in_scale, const float* in_shift, const float* in_alpha,
                                float* out_stats, const void* ep_add, void* stream) {
  XR_CHECK_ARG(in && wpack && out && N > 0 && H > 0 && W > 0, "xr_conv64_direct: null pointer / non-positive dimension");
  XR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "xr_conv64_direct: in_scale and in_shift come together");
  XR_CHECK_ARG(in_alpha == nullptr || in_scale != nullptr, "xr_conv64_direct: in_alpha needs in_scale / in_shift");
  XR_CHECK_ARG(out_stats == nullptr || ep_add == nullptr, "xr_conv64_direct: output statistics and ep_add are exclusive");
  const long long in_bytes = (long long)N * H * W * 64 * 2;
  XR_CHECK_ARG(in_bytes < (1ll << 31), "xr_conv64_direct: input larger than 2 GiB (use xr_conv_igemm)");
  DC64P p{};
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)wpack; p.bias = bias; p.out = (bf16_t*)out; p.ep_add = (const bf16_t*)ep_add;
  p.n_scale = in_scale; p.n_shift = in_shift; p.n_alpha = in_alpha; p.stats = out_stats;
  p.N = N; p.H = H; p.W = W;
  p.tiles_x = cdiv(W, TS);
  p.tiles_img = p.tiles_x * cdiv(H, TS);
  p.ntiles = p.tiles_img * N;
  p.in_bytes = (unsigned)in_bytes;
  p.dbg = g_tune[14];
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  // one persistent workgroup per CU; contiguous tile runs (the tiles of an image stay together: halo rows meet in L2 and
  // the statistics of an image are flushed once)
  p.tpw = cdiv(p.ntiles, cus);
  const int grid = cdiv(p.ntiles, p.tpw);
  hipStream_t st = (hipStream_t)stream;
  const bool norm = in_scale != nullptr;
  // knob 13: 0 (default) = a wave holds half of the weight panel (144 VGPRs, no spills); 1 = the whole panel (288 VGPRs: fewer
  // LDS fragment reads, but hipcc spills ~20-50 registers to scratch at 512)
  const bool wide = g_tune[13] == 1;
  if (transposed) {
    if (norm) return wide ? launch_dconv64<true, true, 2>(p, grid, st) : launch_dconv64<true, true, 1>(p, grid, st);
    return wide ? launch_dconv64<true, false, 2>(p, grid, st) : launch_dconv64<true, false, 1>(p, grid, st);
  }
  if (norm) return wide ? launch_dconv64<false, true, 2>(p, grid, st) : launch_dconv64<false, true, 1>(p, grid, st);
  return wide ? launch_dconv64<false, false, 2>(p, grid, st) : launch_dconv64<false, false, 1>(p, grid, st);
}
