// xr_wgrad8.hip -- 8-wave weight gradient for the wide (K >= 256) bf16 layers: dW[k][tap*C + c] = sum_pix dY[pix][k] * gather(X)[pix][tap][c]
//
// Same contract as xr_conv.hip's sliced wgrad_kernel (the pixel range is cut into slices, each slice writes its own fp32
// [K][Kg] slab, xr_unpack_wgrad sums them).  Why a second kernel: the 4-wave 128x128 kernel prefetches ONE 64-pixel stage
// ahead -- 16 MFMAs per wave, ~0.3 us, well below the L2 / HBM latency -- and drains its LDS-DMA queue (vmcnt(0)) at the barrier
// of every stage; two workgroups per CU hide part of that, MFMA busy stays at 0.31.  Here
//   * one workgroup of 8 waves owns a 128 (k) x 256 (tap*C + c) tile: wave grid 2 x 4, wave tile 64 x 64 as before, but the dY
//     image of a stage is shared by twice as many waves (768 B instead of 1024 B of LDS-DMA traffic per pixel and CU);
//   * the 64-pixel stages go through a ring of THREE LDS buffers (3 x 48 KiB): stage s+2 is issued while stage s is
//     multiplied, so an operand has two stages (~0.6-1 us) to arrive;
//   * the DMA queue is never drained inside the loop: every thread issues exactly six 16-byte pieces per stage (out-of-range
//     pieces are issued with an out-of-range offset and write zeros), s_waitcnt vmcnt(6) retires exactly the stage that is
//     read next, one raw s_barrier per stage orders the ring between the waves;
//   * fragments come from inline-asm ds_read_b64_tr_b16 (the transposing read: both images stay [pixel][channel]) so that
//     hipcc's LDS-DMA alias tracking does not put a vmcnt(0) in front of every LDS read; the reads of k-step ks+1 are in
//     flight under the MFMAs of k-step ks (counted lgkmcnt).
// The gather cursor, the chunk swizzle and the slab epilogue are those of wgrad_kernel (xr_conv.hip).
#include "xr_conv_p.h"

#include <mutex>

namespace {

constexpr int NTW = 512;
constexpr int BP = 64;                       // pixels per stage
constexpr int BR = 128, BC = 256;            // tile: k rows x (tap, c) columns
constexpr int PY = BR * 2;                   // dY image: [64 pixels][128 k], unpadded rows, 16-B chunks XOR-swizzled
constexpr int PX = 128, NSUB = BC / 64;      // X image: four sub-images [64 pixels][64 columns] (one per wave column)
constexpr int SUB = BP * PX;                 // 8 KiB
constexpr int CRY = BR / 8, CRX = PX / 16;   // 16-B chunks per image row
constexpr int NY = BP * CRY / NTW, NX = NSUB;             // DMA pieces per thread and stage: 2 + 4
constexpr int STAGE = BP * PY + NSUB * SUB;  // 48 KiB
constexpr int NRING = 3;
#define XRW_OOR 0x80000000u

typedef short s16x4w_t __attribute__((ext_vector_type(4)));
typedef short s16x8w_t __attribute__((ext_vector_type(8)));

template <int OFF>
__device__ __forceinline__ s16x4w_t lds_tr8(unsigned addr) {
  s16x4w_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// one MFMA fragment (8 pixels x 1 column per lane): two transposing reads four pixel rows apart
template <int R, int KS>
__device__ __forceinline__ bf16x8_t frag(unsigned base) {
  const s16x4w_t lo = lds_tr8<KS * 16 * R>(base);
  const s16x4w_t hi = lds_tr8<KS * 16 * R + 4 * R>(base);
  const s16x8w_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// per-lane byte offset of the fragment that starts at column col0 (a multiple of 32) of an image with pitch R: row =
// 8*h + q (+ pix0, a multiple of 16: same swizzle term), 16-B chunk XOR-swizzled as in xr_conv.hip (pitch >= 256: by
// (row & 3) << 2; pitch 128: by ((row >> 1) & 1) << 2)
template <int R>
__device__ __forceinline__ unsigned frag_off(int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, pp = i & 3;
  const int cgrp = g & 1, h = g >> 1;
  const int row = 8 * h + q;
  const int cb = (col0 + 16 * cgrp + 4 * pp) * 2;
  const int swz = R >= 256 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;   // wg_swz<R> of xr_conv.hip
  return (unsigned)(row * R + ((((cb >> 4) ^ swz) << 4) | (cb & 15)));
}

template <int N>
__device__ __forceinline__ void vm_le() {
  static_assert(N == 0 || N == NY + NX, "unexpected DMA count");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}
static_assert(NY + NX == 6, "vm_le<> spells the count");

__global__ __launch_bounds__(NTW, 1) void wgrad8_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware order (as wgrad_kernel): the tiles of one pixel slice re-read the same dY / X rows and meet in one L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int slice = bid / p.tiles_all, tile = bid - slice * p.tiles_all;
  const int tile_c = tile % p.tiles_c, tile_r = tile / p.tiles_c;
  const int r0 = tile_r * BR, c0 = tile_c * BC;

  // ---- DMA slots: the thread's LDS slot is fixed, it fetches the source chunk that the swizzle maps into it.
  // X: wave w stages pixel rows 8w..8w+7 of all four sub-images, so the four pieces of a lane are the SAME pixel and tap
  // (the tile's 256 columns are one tap: C % 256 == 0), 64 channels apart: ONE gather cursor per lane.
  const int xrow0 = t / CRX, yrow0 = t / CRY;
  const int xch = (t % CRX) ^ (((xrow0 >> 1) & 1) << 2);
  const int ych = (t % CRY) ^ ((yrow0 & 3) << 2);
  const int tap = c0 / p.C;
  const int cch = c0 - tap * p.C + xch * 8;
  const int tr_ = tap / p.S, ts_ = tap - tr_ * p.S;
  const bool col_ok = tap < p.R * p.S;
  const int adv_w = p.c64 * p.stride, adv_h = p.b64 * p.stride, span_w = p.Wo * p.stride, span_h = p.Ho * p.stride;
  const int wlim = span_w + ts_ - p.pad, hlim = span_h + tr_ - p.pad;
  const bool ycol_ok = (r0 + ych * 8) < p.ldy;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
  unsigned y_base[NY];
#pragma unroll
  for (int i = 0; i < NY; ++i) y_base[i] = ycol_ok ? (unsigned)(((yrow0 + (NTW / CRY) * i) * p.ldy + r0 + ych * 8) * 2) : XRW_OOR;

  const int s_begin = slice * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > p.steps_total) s_end = p.steps_total;
  if (s_begin >= s_end) return;   // (whole workgroup: the slice is uniform)

  // gather cursor of the lane's pixel: byte offset of the tap's input pixel, its input row / column for the bounds test; a
  // 64-pixel advance is three adds plus two wrap corrections with launch-uniform deltas
  int xl, xhv, xwv;
  {
    const int m = s_begin * BP + xrow0;
    const int n = fdiv(p.fd_howo, m);
    const int rem = m - n * (int)p.fd_howo.d;
    const int ho = fdiv(p.fd_wo, rem);
    const int wo = rem - ho * (int)p.fd_wo.d;
    xhv = ho * p.stride - p.pad + tr_;
    xwv = wo * p.stride - p.pad + ts_;
    xl = (((n * p.H + xhv) * p.W + xwv) * p.C + cch) * 2;
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // DMA of one stage, in three parts (2 pieces each) so that the address arithmetic of a part runs in the shadow of the MFMA
  // group issued just before it.  Always NY + NX pieces per stage; past the slice end they are out of range (zeros).
  auto dma_y = [&](int step, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned char* sY = smem + buf * STAGE;
    const int mbase = step * BP;
    const bool live = step < s_end;
    const unsigned ysoff = live ? (unsigned)mbase * (unsigned)(p.ldy * 2) : 0u;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int m = mbase + yrow0 + (NTW / CRY) * i;
      const unsigned voff = (live && m < p.M) ? y_base[i] : XRW_OOR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr_t)(sY + (wv * (64 / CRY) + (NTW / CRY) * i) * PY), 16, voff, ysoff, 0, 0);
    }
#else
    (void)step; (void)buf;
#endif
  };
  unsigned x_voff = XRW_OOR;
  auto dma_x_addr = [&](int step) {   // validity + offset of the lane's pixel for stage `step`, then advance the cursor
    const bool ok = step < s_end && col_ok && step * BP + xrow0 < p.M && (unsigned)xhv < (unsigned)p.H && (unsigned)xwv < (unsigned)p.W;
    x_voff = ok ? (unsigned)xl : XRW_OOR;
    int w2 = xwv + adv_w, h2 = xhv + adv_h, l2 = xl + p.d64;
    if (w2 >= wlim) { w2 -= span_w; h2 += p.stride; l2 += p.dwrap_w; }
    if (h2 >= hlim) { h2 -= span_h; l2 += p.dwrap_h; }
    xwv = w2; xhv = h2; xl = l2;
  };
  auto dma_x = [&](int buf, int i0) {   // X pieces i0, i0 + 1 (sub-images i0, i0 + 1: 64 channels = 128 bytes apart)
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned char* sX = smem + buf * STAGE + BP * PY + wv * 1024;
    // the 64-channel step goes into the scalar offset: an instruction offset would also move the LDS destination, and the
    // range check (which supplies the zero padding) looks at the vector offset only
    if (i0 == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(sX + 0 * SUB), 16, x_voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(sX + 1 * SUB), 16, x_voff, 128, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(sX + 2 * SUB), 16, x_voff, 256, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(sX + 3 * SUB), 16, x_voff, 384, 0, 0);
    }
#else
    (void)buf; (void)i0;
#endif
  };
  auto dma_stage = [&](int step, int buf) {
    dma_y(step, buf);
    dma_x_addr(step);
    dma_x(buf, 0);
    dma_x(buf, 2);
  };
  static_assert(NX == 4 && NY == 2, "three parts of two pieces");

  // ---- fragment addresses: wave (wr, wc) owns rows wr*64.. of the dY image and columns wc*64.. of the X image
  const int wr0 = (wave >> 2) * 64, wc0 = (wave & 3) * 64;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned ya[2], xa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ya[i] = lds0 + frag_off<PY>(wr0 + i * 32, lane);
    xa[i] = lds0 + (unsigned)(BP * PY + (wave & 3) * SUB) + frag_off<PX>(i * 32, lane);
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define XRW_RD(KS, SET)                                    \
  fa[SET][0] = frag<PY, KS>(ya[0] + boff);                 \
  fa[SET][1] = frag<PY, KS>(ya[1] + boff);                 \
  fb[SET][0] = frag<PX, KS>(xa[0] + boff);                 \
  fb[SET][1] = frag<PX, KS>(xa[1] + boff);
#define XRW_MMA(SET, PENDING)                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #PENDING ")" ::: "memory");                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  __builtin_amdgcn_s_setprio(1);                                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                       \
  __builtin_amdgcn_sched_barrier(0);

  // ---- ring: stage s lives in buffer (s - s_begin) % 3
  dma_stage(s_begin, 0);
  dma_stage(s_begin + 1, 1);
  int buf = 0;
  for (int step = s_begin; step < s_end; ++step) {
    vm_le<NY + NX>();                       // this thread's pieces of stage `step` have landed (those of step + 1 may be in flight)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();           // ... everybody's have, and everybody is done reading stage step - 1
    __builtin_amdgcn_sched_barrier(0);
    int nb = buf + 2;
    if (nb >= NRING) nb -= NRING;
    const unsigned boff = (unsigned)(buf * STAGE);
    bf16x8_t fa[2][2], fb[2][2];
    XRW_RD(0, 0)
    XRW_RD(1, 1)
    XRW_MMA(0, 8)
    dma_y(step + 2, nb);                    // stage step + 2 goes into the buffer stage step - 1 was read from; each part's
    __builtin_amdgcn_sched_barrier(0);      // address arithmetic runs while the four MFMAs just issued execute
    XRW_RD(2, 0)
    XRW_MMA(1, 8)
    dma_x_addr(step + 2);
    dma_x(nb, 0);
    __builtin_amdgcn_sched_barrier(0);
    XRW_RD(3, 1)
    XRW_MMA(0, 8)
    dma_x(nb, 2);
    __builtin_amdgcn_sched_barrier(0);
    XRW_MMA(1, 0)
    if (++buf == NRING) buf = 0;
  }
  vm_le<0>();   // the trailing out-of-range pieces still write zeros into the ring

  // each slice owns a private [K][Kg] slab: plain coalesced stores (128 B per accumulator row)
  const int lr = lane & 31, lh = lane >> 5;
  float* slab = p.dwp + (size_t)slice * p.K * p.Kg;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = c0 + wc0 + j * 32 + lr;
      if (col >= p.Kg) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = r0 + wr0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < p.K) slab[(size_t)row * p.Kg + col] = acc[i][j][e];
      }
    }
}

}  // namespace

bool xr_wgrad8_eligible(const WgradP& p, int transposed) {
  const int knob = g_tune[13];
  if (knob == 0 || transposed) return false;
  const long long in_bytes = (long long)p.N * p.H * p.W * p.C * 2, dy_bytes = (long long)p.M * p.ldy * 2;
  if (in_bytes >= (1ll << 31) || dy_bytes >= (1ll << 31) || g_tune[2] != 0) return false;
  if (p.C % 256 != 0 || p.Kg % 256 != 0) return false;   // a 256-column tile is one tap (one gather cursor per lane)
  if (knob == 2) return p.K > 64;              // forced (tests / tuning)
  return p.K >= 256 && p.K % 128 == 0;   // measured (tools/wgrad8_bench.py): 7-15 % faster there, slower on the 128-channel layers
}

int xr_wgrad8_launch(WgradP& p, int split, hipStream_t st) {
  p.in_bytes = (unsigned)((long long)p.N * p.H * p.W * p.C * 2);
  p.dy_bytes = (unsigned)((long long)p.M * p.ldy * 2);
  const int howo = p.Ho * p.Wo;
  p.a64 = 64 / howo;
  p.b64 = (64 % howo) / p.Wo;
  p.c64 = (64 % howo) % p.Wo;
  p.d64 = (int)(((long long)p.a64 * p.H * p.W + (long long)p.b64 * p.stride * p.W + (long long)p.c64 * p.stride) * p.C * 2);
  p.dwrap_w = (int)(((long long)p.stride * p.W - (long long)p.Wo * p.stride) * p.C * 2);
  p.dwrap_h = (int)(((long long)p.H * p.W - (long long)p.Ho * p.stride * p.W) * p.C * 2);
  p.fd_howo = make_fd((unsigned)howo);
  p.fd_wo = make_fd((unsigned)p.Wo);
  p.tiles_c = cdiv(p.Kg, BC);
  const int tiles_r = cdiv(p.K, BR);
  p.steps_total = cdiv(p.M, BP);
  if (split < 1) split = 1;
  if (split > p.steps_total) split = p.steps_total;
  p.steps_per_split = cdiv(p.steps_total, split);
  split = cdiv(p.steps_total, p.steps_per_split);
  p.tiles_all = p.tiles_c * tiles_r;
  constexpr int smem = NRING * STAGE;
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  if (attr_err != hipSuccess) {
    xr_set_error("xr_conv_wgrad(8-wave): hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(attr_err));
    return XR_E_LAUNCH;
  }
  hipLaunchKernelGGL(wgrad8_kernel, dim3((unsigned)(p.tiles_all * split)), dim3(NTW), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv_wgrad(8-wave)");
  return split;
}
