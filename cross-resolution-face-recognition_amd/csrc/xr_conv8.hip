// xr_conv8.hip -- 8-wave, 256x256x64-tile implicit-GEMM convolution for the wide (>= 256 output channels) bf16 layers.
//
//   out[m][k] = sum_{tap,c} gather(in)[m][tap][c] * wpack[k][tap*C + c]      (same contract as xr_conv.hip's FAST path)
//
// Why a second kernel: the 4-wave 128x128 kernel re-stages operands every 16 MFMAs per wave and drains the LDS-DMA
// queue (vmcnt(0)) at both barriers of every K-step.  Here a workgroup owns a 256x256 output tile with 8 waves in a
// 2(M) x 4(N) grid, each wave a 128x64 sub-tile = 128 accumulator VGPRs, two waves per SIMD:
//   * the two wave rows run one barrier apart ("ping-pong"): while one row issues its 16 v_mfma_f32_16x16x32_bf16 of a
//     phase at raised priority, the other row of the same SIMDs issues LDS fragment reads and the LDS-DMA prefetch;
//   * a K-tile (64 reduction elements) is 4 phases = the 4 (64-row x 32-col) quadrants of the wave tile; each phase
//     reads only the fragments it needs (12 / 4 / 8 / 4 ds_read_b128) and issues one quarter (16 KiB) of the NEXT
//     K-tile's operands by LDS-DMA into the other LDS buffer;
//   * the DMA queue is never drained inside the loop: counted s_waitcnt vmcnt(4) / vmcnt(2) retire exactly the quarter
//     tiles that are read one phase later, raw s_barrier (no fence) orders them between the waves;
//   * fragments are read with inline-asm ds_read_b128 so that hipcc's LDS-DMA alias tracking does not put a vmcnt(0) in
//     front of every LDS read.
// LDS: 2 buffers x (A 256 rows + B 256 rows) x 128 B = 128 KiB, rows XOR-swizzled by (row>>1)&7 on the 16-B chunk (the
// swizzle is applied on the DMA *source* chunk and on the read address; the LDS image itself is lane-linear).
// Accumulators are kept transposed (channels in registers, pixels across lanes: mfma(B_frag, A_frag)), so that a lane
// owns 4 consecutive output channels of one pixel and the epilogue writes 8-byte packets into a wave-private 128x64 LDS
// image which is then streamed out as full 128-B row segments (bias, fused PReLU-backward as in xr_conv.hip).
#include "xr_conv_p.h"

namespace {

constexpr int NT8 = 512;
constexpr int BM8 = 256, BN8 = 256;
constexpr int BUF8 = (BM8 + BN8) * 128;   // one K-tile of both operands: 64 KiB
constexpr int BOFF8 = BM8 * 128;          // B rows follow the A rows inside a buffer
typedef unsigned v4u8_t __attribute__((ext_vector_type(4)));
#define XR8_OOR 0x80000000u

#define XR8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ bf16x8_t lds_read16(unsigned addr) {
  v4u8_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return __builtin_bit_cast(bf16x8_t, v);
}

template <bool TR>
__global__ __launch_bounds__(NT8, 1) void igemm8_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = fdiv(p.fd_tn, bid);
  const int tile_n = bid - tile_m * p.tiles_n;
  const int n0 = tile_n * BN8;
  const int HW = p.H * p.W;
  const int l8 = lane >> 3;

  // ---- per-thread DMA rows: quarter tile A[mi] = tile rows {i*128 + mi*64 + [0,64)}, i = 0,1 (the rows both wave rows
  // read in the phases of M-half mi); B[ni] = tile columns {g*64 + ni*32 + [0,32)}, g = 0..3.  A quarter is 16 pieces of
  // 8 rows x 128 B = 1 KiB (one wave-instruction each); wave `wid` moves pieces wid and wid + 8.
  int a_off[2][2];
  unsigned a_msk[2][2];
  unsigned b_off[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 128 + mi * 64 + wid * 8 + l8;
      const int cc = (lane & 7) ^ ((row >> 1) & 7);
      bool valid;
      int nb, oh0, ow0;
      decode_pixel<TR>(tile_m * BM8 + row, p.M, p.fd_howo, p.fd_wo, HW, p.stride, p.pad, valid, nb, oh0, ow0);
      a_off[mi][i] = ((nb + oh0 * p.W + ow0) * p.C + cc * 8) * 2;
      unsigned m = 0;
      int ti = 0;
      for (int rr = 0; rr < p.R; ++rr)
        for (int ss = 0; ss < p.S; ++ss, ++ti) {
          const int y = TR ? oh0 - rr : oh0 + rr, x = TR ? ow0 - ss : ow0 + ss;
          if (valid && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) m |= 1u << ti;
        }
      a_msk[mi][i] = m;
    }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = wid + 8 * i;
      const int brow = (g >> 2) * 64 + ni * 32 + (g & 3) * 8 + l8;
      const int cc = (lane & 7) ^ ((brow >> 1) & 7);
      const int col = n0 + brow;
      b_off[ni][i] = col < p.K ? (unsigned)((col * p.Kg + cc * 8) * 2) : XR8_OOR;
    }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, p.w_bytes, 0x00020000);

  // wave-uniform cursor of the K-tile being PREFETCHED: tap index / row / column, channel chunk
  int st_ti = 0, st_rr = 0, st_ss = 0, st_c0 = 0;
  int dby = 0, wk2 = 0;
  unsigned sh = 0;
  auto cursor_eval = [&]() {
    int dpix = st_rr * p.W + st_ss;
    if (TR) dpix = -dpix;
    dby = (dpix * p.C + st_c0) * 2;
    wk2 = (st_ti * p.C + st_c0) * 2;
    sh = (unsigned)st_ti;
  };
  auto cursor_next = [&]() {
    st_c0 += 64;
    if (st_c0 >= p.C) {
      st_c0 = 0;
      ++st_ti;
      if (++st_ss == p.S) { st_ss = 0; ++st_rr; }
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto dma_A = [&](int buf, int mi) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned voff = ((a_msk[mi][i] >> sh) & 1u) ? (unsigned)(a_off[mi][i] + dby) : XR8_OOR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + buf * BUF8 + (i * 128 + mi * 64 + wid * 8) * 128), 16, voff,
                                               0, 0, 0);
    }
#else
    (void)buf; (void)mi;
#endif
  };
  auto dma_B = [&](int buf, int ni) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = wid + 8 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (lds_ptr_t)(smem + buf * BUF8 + BOFF8 + ((g >> 2) * 64 + ni * 32 + (g & 3) * 8) * 128), 16, b_off[ni][i], wk2, 0, 0);
    }
#else
    (void)buf; (void)ni;
#endif
  };

  // ---- fragment reads: lane -> (row l&15 of a 16-row block, 16-B chunk ks*4 + (l>>4)); every block base is a multiple
  // of 16 rows, so the swizzle term (row>>1)&7 depends on the lane only
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned swz = (unsigned)((lr >> 1) & 7);
  unsigned rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) rd[ks] = (unsigned)(lr * 128) + (((unsigned)(ks * 4 + lq) ^ swz) << 4);
  const unsigned ldsA = lds0 + (unsigned)(wr * 128 * 128), ldsB = lds0 + (unsigned)(BOFF8 + wc * 64 * 128);

  bf16x8_t fa[4][2], fb[2][2];
  f32x4_t acc[2][4][4];  // [M-half][16-row block][16-col block]: lane holds pixel l&15, channels 4*(l>>4)..+3
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define XR8_RD_A(buf, mi)                                                                        \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      fa[mb][ks] = lds_read16((unsigned)((buf) * BUF8) + ldsA + (unsigned)(((mi) * 64 + mb * 16) * 128) + rd[ks]);
#define XR8_RD_B(buf, ni)                                                                        \
  _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      fb[nb][ks] = lds_read16((unsigned)((buf) * BUF8) + ldsB + (unsigned)(((ni) * 32 + nb * 16) * 128) + rd[ks]);
#define XR8_MMA(mi, ni)                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)           \
      _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mi][mb][(ni) * 2 + nb] =                           \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nb][ks], fa[mb][ks], acc[mi][mb][(ni) * 2 + nb], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                              \
  __builtin_amdgcn_sched_barrier(0);
#define XR8_BAR()                      \
  __builtin_amdgcn_sched_barrier(0);   \
  __builtin_amdgcn_s_barrier();        \
  __builtin_amdgcn_sched_barrier(0);

  const int nk = p.Kg >> 6;
  // prologue: K-tile 0 into buffer 0 in the order the loop keeps (A0, B0, B1, A1)
  cursor_eval();
  dma_A(0, 0);
  dma_B(0, 0);
  dma_B(0, 1);
  dma_A(0, 1);
  XR8_VMCNT(2);
  XR8_BAR();
  if (wr == 1) { XR8_BAR(); }  // wave row 1 runs one barrier behind wave row 0

  for (int kk = 0; kk < nk; ++kk) {
    const int buf = kk & 1, nbuf = buf ^ 1;
    const bool nxt = kk + 1 < nk;
    if (nxt) {
      cursor_next();
      cursor_eval();
    }
    // ---- phase 0: quadrant (M-half 0, N-half 0)
    XR8_RD_B(buf, 0);
    XR8_RD_A(buf, 0);
    if (nxt) dma_A(nbuf, 0);
    XR8_BAR();
    XR8_MMA(0, 0);
    XR8_BAR();
    // ---- phase 1: (0, 1); retires A1 of this K-tile (read in phase 2)
    XR8_RD_B(buf, 1);
    if (nxt) {
      dma_B(nbuf, 0);
      XR8_VMCNT(4);
    } else {
      XR8_VMCNT(0);
    }
    XR8_BAR();
    XR8_MMA(0, 1);
    XR8_BAR();
    // ---- phase 2: (1, 1)
    XR8_RD_A(buf, 1);
    if (nxt) dma_B(nbuf, 1);
    XR8_BAR();
    XR8_MMA(1, 1);
    XR8_BAR();
    // ---- phase 3: (1, 0); retires A0, B0, B1 of the next K-tile (read from phase 0 on)
    XR8_RD_B(buf, 0);
    if (nxt) {
      dma_A(nbuf, 1);
      XR8_VMCNT(2);
    }
    XR8_BAR();
    XR8_MMA(1, 0);
    XR8_BAR();
  }
  if (wr == 0) { XR8_BAR(); }  // re-align the two wave rows: every operand read is retired, LDS is free for the epilogue

  // ---- epilogue: accumulators (+bias) -> wave-private [128 pixels][64 channels] bf16 image (same chunk swizzle) -> rows
  unsigned char* img = smem + wid * (128 * 128);
  {
    float bv[4][4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = n0 + wc * 64 + nn * 16 + lq * 4 + e;
        bv[nn][e] = (p.bias != nullptr && col < p.K) ? p.bias[col] : 0.f;
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int row = mi * 64 + mb * 16 + lr;
          const int chunk = nn * 2 + (lq >> 1);
          const f32x4_t v = acc[mi][mb][nn];
          uint2 pk;
          pk.x = pack2bf(v[0] + bv[nn][0], v[1] + bv[nn][1]);
          pk.y = pack2bf(v[2] + bv[nn][2], v[3] + bv[nn][3]);
          *reinterpret_cast<uint2*>(img + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + (lq & 1) * 8) = pk;
        }
  }
  // wave-private image: the wave's own LDS writes are ordered before its reads by the compiler's lgkmcnt
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* __restrict__ ep_src = reinterpret_cast<const bf16_t*>(p.ep_src);
  const bool ep = p.ep_src != nullptr;
  const int ch = lane & 7;
  const int ncol = n0 + wc * 64 + ch * 8;
  const int Kw = (p.K + 7) & ~7;
  float dal[8], alv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dal[e] = 0.f, alv[e] = 0.f;
  if (ep && ncol < p.K) ld8(p.ep_alpha + ncol, alv);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + l8;
    const int m = tile_m * BM8 + wr * 128 + row;
    if (m >= p.M || ncol >= Kw) continue;
    const bf16_t* sp = reinterpret_cast<const bf16_t*>(img + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
    bf16_t* dp = out + (size_t)m * p.ldo + ncol;
    if (ep) {
      float d[8], yv[8], o[8];
      ld8(sp, d);
      ld8(ep_src + (size_t)m * p.ldo + ncol, yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool neg = yv[e] <= 0.f;
        o[e] = neg ? d[e] * alv[e] : d[e];
        if (neg) dal[e] += d[e] * yv[e];
      }
      st8(dp, o);
    } else {
      *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
    }
  }
  if (ep) {  // lanes sharing a chunk column are 8 apart
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = dal[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 8 && ncol + e < p.K) atomicAdd(p.ep_dalpha + ncol + e, v);
    }
  }
}

}  // namespace

bool xr_igemm8_eligible(const IgemmP& p, int dtype, int transposed) {
  if (g_tune[7] == 0 || dtype != XR_BF16) return false;
  if (p.ws != nullptr || p.K % 8 != 0 || p.C % 64 != 0 || p.Kg != p.R * p.S * p.C || p.R * p.S > 32) return false;
  if (transposed && p.stride != 1) return false;
  const long long in_bytes = (long long)p.N * p.H * p.W * p.C * 2, w_bytes = (long long)p.K * p.Kg * 2;
  if (in_bytes >= (1ll << 31) || w_bytes >= (1ll << 31)) return false;
  if (g_tune[7] == 2) return true;  // forced (tests / tuning)
  // auto: full-width column tiles and enough row tiles to occupy most of the 256 CUs
  const long long tiles = (long long)cdiv(p.M, BM8) * cdiv(p.K, BN8);
  return p.K % BN8 == 0 && tiles >= 160;
}

int xr_igemm8_launch(IgemmP& p, int transposed, hipStream_t st) {
  p.tiles_n = cdiv((p.K + 7) / 8 * 8, BN8);
  const int tiles_m = cdiv(p.M, BM8);
  p.cls = 0; p.tpc = 0; p.Mc = 0; p.ksplit = 0;
  p.fd_howo = make_fd((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fd((unsigned)p.Wo);
  p.fd_tn = make_fd((unsigned)p.tiles_n);
  p.in_bytes = (unsigned)((long long)p.N * p.H * p.W * p.C * 2);
  p.w_bytes = (unsigned)((long long)p.K * p.Kg * 2);
  constexpr int smem = 2 * BUF8;
  static bool attr_done[2] = {false, false};
  const void* fn = transposed ? reinterpret_cast<const void*>(igemm8_kernel<true>) : reinterpret_cast<const void*>(igemm8_kernel<false>);
  if (!attr_done[transposed ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      xr_set_error("xr_conv_igemm(8-wave): hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
      return XR_E_LAUNCH;
    }
    attr_done[transposed ? 1 : 0] = true;
  }
  if (transposed) hipLaunchKernelGGL(igemm8_kernel<true>, dim3((unsigned)tiles_m * p.tiles_n), dim3(NT8), smem, st, p);
  else hipLaunchKernelGGL(igemm8_kernel<false>, dim3((unsigned)tiles_m * p.tiles_n), dim3(NT8), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv_igemm(8-wave)");
  return XR_OK;
}
