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
typedef unsigned v4u8_t __attribute__((ext_vector_type(4)));
#define XR8_OOR 0x80000000u

template <int N>
__device__ __forceinline__ void vmcnt_le() {  // wait until at most N of this wave's vector-memory operations are pending
  static_assert(N >= 0 && N <= 12, "unexpected DMA count");
#define XR8_VM(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XR8_VM(1); XR8_VM(2); XR8_VM(3); XR8_VM(4); XR8_VM(5); XR8_VM(6); XR8_VM(7); XR8_VM(8); XR8_VM(9); XR8_VM(10); XR8_VM(11);
  XR8_VM(12);
#undef XR8_VM
}

__device__ __forceinline__ bf16x8_t lds_read16(unsigned addr) {
  v4u8_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return __builtin_bit_cast(bf16x8_t, v);
}

// Tile configuration: WRN wave rows x WCN = 8 / WRN wave columns; a wave owns (2 * MB * 16) rows x 64 columns, one
// phase = one quadrant of MB x 2 MFMA blocks x 2 k-steps.
//   <2, 4>: 256 x 256 (K % 256 == 0, many rows)   <4, 2>: 256 x 128 (128-channel layers)   <2, 2>: 128 x 256 (7x7 maps)
// MB1 < MB trims the second M-half of every wave to MB1 blocks: the LDS image keeps its 2 * MB * 16 rows per wave row (the
// trimmed rows are zero-filled and never multiplied), the tile covers WRN * (MB + MB1) * 16 output rows.  <2, 4, 3> = 224
// rows turns the 196 (x 256-row) tiles of the 14x14 layers at batch 256 into 224 tiles of 7/8 the work: one round on 256 CUs
// either way.
template <int WRN, int MB, int MB1 = MB>
struct Cfg8 {
  static constexpr int WCN = 8 / WRN;
  static constexpr int QR = MB * 16, WTR = 2 * QR;     // quadrant / wave-tile rows (LDS image)
  static constexpr int WV = (MB + MB1) * 16;           // output rows per wave row
  static constexpr int BMV = WRN * WV;                 // output rows per tile
  static constexpr int BM = WRN * WTR, BN = WCN * 64;
  static constexpr int BUF = (BM + BN) * 128;           // one K-tile of both operands
  static constexpr int BOFF = BM * 128;                 // B rows follow the A rows inside a buffer
  static constexpr int PA = WRN * QR / 64, PB = WCN / 2;  // 1-KiB DMA pieces per wave per A / B quarter tile
  static constexpr int IMG = WTR * 128;                 // wave-private epilogue image
  static_assert(2 * BUF >= 8 * IMG, "epilogue images must fit in the operand buffers");
};

template <bool TR, int WRN, int MB, int MB1, bool DEEP>
__global__ __launch_bounds__(NT8, 1) void igemm8_kernel(IgemmP p) {
  using C8 = Cfg8<WRN, MB, MB1>;
  constexpr int WCN = C8::WCN, QR = C8::QR, WTR = C8::WTR, BN8 = C8::BN, BUF8 = C8::BUF, BOFF8 = C8::BOFF;
  constexpr int WV = C8::WV, BMV = C8::BMV;
  constexpr int PA = C8::PA, PB = C8::PB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wid / WCN, wc = wid % WCN;
  const int grp = wid >> 2;  // waves 0-3 / 4-7 sit on the same four SIMDs: the two ping-pong groups
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
  int a_off[2][PA];
  unsigned a_msk[2][PA];
  unsigned b_off[2][PB];
  auto a_piece_row = [&](int mi, int i) {  // first tile row of A piece (wid + 8 i) of quarter mi
    const int q = (wid + 8 * i) * 8;       // row inside the quarter's row list, QR rows per wave row
    return (q / QR) * WTR + mi * QR + (q % QR);
  };
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = a_piece_row(mi, i) + l8;   // LDS image row
      const int cc = (lane & 7) ^ ((row >> 1) & 7);
      const int wrow = row / WTR, rel = row % WTR;  // wave row, row inside the wave tile
      bool valid;
      int nb, oh0, ow0;
      decode_pixel<TR>(rel < WV ? tile_m * BMV + wrow * WV + rel : p.M, p.M, p.fd_howo, p.fd_wo, HW, p.stride, p.pad, valid, nb,
                       oh0, ow0);
      a_off[mi][i] = ((nb + oh0 * p.W + ow0) * p.C + cc * 8) * 2;
      unsigned xm = 0, m = 0;   // separable validity: column mask replicated for every valid tap row
      for (int ss = 0; ss < p.S; ++ss) {
        const int x = TR ? ow0 - ss : ow0 + ss;
        if ((unsigned)x < (unsigned)p.W) xm |= 1u << ss;
      }
      if (!valid) xm = 0;
      for (int rr = 0; rr < p.R; ++rr) {
        const int y = TR ? oh0 - rr : oh0 + rr;
        if ((unsigned)y < (unsigned)p.H) m |= xm << (rr * p.S);
      }
      a_msk[mi][i] = m;
    }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int i = 0; i < PB; ++i) {
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
  int dby_b = 0, wk2_b = 0;   // DEEP: (dby, wk2, sh) address K-tile t+2 (the A0 / B0 quarters), the _b set K-tile t+1 (B1 / A1)
  unsigned sh_b = 0;
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
  auto dma_A = [&](int buf, int mi, bool lag = false) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned sh_ = lag ? sh_b : sh;
    const int dby_ = lag ? dby_b : dby;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const unsigned voff = ((a_msk[mi][i] >> sh_) & 1u) ? (unsigned)(a_off[mi][i] + dby_) : XR8_OOR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + buf * BUF8 + a_piece_row(mi, i) * 128), 16, voff, 0, 0, 0);
    }
#else
    (void)buf; (void)mi; (void)lag;
#endif
  };
  auto dma_B = [&](int buf, int ni, bool lag = false) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int wk_ = lag ? wk2_b : wk2;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int g = wid + 8 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (lds_ptr_t)(smem + buf * BUF8 + BOFF8 + ((g >> 2) * 64 + ni * 32 + (g & 3) * 8) * 128), 16, b_off[ni][i], wk_, 0, 0);
    }
#else
    (void)buf; (void)ni; (void)lag;
#endif
  };

  // ---- fragment reads: lane -> (row l&15 of a 16-row block, 16-B chunk ks*4 + (l>>4)); every block base is a multiple
  // of 16 rows, so the swizzle term (row>>1)&7 depends on the lane only
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned swz = (unsigned)((lr >> 1) & 7);
  unsigned rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) rd[ks] = (unsigned)(lr * 128) + (((unsigned)(ks * 4 + lq) ^ swz) << 4);
  const unsigned ldsA = lds0 + (unsigned)(wr * WTR * 128), ldsB = lds0 + (unsigned)(BOFF8 + wc * 64 * 128);

  bf16x8_t fa[MB][2], fb[2][2];
  bf16x8_t fb0[2][2];  // DEEP keeps the N-half-0 fragments for phase 3 instead of re-reading them (frees the B0 quarter early)
  f32x4_t acc[2][MB][4];  // [M-half][16-row block][16-col block]: lane holds pixel l&15, channels 4*(l>>4)..+3
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define XR8_RD_A(buf, mi)                                                                        \
  _Pragma("unroll") for (int mb = 0; mb < ((mi) ? MB1 : MB); ++mb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      fa[mb][ks] = lds_read16((unsigned)((buf) * BUF8) + ldsA + (unsigned)(((mi) * QR + mb * 16) * 128) + rd[ks]);
#define XR8_RD_B(buf, ni)                                                                        \
  _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      fb[nb][ks] = lds_read16((unsigned)((buf) * BUF8) + ldsB + (unsigned)(((ni) * 32 + nb * 16) * 128) + rd[ks]);
#define XR8_MMA(mi, ni)                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mb = 0; mb < ((mi) ? MB1 : MB); ++mb) \
      _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mi][mb][(ni) * 2 + nb] =                           \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nb][ks], fa[mb][ks], acc[mi][mb][(ni) * 2 + nb], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                              \
  __builtin_amdgcn_sched_barrier(0);
#define XR8_BAR()                      \
  __builtin_amdgcn_sched_barrier(0);   \
  __builtin_amdgcn_s_barrier();        \
  __builtin_amdgcn_sched_barrier(0);

  const int nk = p.Kg >> 6;
  if constexpr (DEEP) {
    // Deeper prefetch (5-6 phases instead of 3-4; four quarter tiles in flight): a quarter is refilled two phases after its
    // last read -- with the NEXT-BUT-ONE K-tile for A0 / B0 (same buffer), the next K-tile for B1 / A1 (other buffer):
    //   phase 0: read B0, A0 | issue B1(t+1) | wait B1(t)        phase 2: read A1 | issue A0(t+2)
    //   phase 1: read B1     | issue A1(t+1) | wait A1(t)        phase 3: (B0 from registers) | issue B0(t+2) | wait A0,B0(t+1)
    // Counted waits leave 2 PA + 2 PB pieces in flight while a next K-tile exists.
#define XR8_RD_B0(buf)                                                                           \
  _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      fb0[nb][ks] = lds_read16((unsigned)((buf) * BUF8) + ldsB + (unsigned)((nb * 16) * 128) + rd[ks]);
#define XR8_MMA_B0(mi)                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mb = 0; mb < ((mi) ? MB1 : MB); ++mb) \
      _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mi][mb][nb] =                                      \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[nb][ks], fa[mb][ks], acc[mi][mb][nb], 0, 0, 0);          \
  __builtin_amdgcn_s_setprio(0);                                                                              \
  __builtin_amdgcn_sched_barrier(0);
    cursor_eval();              // K-tile 0
    dma_A(0, 0);
    dma_B(0, 0);
    dma_B(0, 1);
    dma_A(0, 1);
    if (nk > 1) {
      cursor_next();
      cursor_eval();            // K-tile 1
      dma_A(1, 0);
      dma_B(1, 0);
      vmcnt_le<2 * PA + 2 * PB>();
    } else {
      vmcnt_le<PA + PB>();
    }
    XR8_BAR();
    if (grp == 1) { XR8_BAR(); }
    for (int kk = 0; kk < nk; ++kk) {
      const int buf = kk & 1, nbuf = buf ^ 1;
      const bool n1 = kk + 1 < nk, n2 = kk + 2 < nk;
      // cursor bookkeeping: (dby, wk2, sh) currently describe K-tile kk+1 (when it exists); keep them as the lagging set and
      // advance the leading set to K-tile kk+2
      dby_b = dby; wk2_b = wk2; sh_b = sh;
      if (n2) {
        cursor_next();
        cursor_eval();
      }
      // ---- phase 0
      XR8_RD_B0(buf);
      XR8_RD_A(buf, 0);
      if (n1) {
        dma_B(nbuf, 1, true);
        vmcnt_le<2 * PA + 2 * PB>();
      } else {
        vmcnt_le<PA>();
      }
      XR8_BAR();
      XR8_MMA_B0(0);
      XR8_BAR();
      // ---- phase 1
      XR8_RD_B(buf, 1);
      if (n1) {
        dma_A(nbuf, 1, true);
        vmcnt_le<2 * PA + 2 * PB>();
      } else {
        vmcnt_le<0>();
      }
      XR8_BAR();
      XR8_MMA(0, 1);
      XR8_BAR();
      // ---- phase 2
      XR8_RD_A(buf, 1);
      if (n2) dma_A(buf, 0);
      XR8_BAR();
      XR8_MMA(1, 1);
      XR8_BAR();
      // ---- phase 3
      if (n2) {
        dma_B(buf, 0);
        vmcnt_le<2 * PA + 2 * PB>();
      } else if (n1) {
        vmcnt_le<PA + PB>();
      }
      XR8_BAR();
      XR8_MMA_B0(1);
      XR8_BAR();
    }
  } else {
  // prologue: K-tile 0 into buffer 0 in the order the loop keeps (A0, B0, B1, A1)
  cursor_eval();
  dma_A(0, 0);
  dma_B(0, 0);
  dma_B(0, 1);
  dma_A(0, 1);
  vmcnt_le<PA>();
  XR8_BAR();
  if (grp == 1) { XR8_BAR(); }  // group 1 runs one barrier behind group 0

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
      vmcnt_le<PA + PB>();
    } else {
      vmcnt_le<0>();
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
      vmcnt_le<PA>();
    }
    XR8_BAR();
    XR8_MMA(1, 0);
    XR8_BAR();
  }
  }
  if (grp == 0) { XR8_BAR(); }  // re-align the two groups: every operand read is retired, LDS is free for the epilogue

  // ---- epilogue: accumulators (+bias) -> wave-private [128 pixels][64 channels] bf16 image (same chunk swizzle) -> rows
  unsigned char* img = smem + wid * C8::IMG;
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
      for (int mb = 0; mb < (mi ? MB1 : MB); ++mb)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int row = mi * QR + mb * 16 + lr;
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
  const bool bnr = p.ep_red != nullptr;               // BatchNorm-backward partial sums (ep_src = BN input)
  const bool ep = p.ep_src != nullptr && !bnr;       // PReLU backward
  float bs0[8], bs1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs0[e] = 0.f, bs1[e] = 0.f;
  const int ch = lane & 7;
  const int ncol = n0 + wc * 64 + ch * 8;
  const int Kw = (p.K + 7) & ~7;
  float dal[8], alv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dal[e] = 0.f, alv[e] = 0.f;
  bf16_t* __restrict__ out2 = reinterpret_cast<bf16_t*>(p.ep2_out);
  const bool ep2 = p.ep2_out != nullptr;
  if ((ep || ep2) && ncol < p.K) ld8(p.ep_alpha + ncol, alv);
#pragma unroll 4
  for (int it = 0; it < WV / 8; ++it) {
    const int row = it * 8 + l8;
    const int m = tile_m * BMV + wr * WV + row;
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
    } else if (p.ep_add != nullptr) {   // gradient of an identity branch summed here
      float d[8], av[8];
      ld8(sp, d);
      ld8(reinterpret_cast<const bf16_t*>(p.ep_add) + (size_t)m * p.ldo + ncol, av);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] += av[e];
      st8(dp, d);
    } else {
      *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
      if (bnr) {  // sums of d and d * x per channel (d as rounded for the output tensor)
        float d[8], xv[8];
        ld8(sp, d);
        if (ep_src != nullptr) {
          ld8(ep_src + (size_t)m * p.ldo + ncol, xv);
        } else {   // statistics of the output itself (sum, sum of squares) for a BatchNorm that follows
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = d[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bs0[e] += d[e];
          bs1[e] += d[e] * xv[e];
        }
      }
      if (ep2) {  // second output: PReLU of the value just produced (what the next convolution consumes)
        float d[8], o[8];
        ld8(sp, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = d[e] > 0.f ? d[e] : d[e] * alv[e];
        st8(out2 + (size_t)m * p.ldo + ncol, o);
      }
    }
  }
  if (ep) {
    // dalpha: fold lanes (same chunk column = 8 apart), then the wave rows through LDS, then ONE atomic per column and
    // workgroup issued as full 64-lane instructions -- per-wave atomics (8 x 64 per tile onto the same few cache lines of
    // dalpha, from every CU at once) serialise at the memory side and cost more than the whole main loop
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = dal[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      dal[e] = v;
    }
    __syncthreads();  // every wave is done with its epilogue image
    float* red = reinterpret_cast<float*>(smem);
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wid * 64 + lane * 8 + e] = dal[e];
    }
    __syncthreads();
    for (int c = t; c < BN8; c += NT8) {
      const int cwc = c >> 6, cc = c & 63;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < WRN; ++r) sum += red[(r * WCN + cwc) * 64 + cc];
      if (n0 + c < p.K) atomicAdd(p.ep_dalpha + (size_t)(tile_m % p.ep_spread) * p.K + n0 + c, sum);
    }
  }
  if (bnr) {  // same fold as dalpha (lanes -> wave rows through LDS -> one atomic per column and workgroup), two vectors
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v ? bs1[e] : bs0[e];
        x += __shfl_xor(x, 8, 64);
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        f[e] = x;
      }
      __syncthreads();
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wid * 64 + lane * 8 + e] = f[e];
      }
      __syncthreads();
      for (int c = t; c < BN8; c += NT8) {
        const int cwc = c >> 6, cc = c & 63;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < WRN; ++r) sum += red[(r * WCN + cwc) * 64 + cc];
        if (n0 + c < p.K) atomicAdd(p.ep_red + ((size_t)v * p.ep_spread + (tile_m % p.ep_spread)) * p.K + n0 + c, sum);
      }
    }
  }
}

}  // namespace

// tile configuration for a problem: 0 none, 1 = 256x256, 2 = 256x128, 3 = 128x256, 4 = 224x256, 5 = 512x128, 6 = 448x128
static int igemm8_config(const IgemmP& p, int dtype, int transposed) {
  const int knob = g_tune[7];
  if (knob == 0 || dtype != XR_BF16) return 0;
  if (p.ws != nullptr || p.K % 8 != 0 || p.C % 64 != 0 || p.Kg != p.R * p.S * p.C || p.R * p.S > 32) return 0;
  if (transposed && p.stride != 1) return 0;
  const long long in_bytes = (long long)p.N * p.H * p.W * p.C * 2, w_bytes = (long long)p.K * p.Kg * 2;
  if (in_bytes >= (1ll << 31) || w_bytes >= (1ll << 31)) return 0;
  if (knob >= 3 && knob <= 8) return knob - 2;  // forced configuration (tests / tuning)
  // auto (knob 1; knob 2 drops the occupancy thresholds).  One workgroup per CU: time ~ rounds x rows per tile.  Only the
  // 128x64-per-wave configurations beat the 4-wave kernel (measured: tools/conv_bench.py), so 256x128 / 128x256 stay manual.
  const bool force = knob == 2;
  if (p.Kg < 8 * 64 && !force) return 0;  // short reductions (1x1 shortcuts) cannot amortise the 64-128 KiB prologue
  if (p.K % 256 == 0) {
    const long long t256 = (long long)cdiv(p.M, 256) * (p.K / 256), t224 = (long long)cdiv(p.M, 224) * (p.K / 256);
    if (t256 < 160 && !force) return 0;
    if (knob == 10) return 1;
    return cdiv(t224, 256) * 224 < cdiv(t256, 256) * 256 ? 4 : 1;
  }
  if (p.K % 128 == 0 && knob != 9 && knob != 10) {
    const long long t512 = (long long)cdiv(p.M, 512) * (p.K / 128), t448 = (long long)cdiv(p.M, 448) * (p.K / 128);
    if (t512 < 160 && !force) return 0;
    return cdiv(t448, 256) * 448 < cdiv(t512, 256) * 512 ? 6 : 5;
  }
  return 0;
}

bool xr_igemm8_eligible(const IgemmP& p, int dtype, int transposed) { return igemm8_config(p, dtype, transposed) != 0; }

template <bool TR, int WRN, int MB, int MB1, bool DEEP>
static int igemm8_launch_cfg2(IgemmP& p, hipStream_t st) {
  using C8 = Cfg8<WRN, MB, MB1>;
  p.tiles_n = cdiv((p.K + 7) / 8 * 8, C8::BN);
  const int tiles_m = cdiv(p.M, C8::BMV);
  p.fd_tn = make_fd((unsigned)p.tiles_n);
  constexpr int smem = 2 * C8::BUF;
  auto kern = igemm8_kernel<TR, WRN, MB, MB1, DEEP>;
  // once per template instance, safe from concurrent autograd threads; a failure is remembered and reported on every call
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  if (attr_err != hipSuccess) {
    xr_set_error("xr_conv_igemm(8-wave): hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(attr_err));
    return XR_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles_m * p.tiles_n), dim3(NT8), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv_igemm(8-wave)");
  return XR_OK;
}

template <bool TR, int WRN, int MB, int MB1 = MB>
static int igemm8_launch_cfg(IgemmP& p, hipStream_t st) {
  if (g_tune[12]) return igemm8_launch_cfg2<TR, WRN, MB, MB1, true>(p, st);
  return igemm8_launch_cfg2<TR, WRN, MB, MB1, false>(p, st);
}

int xr_igemm8_launch(IgemmP& p, int transposed, hipStream_t st) {
  const int cfg = igemm8_config(p, XR_BF16, transposed);
  p.cls = 0; p.tpc = 0; p.Mc = 0; p.ksplit = 0;
  p.fd_howo = make_fd((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fd((unsigned)p.Wo);
  p.in_bytes = (unsigned)((long long)p.N * p.H * p.W * p.C * 2);
  p.w_bytes = (unsigned)((long long)p.K * p.Kg * 2);
  switch (cfg) {
    case 1: return transposed ? igemm8_launch_cfg<true, 2, 4>(p, st) : igemm8_launch_cfg<false, 2, 4>(p, st);
    case 2: return transposed ? igemm8_launch_cfg<true, 4, 2>(p, st) : igemm8_launch_cfg<false, 4, 2>(p, st);
    case 3: return transposed ? igemm8_launch_cfg<true, 2, 2>(p, st) : igemm8_launch_cfg<false, 2, 2>(p, st);
    case 4: return transposed ? igemm8_launch_cfg<true, 2, 4, 3>(p, st) : igemm8_launch_cfg<false, 2, 4, 3>(p, st);
    case 5: return transposed ? igemm8_launch_cfg<true, 4, 4>(p, st) : igemm8_launch_cfg<false, 4, 4>(p, st);
    case 6: return transposed ? igemm8_launch_cfg<true, 4, 4, 3>(p, st) : igemm8_launch_cfg<false, 4, 4, 3>(p, st);
  }
  xr_set_error("xr_conv_igemm(8-wave): problem not eligible");
  return XR_E_INVALID;
}
