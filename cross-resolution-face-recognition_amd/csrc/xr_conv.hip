// xr_conv.hip -- implicit-GEMM convolution family for gfx950 (CDNA4, wave64, MFMA 32x32x16 bf16).
//
// One gather engine serves Conv2d forward, Conv2d input-gradient, ConvTranspose2d forward/backward and
// Linear (as a full-extent convolution):
//   out[m][k] = sum_{tap,c} gather(in)[m][tap][c] * wpack[k][tap*C + c]        (xr_conv_igemm)
//   dwp[k][tap*C + c] += sum_m dy[m][k] * gather(in)[m][tap][c]                (xr_conv_wgrad)
// Activations are NHWC so a tap's C-slice is one contiguous, coalesced 16-B-chunk row; both operands are
// staged through LDS in a K-contiguous XOR-swizzled image (conflict-free ds_read_b128 for the 32x32x16
// operand map) -- for the weight gradient, where the reduction runs over pixels, the images stay
// [pixel][channel] and the operands are fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16.
// XR_F32 mode keeps fp32 tensors in HBM and runs each product as hi*hi + hi*lo + lo*hi split-bf16 MFMAs.
//
// Replaces the ATen conv calls issued by /root/reference model/FSRnet.py:79,85,110,312,318,345,351,384,391,
// 392,432,436,439; SUPER_RESOLUTION/model/model_irse.py:56-60,140,147; model/resnet.py:9-16,158,170.
#include "xr_conv_p.h"
#include <string.h>
#include <type_traits>
#include <vector>

namespace {

constexpr int BK = 64;        // reduction elements per LDS stage
constexpr int NT = 256;       // threads per workgroup (4 waves)

// input coordinate for tap (r,s); returns validity and pixel offset hi*W+wi
template <bool TR>
__device__ __forceinline__ bool tap_coord(int oh0, int ow0, int r, int s, int stride, int H, int W, int& pix) {
  int hi, wi;
  if (TR) {
    int th = oh0 - r, tw = ow0 - s;
    if (th < 0 || tw < 0) return false;
    if (stride == 1) {
      hi = th; wi = tw;
    } else {
      hi = th / stride; wi = tw / stride;
      if (hi * stride != th || wi * stride != tw) return false;
    }
  } else {
    hi = oh0 + r; wi = ow0 + s;
  }
  if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) return false;
  pix = hi * W + wi;
  return true;
}

// LDS operand image: rows of 64 bf16 (128 B), 16-B chunk index XOR-swizzled by (row>>1)&7
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// same idea for 32-element (64 B) stage rows: 4 chunks per row, swizzled by (row>>2)&3 -> any 16 consecutive rows of
// one chunk column land on 16 distinct 16-B slots of the 256-B bank row
template <int BKT>
__device__ __forceinline__ int lds_off_t(int row, int chunk) {
  if constexpr (BKT == 64) return lds_off(row, chunk);
  else return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// MODE: 0 bf16 tensors (one plane); 1 fp32 tensors split into three bf16 planes, six plane-pair MFMAs per product (all 24
// significand bits); 2 fp32 tensors split into two planes, three MFMAs (hi*hi, hi*lo, lo*hi: ~16 significand bits -- XR_F32X2)
__host__ __device__ constexpr int ns_of(int mode) { return mode == 1 ? 3 : (mode == 2 ? 2 : 1); }

// fp32 -> NP bf16 planes with x ~= sum_p plane[p] (3 planes carry all 24 significand bits)
template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&pl)[NP]) {
  unsigned w[NP][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = v[2 * i], b = v[2 * i + 1];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const bf16_t ha = f2bf(a), hb = f2bf(b);
      w[q][i] = (unsigned)ha | ((unsigned)hb << 16);
      a -= bf2f(ha);
      b -= bf2f(hb);
    }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) pl[q] = make_uint4(w[q][0], w[q][1], w[q][2], w[q][3]);
}

// split-precision product: all plane pairs (s, t) with s + t < NS, smallest terms first
template <int NS>
__device__ __forceinline__ f32x16_t mfma_split(const bf16x8_t (&a)[NS], const bf16x8_t (&b)[NS], f32x16_t acc) {
#pragma unroll
  for (int d = NS - 1; d >= 0; --d)
#pragma unroll
    for (int s = 0; s <= d; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[d - s], acc, 0, 0, 0);
  return acc;
}

// ------------------------------------------------------------------------------------------------ forward/dgrad
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define XR_OOR 0x80000000u  // buffer offset beyond every descriptor extent used here: the load returns zeros

// FAST: taps address the input linearly (forward conv of any stride, stride-1 dgrad, class-wise strided dgrad) and
// C % 64 == 0, so a K-stage is one (tap, 64-channel chunk): per-row byte offsets and per-row tap-validity bitmasks are
// computed once per tile, the per-stage part is scalar, and out-of-image taps are fetched as zeros by the buffer
// range check -- ~4 VALU per gathered row per stage instead of ~15.
// DMA (bf16 FAST only): both operand tiles go global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, zero fill by
// the descriptor range check), double-buffered: the copy of stage k+1 runs while the MFMAs consume stage k, and neither
// VGPRs nor ds_write issue slots (the LDS write port is ~80 B/clk on gfx950) are spent on staging.  The LDS image is
// lane-linear, so the XOR swizzle is applied to the SOURCE chunk each lane fetches.
template <int MODE, int BM, int BN, int WM, bool TR, bool FAST, int DMA, int BKT>
__global__ __launch_bounds__(NT, (FAST && MODE == 0 && DMA == 0) ? (BM > 128 ? 2 : 4) : 1) void igemm_kernel(IgemmP p) {
  constexpr int NBUF = DMA ? DMA : 1;
  static_assert(DMA == 0 || (FAST && MODE == 0), "DMA staging needs the FAST bf16 path");
  static_assert(BKT == 64 || (BKT == 32 && DMA != 0), "32-element stages exist for the DMA path only");
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NS = ns_of(MODE);
  constexpr int CPW = BKT / 8;        // 16-B chunks per stage row
  constexpr int RPP = NT / CPW;       // tile rows covered by one pass of the 256 threads
  constexpr int RA = BM / RPP, RB = BN / RPP;
  constexpr int ROWB = BKT * 2;       // stage row bytes
  using in_t = typename std::conditional<MODE != 0, float, bf16_t>::type;
  using out_t = in_t;
  constexpr int PADE = 16 / sizeof(out_t);

  constexpr int STAGE = NS * (BM + BN) * ROWB;       // one LDS stage (A rows then B rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  // XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs; give each XCD a
  // contiguous run of tiles so neighbouring tiles (shared halo rows / shared weight panel) meet in one L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = fdiv(p.fd_tn, bid);
  const int tile_n = bid - tile_m * p.tiles_n;
  const int n0 = tile_n * BN;
  const int HW = p.H * p.W;
  const in_t* __restrict__ in = reinterpret_cast<const in_t*>(p.in);
  // class mode bookkeeping (wave-uniform)
  const bool cls_mode = TR && p.cls;
  int ph = 0, pw = 0, r0 = 0, s0 = 0, nr = 0, ns = 0, t_in = tile_m;
  if (cls_mode) {
    const int c = tile_m / p.tpc;
    t_in = tile_m - c * p.tpc;
    ph = c / p.stride; pw = c - ph * p.stride;
    r0 = (ph + p.pad) % p.stride; s0 = (pw + p.pad) % p.stride;
    nr = r0 < p.R ? (p.R - r0 + p.stride - 1) / p.stride : 0;
    ns = s0 < p.S ? (p.S - s0 + p.stride - 1) / p.stride : 0;
  }
  // tile row -> output pixel (memory index m_out, or -1) and gather origin
  auto decode_row = [&](int row, bool& valid, int& nb, int& oh0, int& ow0) -> int {
    if (!cls_mode) {
      const int m = tile_m * BM + row;
      decode_pixel<TR>(m, p.M, p.fd_howo, p.fd_wo, HW, p.stride, p.pad, valid, nb, oh0, ow0);
      return valid ? m : -1;
    }
    const int idx = t_in * BM + row;
    valid = idx < p.Mc;
    const int ii = valid ? idx : 0;
    const int n = fdiv(p.fd_hqwq, ii);
    const int rem = ii - n * (int)p.fd_hqwq.d;
    const int hq = fdiv(p.fd_wq, rem), wq = rem - hq * (int)p.fd_wq.d;
    const int ho = hq * p.stride + ph, wo = wq * p.stride + pw;
    nb = n * HW; oh0 = ho + p.pad; ow0 = wo + p.pad;
    return valid ? (n * p.Ho + ho) * p.Wo + wo : -1;
  };

  const int rbase = t / CPW;
  // 16-B chunk column inside the K-stage this thread fetches; with DMA staging the thread's LDS slot is fixed
  // (lane-linear image), so it fetches the chunk that the swizzle maps INTO that slot
  const int swz = BKT == 64 ? ((rbase >> 1) & 7) : ((rbase >> 2) & 3);
  const int cc = DMA ? ((t & (CPW - 1)) ^ swz) : (t & (CPW - 1));
  bool a_valid[RA];
  int a_nb[RA], a_oh[RA], a_ow[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) (void)decode_row(rbase + RPP * i, a_valid[i], a_nb[i], a_oh[i], a_ow[i]);

  float a_f[MODE ? RA : 1][8];
  uint4 a_u[MODE ? 1 : RA];
  uint4 b_u[NS][RB];
  const int taps = p.R * p.S;

  // ---- FAST path state
  constexpr int ESZ = (int)sizeof(in_t);
  int a_off[FAST ? RA : 1];
  unsigned a_mlo[FAST ? RA : 1];
  unsigned b_off[FAST ? RB : 1];
  __amdgpu_buffer_rsrc_t rsA, rsB;
  int st_ti = 0, st_rr = 0, st_ss = 0, st_c0 = 0;   // wave-uniform stage cursor: tap index, tap row/col, channel chunk
  const int nsw = cls_mode ? ns : p.S;
  if constexpr (FAST) {
    const int nrw = cls_mode ? nr : p.R;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int ah = a_oh[i], aw = a_ow[i];
      if (cls_mode) {
        ah = fdiv(p.fd_st, ah - r0);
        aw = fdiv(p.fd_st, aw - s0);
      }
      a_off[i] = ((a_nb[i] + ah * p.W + aw) * p.C + cc * 8) * ESZ;
      // tap (rr, ss) is inside the image iff its row and its column are: build the column mask once and replicate it
      // for every valid tap row (R + S comparisons per gathered row instead of R * S -- the prologue is a visible share
      // of the instruction stream on the nine-stage 64-channel layers)
      unsigned xm = 0, lo = 0;
      for (int ss = 0; ss < nsw; ++ss) {
        const int x = TR ? aw - ss : aw + ss;
        if ((unsigned)x < (unsigned)p.W) xm |= 1u << ss;
      }
      if (!a_valid[i]) xm = 0;
      for (int rr = 0; rr < nrw; ++rr) {
        const int y = TR ? ah - rr : ah + rr;
        if ((unsigned)y < (unsigned)p.H) lo |= xm << (rr * nsw);
      }
      a_mlo[i] = lo;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = n0 + rbase + RPP * i;
      b_off[i] = row < p.K ? (unsigned)((row * p.Kg + cc * 8) * 2) : XR_OOR;
    }
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  }
  auto load_stage_fast = [&]() {
    int dpix = st_rr * p.W + st_ss;
    if (TR) dpix = -dpix;
    const int dby = (dpix * p.C + st_c0) * ESZ;
    const int wk = cls_mode ? ((r0 + st_rr * p.stride) * p.S + (s0 + st_ss * p.stride)) * p.C + st_c0 : st_ti * p.C + st_c0;
    const unsigned sh = (unsigned)st_ti;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const unsigned voff = ((a_mlo[i] >> sh) & 1u) ? (unsigned)(a_off[i] + dby) : XR_OOR;
      if constexpr (MODE != 0) {
        const v4u_t u0 = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, 0, 0);
        const v4u_t u1 = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff + 16u, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a_f[i][e] = __uint_as_float(u0[e]);
          a_f[i][4 + e] = __uint_as_float(u1[e]);
        }
      } else {
        const v4u_t u = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, 0, 0);
        a_u[i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
    }
    const int plane = p.K * p.Kg * 2;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const v4u_t u = __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], wk * 2 + q * plane, 0);
        b_u[q][i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
    // advance the cursor to the next stage
    st_c0 += BKT;
    if (st_c0 >= p.C) {
      st_c0 = 0;
      ++st_ti;
      if (++st_ss == nsw) { st_ss = 0; ++st_rr; }
    }
  };

  // LDS-DMA staging of the stage under the cursor into buffer `buf`
  auto dma_stage = [&](int buf) {
    if constexpr (DMA) {
      typedef __attribute__((address_space(3))) void* lds_ptr_t;
      constexpr int RPW = 1024 / ROWB;           // stage rows one 1-KiB wave-instruction fills
      const int wrow = (t >> 6) * RPW;           // first tile row this wave fills (wave-uniform; hipcc broadcasts it for M0)
      unsigned char* sA = smem + buf * STAGE;
      unsigned char* sB = sA + BM * ROWB;
      int dpix = st_rr * p.W + st_ss;
      if (TR) dpix = -dpix;
      const int dby = (dpix * p.C + st_c0) * ESZ;
      const int wk = cls_mode ? ((r0 + st_rr * p.stride) * p.S + (s0 + st_ss * p.stride)) * p.C + st_c0 : st_ti * p.C + st_c0;
      const unsigned sh = (unsigned)st_ti;
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass of hipcc does not know that gfx950 allows 16-byte LDS-DMA
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const unsigned voff = ((a_mlo[i] >> sh) & 1u) ? (unsigned)(a_off[i] + dby) : XR_OOR;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + (wrow + RPP * i) * ROWB), 16, voff, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(sB + (wrow + RPP * i) * ROWB), 16, b_off[i], wk * 2, 0, 0);
#else
      (void)sA; (void)sB; (void)dby; (void)wk; (void)sh; (void)wrow;
#endif
      st_c0 += BKT;
      if (st_c0 >= p.C) {
        st_c0 = 0;
        ++st_ti;
        if (++st_ss == nsw) { st_ss = 0; ++st_rr; }
      }
    }
  };

  auto load_stage = [&](int kk) {
    if constexpr (FAST) {
      load_stage_fast();
      return;
    }
    int k0, c, r, s;
    bool tap_ok;
    if (cls_mode) {  // only the taps this output class can see; C % 64 == 0 so a stage never straddles taps
      const int cpt = p.C >> 6;
      const int ti = kk / cpt;   // wave-uniform scalar division, once per stage
      c = (kk - ti * cpt) * BK + cc * 8;
      const int ri = ti / ns, si = ti - ri * ns;
      r = r0 + ri * p.stride; s = s0 + si * p.stride;
      k0 = (r * p.S + s) * p.C + c;
      tap_ok = true;
    } else {
      k0 = kk * BK + cc * 8;
      const int tap = fdiv(p.fd_c, k0);
      c = k0 - tap * p.C;
      r = fdiv(p.fd_s, tap);
      s = tap - r * p.S;
      tap_ok = tap < taps;
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int pix = 0;
      bool ok = a_valid[i] && tap_ok && tap_coord<TR>(a_oh[i], a_ow[i], r, s, p.stride, p.H, p.W, pix);
      const in_t* src = in + ((size_t)(a_nb[i] + pix) * p.C + c);
      if constexpr (MODE != 0) {
        if (ok) {
          ld8(src, a_f[i]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) a_f[i][e] = 0.f;
        }
      } else {
        a_u[i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = n0 + rbase + 32 * i;
      const bool ok = row < p.K;
      const size_t off = (size_t)row * p.Kg + k0;
#pragma unroll
      for (int q = 0; q < NS; ++q)
        b_u[q][i] = ok ? *reinterpret_cast<const uint4*>(p.w + (size_t)q * p.K * p.Kg + off) : make_uint4(0, 0, 0, 0);
    }
  };

  auto store_stage = [&](int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + NS * BM * 128;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int off = lds_off(rbase + 32 * i, cc);
      if constexpr (MODE != 0) {
        uint4 pl[NS];
        split8<NS>(a_f[i], pl);
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint4*>(sA + q * BM * 128 + off) = pl[q];
      } else {
        *reinterpret_cast<uint4*>(sA + off) = a_u[i];
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int off = lds_off(rbase + 32 * i, cc);
#pragma unroll
      for (int q = 0; q < NS; ++q) *reinterpret_cast<uint4*>(sB + q * BN * 128 + off) = b_u[q][i];
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
  const int lr = lane & 31, lh = lane >> 5;
  int nk = cls_mode ? nr * ns * (p.C / BKT) : p.Kg / BKT;
  int kbeg = 0;
  if (p.ksplit > 0) {
    kbeg = blockIdx.y * p.ksplit;
    nk = (kbeg + p.ksplit < nk) ? kbeg + p.ksplit : nk;
  }
  if constexpr (FAST) {
    const int cpt = p.C / BKT;
    st_ti = kbeg / cpt;
    st_c0 = (kbeg - st_ti * cpt) * BKT;
    st_rr = st_ti / nsw;
    st_ss = st_ti - st_rr * nsw;
  }

  auto compute_stage = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + NS * BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8_t fa[TM][NS], fb[TN][NS];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int off = lds_off_t<BKT>(wm0 + i * 32 + lr, ks * 2 + lh);
#pragma unroll
        for (int s = 0; s < NS; ++s) fa[i][s] = *reinterpret_cast<const bf16x8_t*>(sA + s * BM * ROWB + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int off = lds_off_t<BKT>(wn0 + j * 32 + lr, ks * 2 + lh);
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[j][s] = *reinterpret_cast<const bf16x8_t*>(sB + s * BN * ROWB + off);
      }
      __builtin_amdgcn_s_setprio(1);   // unconditional: a run-time switch here compiled to a scalar branch per MFMA group
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_split<NS>(fa[i], fb[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  if constexpr (DMA == 2) {
    // stage kk lives in buffer (kk-kbeg)&1.  The DMA of stage kk+1 into the other buffer is issued before the MFMAs
    // of stage kk; the barrier (hipcc drains vmcnt(0) in front of it because LDS-DMA is pending) then orders RAW
    // (next stage's reads) and WAR (the buffer just consumed becomes the next DMA target).
    if (kbeg < nk) dma_stage(0);
    __syncthreads();
    for (int kk = kbeg; kk < nk; ++kk) {
      const int buf = (kk - kbeg) & 1;
      if (kk + 1 < nk) dma_stage(buf ^ 1);
      compute_stage(buf);
      __syncthreads();
    }
  } else if constexpr (NBUF == 2) {
    __builtin_trap();
  } else {
    if (kbeg < nk) load_stage(kbeg);
    for (int kk = kbeg; kk < nk; ++kk) {
      store_stage(0);
      __syncthreads();
      if (kk + 1 < nk) load_stage(kk + 1);
      compute_stage(0);
      __syncthreads();
    }
  }

  if (p.ws != nullptr) {  // split-K: fp32 atomics into the workspace; bias/cast happen in xr_bias_cast
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + lr;
        if (col >= p.ldo) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = tile_m * BM + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) atomicAdd(p.ws + (size_t)m * p.ldo + col, acc[i][j][e]);
        }
      }
    return;
  }
  // epilogue: accumulators (+bias) -> LDS row image -> coalesced 16-B-chunk stores
  out_t* stage = reinterpret_cast<out_t*>(smem);
  constexpr int PITCH = BN + PADE;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn0 + j * 32 + lr;
      const float bv = (p.bias != nullptr && n0 + col < p.K) ? p.bias[n0 + col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        XrT<out_t>::st(stage + row * PITCH + col, acc[i][j][e] + bv);
      }
    }
  __syncthreads();
  out_t* __restrict__ out = reinterpret_cast<out_t*>(p.out);
  constexpr int CPR = BN / 8;  // chunks per tile row
  const int Kw = (p.K + 7) & ~7;
  const bool bnr = p.ep_red != nullptr;                    // BatchNorm-backward partial sums (ep_src = BN input)
  const bool ep = p.ep_src != nullptr && !bnr;   // PReLU backward; requires K % 8 == 0 (checked by the host)
  float bs0[8], bs1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs0[e] = 0.f, bs1[e] = 0.f;
  const out_t* __restrict__ ep_src = reinterpret_cast<const out_t*>(p.ep_src);
  float dal[8], alv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dal[e] = 0.f, alv[e] = 0.f;
  out_t* __restrict__ out2 = reinterpret_cast<out_t*>(p.ep2_out);
  const bool ep2 = p.ep2_out != nullptr;   // requires K % 8 == 0 (checked by the host)
  if ((ep || ep2) && n0 + (t % CPR) * 8 < p.K) ld8(p.ep_alpha + n0 + (t % CPR) * 8, alv);
  for (int idx = t; idx < BM * CPR; idx += NT) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int ncol = n0 + ch * 8;
    if (ncol >= Kw) continue;  // Kw = K rounded up to 8: padding channels are written as zeros
    int m;
    if (cls_mode) {
      bool v; int a, b, c;
      m = decode_row(row, v, a, b, c);
    } else {
      m = tile_m * BM + row;
      if (m >= p.M) m = -1;
    }
    if (m < 0) continue;
    const out_t* sp = stage + row * PITCH + ch * 8;
    size_t oo = (size_t)m * p.ldo + ncol;   // element offset of this chunk in out / ep_src / ep_add / ep2_out
    if (p.d2s_c != 0) {   // depth-to-space: column block -> sub-pixel (a, b) of the 2x up-sampled grid
      const int cl = fdiv(p.fd_d2s, ncol), cch = ncol - cl * p.d2s_c;
      const int n_ = fdiv(p.fd_howo, m), rem_ = m - n_ * (int)p.fd_howo.d;
      const int ho_ = fdiv(p.fd_wo, rem_), wo_ = rem_ - ho_ * (int)p.fd_wo.d;
      oo = ((size_t)(n_ * 2 * p.Ho + 2 * ho_ + (cl >> 1)) * (2 * p.Wo) + 2 * wo_ + (cl & 1)) * p.d2s_c + cch;
    }
    out_t* dp = out + oo;
    if (ep) {
      float d[8], yv[8], o[8];
      ld8(sp, d);
      ld8(ep_src + oo, yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool neg = yv[e] <= 0.f;
        o[e] = neg ? d[e] * alv[e] : d[e];
        if (neg) dal[e] += d[e] * yv[e];
      }
      st8(dp, o);
      continue;
    }
    if (bnr) {  // sums of d and d * x per channel (d as rounded for the output tensor)
      float d[8], xv[8];
      ld8(sp, d);
      if (ep_src != nullptr) {
        ld8(ep_src + oo, xv);
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
      st8(out2 + oo, o);
    }
    if (p.ep_add != nullptr) {   // gradient of an identity branch summed here instead of by a separate elementwise pass
      float d[8], av[8];
      ld8(sp, d);
      ld8(reinterpret_cast<const out_t*>(p.ep_add) + oo, av);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] += av[e];
      st8(dp, d);
      continue;
    }
    if (ncol + 8 <= Kw) {
      if constexpr (sizeof(out_t) == 2) {
        *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
      } else {
        *reinterpret_cast<float4*>(dp) = *reinterpret_cast<const float4*>(sp);
        *reinterpret_cast<float4*>(dp + 4) = *reinterpret_cast<const float4*>(sp + 4);
      }
    } else {
      for (int e = 0; e < 8 && ncol + e < Kw; ++e) dp[e] = sp[e];
    }
  }
  if (ep) {  // fold the per-thread dalpha partials: threads sharing a chunk column are CPR apart
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(t / CPR) * BN + (t % CPR) * 8 + e] = dal[e];
    __syncthreads();
    for (int c = t; c < BN; c += NT) {
      float sum = 0.f;
      for (int r = 0; r < NT / CPR; ++r) sum += red[r * BN + c];
      if (n0 + c < p.K) atomicAdd(p.ep_dalpha + (size_t)(tile_m % p.ep_spread) * p.K + n0 + c, sum);
    }
  }
  if (bnr) {  // same fold as dalpha, two vectors
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(t / CPR) * BN + (t % CPR) * 8 + e] = v ? bs1[e] : bs0[e];
      __syncthreads();
      for (int c = t; c < BN; c += NT) {
        float sum = 0.f;
        for (int r = 0; r < NT / CPR; ++r) sum += red[r * BN + c];
        if (n0 + c < p.K) atomicAdd(p.ep_red + ((size_t)v * p.ep_spread + (tile_m % p.ep_spread)) * p.K + n0 + c, sum);
      }
    }
  }
}

template <int MODE, int BM, int BN, int DMA, int BKT>
constexpr size_t igemm_smem() {
  constexpr int NS = ns_of(MODE);
  constexpr size_t ops = (size_t)NS * (BM + BN) * (BKT * 2) * (DMA ? DMA : 1);
  constexpr size_t esz = MODE ? 4 : 2;
  constexpr size_t stg = (size_t)BM * (BN + 16 / esz) * esz;
  return ops > stg ? ops : stg;
}

}  // namespace
XrTune g_tune[20] = {3, 1, 0, 0, 1, 1, 18, 1, 1024, 1024, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0};  // [0] igemm gather path: 0 generic, 1 FAST (register staging), 2 FAST + LDS-DMA, 3 auto; [2] != 0 disables wgrad FAST; [3] != 0: 128x64 tiles for every K; [4], [5] unused (s_setprio around the MFMA groups is unconditional); [6] DMA threshold (K stages); [7] 8-wave 256x256 kernel (xr_conv8.hip): 0 off, 1 auto, 2 whenever eligible; [8]/[9] reduce-kernel block targets (xr_norm.hip); [12] 8-wave kernel prefetch schedule: 1 = deep (5-6 phases ahead, default), 0 = shallow; [13] 8-wave ring weight gradient (xr_wgrad8.hip): 0 off, 1 auto (K >= 256), 2 whenever eligible; [11] wgrad LDS-DMA staging (0 off: faster on warm inputs in tools/conv_bench.py, slower inside the training step, where operands come from HBM), 1 auto, 2 always
namespace {

template <int MODE, int BM, int BN, int WM, bool TR, bool FAST, int DMA, int BKT = 64>
int launch_igemm_f(IgemmP& p, hipStream_t st);

template <int MODE, int BM, int BN, int WM, bool TR>
int launch_igemm(IgemmP& p, hipStream_t st) {
  const bool cls_ok = TR && p.stride > 1 && p.Ho % p.stride == 0 && p.Wo % p.stride == 0 && p.C % 64 == 0 && p.ws == nullptr;
  const long long in_bytes = (long long)p.N * p.H * p.W * p.C * (MODE ? 4 : 2);
  const long long w_bytes = (long long)ns_of(MODE) * p.K * p.Kg * 2;
  int ntaps = p.R * p.S;
  if (cls_ok) ntaps = ((p.R + p.stride - 1) / p.stride) * ((p.S + p.stride - 1) / p.stride);
  const bool fast = g_tune[0] && p.C % 64 == 0 && (!TR || p.stride == 1 || cls_ok) && ntaps <= 32 &&
                    in_bytes < (1ll << 31) && w_bytes < (1ll << 31) && p.Kg == p.R * p.S * p.C;
  p.in_bytes = (unsigned)(in_bytes < (1ll << 31) ? in_bytes : 0x7FFFFFFF);
  p.w_bytes = (unsigned)(w_bytes < (1ll << 31) ? w_bytes : 0x7FFFFFFF);
  if constexpr (MODE == 0) {
    // LDS-DMA staging pays once the K loop is long enough to amortise the lower occupancy (2 x 32 KB stages per WG):
    // measured cross-over between the 128- and 256-channel 3x3 layers (18 vs 36 stages)
    const bool dma = g_tune[0] == 2 || (g_tune[0] == 3 && p.Kg / BK >= g_tune[6]);
    if (fast && dma) return launch_igemm_f<MODE, BM, BN, WM, TR, true, 2>(p, st);
  }
  if (fast) return launch_igemm_f<MODE, BM, BN, WM, TR, true, 0>(p, st);
  return launch_igemm_f<MODE, BM, BN, WM, TR, false, 0>(p, st);
}

template <int MODE, int BM, int BN, int WM, bool TR, bool FAST, int DMA, int BKT>
int launch_igemm_f(IgemmP& p, hipStream_t st) {
  p.tiles_n = cdiv((p.K + 7) / 8 * 8, BN);
  int tiles_m = cdiv(p.M, BM);
  p.cls = 0; p.tpc = 0; p.Mc = 0;
  if (TR && p.stride > 1 && p.Ho % p.stride == 0 && p.Wo % p.stride == 0 && p.C % 64 == 0 && p.ws == nullptr) {
    p.cls = 1;
    p.Mc = p.N * (p.Ho / p.stride) * (p.Wo / p.stride);
    p.tpc = cdiv(p.Mc, BM);
    tiles_m = p.tpc * p.stride * p.stride;
  }
  p.fd_howo = make_fd((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fd((unsigned)p.Wo);
  p.fd_c = make_fd((unsigned)p.C);
  p.fd_s = make_fd((unsigned)p.S);
  p.fd_hqwq = make_fd((unsigned)((p.Ho / p.stride) * (p.Wo / p.stride) > 0 ? (p.Ho / p.stride) * (p.Wo / p.stride) : 1));
  p.fd_wq = make_fd((unsigned)(p.Wo / p.stride > 0 ? p.Wo / p.stride : 1));
  p.fd_tn = make_fd((unsigned)p.tiles_n);
  p.fd_st = make_fd((unsigned)p.stride);
  p.fd_d2s = make_fd((unsigned)(p.d2s_c > 0 ? p.d2s_c : 1));
  int gy = 1;
  if (p.ws != nullptr) {
    const int nk = p.Kg / BKT;
    gy = p.ksplit;                 // requested number of slices
    if (gy > nk) gy = nk;
    p.ksplit = cdiv(nk, gy);       // K-stages per slice
    gy = cdiv(nk, p.ksplit);
  } else {
    p.ksplit = 0;
  }
  constexpr size_t smem = igemm_smem<MODE, BM, BN, DMA, BKT>();
  auto kern = igemm_kernel<MODE, BM, BN, WM, TR, FAST, DMA, BKT>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) {
      xr_set_error("xr_conv_igemm: hipFuncSetAttribute(%zu) failed: %s", smem, hipGetErrorString(e));
      return XR_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles_m * p.tiles_n, (unsigned)gy), dim3(NT), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv_igemm");
  return XR_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// (WgradP: xr_conv_p.h)

// transposing fragment fetch from a [pixel][col] LDS image (pitch bytes): 8 pixels x 1 column per lane
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* img, int pitch, int pix0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, pp = i & 3;
  const int cgrp = g & 1, h = g >> 1;
  const int byte = (pix0 + 8 * h + q) * pitch + (col0 + 16 * cgrp + 4 * pp) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(img + byte));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(img + byte + 4 * pitch));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// same fetch from an unpadded image (row = R bytes, a power of two >= 128) whose 16-B chunks were XOR-swizzled on the way in
// (LDS-DMA writes lane-linear rows, so the swizzle lives in the DMA source chunk and here): one ds_read_b64_tr_b16 half
// (32 lanes) touches 4 pixel rows x 64 B, and the swizzle puts those four 64-B segments on distinct quarters of the 256-B
// bank row -- R >= 256: chunk ^= (row & 3) << 2;  R == 128: chunk ^= ((row >> 1) & 1) << 2
template <int R>
__device__ __forceinline__ int wg_swz(int row) { return R >= 256 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; }
template <int R>
__device__ __forceinline__ bf16x8_t tr_frag_sw(const unsigned char* img, int pix0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, pp = i & 3;
  const int cgrp = g & 1, h = g >> 1;
  const int row = pix0 + 8 * h + q;                 // row + 4 has the same swizzle term
  const int cb = (col0 + 16 * cgrp + 4 * pp) * 2;   // byte column of this lane's 8-B packet
  const int byte = row * R + ((((cb >> 4) ^ wg_swz<R>(row)) << 4) | (cb & 15));
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(img + byte));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(img + byte + 4 * R));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// FAST (forward-gather only): the per-slot pixel cursor (n, ho, wo) is advanced incrementally by the scalar
// decomposition of 64 pixels instead of being re-derived with two divisions per row per step, and both operands
// come in through buffer loads whose range check supplies the zero padding.
template <int MODE, int BR, int BC, bool TR, int NBUF, bool FAST>
__global__ __launch_bounds__(NT) void wgrad_kernel(WgradP p) {
  constexpr int BP = 64;  // pixels per stage
  constexpr int NS = ns_of(MODE);
  constexpr int WC = BC / 64, WR = BR / 64;  // wave grid (wave tile 64x64)
  static_assert(WC * WR == 4, "4 waves");
  // NBUF == 2 on the bf16 FAST path = LDS-DMA staging: both images go global -> LDS by buffer_load ... lds (no staging
  // VGPRs, no ds_write -- the LDS write port, ~80 B/clk, is the busiest unit of the register-staged variant), double
  // buffered, unpadded rows with the chunk swizzle of tr_frag_sw
  constexpr bool DMA = NBUF == 2 && FAST && MODE == 0;
  constexpr int PY = DMA ? BR * 2 : BR * 2 + 64, PX = DMA ? BC * 2 : BC * 2 + 64;  // padded pitches: == 64 (mod 256) -> conflict-free tr reads
  constexpr int CRY = BR / 8, CRX = BC / 8;          // chunks per image row
  constexpr int NY = BP * CRY / NT, NX = BP * CRX / NT;
  using in_t = typename std::conditional<MODE != 0, float, bf16_t>::type;

  constexpr int STAGE = NS * BP * (PY + PX);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // 1-D grid, XCD-aware order: each XCD gets a contiguous run of (slice, tile) work items, so the tiles of one pixel
  // slice -- which all re-read the same dY / X rows -- meet in one L2 instead of being dealt over all eight
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int slice = bid / p.tiles_all, tile = bid - slice * p.tiles_all;
  const int tile_c = tile % p.tiles_c, tile_r = tile / p.tiles_c;
  const int r0 = tile_r * BR, c0 = tile_c * BC;
  const int HW = p.H * p.W;
  const in_t* __restrict__ in = reinterpret_cast<const in_t*>(p.in);
  const in_t* __restrict__ dy = reinterpret_cast<const in_t*>(p.dy);

  // X-gather column decode: fixed per thread for the whole kernel (with DMA staging the thread's LDS slot is fixed and it
  // fetches the source chunk that the swizzle maps into that slot; the row term is the same for all of a thread's rows)
  const int xrow0 = t / CRX, yrow0 = t / CRY;
  const int xch = DMA ? (t % CRX) ^ wg_swz<PX>(xrow0) : t % CRX;
  const int ych = DMA ? (t % CRY) ^ wg_swz<PY>(yrow0) : t % CRY;
  const int j0 = c0 + xch * 8;
  const int tap = j0 / p.C;
  const int cch = j0 - tap * p.C;
  const int tr_ = tap / p.S, ts_ = tap - tr_ * p.S;
  const bool col_ok = (tap < p.R * p.S) && (j0 < p.Kg);
  // cursor constants: per-stage advance of the input row / column, wrap limits of this thread's tap, wrap spans
  const int adv_w = p.c64 * p.stride, adv_h = p.b64 * p.stride, span_w = p.Wo * p.stride, span_h = p.Ho * p.stride;
  const int wlim = span_w + ts_ - p.pad, hlim = span_h + tr_ - p.pad;
  const bool ycol_ok = (r0 + ych * 8) < p.ldy;

  float y_f[MODE ? NY : 1][8], x_f[MODE ? NX : 1][8];
  uint4 y_u[MODE ? 1 : NY], x_u[MODE ? 1 : NX];

  // ---- FAST path state
  constexpr int ESZ = (int)sizeof(in_t);
  // per-slot gather cursor kept in the form the load needs: byte offset of the tap's input pixel (xl), its input row /
  // column (xhv, xwv) for the bounds test; a 64-pixel advance is three adds plus two conditional wrap corrections with
  // launch-uniform deltas (was: re-deriving the address from (n, ho, wo) with two multiplies per slot and stage)
  int xl[FAST ? NX : 1], xhv[FAST ? NX : 1], xwv[FAST ? NX : 1];
  unsigned y_base[FAST ? NY : 1];
  __amdgpu_buffer_rsrc_t rsX, rsY;
  if constexpr (FAST) {
    rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
    rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NY; ++i)
      y_base[i] = ycol_ok ? (unsigned)(((yrow0 + (NT / CRY) * i) * p.ldy + r0 + ych * 8) * ESZ) : XR_OOR;
  }
  auto init_cursor = [&](int step) {
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int m = step * BP + xrow0 + (NT / CRX) * i;
        const int n = fdiv(p.fd_howo, m);
        const int rem = m - n * (int)p.fd_howo.d;
        const int ho = fdiv(p.fd_wo, rem);
        const int wo = rem - ho * (int)p.fd_wo.d;
        xhv[i] = ho * p.stride - p.pad + tr_;
        xwv[i] = wo * p.stride - p.pad + ts_;
        xl[i] = (((n * p.H + xhv[i]) * p.W + xwv[i]) * p.C + cch) * ESZ;
      }
    }
  };
  auto load_stage_fast = [&](int step) {
    const int mbase = step * BP;
    const unsigned ysoff = (unsigned)mbase * (unsigned)(p.ldy * ESZ);
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int m = mbase + yrow0 + (NT / CRY) * i;
      const unsigned voff = m < p.M ? y_base[i] : XR_OOR;
      if constexpr (MODE != 0) {
        const v4u_t u0 = __builtin_amdgcn_raw_buffer_load_b128(rsY, voff, ysoff, 0);
        const v4u_t u1 = __builtin_amdgcn_raw_buffer_load_b128(rsY, voff + 16u, ysoff, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { y_f[i][e] = __uint_as_float(u0[e]); y_f[i][4 + e] = __uint_as_float(u1[e]); }
      } else {
        const v4u_t u = __builtin_amdgcn_raw_buffer_load_b128(rsY, voff, ysoff, 0);
        y_u[i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const bool ok = col_ok && mbase + xrow0 + (NT / CRX) * i < p.M && (unsigned)xhv[i] < (unsigned)p.H &&
                      (unsigned)xwv[i] < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)xl[i] : XR_OOR;
      if constexpr (MODE != 0) {
        const v4u_t u0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, voff, 0, 0);
        const v4u_t u1 = __builtin_amdgcn_raw_buffer_load_b128(rsX, voff + 16u, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x_f[i][e] = __uint_as_float(u0[e]); x_f[i][4 + e] = __uint_as_float(u1[e]); }
      } else {
        const v4u_t u = __builtin_amdgcn_raw_buffer_load_b128(rsX, voff, 0, 0);
        x_u[i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
      // advance this slot's pixel cursor by 64 pixels
      int wv = xwv[i] + adv_w, hv = xhv[i] + adv_h, l = xl[i] + p.d64;
      if (wv >= wlim) { wv -= span_w; hv += p.stride; l += p.dwrap_w; }
      if (hv >= hlim) { hv -= span_h; l += p.dwrap_h; }
      xwv[i] = wv; xhv[i] = hv; xl[i] = l;
    }
  };

  auto dma_stage = [&](int step, int buf) {
    if constexpr (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)
      typedef __attribute__((address_space(3))) void* lds_ptr_t;
      unsigned char* sY = smem + buf * STAGE;
      unsigned char* sX = sY + BP * PY;
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      const int mbase = step * BP;
      const unsigned ysoff = (unsigned)mbase * (unsigned)(p.ldy * ESZ);
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int m = mbase + yrow0 + (NT / CRY) * i;
        const unsigned voff = m < p.M ? y_base[i] : XR_OOR;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr_t)(sY + (wv * (64 / CRY) + (NT / CRY) * i) * PY), 16, voff, ysoff, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const bool ok = col_ok && mbase + xrow0 + (NT / CRX) * i < p.M && (unsigned)xhv[i] < (unsigned)p.H &&
                        (unsigned)xwv[i] < (unsigned)p.W;
        const unsigned voff = ok ? (unsigned)xl[i] : XR_OOR;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(sX + (wv * (64 / CRX) + (NT / CRX) * i) * PX), 16, voff, 0, 0, 0);
        int wv = xwv[i] + adv_w, hv = xhv[i] + adv_h, l = xl[i] + p.d64;
        if (wv >= wlim) { wv -= span_w; hv += p.stride; l += p.dwrap_w; }
        if (hv >= hlim) { hv -= span_h; l += p.dwrap_h; }
        xwv[i] = wv; xhv[i] = hv; xl[i] = l;
      }
#else
      (void)step; (void)buf;
#endif
    }
  };

  auto load_stage = [&](int step) {
    if constexpr (FAST) {
      load_stage_fast(step);
      return;
    }
    const int mbase = step * BP;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int m = mbase + yrow0 + (NT / CRY) * i;
      const bool ok = ycol_ok && m < p.M;
      const in_t* src = dy + ((size_t)m * p.ldy + r0 + ych * 8);
      if constexpr (MODE != 0) {
        if (ok) ld8(src, y_f[i]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) y_f[i][e] = 0.f;
        }
      } else {
        y_u[i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = mbase + xrow0 + (NT / CRX) * i;
      bool valid; int nb, oh0, ow0, pix = 0;
      decode_pixel<TR>(m, p.M, p.fd_howo, p.fd_wo, HW, p.stride, p.pad, valid, nb, oh0, ow0);
      const bool ok = valid && col_ok && tap_coord<TR>(oh0, ow0, tr_, ts_, p.stride, p.H, p.W, pix);
      const in_t* src = in + ((size_t)(nb + pix) * p.C + cch);
      if constexpr (MODE != 0) {
        if (ok) ld8(src, x_f[i]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) x_f[i][e] = 0.f;
        }
      } else {
        x_u[i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sY = smem + buf * STAGE;
    unsigned char* sX = sY + NS * BP * PY;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int off = (yrow0 + (NT / CRY) * i) * PY + ych * 16;
      if constexpr (MODE != 0) {
        uint4 pl[NS];
        split8<NS>(y_f[i], pl);
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint4*>(sY + q * BP * PY + off) = pl[q];
      } else {
        *reinterpret_cast<uint4*>(sY + off) = y_u[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int off = (xrow0 + (NT / CRX) * i) * PX + xch * 16;
      if constexpr (MODE != 0) {
        uint4 pl[NS];
        split8<NS>(x_f[i], pl);
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint4*>(sX + q * BP * PX + off) = pl[q];
      } else {
        *reinterpret_cast<uint4*>(sX + off) = x_u[i];
      }
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wr0 = (wave / WC) * 64, wc0 = (wave % WC) * 64;
  const int s_begin = slice * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > p.steps_total) s_end = p.steps_total;

  auto compute_stage = [&](int buf) {
    const unsigned char* sY = smem + buf * STAGE;
    const unsigned char* sX = sY + NS * BP * PY;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t fa[2][NS], fb[2][NS];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fa[i][s] = DMA ? tr_frag_sw<PY>(sY, ks * 16, wr0 + i * 32, lane) : tr_frag(sY + s * BP * PY, PY, ks * 16, wr0 + i * 32, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fb[j][s] = DMA ? tr_frag_sw<PX>(sX, ks * 16, wc0 + j * 32, lane) : tr_frag(sX + s * BP * PX, PX, ks * 16, wc0 + j * 32, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_split<NS>(fa[i], fb[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  if (s_begin < s_end) {
    init_cursor(s_begin);
    if constexpr (DMA) {
      // stage s lives in buffer (s - s_begin) & 1; the DMA of stage s+1 is issued before the MFMAs of stage s, and the
      // barrier (hipcc drains vmcnt(0) in front of it because LDS-DMA is pending) orders RAW and WAR
      dma_stage(s_begin, 0);
      __syncthreads();
      for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        if (step + 1 < s_end) dma_stage(step + 1, buf ^ 1);
        compute_stage(buf);
        __syncthreads();
      }
    } else if constexpr (NBUF == 2) {
      load_stage(s_begin);
      store_stage(0);
      __syncthreads();
      for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        if (step + 1 < s_end) load_stage(step + 1);
        compute_stage(buf);
        if (step + 1 < s_end) store_stage(buf ^ 1);
        __syncthreads();
      }
    } else {
      load_stage(s_begin);
      for (int step = s_begin; step < s_end; ++step) {
        store_stage(0);
        __syncthreads();
        if (step + 1 < s_end) load_stage(step + 1);
        compute_stage(0);
        __syncthreads();
      }
    }
  }

  // each pixel-range slice owns a private [K][Kg] slab: plain coalesced stores (128 B per accumulator row), no
  // atomics -- the slices are summed by xr_unpack_wgrad while it converts to the parameter layout
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

template <int MODE, int BR, int BC, bool TR, int NBUF, bool FAST>
int launch_wgrad_nb(WgradP& p, int split, hipStream_t st);

template <int MODE, int BR, int BC, bool TR>
int launch_wgrad(WgradP& p, int split, hipStream_t st) {
  const long long esz = MODE ? 4 : 2;
  const long long in_bytes = (long long)p.N * p.H * p.W * p.C * esz, dy_bytes = (long long)p.M * p.ldy * esz;
  const bool fast = !TR && g_tune[2] == 0 && in_bytes < (1ll << 31) && dy_bytes < (1ll << 31);
  p.in_bytes = (unsigned)(in_bytes < (1ll << 31) ? in_bytes : 0x7FFFFFFF);
  p.dy_bytes = (unsigned)(dy_bytes < (1ll << 31) ? dy_bytes : 0x7FFFFFFF);
  const int howo = p.Ho * p.Wo;
  p.a64 = 64 / howo;
  p.b64 = (64 % howo) / p.Wo;
  p.c64 = (64 % howo) % p.Wo;
  p.d64 = (int)(((long long)p.a64 * p.H * p.W + (long long)p.b64 * p.stride * p.W + (long long)p.c64 * p.stride) * p.C * esz);
  p.dwrap_w = (int)(((long long)p.stride * p.W - (long long)p.Wo * p.stride) * p.C * esz);
  p.dwrap_h = (int)(((long long)p.H * p.W - (long long)p.Ho * p.stride * p.W) * p.C * esz);
  if constexpr (!TR) {
    if constexpr (MODE == 0) {
      // LDS-DMA staging: measured per layer shape (tools/conv_bench.py) -- pays on the stride-1 gathers with >= 128 input
      // channels and on all >= 256-channel layers, loses on the strided / 64-channel gathers except 64 -> 64 stride 1
      const bool dma = g_tune[11] == 2 ||
                       (g_tune[11] == 1 && (p.C >= 256 || (p.stride == 1 && (p.C >= 128 || p.K <= 64))));
      if (fast && dma) return launch_wgrad_nb<MODE, BR, BC, TR, 2, true>(p, split, st);
    }
    if (fast) return launch_wgrad_nb<MODE, BR, BC, TR, 1, true>(p, split, st);
  }
  return launch_wgrad_nb<MODE, BR, BC, TR, 1, false>(p, split, st);
}

template <int MODE, int BR, int BC, bool TR, int NBUF, bool FAST>
int launch_wgrad_nb(WgradP& p, int split, hipStream_t st) {
  constexpr int NS = ns_of(MODE);
  constexpr bool DMA = NBUF == 2 && FAST && MODE == 0;
  constexpr size_t smem = DMA ? (size_t)2 * 64 * (BR * 2 + BC * 2) : (size_t)NS * 64 * ((BR * 2 + 64) + (BC * 2 + 64)) * NBUF;
  p.fd_howo = make_fd((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fd((unsigned)p.Wo);
  p.tiles_c = cdiv(p.Kg, BC);
  const int tiles_r = cdiv(p.K, BR);
  p.steps_total = cdiv(p.M, 64);
  if (split < 1) split = 1;
  if (split > p.steps_total) split = p.steps_total;
  p.steps_per_split = cdiv(p.steps_total, split);
  split = cdiv(p.steps_total, p.steps_per_split);
  auto kern = wgrad_kernel<MODE, BR, BC, TR, NBUF, FAST>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) {
      xr_set_error("xr_conv_wgrad: hipFuncSetAttribute(%zu) failed: %s", smem, hipGetErrorString(e));
      return XR_E_LAUNCH;
    }
  }
  p.tiles_all = p.tiles_c * tiles_r;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_all * split)), dim3(NT), smem, st, p);
  XR_CHECK_LAUNCH("xr_conv_wgrad");
  return split;  // number of slabs written (>= 1)
}

// ------------------------------------------------------------------------------------------------ pack / unpack
__global__ void pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes, int A2,
                                   int taps, int B, int Bp, int Kg, int64_t sa1, int64_t sa2, int64_t st_, int64_t sb,
                                   int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = i / Kg;
    const int j = (int)(i - a * Kg);
    const int tp = j / Bp, b = j - tp * Bp;
    float v = 0.f;
    if (tp < taps && b < B) {
      const int64_t a1 = a / A2, a2 = a - a1 * A2;
      v = src[a1 * sa1 + a2 * sa2 + tp * st_ + b * sb];
    }
    for (int q = 0; q < nplanes; ++q) {
      const bf16_t h = f2bf(v);
      dst[(int64_t)q * total + i] = h;
      v -= bf2f(h);
    }
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ packed, float* __restrict__ dst, int A2, int taps, int B,
                                    int Bp, int Kg, int64_t sa1, int64_t sa2, int64_t st_, int64_t sb, int accumulate,
                                    int nslices, int64_t slice_stride, int64_t total, int spg) {
  // blockIdx.y selects a group of `spg` slices; with more than one group the groups meet in dst by atomicAdd
  const int s_beg = blockIdx.y * spg;
  const int s_end = (s_beg + spg < nslices) ? s_beg + spg : nslices;
  const bool atomic = gridDim.y > 1;
  // iterate over destination-meaningful elements (a, tap, b)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    const int64_t r = i / B;
    const int tp = (int)(r % taps);
    const int64_t a = r / taps;
    const int64_t a1 = a / A2, a2 = a - a1 * A2;
    const float* src = packed + a * Kg + (int64_t)tp * Bp + b;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int sl = s_beg;
    for (; sl + 3 < s_end; sl += 4) {  // four independent loads in flight
      v0 += src[sl * slice_stride];
      v1 += src[(sl + 1) * slice_stride];
      v2 += src[(sl + 2) * slice_stride];
      v3 += src[(sl + 3) * slice_stride];
    }
    for (; sl < s_end; ++sl) v0 += src[sl * slice_stride];
    const float v = (v0 + v1) + (v2 + v3);
    float* d = dst + a1 * sa1 + a2 * sa2 + tp * st_ + b * sb;
    if (atomic) atomicAdd(d, v);
    else *d = (accumulate & 1) ? (*d + v) : v;
  }
}

// B % 4 == 0: one thread sums four consecutive packed columns over its slice group with 16-B loads (four slices in
// flight).  blockIdx.y selects a group of `spg` slices; with more than one group the groups meet in dst by atomicAdd
// (dst then holds the running gradient or was zeroed by the launcher).
__global__ void unpack_wgrad4_kernel(const float* __restrict__ packed, float* __restrict__ dst, int A2, int taps, int B4,
                                     int Bp, int Kg, int64_t sa1, int64_t sa2, int64_t st_, int64_t sb, int accumulate,
                                     int nslices, int64_t slice_stride, int64_t total4, int spg) {
  const int s_beg = blockIdx.y * spg;
  const int s_end = (s_beg + spg < nslices) ? s_beg + spg : nslices;
  const bool atomic = gridDim.y > 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B4) * 4;
    const int64_t r = i / B4;
    const int tp = (int)(r % taps);
    const int64_t a = r / taps;
    const int64_t a1 = a / A2, a2 = a - a1 * A2;
    const float* src = packed + a * Kg + (int64_t)tp * Bp + b;
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int sl = s_beg;
    for (; sl + 3 < s_end; sl += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(src + (sl + q) * slice_stride);
        v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;
      }
    }
    for (; sl < s_end; ++sl) {
      const float4 x = *reinterpret_cast<const float4*>(src + sl * slice_stride);
      v[0].x += x.x; v[0].y += x.y; v[0].z += x.z; v[0].w += x.w;
    }
    const float o[4] = {(v[0].x + v[1].x) + (v[2].x + v[3].x), (v[0].y + v[1].y) + (v[2].y + v[3].y),
                        (v[0].z + v[1].z) + (v[2].z + v[3].z), (v[0].w + v[1].w) + (v[2].w + v[3].w)};
    float* d = dst + a1 * sa1 + a2 * sa2 + tp * st_ + (int64_t)b * sb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (atomic) atomicAdd(d + e * sb, o[e]);
      else d[e * sb] = (accumulate & 1) ? (d[e * sb] + o[e]) : o[e];
    }
  }
}

// ---- LDS-tiled layout converters for the two conv-shaped cases (both sides of the transpose stay coalesced) ----------
// (1) parameter [A][B][taps] (taps contiguous: Conv2d [K][C][R][S], Linear [K][C*HW])  <->  pack [A][tap][Bp]
//     block = (row a, 64 consecutive b): 64*taps contiguous floats on the parameter side, 64-wide rows on the pack side
__device__ __forceinline__ void pack_fwdform_body(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes, int taps,
                                                  int B, int Bp, int Kg, int64_t plane, int bx, int by, float* tile) {
  const int a = by, b0 = bx * 64, t = threadIdx.x;
  const int nb = (B - b0) < 64 ? (B - b0) : 64;
  const float* sp = src + ((int64_t)a * B + b0) * taps;
  for (int j = t; j < nb * taps; j += 256) tile[j] = sp[j];
  __syncthreads();
  const int nbp = (Bp - b0) < 64 ? (Bp - b0) : 64;  // includes the zero padding columns b in [B, Bp)
  for (int e = t; e < 64 * taps; e += 256) {
    const int tp = e >> 6, bl = e & 63;
    if (bl >= nbp) continue;
    float v = bl < nb ? tile[bl * taps + tp] : 0.f;
    bf16_t* dp = dst + (int64_t)a * Kg + (int64_t)tp * Bp + b0 + bl;
    for (int q = 0; q < nplanes; ++q) {
      const bf16_t h = f2bf(v);
      dp[(int64_t)q * plane] = h;
      v -= bf2f(h);
    }
  }
  if (bx == 0)
    for (int j = taps * Bp + t; j < Kg; j += 256)
      for (int q = 0; q < nplanes; ++q) dst[(int64_t)q * plane + (int64_t)a * Kg + j] = 0;
}
__global__ __launch_bounds__(256) void pack_fwdform_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes,
                                                           int taps, int B, int Bp, int Kg, int64_t plane) {
  extern __shared__ float tile[];
  pack_fwdform_body(src, dst, nplanes, taps, B, Bp, Kg, plane, blockIdx.x, blockIdx.y, tile);
}

__global__ __launch_bounds__(256) void unpack_fwdform_kernel(const float* __restrict__ packed, float* __restrict__ dst, int taps,
                                                             int B, int Bp, int Kg, int accumulate, int nslices,
                                                             int64_t slice_stride, int spg) {
  // block = (row a, 64 consecutive b, slice group z): the group's slices are summed with four independent loads in flight
  // (coalesced 256-B rows of the packed layout), transposed through LDS, and leave as one contiguous run of 64 * taps floats
  // of the parameter layout.  With more than one slice group the groups meet in dst by atomicAdd (dst then holds the running
  // gradient or was zeroed by the launcher).
  extern __shared__ float tile[];
  const int a = blockIdx.y, b0 = blockIdx.x * 64, t = threadIdx.x;
  const int s_beg = blockIdx.z * spg;
  const int s_end = (s_beg + spg < nslices) ? s_beg + spg : nslices;
  const bool atomic = gridDim.z > 1;
  const int nb = (B - b0) < 64 ? (B - b0) : 64;
  // (a 16-byte-load form of this loop -- 16 lanes per tap row, 16 tap rows per pass -- was 12 % faster in isolation, 20.4 -> 17.9 us
  // cold, and 70-80 % SLOWER inside the FHN steps, where it runs beside the persistent direct weight-gradient workgroups: fewer
  // active lanes per block hide less latency under contention.  Kept: one 4-byte element per lane, four slices in flight.)
  for (int e = t; e < 64 * taps; e += 256) {
    const int tp = e >> 6, bl = e & 63;
    if (bl >= nb) continue;
    const float* sp = packed + (int64_t)a * Kg + (int64_t)tp * Bp + b0 + bl;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int sl = s_beg;
    for (; sl + 3 < s_end; sl += 4) {
      v0 += sp[sl * slice_stride];
      v1 += sp[(sl + 1) * slice_stride];
      v2 += sp[(sl + 2) * slice_stride];
      v3 += sp[(sl + 3) * slice_stride];
    }
    for (; sl < s_end; ++sl) v0 += sp[sl * slice_stride];
    tile[bl * taps + tp] = (v0 + v1) + (v2 + v3);
  }
  __syncthreads();
  float* dp = dst + ((int64_t)a * B + b0) * taps;
  if (atomic) {
    for (int j = t; j < nb * taps; j += 256) atomicAdd(dp + j, tile[j]);
  } else {
    for (int j = t; j < nb * taps; j += 256) dp[j] = accumulate ? dp[j] + tile[j] : tile[j];
  }
}

// (2) parameter [B][A][taps] -> pack [A][tap][Bp]  (the input-gradient pack of a Conv2d: rows = input channel a,
//     columns = (tap, output channel b)): block = (4 rows a, 64 columns b)
__device__ __forceinline__ void pack_dgradform_body(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes, int taps,
                                                    int A, int B, int Bp, int Kg, int64_t plane, int bx, int by, float* tile) {
  // tile: [64][4*taps + 1]
  const int a0 = by * 4, b0 = bx * 64, t = threadIdx.x;
  const int na = (A - a0) < 4 ? (A - a0) : 4, nb = (B - b0) < 64 ? (B - b0) : 64;
  const int rowlen = na * taps, pitch = 4 * taps + 1;
  for (int e = t; e < 64 * rowlen; e += 256) {
    const int bl = e / rowlen, r = e - bl * rowlen;
    if (bl < nb) tile[bl * pitch + r] = src[((int64_t)(b0 + bl) * A + a0) * taps + r];
  }
  __syncthreads();
  const int nbp = (Bp - b0) < 64 ? (Bp - b0) : 64;
  for (int e = t; e < na * taps * 64; e += 256) {
    const int bl = e & 63, rt = e >> 6;      // rt = al * taps + tp
    if (bl >= nbp) continue;
    const int al = rt / taps, tp = rt - al * taps;
    float v = bl < nb ? tile[bl * pitch + rt] : 0.f;
    bf16_t* dp = dst + (int64_t)(a0 + al) * Kg + (int64_t)tp * Bp + b0 + bl;
    for (int q = 0; q < nplanes; ++q) {
      const bf16_t h = f2bf(v);
      dp[(int64_t)q * plane] = h;
      v -= bf2f(h);
    }
  }
  if (bx == 0)
    for (int al = 0; al < na; ++al)
      for (int j = taps * Bp + t; j < Kg; j += 256)
        for (int q = 0; q < nplanes; ++q) dst[(int64_t)q * plane + (int64_t)(a0 + al) * Kg + j] = 0;
}
__global__ __launch_bounds__(256) void pack_dgradform_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes,
                                                             int taps, int A, int B, int Bp, int Kg, int64_t plane) {
  extern __shared__ float tile[];
  pack_dgradform_body(src, dst, nplanes, taps, A, B, Bp, Kg, plane, blockIdx.x, blockIdx.y, tile);
}

// ---- batched pack: one launch refreshes every registered parameter pack (xr_pack_plan / xr_pack_run) ------------------
struct PackEntry {          // 128 bytes, device table
  const float* src;
  bf16_t* dst;
  int64_t sa1, sa2, st, sb, total;
  int nplanes, A1, A2, taps, B, Bp, Kg, form;   // form: 0 generic, 1 fwd-form, 2 dgrad-form
  int blk0, nblk, gx, pad_;
  int64_t pad2_[3];
};
static_assert(sizeof(PackEntry) == 128, "PackEntry layout");

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackEntry* __restrict__ table, int n) {
  extern __shared__ float tile[];
  int lo = 0, hi = n - 1;  // last entry with blk0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = table[lo];
  const int lb = blockIdx.x - e.blk0;
  if (e.form == 1) {
    pack_fwdform_body(e.src, e.dst, e.nplanes, e.taps, e.B, e.Bp, e.Kg, e.total, lb % e.gx, lb / e.gx, tile);
  } else if (e.form == 2) {
    pack_dgradform_body(e.src, e.dst, e.nplanes, e.taps, e.A1, e.B, e.Bp, e.Kg, e.total, lb % e.gx, lb / e.gx, tile);
  } else {
    for (int64_t i = lb * 256ll + threadIdx.x; i < e.total; i += e.nblk * 256ll) {
      const int64_t a = i / e.Kg;
      const int j = (int)(i - a * e.Kg);
      const int tp = j / e.Bp, b = j - tp * e.Bp;
      float v = 0.f;
      if (tp < e.taps && b < e.B) {
        const int64_t a1 = a / e.A2, a2 = a - a1 * e.A2;
        v = e.src[a1 * e.sa1 + a2 * e.sa2 + tp * e.st + b * e.sb];
      }
      for (int q = 0; q < e.nplanes; ++q) {
        const bf16_t h = f2bf(v);
        e.dst[(int64_t)q * e.total + i] = h;
        v -= bf2f(h);
      }
    }
  }
}

// (3) Linear input-gradient pack: dst[(p*C + c)][k] = W[k][c*HW + p]  (rows in NHWC-flatten order, k contiguous).
//     block = (64 k, 8 c): each k contributes 8*HW contiguous source floats; each (p, c) row gets 64 contiguous k
__global__ __launch_bounds__(256) void pack_lindgrad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int nplanes,
                                                            int HW, int C, int K, int Bp, int Kg, int64_t plane) {
  extern __shared__ float tile[];  // [64][8*HW + 1]
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 8, t = threadIdx.x;
  const int nk = (K - k0) < 64 ? (K - k0) : 64, nc = (C - c0) < 8 ? (C - c0) : 8;
  const int rowlen = nc * HW, pitch = 8 * HW + 1;
  for (int e = t; e < 64 * rowlen; e += 256) {
    const int kl = e / rowlen, r = e - kl * rowlen;
    if (kl < nk) tile[kl * pitch + r] = src[((int64_t)(k0 + kl) * C + c0) * HW + r];
  }
  __syncthreads();
  const int nkp = (Bp - k0) < 64 ? (Bp - k0) : 64;
  for (int e = t; e < rowlen * 64; e += 256) {
    const int kl = e & 63, r = e >> 6;   // r = cl*HW + p
    if (kl >= nkp) continue;
    const int cl = r / HW, pp = r - cl * HW;
    float v = kl < nk ? tile[kl * pitch + r] : 0.f;
    bf16_t* dp = dst + ((int64_t)pp * C + c0 + cl) * Kg + k0 + kl;
    for (int q = 0; q < nplanes; ++q) {
      const bf16_t h = f2bf(v);
      dp[(int64_t)q * plane] = h;
      v -= bf2f(h);
    }
  }
  if (blockIdx.x == 0)  // zero tail columns [Bp, Kg) of the rows this block owns
    for (int r = 0; r < rowlen; ++r) {
      const int cl = r / HW, pp = r - cl * HW;
      for (int j = Bp + t; j < Kg; j += 256)
        for (int q = 0; q < nplanes; ++q) dst[(int64_t)q * plane + ((int64_t)pp * C + c0 + cl) * Kg + j] = 0;
    }
}

template <typename T>
__global__ void bias_cast_kernel(const float* __restrict__ ws, const float* __restrict__ bias, T* __restrict__ out, int64_t M,
                                 int K, int ld) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M * ld; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld);
    float v = ws[i];
    if (bias != nullptr && c < K) v += bias[c];
    XrT<T>::st(out + i, c < ((K + 7) & ~7) ? v : XrT<T>::ld(out + i));
  }
}

}  // namespace

extern "C" int xr_pack_plan(const int64_t* entries, int n, void* table_dev, int* smem_out, void* stream) {
  XR_CHECK_ARG(entries && table_dev && smem_out && n > 0, "xr_pack_plan: null pointer or empty plan");
  std::vector<PackEntry> tab((size_t)n);
  int blk = 0, smem = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t* f = entries + (size_t)i * 14;
    PackEntry& e = tab[(size_t)i];
    memset(&e, 0, sizeof(e));
    e.src = reinterpret_cast<const float*>((uintptr_t)f[0]);
    e.dst = reinterpret_cast<bf16_t*>((uintptr_t)f[1]);
    e.nplanes = (int)f[2]; e.A1 = (int)f[3]; e.A2 = (int)f[4]; e.taps = (int)f[5]; e.B = (int)f[6]; e.Bp = (int)f[7]; e.Kg = (int)f[8];
    e.sa1 = f[9]; e.sa2 = f[10]; e.st = f[11]; e.sb = f[12];
    XR_CHECK_ARG(e.src && e.dst && (e.nplanes >= 1 && e.nplanes <= 3) && e.A1 > 0 && e.A2 > 0 && e.taps > 0 && e.B > 0 &&
                     e.Bp >= e.B && e.Bp % 8 == 0 && e.Kg % 64 == 0 && e.Kg >= e.taps * e.Bp,
                 "xr_pack_plan: bad entry %d", i);
    e.total = (int64_t)e.A1 * e.A2 * e.Kg;
    int need = 0;
    e.form = 0;
    if (e.A2 == 1 && e.st == 1 && e.taps <= 64) {
      if (e.sb == e.taps && e.sa1 == (int64_t)e.B * e.taps) {
        e.form = 1; e.gx = cdiv(e.Bp, 64); e.nblk = e.gx * e.A1; need = 64 * e.taps * 4;
      } else if (e.sa1 == e.taps && e.sb == (int64_t)e.A1 * e.taps) {
        e.form = 2; e.gx = cdiv(e.Bp, 64); e.nblk = e.gx * cdiv(e.A1, 4); need = 64 * (4 * e.taps + 1) * 4;
      }
    }
    if (e.form != 0 && need > 48 * 1024) { e.form = 0; need = 0; }
    if (e.form == 0) {
      e.gx = 1;
      e.nblk = (int)((e.total + 256 * 16 - 1) / (256 * 16));
      if (e.nblk > 1024) e.nblk = 1024;
    }
    e.blk0 = blk;
    blk += e.nblk;
    if (need > smem) smem = need;
  }
  if (hipMemcpyAsync(table_dev, tab.data(), sizeof(PackEntry) * (size_t)n, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {  // the host vector dies at return
    xr_set_error("xr_pack_plan: table upload failed");
    return XR_E_LAUNCH;
  }
  *smem_out = smem;
  return blk;
}

extern "C" int xr_pack_run(const void* table_dev, int n, int blocks, int smem, void* stream) {
  XR_CHECK_ARG(table_dev && n > 0 && blocks > 0 && smem >= 0 && smem <= 48 * 1024, "xr_pack_run: bad arguments");
  hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)blocks), dim3(256), (size_t)smem, (hipStream_t)stream,
                     reinterpret_cast<const PackEntry*>(table_dev), n);
  XR_CHECK_LAUNCH("xr_pack_run");
  return XR_OK;
}

extern "C" int xr_tune(int knob, int value) {
  XR_CHECK_ARG(knob >= 0 && knob < 20, "xr_tune: knob out of range");
  g_tune[knob] = value;
  return XR_OK;
}

extern "C" int xr_set_deterministic(int on) {
  g_tune[16] = on ? 1 : 0;
  return XR_OK;
}

extern "C" int xr_bias_cast(int dtype, const float* ws, const float* bias, void* out, int64_t M, int K, int ld, void* stream) {
  XR_CHECK_ARG((dtype == XR_BF16 || dtype == XR_F32) && ws && out && M > 0 && K > 0 && ld >= K, "xr_bias_cast: bad arguments");
  int blocks = (int)((M * ld + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == XR_BF16)
    hipLaunchKernelGGL(bias_cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, bias, (bf16_t*)out, M, K, ld);
  else
    hipLaunchKernelGGL(bias_cast_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, bias, (float*)out, M, K, ld);
  XR_CHECK_LAUNCH("xr_bias_cast");
  return XR_OK;
}

extern "C" int xr_pack_weight(const float* src, void* dst, int nplanes, int A1, int A2, int taps, int B, int Bp,
                              int Kg, int64_t sa1, int64_t sa2, int64_t st_, int64_t sb, void* stream) {
  XR_CHECK_ARG(src && dst && nplanes >= 1 && nplanes <= 3, "xr_pack_weight: null pointer or nplanes not in {1,2,3}");
  XR_CHECK_ARG(A1 > 0 && A2 > 0 && taps > 0 && B > 0 && Bp >= B && Bp % 8 == 0 && Kg % 64 == 0 && Kg >= taps * Bp,
               "xr_pack_weight: bad dims A1=%d A2=%d taps=%d B=%d Bp=%d Kg=%d", A1, A2, taps, B, Bp, Kg);
  hipStream_t st_h = (hipStream_t)stream;
  const int64_t total = (int64_t)A1 * A2 * Kg;
  if (A2 == 1 && st_ == 1 && taps <= 64 && A1 <= 65535) {
    if (sb == taps && sa1 == (int64_t)B * taps) {  // parameter [A][B][taps]
      hipLaunchKernelGGL(pack_fwdform_kernel, dim3(cdiv(Bp, 64), A1), dim3(256), (size_t)64 * taps * sizeof(float), st_h, src,
                         (bf16_t*)dst, nplanes, taps, B, Bp, Kg, total);
      XR_CHECK_LAUNCH("xr_pack_weight");
      return XR_OK;
    }
    if (sa1 == taps && sb == (int64_t)A1 * taps) {  // parameter [B][A][taps]
      hipLaunchKernelGGL(pack_dgradform_kernel, dim3(cdiv(Bp, 64), cdiv(A1, 4)), dim3(256),
                         (size_t)64 * (4 * taps + 1) * sizeof(float), st_h, src, (bf16_t*)dst, nplanes, taps, A1, B, Bp, Kg, total);
      XR_CHECK_LAUNCH("xr_pack_weight");
      return XR_OK;
    }
  }
  if (taps == 1 && sa1 == 1 && sa2 == (int64_t)A1 && sb == (int64_t)A1 * A2 && A1 <= 64 && A2 <= 65535 * 8) {
    // Linear dgrad pack: A1 = HW, A2 = C, B = K
    const size_t smem = (size_t)64 * (8 * A1 + 1) * sizeof(float);
    if (smem > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pack_lindgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL(pack_lindgrad_kernel, dim3(cdiv(Bp, 64), cdiv(A2, 8)), dim3(256), smem, st_h, src, (bf16_t*)dst, nplanes, A1,
                       A2, B, Bp, Kg, total);
    XR_CHECK_LAUNCH("xr_pack_weight");
    return XR_OK;
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st_h, src, (bf16_t*)dst, nplanes, A2, taps, B, Bp, Kg, sa1, sa2, st_, sb, total);
  XR_CHECK_LAUNCH("xr_pack_weight");
  return XR_OK;
}

extern "C" int xr_unpack_wgrad(const float* packed, float* dst, int A1, int A2, int taps, int B, int Bp, int Kg,
                               int64_t sa1, int64_t sa2, int64_t st_, int64_t sb, int accumulate, int nslices,
                               void* stream) {
  XR_CHECK_ARG(packed && dst && nslices >= 1, "xr_unpack_wgrad: null pointer / nslices < 1");
  XR_CHECK_ARG(A1 > 0 && A2 > 0 && taps > 0 && B > 0 && Bp >= B && Kg >= taps * Bp, "xr_unpack_wgrad: bad dims");
  const int64_t total = (int64_t)A1 * A2 * taps * B;
  if (A2 == 1 && st_ == 1 && sb == taps && sa1 == (int64_t)B * taps && taps >= 2 && taps <= 64 && A1 <= 65535 && B >= 32) {
    // Conv2d [K][C][R][S] / Linear-as-7x7-conv parameters: the taps are the fastest axis of the parameter and the slowest of
    // the pack -- LDS-tiled so both sides stay coalesced (a strided read-modify-write of the parameter layout ran at 1.3 TB/s)
    const int tiles = cdiv(B, 64) * A1;
    int groups = 1;
    if (tiles < 1024 && nslices > 8) groups = cdiv(nslices, g_tune[10] > 0 ? (int)g_tune[10] : (tiles < 128 ? 8 : 16));   // few output tiles, many slices (knob 10: slices per group)
    if (groups > 64) groups = 64;
    if (XR_DET()) groups = 1;   // the slice groups meet in dst by atomics: one group sums the slices in order
    const int spg = cdiv(nslices, groups);
    groups = cdiv(nslices, spg);
    if (groups > 1 && !(accumulate & 1)) {
      if (hipMemsetAsync(dst, 0, (size_t)total * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        xr_set_error("xr_unpack_wgrad: memset failed");
        return XR_E_LAUNCH;
      }
    }
    hipLaunchKernelGGL(unpack_fwdform_kernel, dim3(cdiv(B, 64), A1, groups), dim3(256), (size_t)64 * taps * sizeof(float),
                       (hipStream_t)stream, packed, dst, taps, B, Bp, Kg, accumulate & 1, nslices, (int64_t)A1 * Kg, spg);
    XR_CHECK_LAUNCH("xr_unpack_wgrad");
    return XR_OK;
  }
  if (B % 4 == 0 && Bp % 4 == 0 && Kg % 4 == 0) {
    const int64_t total4 = total / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    // few outputs but many slices (the 64-channel layers): spread the slices over blockIdx.y and meet by atomics
    int groups = 1;
    if (blocks < 512 && nslices > 16) groups = (nslices + 15) / 16;
    if (XR_DET()) groups = 1;
    const int spg = (nslices + groups - 1) / groups;
    groups = (nslices + spg - 1) / spg;
    if (groups > 1 && !(accumulate & 1)) {
      if (hipMemsetAsync(dst, 0, (size_t)total * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        xr_set_error("xr_unpack_wgrad: memset failed");
        return XR_E_LAUNCH;
      }
    }
    hipLaunchKernelGGL(unpack_wgrad4_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, packed, dst, A2, taps, B / 4,
                       Bp, Kg, sa1, sa2, st_, sb, accumulate, nslices, (int64_t)A1 * A2 * Kg, total4, spg);
    XR_CHECK_LAUNCH("xr_unpack_wgrad");
    return XR_OK;
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  int groups = 1;
  if (blocks < 512 && nslices > 16) groups = (nslices + 15) / 16;
  if (XR_DET()) groups = 1;
  const int spg = (nslices + groups - 1) / groups;
  groups = (nslices + spg - 1) / spg;
  if (groups > 1 && !(accumulate & 1)) {
    if (hipMemsetAsync(dst, 0, (size_t)total * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      xr_set_error("xr_unpack_wgrad: memset failed");
      return XR_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, packed, dst, A2, taps, B, Bp,
                     Kg, sa1, sa2, st_, sb, accumulate, nslices, (int64_t)A1 * A2 * Kg, total, spg);
  XR_CHECK_LAUNCH("xr_unpack_wgrad");
  return XR_OK;
}

extern "C" int xr_conv_igemm(int dtype, const void* in, const void* w, const float* bias, void* out,
                             int N, int H, int W, int C, int Ho, int Wo, int K, int R, int S, int stride, int pad,
                             int transposed, int Kg, int ldo, float* splitk_ws, int splitk, const void* ep_src,
                             const float* ep_alpha, float* ep_dalpha, int ep_spread, void* ep2_out, float* ep_red, const void* ep_add, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32 || dtype == XR_F32X2, "xr_conv_igemm: bad dtype %d", dtype);
  XR_CHECK_ARG(ep_src == nullptr || ep_red != nullptr ||
                   (ep_alpha && ep_dalpha && K % 8 == 0 && splitk_ws == nullptr && bias == nullptr),
               "xr_conv_igemm: fused PReLU-backward epilogue needs alpha, dalpha, K %% 8 == 0, no bias, no split-K");
  XR_CHECK_ARG(ep_red == nullptr || (K % 8 == 0 && splitk_ws == nullptr && ep2_out == nullptr && out),
               "xr_conv_igemm: fused per-channel reductions (ep_red) need K %% 8 == 0, no split-K, no second output");
  XR_CHECK_ARG(ep2_out == nullptr || (ep_alpha && ep_src == nullptr && K % 8 == 0 && splitk_ws == nullptr && out),
               "xr_conv_igemm: fused PReLU-forward output needs alpha, K %% 8 == 0, no PReLU-backward epilogue, no split-K");
  XR_CHECK_ARG(ep_add == nullptr || (K % 8 == 0 && ep_src == nullptr && ep2_out == nullptr && ep_red == nullptr && splitk_ws == nullptr && out),
               "xr_conv_igemm: fused gradient sum (ep_add) needs K %% 8 == 0 and no other epilogue fusion / split-K");
  XR_CHECK_ARG((splitk_ws == nullptr) == (splitk <= 1), "xr_conv_igemm: split-K needs both a workspace and splitk > 1");
  XR_CHECK_ARG(in && w && (out || splitk_ws), "xr_conv_igemm: null pointer");
  XR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
               "xr_conv_igemm: non-positive dimension");
  XR_CHECK_ARG(C % 8 == 0, "xr_conv_igemm: C=%d must be a multiple of 8 (pad channels)", C);
  XR_CHECK_ARG(Kg % 64 == 0 && Kg >= R * S * C, "xr_conv_igemm: Kg=%d must be a multiple of 64 and >= R*S*C=%d", Kg,
               R * S * C);
  XR_CHECK_ARG(ldo >= K && ldo % 8 == 0, "xr_conv_igemm: ldo=%d must be >= K=%d and a multiple of 8", ldo, K);
  XR_CHECK_ARG((long long)N * Ho * Wo < (1ll << 31) && (long long)N * H * W * C < (1ll << 40), "xr_conv_igemm: too large");
  if (!transposed) {
    XR_CHECK_ARG((H + 2 * pad - R) / stride + 1 == Ho && (W + 2 * pad - S) / stride + 1 == Wo,
                 "xr_conv_igemm: output %dx%d inconsistent with input %dx%d k%d s%d p%d", Ho, Wo, H, W, R, stride, pad);
  } else {
    XR_CHECK_ARG((Ho + 2 * pad - R) / stride + 1 == H && (Wo + 2 * pad - S) / stride + 1 == W,
                 "xr_conv_igemm(transposed): output %dx%d inconsistent with input %dx%d k%d s%d p%d", Ho, Wo, H, W, R,
                 stride, pad);
  }
  IgemmP p{in, (const bf16_t*)w, bias, out, N, H, W, C, Ho, Wo, K, R, S, stride, pad, Kg, ldo,
           N * Ho * Wo, 0, 0, 0, 0, splitk_ws, splitk > 1 ? splitk : 0, {}, {}, {}, {}, {}, {}, {}, {}, 0, 0, ep_src, ep_alpha, ep_dalpha, ep_spread > 0 ? ep_spread : 1, ep2_out, ep_red, ep_add, g_tune[4]};
  hipStream_t st = (hipStream_t)stream;
  if (xr_igemm8_eligible(p, dtype, transposed)) return xr_igemm8_launch(p, transposed, st);
  const bool wide = K > 64 && g_tune[3] == 0;
  if (dtype == XR_BF16) {
    if (wide) return transposed ? launch_igemm<0, 128, 128, 2, true>(p, st) : launch_igemm<0, 128, 128, 2, false>(p, st);
    // <= 32 output columns (the 64 -> 3 image heads of FSRNet, model/FSRnet.py:326,455; 1x1 parsing head): a 128 x 32 tile --
    // the narrow-tile kernel is instruction-issue bound, and half of a 64-column tile's weight staging / MFMAs / epilogue
    // would be spent on padding
    if (K <= 32 && g_tune[15] == 0)
      return transposed ? launch_igemm<0, 128, 32, 4, true>(p, st) : launch_igemm<0, 128, 32, 4, false>(p, st);
    return transposed ? launch_igemm<0, 128, 64, 4, true>(p, st) : launch_igemm<0, 128, 64, 4, false>(p, st);
  }
  if (dtype == XR_F32X2) {
    if (wide) return transposed ? launch_igemm<2, 128, 128, 2, true>(p, st) : launch_igemm<2, 128, 128, 2, false>(p, st);
    return transposed ? launch_igemm<2, 128, 64, 4, true>(p, st) : launch_igemm<2, 128, 64, 4, false>(p, st);
  }
  if (wide) return transposed ? launch_igemm<1, 128, 128, 2, true>(p, st) : launch_igemm<1, 128, 128, 2, false>(p, st);
  return transposed ? launch_igemm<1, 128, 64, 4, true>(p, st) : launch_igemm<1, 128, 64, 4, false>(p, st);
}

// ---- stride-2 3x3 input gradient as ONE dense stride-1 2x2 gather (xr_conv_dgrad_s2)
namespace {
// dst[q][(cls * C2 + c)][t * Kp + k] = plane q of w[k][c][r][s] for the window tap t = 2 dr + ds that sub-pixel class cls = 2 a + b
// can see (a = 0: dr = 0 -> r = 1; a = 1: dr = 0 -> r = 2, dr = 1 -> r = 0; the same for b / ds / s), 0 elsewhere
__global__ void pack_s2dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int nplanes, int K, int C, int Kp, int C2,
                                    int Kg, int64_t plane) {
  const int64_t total = (int64_t)4 * C2 * Kg;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kg), col = (int)(i - (int64_t)row * Kg);
    const int cls = row / C2, c = row - cls * C2;
    const int tp = col / Kp, k = col - tp * Kp;
    float v = 0.f;
    if (c < C && k < K && tp < 4) {
      const int a = cls >> 1, b = cls & 1, dr = tp >> 1, ds = tp & 1;
      const int r = a ? (dr ? 0 : 2) : (dr ? -1 : 1), s_ = b ? (ds ? 0 : 2) : (ds ? -1 : 1);
      if (r >= 0 && s_ >= 0) v = w[(((size_t)k * C + c) * 3 + r) * 3 + s_];
    }
    for (int q = 0; q < nplanes; ++q) {
      const bf16_t h = f2bf(v);
      dst[(int64_t)q * plane + i] = h;
      v -= bf2f(h);
    }
  }
}
}  // namespace

extern "C" int xr_pack_dgrad_s2(const float* w, void* dst, int nplanes, int K, int C, int Kp, int C2, int Kg, void* stream) {
  XR_CHECK_ARG(w && dst && nplanes >= 1 && nplanes <= 3 && K > 0 && C > 0 && Kp >= K && C2 >= C && Kp % 8 == 0 && C2 % 8 == 0,
               "xr_pack_dgrad_s2: bad arguments");
  XR_CHECK_ARG(Kg % 64 == 0 && Kg >= 4 * Kp, "xr_pack_dgrad_s2: Kg must be a multiple of 64 and >= 4 * Kp");
  const int64_t total = (int64_t)4 * C2 * Kg;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_s2dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, nplanes, K, C, Kp, C2, Kg,
                     total);
  XR_CHECK_LAUNCH("xr_pack_dgrad_s2");
  return XR_OK;
}

extern "C" int xr_conv_dgrad_s2(int dtype, const void* dy, const void* wpack, void* dx, int N, int Ho, int Wo, int Kp, int C2, int Kg,
                                const void* ep_src, const float* ep_alpha4, float* ep_dalpha, int ep_spread, float* ep_red,
                                const void* ep_add, void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32 || dtype == XR_F32X2, "xr_conv_dgrad_s2: bad dtype %d", dtype);
  XR_CHECK_ARG(dy && wpack && dx && N > 0 && Ho > 0 && Wo > 0 && Kp > 0 && C2 > 0, "xr_conv_dgrad_s2: null pointer / non-positive dimension");
  XR_CHECK_ARG(Kp % 8 == 0 && C2 % 8 == 0 && Kg % 64 == 0 && Kg >= 4 * Kp, "xr_conv_dgrad_s2: Kp / C2 multiples of 8, Kg >= 4 * Kp");
  XR_CHECK_ARG((long long)N * Ho * Wo < (1ll << 29), "xr_conv_dgrad_s2: too many pixels");
  XR_CHECK_ARG(ep_src == nullptr || ep_red != nullptr || (ep_alpha4 && ep_dalpha), "xr_conv_dgrad_s2: PReLU-backward epilogue needs alpha and dalpha");
  XR_CHECK_ARG(ep_add == nullptr || (ep_src == nullptr && ep_red == nullptr), "xr_conv_dgrad_s2: ep_add excludes the other epilogues");
  IgemmP p{dy, (const bf16_t*)wpack, nullptr, dx, N, Ho, Wo, Kp, Ho, Wo, 4 * C2, 2, 2, 1, 0, Kg, 4 * C2,
           N * Ho * Wo, 0, 0, 0, 0, nullptr, 0, {}, {}, {}, {}, {}, {}, {}, {}, 0, 0, ep_src, ep_alpha4, ep_dalpha, ep_spread > 0 ? ep_spread : 1, nullptr, ep_red, ep_add, g_tune[4]};
  p.d2s_c = C2;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XR_BF16) return launch_igemm<0, 128, 128, 2, false>(p, st);
  if (dtype == XR_F32X2) return launch_igemm<2, 128, 128, 2, false>(p, st);
  return launch_igemm<1, 128, 128, 2, false>(p, st);
}

extern "C" int xr_conv_wgrad(int dtype, const void* in, const void* dy, float* dwp, int N, int H, int W, int C, int Ho,
                             int Wo, int K, int R, int S, int stride, int pad, int transposed, int ldy, int Kg, int split,
                             void* stream) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32 || dtype == XR_F32X2, "xr_conv_wgrad: bad dtype %d", dtype);
  XR_CHECK_ARG(in && dy && dwp, "xr_conv_wgrad: null pointer");
  XR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0,
               "xr_conv_wgrad: non-positive dimension");
  XR_CHECK_ARG(C % 8 == 0 && ldy % 8 == 0 && ldy >= K, "xr_conv_wgrad: C=%d / ldy=%d must be multiples of 8, ldy >= K", C,
               ldy);
  XR_CHECK_ARG(Kg % 64 == 0 && Kg >= R * S * C, "xr_conv_wgrad: bad Kg=%d", Kg);
  XR_CHECK_ARG((long long)N * Ho * Wo < (1ll << 31), "xr_conv_wgrad: too many pixels");
  WgradP p{in, dy, dwp, N, H, W, C, Ho, Wo, K, R, S, stride, pad, ldy, Kg, N * Ho * Wo, 0, 0, 0, 0, {}, {}, 0, 0, 0, 0, 0, 0, g_tune[5], 0, 0};
  hipStream_t st = (hipStream_t)stream;
  // 128 x 128 tile for K > 64 and for short reduction rows (Kg <= 128: the 3 -> 64 stems, 1x1 convolutions of <= 128 channels):
  // the 64 x 256 tile would stage 256 columns of which at most 128 exist (stem weight gradient 253 -> 213 us)
  const bool tall = K > 64 || Kg <= 128;
  if (dtype == XR_BF16 && xr_wgrad8_eligible(p, transposed)) return xr_wgrad8_launch(p, split, st);   // wide layers: xr_wgrad8.hip
  if (dtype == XR_BF16) {
    if (tall) return transposed ? launch_wgrad<0, 128, 128, true>(p, split, st) : launch_wgrad<0, 128, 128, false>(p, split, st);
    return transposed ? launch_wgrad<0, 64, 256, true>(p, split, st) : launch_wgrad<0, 64, 256, false>(p, split, st);
  }
  if (dtype == XR_F32X2) {
    if (tall) return transposed ? launch_wgrad<2, 128, 128, true>(p, split, st) : launch_wgrad<2, 128, 128, false>(p, split, st);
    return transposed ? launch_wgrad<2, 64, 256, true>(p, split, st) : launch_wgrad<2, 64, 256, false>(p, split, st);
  }
  if (tall) return transposed ? launch_wgrad<1, 128, 128, true>(p, split, st) : launch_wgrad<1, 128, 128, false>(p, split, st);
  return transposed ? launch_wgrad<1, 64, 256, true>(p, split, st) : launch_wgrad<1, 64, 256, false>(p, split, st);
}
