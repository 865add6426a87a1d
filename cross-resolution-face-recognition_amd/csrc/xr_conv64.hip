// xr_conv64.hip -- weights-stationary direct 3x3 convolution for the 64 -> 64 channel, stride-1, pad-1 bf16 layers
// (every FSRNet body layer: model/FSRnet.py:79,85 at 112x112 and 28x28; IR / ResNet stage 1: model_irse.py:59,
// model/resnet.py:9-12 at 112x112 / 56x56).  Forward and input gradient (the same kernel with mirrored tap offsets).
//
// Why not the implicit GEMM of xr_conv.hip: with C = K = 64 a K-loop has only nine stages, the im2col gather re-reads the
// input nine times through the texture path and the 128x64 tiles are instruction-issue bound (~13 VALU per MFMA).  Here
//   * the 64 x 576 weight panel lives in REGISTERS as MFMA A-operand fragments for the whole kernel: persistent workgroups
//     (one per CU, 4 waves = one per SIMD), wave = (32-channel half, 4 of the 8 pixel blocks) -> 144 weight VGPRs, 64
//     accumulator registers (the eight-wave form, 2 pixel blocks per wave and two waves per SIMD, reads every pixel fragment
//     from LDS twice as often per MFMA and measured slower);
//   * a workgroup walks a contiguous run of 16x16-pixel output tiles; the 18x18-pixel input tile with halo is staged ONCE
//     into LDS (global -> registers -> LDS, double buffered) and all nine taps are read from LDS with ds_read_b128;
//   * staging goes through registers because it can TRANSFORM: y = prelu(x * scale[n][c] + shift[n][c], alpha[c]) -- the
//     InstanceNorm apply + PReLU of the producing layer (model/FSRnet.py:81-84) is folded into the consumer's load, once
//     per input element (an im2col gather would pay it nine times), so the normalised activation never exists in HBM;
//   * MFMA is v_mfma_f32_32x32x16_bf16 with the 32 "rows" = 2 image rows x 16 columns: the LDS image is pixel-major
//     (128 B per pixel) with the 16-B chunk XOR-swizzled by (halo column >> 1) & 7 -- conflict-free ds_read_b128 for every
//     tap shift (the 16-lane groups of a b128 read then see each 16-B slot of the 256-B bank row once);
//   * the kernel is a hand-laid software pipeline: the 144 MFMAs a wave issues per tile are the clock; every MFMA is followed
//     by a slot of "side work" that issues under it (an MFMA holds the vector issue port for 8 of its 32 cycles): the
//     fragment read of the next stage, the global loads of tile t+1, the stream-out of tile t-1's output image (LDS -> HBM,
//     + statistics / residual sum) and the transform + LDS write of tile t+1.  Only the accumulator -> LDS epilogue and two
//     barriers per tile are not overlapped.  The element-wise side work is written on register PAIRS (v_pk_fma / v_pk_mul /
//     v_pk_add, one v_cvt_pk_bf16_f32 per packed dword): 26-28 % fewer VALU instructions per tile in the fused variants than
//     the scalar form -- worth 5 % on EP 4 and nothing measurable elsewhere: the loop is bound by dependent-issue stalls of its
//     single wave (SQ_WAIT_INST_ANY 30 %, s_waitcnt 20 % of wave cycles), not by the instruction count;
//   * accumulators are kept transposed (D[channel][pixel]): a lane owns 4 consecutive channels of a pixel, the epilogue
//     writes 8-byte packets into an LDS image from which full 128-B pixel rows are streamed out;
//   * epilogue fusions: bias; per-image sum / sum of squares of the (rounded) output for the InstanceNorm that follows
//     (accumulated in registers across the tiles of one image, one atomic per channel and image change); residual-gradient
//     sum (out += ep_add); InstanceNorm + PReLU BACKWARD reductions (EP == 3): when the tensor being written is the gradient
//     dy1 of y1 = prelu(c1 * scale[n] + shift[n]) -- conv2's input gradient in the FSRNet block -- the stream-out also loads the
//     matching c1 chunk and accumulates sum dz, sum dz * c1 and the PReLU slope term per image and channel, the three sums of
//     xr_affine_act_bwd_reduce: that pass (two full-tensor reads) disappears.  EP == 4 chains two residual blocks in the backward
//     pass: the kernel computes dout = conv + ep_add (the gradient entering the PREVIOUS block's tail out = prelu(c2 * scale +
//     shift + x)), but stores dz = dout * prelu'(z) -- the gradient of the tail's pre-activation, which is both that block's
//     residual-branch gradient and the input of its InstanceNorm backward -- together with the three sums over (dz, c2): the
//     previous block then needs neither its reduce pass nor the residual half of its apply pass.
#include "xr_common.h"
#include <type_traits>


// slot table of the on-load transform (NORM): chunk i is transformed at XB + XS i + XP q (q = 0..3) and written at XB + XS i + XW
#ifndef XR64_TWO_PHASE
#define XR64_TWO_PHASE 0   // measured: 7.74 -> 7.69 VALU per MFMA only (the spills are not the staging registers); off
#endif
#ifndef XR64_XB
#define XR64_XB 56
#define XR64_XS 8
#define XR64_XP 2
#define XR64_XW 7
#endif

namespace {

constexpr int NT = 256;
constexpr int TS = 16;                    // output tile edge
constexpr int HS = TS + 2;                // halo tile edge
constexpr int HPIX = HS * HS;             // 324 halo pixels
constexpr int INBUF = (HPIX + 1) * 128;   // one input stage (bytes); row 324 is a dump slot for the lanes of the ragged last chunk
constexpr int NCH = (HPIX * 8 + NT - 1) / NT;  // 16-B chunks a thread stages per tile (11 at 256 threads)
constexpr int RPP = NT / 8;               // pixel rows one pass of the workgroup covers (32)
constexpr int NOUT = TS * TS / RPP;       // output-image row groups per thread (8)
constexpr int OPITCH = 144;               // output image pitch (bytes): 16-B aligned, rows 4 banks apart
constexpr int OUTIMG = TS * TS * OPITCH;  // 36,864 B
constexpr int OUTBASE = 2 * INBUF;
constexpr int SMEM = 2 * INBUF + OUTIMG;  // 120,064 B

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define XR64_OOR 0x80000000u

// compile-time loop: f(std::integral_constant<int, i>) for i in [B, E) -- the pipeline below indexes register arrays with the
// loop counter, which must therefore never become a run-time value (a failed "#pragma unroll" would send them to scratch)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct DC64P {
  const bf16_t* in;
  const bf16_t* w;        // [64][576] bf16: row = GEMM output channel, column = tap * 64 + reduction channel
  const float* bias;      // [64] or null
  bf16_t* out;
  const bf16_t* ep_add;   // laid out like out: residual gradient (EP == 2, 4) or the norm input c1 (EP == 3)
  const bf16_t* ep_c;     // EP == 4: the previous block's c2 (norm input of its tail) and x (its residual input)
  const bf16_t* ep_x;
  bf16_t* out2;           // EP == 5: second output prelu(out, n_alpha), laid out like out
  const float* n_scale;   // [N][64] per-image affine applied to the input on load (NORM)
  const float* n_shift;
  const float* n_alpha;   // [64] PReLU slope applied after the affine, or null (no activation)
  float* stats;           // [2][N][64]: sum, sum of squares of the output per image and channel (EP == 1);
                          // [3][N][64]: sum dz, sum dz * c1, sum out * z * [z <= 0] (EP == 3; n_scale / n_shift / n_alpha then
                          // describe z = c1 * scale + shift, dz = out * prelu'(z) -- the on-load transform is off)
  int N, H, W, tiles_x, tiles_img, ntiles, tpw;
  unsigned io_bytes;      // extent of in / out / ep_add (same shape)
  int dbg;                // tuning knob 14, timing experiments only (results are wrong): bit 0 no halo loads, bit 1 no
                          // accumulator -> LDS epilogue, bit 2 no output stores
};

struct TileGeo {          // wave-uniform description of one tile
  int n, y0, x0;
  int base;               // byte offset of pixel (y0, x0) of image n
  bool interior;          // the whole halo lies inside the image
};

// EP: 0 plain, 1 per-image output statistics, 2 out += ep_add, 3 InstanceNorm / PReLU backward reductions of the output,
// 4 out = prelu'(tail) * (conv + ep_add) with the reductions of the previous block's tail
template <bool TR, bool NORM, int EP>
__global__ __launch_bounds__(NT, 1) void dconv64_kernel(DC64P p) {
  constexpr bool ADD = EP >= 2 && EP <= 4;   // a second input laid out like the output is loaded in the epilogue
  constexpr bool RED = EP == 3 || EP == 4;   // backward reductions over the output

  constexpr int NPB = 4;   // pixel blocks (32 pixels = 2 image rows x 16 columns) per wave
  constexpr int NSLOT = 36 * NPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;     // pixel-block half, 32-channel half
  const int lp = lane & 31, kg = lane >> 5;

  int tile = blockIdx.x * p.tpw;
  int tile_end = tile + p.tpw;
  if (tile_end > p.ntiles) tile_end = p.ntiles;
  if (tile >= tile_end) return;

  // ---- weight panel -> registers (MFMA A operand: row = channel lp of the half, 8 reduction elements 8*kg..+7 of the step)
  bf16x8_t wf[9][4];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wf[tp][ks] = *reinterpret_cast<const bf16x8_t*>(p.w + (size_t)(wn * 32 + lp) * 576 + tp * 64 + ks * 16 + kg * 8);

  // ---- tile-invariant staging constants of this thread: chunk column cc is fixed (NT % 8 == 0), halo pixels (t >> 3) + RPP i
  const int cc = t & 7;
  int rel[NCH];      // byte offset of the chunk relative to the tile's origin pixel (y0, x0)
  int ldso[NCH];     // LDS byte offset inside an input stage (lanes beyond the 324 halo pixels: the dump row)
  unsigned vm_all = 0;   // bit i: chunk i of this thread is a real halo pixel
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int hp = (t >> 3) + RPP * i;
    const int hy = (hp * 57) >> 10, hx = hp - hy * HS;   // hp / 18, exact for hp < 400
    rel[i] = (((hy - 1) * p.W + (hx - 1)) * 64 + cc * 8) * 2;
    ldso[i] = hp < HPIX ? hp * 128 + ((cc ^ ((hx >> 1) & 7)) << 4) : HPIX * 128;
    if (hp < HPIX) vm_all |= 1u << i;
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in), 0, p.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out2 = __builtin_amdgcn_make_buffer_rsrc(EP == 5 ? p.out2 : p.out, 0, p.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_add =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ADD ? p.ep_add : p.in), 0, p.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(EP == 4 ? p.ep_c : p.in), 0, p.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(EP == 4 ? p.ep_x : p.in), 0, p.io_bytes, 0x00020000);
  // per-channel coefficients of this thread's chunk column, as register PAIRS: the element-wise work below is written on
  // f32x2_t so that it issues as v_pk_fma / v_pk_mul / v_pk_add (the fused variants of this kernel are VALU-issue bound:
  // one wave per SIMD issues the 144 MFMAs of a tile AND its side work)
  f32x2_t sc[4], sh[4], al[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) sc[q] = (f32x2_t){1.f, 1.f}, sh[q] = (f32x2_t){0.f, 0.f}, al[q] = (f32x2_t){1.f, 1.f};
  auto ld8p = [&](const float* src, f32x2_t (&v)[4]) {
    float tmp[8];
    ld8(src, tmp);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (f32x2_t){tmp[2 * q], tmp[2 * q + 1]};
  };
  if ((NORM || RED || EP == 5) && p.n_alpha != nullptr) ld8p(p.n_alpha + cc * 8, al);
  f32x2_t alm1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) alm1[q] = al[q] - (f32x2_t){1.f, 1.f};
  int n_staged = -1;

  auto geo_of = [&](int tl) {
    TileGeo g;
    g.n = tl / p.tiles_img;
    const int r = tl - g.n * p.tiles_img;
    const int ty = r / p.tiles_x;
    g.y0 = ty * TS;
    g.x0 = (r - ty * p.tiles_x) * TS;
    g.base = ((g.n * p.H + g.y0) * p.W + g.x0) * 128;
    g.interior = g.y0 > 0 && g.x0 > 0 && g.y0 + TS < p.H && g.x0 + TS < p.W;
    return g;
  };

  // The pipelined region below must stay free of control flow around memory operations (hipcc's s_waitcnt insertion falls
  // back to vmcnt(0) at every join, which would drain the stores in flight in front of every LDS write): all conditions are
  // folded into per-tile bit masks computed at the top of an iteration and applied with selects / out-of-range offsets.
  // ---- side work 1: global loads of a tile's halo chunks into registers
  v4u_t st[NCH];
  auto chunk_mask = [&](const TileGeo& g, bool live) {   // bit i: chunk i lies inside the image (and the tile exists)
    unsigned m = live ? vm_all : 0u;
    if (live && !g.interior) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int hp = (t >> 3) + RPP * i;
        const int hy = (hp * 57) >> 10, hx = hp - hy * HS;
        if (!((unsigned)(g.y0 - 1 + hy) < (unsigned)p.H && (unsigned)(g.x0 - 1 + hx) < (unsigned)p.W)) m &= ~(1u << i);
      }
    }
    return m;
  };
  auto load_chunk = [&](int base, unsigned vm, int i) {
    const unsigned voff = ((vm >> i) & 1u) ? (unsigned)(base + rel[i]) : XR64_OOR;
    st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
  };
  // ---- side work 3: (transform and) write a staged chunk into an input stage
  auto xform_part = [&](unsigned vm, int i, int q) {   // NORM: dword q (two channels) of staged chunk i, in place
#ifndef XR64_NOXFORM   // timing experiments only (tools/build_variant.sh): the on-load transform's arithmetic switched off
    if constexpr (NORM) {
      const bool ok = (vm >> i) & 1u;   // zero padding applies AFTER the transform
      const f32x2_t z = __builtin_elementwise_fma(unpack2bf(st[i][q]), sc[q], sh[q]);
#ifdef XR64_PRELU_SELECT
      const f32x2_t za = z * al[q];
      unsigned o = ok ? pack2bf(z.x > 0.f ? z.x : za.x, z.y > 0.f ? z.y : za.y) : 0u;
#else
      // prelu(z) = z + (alpha - 1) * min(z, 0): two v_min + one packed fma instead of a packed multiply, two compares and two selects
      // (each compare -> select pair also costs the VCC wait states)
      unsigned o = ok ? pack2bf(__builtin_elementwise_fma(alm1[q], (f32x2_t){fminf(z.x, 0.f), fminf(z.y, 0.f)}, z)) : 0u;
#endif
      // pin the computation to THIS slot: IR-level sinking would otherwise move all four parts down to the LDS write
      // (sched_barrier only fences the machine scheduler)
      asm volatile("" : "+v"(o));
      st[i][q] = o;
    }
#endif
  };
  auto write_chunk = [&](int buf, int i) { *reinterpret_cast<v4u_t*>(smem + buf * INBUF + ldso[i]) = st[i]; };

  // ---- side work 2: stream the previous tile's output image out.  thread = (chunk column cc, image rows (t >> 7) + 4 i,
  // column (t >> 3) & 15)
  constexpr int ORS = RPP / TS;                                    // image rows between a thread's row groups (4)
  const int orow0 = t >> 7, ocol = (t >> 3) & 15;
  const int oimg = OUTBASE + (t >> 3) * OPITCH + cc * 16;          // + i * RPP * OPITCH
  const int orel0 = ((orow0 * p.W + ocol) * 64 + cc * 8) * 2;      // + i * ORS * W * 128
  const int orstep = ORS * p.W * 128;
  f32x2_t bs[4], bss[4], b3[4];   // EP == 1 / 3 / 4: sums of this thread's channel chunk (as pairs), current image
#pragma unroll
  for (int q = 0; q < 4; ++q) bs[q] = bss[q] = b3[q] = (f32x2_t){0.f, 0.f};
  int n_stats = -1;
  v4u_t ov[NOUT], addv[NOUT], cv[EP == 4 ? NOUT : 1], xv[EP == 4 ? NOUT : 1];
  auto out_mask = [&](const TileGeo& g, bool live) {   // bit i: row group i of this thread lies inside the image
    unsigned m = 0;
    if (live && g.x0 + ocol < p.W) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i)
        if (g.y0 + orow0 + ORS * i < p.H) m |= 1u << i;
    }
    return m;
  };
  auto out_voff = [&](int base, unsigned om, int i) {
    return ((om >> i) & 1u) ? (unsigned)(base + orel0 + i * orstep) : XR64_OOR;
  };
  auto add_load = [&](int base, unsigned om, int i) {
    if constexpr (ADD) addv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, out_voff(base, om, i), 0, 0);
  };
  auto tail_load = [&](int base, unsigned om, int i, int which) {
    if constexpr (EP == 4) {
      if (which == 0) cv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, out_voff(base, om, i), 0, 0);
      else xv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, out_voff(base, om, i), 0, 0);
    }
  };
  auto out_read = [&](int i) { ov[i] = *reinterpret_cast<const v4u_t*>(smem + oimg + i * RPP * OPITCH); };
  // The element-wise work of a row group is cut into four PARTS (dword q = two channels each, results left in place in ov[i]) so
  // that the pipeline can give every part its own MFMA slot: a slot hides ~7 VALU instructions under its MFMA, a whole row group
  // of the reducing variants is 60-100 -- issued in one slot it stalls the matrix pipe for that long.
  v4u_t ov2[EP == 5 ? NOUT : 1];
  auto out_part = [&](unsigned om, int i, int q) {
    const bool ok = (om >> i) & 1u;   // rows outside the image contribute nothing to the sums (their loads returned zeros as well)
    if constexpr (EP == 2) {
      unsigned o = pack2bf(unpack2bf(ov[i][q]) + unpack2bf(addv[i][q]));
      asm volatile("" : "+v"(o));
      ov[i][q] = o;
    }
    if constexpr (EP == 4) {
      // dout = conv + residual gradient, rounded to bf16 as the tensor the reduce pass would have read
      const unsigned dw = pack2bf(unpack2bf(ov[i][q]) + unpack2bf(addv[i][q]));
      const f32x2_t d = unpack2bf(ok ? dw : 0u), c = unpack2bf(cv[i][q]);
      const f32x2_t z = __builtin_elementwise_fma(c, sc[q], sh[q]) + unpack2bf(xv[i][q]);
      const f32x2_t da = d * al[q];
      const f32x2_t dz = {z.x > 0.f ? d.x : da.x, z.y > 0.f ? d.y : da.y};
      bs[q] += dz;
      bss[q] = __builtin_elementwise_fma(dz, c, bss[q]);
      // d * z where z <= 0 (min + fma instead of compare, select, multiply, add)
      b3[q] = __builtin_elementwise_fma(d, (f32x2_t){fminf(z.x, 0.f), fminf(z.y, 0.f)}, b3[q]);
      unsigned o = pack2bf(dz);
      // pin the result AND the sums to this slot: left free, LLVM sinks the accumulator updates of every row group towards the
      // flush and the kernel spills 220 registers to scratch (4x slower)
      asm volatile("" : "+v"(o), "+v"(bs[q]), "+v"(bss[q]), "+v"(b3[q]));
      ov[i][q] = o;
    }
    if constexpr (EP == 5) {   // second output: PReLU of the value being stored (what the next convolution consumes)
      const f32x2_t a = unpack2bf(ov[i][q]), aa = a * al[q];
      unsigned o = pack2bf(a.x > 0.f ? a.x : aa.x, a.y > 0.f ? a.y : aa.y);
      asm volatile("" : "+v"(o));
      ov2[i][q] = o;
    }
    if constexpr (EP == 1) {
      const f32x2_t a = unpack2bf(ok ? ov[i][q] : 0u);
      bs[q] += a;
      bss[q] = __builtin_elementwise_fma(a, a, bss[q]);
      asm volatile("" : "+v"(bs[q]), "+v"(bss[q]));
    }
    if constexpr (EP == 3) {
      const f32x2_t d = unpack2bf(ok ? ov[i][q] : 0u), c = unpack2bf(addv[i][q]);
      const f32x2_t z = __builtin_elementwise_fma(c, sc[q], sh[q]);
      const f32x2_t da = d * al[q];
      const f32x2_t dz = {z.x > 0.f ? d.x : da.x, z.y > 0.f ? d.y : da.y};
      bs[q] += dz;
      bss[q] = __builtin_elementwise_fma(dz, c, bss[q]);
      b3[q] = __builtin_elementwise_fma(d, (f32x2_t){fminf(z.x, 0.f), fminf(z.y, 0.f)}, b3[q]);   // d * z where z <= 0
      asm volatile("" : "+v"(bs[q]), "+v"(bss[q]), "+v"(b3[q]));   // pinned (see EP == 4)
    }
  };
  auto out_emit = [&](int base, unsigned om, int i) {
    const unsigned voff = out_voff(base, om, i);
    __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_out, voff, 0, 0);
    if constexpr (EP == 5) __builtin_amdgcn_raw_buffer_store_b128(ov2[i], rs_out2, voff, 0, 0);
  };
  auto out_store = [&](int base, unsigned om, int i) {   // the whole row group at once (drain after the last tile)
#pragma unroll
    for (int q = 0; q < 4; ++q) out_part(om, i, q);
    out_emit(base, om, i);
  };
  auto flush_stats = [&]() {
    // fold the 64 threads that share a chunk column (lanes 8 apart); the 8 waves meet in the atomics (once per image)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = bs[e >> 1][e & 1], b = bss[e >> 1][e & 1];
      a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 8, 64); b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      float c3 = 0.f;
      if constexpr (RED) {
        c3 = b3[e >> 1][e & 1];
        c3 += __shfl_xor(c3, 8, 64); c3 += __shfl_xor(c3, 16, 64); c3 += __shfl_xor(c3, 32, 64);
      }
      if (lane < 8 && n_stats >= 0) {
        atomicAdd(p.stats + (size_t)n_stats * 64 + cc * 8 + e, a);
        atomicAdd(p.stats + ((size_t)p.N + n_stats) * 64 + cc * 8 + e, b);
        if constexpr (RED) atomicAdd(p.stats + ((size_t)2 * p.N + n_stats) * 64 + cc * 8 + e, c3);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) bs[q] = bss[q] = b3[q] = (f32x2_t){0.f, 0.f};
  };
  auto stats_image = [&](int n) {   // wave-uniform: called before the first stream-out item of a tile
    if ((EP == 1 || RED) && n != n_stats) {
      if (n_stats >= 0) flush_stats();
      n_stats = n;
      if constexpr (RED) {      // coefficients of z = c1 * scale + shift for the image being streamed out
        ld8p(p.n_scale + (size_t)n * 64 + cc * 8, sc);
        ld8p(p.n_shift + (size_t)n * 64 + cc * 8, sh);
      }
    }
  };
  auto norm_image = [&](int n) {    // wave-uniform: per-image transform coefficients of the tile about to be staged
    if (NORM && n != n_staged) {
      ld8p(p.n_scale + (size_t)n * 64 + cc * 8, sc);
      ld8p(p.n_shift + (size_t)n * 64 + cc * 8, sh);
      n_staged = n;
    }
  };

  // ---- fragment read addressing: lane -> pixel (row lp >> 4 of the block's two image rows, column lp & 15); everything
  // but the stage base and the (block, tap row) immediates is folded into twelve per-lane offsets
  const int lrow = lp >> 4, lcol = lp & 15;
  unsigned xb[3][4];   // [tap column][k-step], for the stage being read (toggled between the two stages in place)
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      xb[s][ks] = (unsigned)(((wm * NPB * 2 + lrow) * HS + lcol + s) * 128 + (((2 * ks + kg) ^ (((lcol + s) >> 1) & 7)) << 4));

  f32x16_t acc[NPB];

  // prologue: first tile into stage 0
  TileGeo cur = geo_of(tile);
  norm_image(cur.n);
  {
    const unsigned vm0 = chunk_mask(cur, true);
#pragma unroll
    for (int i = 0; i < NCH; ++i) load_chunk(cur.base, vm0, i);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) xform_part(vm0, i, q);
      write_chunk(0, i);
    }
  }
  __syncthreads();

  TileGeo prv = cur;
  bool have_prev = false;
  for (int it = 0; tile < tile_end; ++tile, ++it) {
    const int buf = it & 1;
    const bool more = tile + 1 < tile_end;
    TileGeo nxt = cur;
    if (more) {
      nxt = geo_of(tile + 1);
      norm_image(nxt.n);
    }
    if (have_prev) stats_image(prv.n);
    const unsigned vm = chunk_mask(nxt, more && !(p.dbg & 1));   // chunks of tile t+1 (0: there is none -> loads hit nothing)
    const unsigned om = out_mask(prv, have_prev && !(p.dbg & 4));   // row groups of tile t-1
    const int nbase = nxt.base, pbase = prv.base;
    if (it > 0) {
      const int d = buf ? INBUF : -INBUF;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[s][ks] += d;
    }

    // ---- 36 stages (tap, k-step) x NPB MFMAs; slot k = NPB g + m follows MFMA m of stage g
    bf16x8_t xf[2][NPB];
    auto frag_read = [&](auto G, auto I) {
      constexpr int g = decltype(G)::value, i = decltype(I)::value;
      constexpr int tp = g >> 2, ks = g & 3;
      constexpr int r = TR ? 2 - tp / 3 : tp / 3, s = TR ? 2 - tp % 3 : tp % 3;
      xf[g & 1][i] = *reinterpret_cast<const bf16x8_t*>(smem + xb[s][ks] + (i * 2 + r) * HS * 128);
    };
    auto side = [&](auto K) {
      constexpr int k = decltype(K)::value;   // 0 .. NSLOT - 1 (144 slots per wave and tile)
      // global loads of tile t+1: slots 1, 4, ..., 31
      if constexpr (EP == 4) {
        // two staging phases (chunks 0-5 load at 1.. and are written at 60..; chunks 6-10 load at 75.., written at 122..): 24
        // staging registers live instead of 44 -- this variant also keeps three epilogue tensors in flight
        if constexpr (k >= 1 && k < 1 + 3 * 6 && (k - 1) % 3 == 0) load_chunk(nbase, vm, (k - 1) / 3);
        if constexpr (k >= 75 && k < 75 + 3 * (NCH - 6) && (k - 75) % 3 == 0) load_chunk(nbase, vm, 6 + (k - 75) / 3);
        if constexpr (k >= 60 && k < 60 + 4 * 6 && (k - 60) % 4 == 0) write_chunk(buf ^ 1, (k - 60) / 4);
        if constexpr (k >= 122 && k < 122 + 4 * (NCH - 6) && (k - 122) % 4 == 0) write_chunk(buf ^ 1, 6 + (k - 122) / 4);
      } else if constexpr (NORM && XR64_TWO_PHASE) {
        // two staging phases here as well: chunk 6 + j is loaded right after chunk j has left for LDS (slot XB + XS j + XW + 1) and
        // transformed XS * 6 - XW - 1 = 40 slots later -- 24 staging registers live instead of 44, fewer AGPR spill reloads
        if constexpr (k >= 1 && k < 1 + 3 * 6 && (k - 1) % 3 == 0) load_chunk(nbase, vm, (k - 1) / 3);
        if constexpr (k >= XR64_XB + XR64_XW + 1 && k < XR64_XB + XR64_XW + 1 + XR64_XS * (NCH - 6) && (k - XR64_XB - XR64_XW - 1) % XR64_XS == 0)
          load_chunk(nbase, vm, 6 + (k - XR64_XB - XR64_XW - 1) / XR64_XS);
      } else {
        if constexpr (k >= 1 && k < 1 + 3 * NCH && (k - 1) % 3 == 0) load_chunk(nbase, vm, (k - 1) / 3);
      }
      // stream-out of tile t-1, row group i of NOUT: (epilogue loads at LD + LS i,) LDS read at RD + ST i, the four element-wise
      // parts at RD + ST i + OFS + PS q, the store with the last part.  Variants with epilogue loads leave >= 44 slots (~1.5 us)
      // between a group's loads and its first part and keep 4-5 groups in flight.
      {
        constexpr bool HEAVY = EP == 3 || EP == 4;          // reducing epilogues: 16-26 VALU per part -> a part every third slot
        constexpr int LD = 2, LS = EP == 4 ? 12 : 9;        // EP 4 issues three loads per group (slots LD + LS i + 0 / 1 / 2)
        constexpr int RD = ADD ? 44 : 36, ST = ADD ? 12 : 8, PS = HEAVY ? 3 : ((ADD || NORM) ? 2 : 1);
        constexpr int OFS = NORM ? 3 : 2;                   // NORM: parts on odd slots, the on-load transform parts sit on even ones
        static_assert(RD + ST * (NOUT - 1) + OFS + PS * 3 < NSLOT, "stream-out schedule runs past the tile");
        static_for<0, NOUT>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (ADD && k == LD + LS * i) add_load(pbase, om, i);
          if constexpr (EP == 4 && k == LD + LS * i + 1) tail_load(pbase, om, i, 0);
          if constexpr (EP == 4 && k == LD + LS * i + 2) tail_load(pbase, om, i, 1);
          if constexpr (k == RD + ST * i) out_read(i);
          static_for<0, 4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if constexpr (k == RD + ST * i + OFS + PS * q) out_part(om, i, q);
          });
          if constexpr (k == RD + ST * i + OFS + PS * 3) out_emit(pbase, om, i);
        });
      }
      // transform + LDS write of tile t+1 (the loads left >= 55 slots earlier).  NORM: chunk i is transformed two channels
      // at a time at slots 56 + 8 i + {0, 2, 4, 6} and written at 56 + 8 i + 7; otherwise written at 100 + 4 i
      if constexpr (NORM) {
        constexpr int XB = XR64_XB, XS = XR64_XS, XP = XR64_XP, XW = XR64_XW;   // first slot, slots per chunk, part stride, write offset
        static_assert(XB + XS * (NCH - 1) + XW < NSLOT && 3 * XP < XW && XW < XS + XP, "on-load transform schedule");
        if constexpr (k >= XB && k < XB + XS * NCH && (k - XB) % XS % XP == 0 && (k - XB) % XS / XP < 4)
          xform_part(vm, (k - XB) / XS, (k - XB) % XS / XP);
        if constexpr (k >= XB && k < XB + XS * NCH && (k - XB) % XS == XW) write_chunk(buf ^ 1, (k - XB) / XS);
      } else if constexpr (EP != 4) {
        if constexpr (k >= 100 && k < 100 + 4 * NCH && (k - 100) % 4 == 0) write_chunk(buf ^ 1, (k - 100) / 4);
      }
    };
    static_for<0, NPB>([&](auto I) { frag_read(std::integral_constant<int, 0>{}, I); });
    static_for<0, 36>([&](auto G) {
      constexpr int g = decltype(G)::value;
      static_for<0, NPB>([&](auto M) {
        constexpr int m = decltype(M)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 0) {
          const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][0], xf[0][m], z, 0, 0, 0);
        } else {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g >> 2][g & 3], xf[g & 1][m], acc[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g + 1 < 36) frag_read(std::integral_constant<int, g + 1>{}, M);   // consumed by MFMA m of the next stage
        side(std::integral_constant<int, g * NPB + m>{});
      });
    });
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // stage `buf` and the output image are free; stage buf ^ 1 is complete

    // ---- accumulators (+bias) -> [256 pixels][64 channels] bf16 output image
    if (!(p.dbg & 2)) {
      unsigned char* img = smem + OUTBASE;
      const bool hb = p.bias != nullptr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch0 = wn * 32 + 8 * q + 4 * kg;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (hb) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch0);
          bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
          const int prow = (wm * NPB + i) * 32 + lp;
          uint2 pk;
          pk.x = pack2bf(acc[i][4 * q] + bv[0], acc[i][4 * q + 1] + bv[1]);
          pk.y = pack2bf(acc[i][4 * q + 2] + bv[2], acc[i][4 * q + 3] + bv[3]);
          *reinterpret_cast<uint2*>(img + prow * OPITCH + ch0 * 2) = pk;
        }
      }
    }
    __syncthreads();   // output image complete
    prv = cur;
    cur = nxt;
    have_prev = true;
  }
  // drain: the last tile's output image
  stats_image(prv.n);
  {
    const unsigned om = out_mask(prv, true);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      add_load(prv.base, om, i);
      tail_load(prv.base, om, i, 0);
      tail_load(prv.base, om, i, 1);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      out_read(i);
      out_store(prv.base, om, i);
    }
  }
  if (EP == 1 || RED) flush_stats();
}

template <bool TR, bool NORM, int EP>
int launch_dconv64(DC64P& p, int grid, hipStream_t st) {
  auto kern = dconv64_kernel<TR, NORM, EP>;
  // (idempotent attribute: a racing second caller only repeats it)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  if (e != hipSuccess) {
    xr_set_error("xr_conv64_direct: hipFuncSetAttribute(%d) failed: %s", SMEM, hipGetErrorString(e));
    return XR_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), SMEM, st, p);
  XR_CHECK_LAUNCH("xr_conv64_direct");
  return XR_OK;
}

template <bool TR, bool NORM>
int launch_dconv64_ep(DC64P& p, int grid, hipStream_t st) {
  if (p.stats != nullptr && p.ep_add != nullptr) {
    if constexpr (!NORM) {
      if (p.ep_c != nullptr) return launch_dconv64<TR, false, 4>(p, grid, st);
      return launch_dconv64<TR, false, 3>(p, grid, st);
    } else {
      return XR_E_INVALID;
    }
  }
  if (p.out2 != nullptr) {
    if constexpr (!TR && !NORM) return launch_dconv64<false, false, 5>(p, grid, st);
    else return XR_E_INVALID;
  }
  if (p.stats != nullptr) return launch_dconv64<TR, NORM, 1>(p, grid, st);
  if (p.ep_add != nullptr) return launch_dconv64<TR, NORM, 2>(p, grid, st);
  return launch_dconv64<TR, NORM, 0>(p, grid, st);
}

}  // namespace

static int dconv64_cus() {
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

static int dconv64_run(DC64P& p, int transposed, bool norm, hipStream_t st) {
  p.tiles_x = cdiv(p.W, TS);
  p.tiles_img = p.tiles_x * cdiv(p.H, TS);
  p.ntiles = p.tiles_img * p.N;
  p.dbg = g_tune[14];
  // one persistent workgroup per CU; contiguous tile runs (the tiles of an image stay together: halo rows meet in L2 and
  // the statistics of an image are flushed once)
  p.tpw = cdiv(p.ntiles, dconv64_cus());
  const int grid = cdiv(p.ntiles, p.tpw);
  if (transposed) return norm ? launch_dconv64_ep<true, true>(p, grid, st) : launch_dconv64_ep<true, false>(p, grid, st);
  return norm ? launch_dconv64_ep<false, true>(p, grid, st) : launch_dconv64_ep<false, false>(p, grid, st);
}

extern "C" int xr_conv64_direct(const void* in, const void* wpack, const float* bias, void* out, int N, int H, int W,
                                int transposed, const float* in_scale, const float* in_shift, const float* in_alpha,
                                float* out_stats, const void* ep_add, void* stream) {
  XR_CHECK_ARG(in && wpack && out && N > 0 && H > 0 && W > 0, "xr_conv64_direct: null pointer / non-positive dimension");
  XR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "xr_conv64_direct: in_scale and in_shift come together");
  XR_CHECK_ARG(in_alpha == nullptr || in_scale != nullptr, "xr_conv64_direct: in_alpha needs in_scale / in_shift");
  XR_CHECK_ARG(out_stats == nullptr || ep_add == nullptr, "xr_conv64_direct: output statistics and ep_add are exclusive");
  const long long io_bytes = (long long)N * H * W * 64 * 2;
  XR_CHECK_ARG(io_bytes < (1ll << 31), "xr_conv64_direct: tensor larger than 2 GiB (use xr_conv_igemm)");
  DC64P p{};
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)wpack; p.bias = bias; p.out = (bf16_t*)out; p.ep_add = (const bf16_t*)ep_add;
  p.n_scale = in_scale; p.n_shift = in_shift; p.n_alpha = in_alpha; p.stats = out_stats;
  p.N = N; p.H = H; p.W = W;
  p.io_bytes = (unsigned)io_bytes;
  return dconv64_run(p, transposed, in_scale != nullptr, (hipStream_t)stream);
}

extern "C" int xr_conv64_direct_prelu(const void* in, const void* wpack, void* out, void* out2, const float* alpha, int N, int H,
                                      int W, void* stream) {
  XR_CHECK_ARG(in && wpack && out && out2 && alpha && N > 0 && H > 0 && W > 0, "xr_conv64_direct_prelu: null pointer / non-positive dimension");
  const long long io_bytes = (long long)N * H * W * 64 * 2;
  XR_CHECK_ARG(io_bytes < (1ll << 31), "xr_conv64_direct_prelu: tensor larger than 2 GiB (use xr_conv_igemm)");
  DC64P p{};
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)wpack; p.out = (bf16_t*)out; p.out2 = (bf16_t*)out2; p.n_alpha = alpha;
  p.N = N; p.H = H; p.W = W;
  p.io_bytes = (unsigned)io_bytes;
  return dconv64_run(p, 0, false, (hipStream_t)stream);
}

extern "C" int xr_conv64_direct_bwdred(const void* in, const void* wpack, void* out, int N, int H, int W, int transposed,
                                       const void* red_src, const float* red_scale, const float* red_shift, const float* red_alpha,
                                       float* red, void* stream) {
  XR_CHECK_ARG(in && wpack && out && N > 0 && H > 0 && W > 0, "xr_conv64_direct_bwdred: null pointer / non-positive dimension");
  XR_CHECK_ARG(red_src && red_scale && red_shift && red, "xr_conv64_direct_bwdred: the reduction needs its source, scale, shift and sums");
  const long long io_bytes = (long long)N * H * W * 64 * 2;
  XR_CHECK_ARG(io_bytes < (1ll << 31), "xr_conv64_direct_bwdred: tensor larger than 2 GiB");
  DC64P p{};
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)wpack; p.out = (bf16_t*)out; p.ep_add = (const bf16_t*)red_src;
  p.n_scale = red_scale; p.n_shift = red_shift; p.n_alpha = red_alpha; p.stats = red;
  p.N = N; p.H = H; p.W = W;
  p.io_bytes = (unsigned)io_bytes;
  return dconv64_run(p, transposed, false, (hipStream_t)stream);
}

extern "C" int xr_conv64_direct_tailred(const void* in, const void* wpack, void* out, int N, int H, int W, int transposed,
                                        const void* ep_add, const void* tail_c, const void* tail_x, const float* tail_scale,
                                        const float* tail_shift, const float* tail_alpha, float* red, void* stream) {
  XR_CHECK_ARG(in && wpack && out && N > 0 && H > 0 && W > 0, "xr_conv64_direct_tailred: null pointer / non-positive dimension");
  XR_CHECK_ARG(ep_add && tail_c && tail_x && tail_scale && tail_shift && red,
               "xr_conv64_direct_tailred: needs the residual gradient, the tail's c / x / scale / shift and the sums");
  const long long io_bytes = (long long)N * H * W * 64 * 2;
  XR_CHECK_ARG(io_bytes < (1ll << 31), "xr_conv64_direct_tailred: tensor larger than 2 GiB");
  DC64P p{};
  p.in = (const bf16_t*)in; p.w = (const bf16_t*)wpack; p.out = (bf16_t*)out; p.ep_add = (const bf16_t*)ep_add;
  p.ep_c = (const bf16_t*)tail_c; p.ep_x = (const bf16_t*)tail_x;
  p.n_scale = tail_scale; p.n_shift = tail_shift; p.n_alpha = tail_alpha; p.stats = red;
  p.N = N; p.H = H; p.W = W;
  p.io_bytes = (unsigned)io_bytes;
  return dconv64_run(p, transposed, false, (hipStream_t)stream);
}
