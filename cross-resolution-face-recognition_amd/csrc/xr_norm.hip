// xr_norm.hip -- normalisation / activation / SE family on NHWC tensors viewed as [G][rows][C].
//
// HBM-bound kernels: every thread owns one fixed 8-channel chunk (16 B of bf16 / 32 B of fp32) so the
// per-channel coefficients live in registers, rows are walked with fully coalesced 16-B-per-lane accesses,
// and the per-(group,channel) reductions are done register -> LDS -> one atomic per block.
//
// Replaces aten::batch_norm / instance_norm / prelu / relu / add / adaptive_avg_pool2d / sigmoid / mul call
// sites: /root/reference model/FSRnet.py:81-98,112-135,319,347,385,434; SUPER_RESOLUTION/model/model_irse.py:
// 23-46,56-66,76-91,141-148; model/resnet.py:24-47,159,167,173.
#include "xr_common.h"


namespace {

constexpr int NT = 256;

struct Geo {
  int G, rows, C, cpr, rpb, active, rows_per_block;
};

static Geo make_geo(int G, int rows, int C, int target_blocks, bool reduces = false) {
  Geo g;
  g.G = G; g.rows = rows; g.C = C;
  g.cpr = C / 8;
  g.rpb = NT / g.cpr;
  g.active = g.rpb * g.cpr;
  int nb = target_blocks / (G > 0 ? G : 1);
  if (nb < 1) nb = 1;
  // deterministic mode: ONE block per group, so every (group, channel) sum is formed by one block in a fixed order (shuffle
  // ladder, LDS hop) and lands in zeroed memory through a single add -- hosts then ask for one group per image / per tile and
  // fold the groups in order (xr_norm_finalize_pivot / xr_norm_bwd_coeffs fold, xr_reduce_groups)
  if (XR_DET() && reduces) nb = 1;
  int rpbk = cdiv(rows, nb);
  rpbk = cdiv(rpbk, g.rpb) * g.rpb;
  if (rpbk < g.rpb) rpbk = g.rpb;
  g.rows_per_block = rpbk;
  return g;
}
static dim3 geo_grid(const Geo& g) { return dim3((unsigned)cdiv(g.rows, g.rows_per_block), (unsigned)g.G); }

__device__ __forceinline__ float act_fwd(float z, float a, int act) {
  if (act == XR_ACT_PRELU) return z > 0.f ? z : a * z;
  if (act == XR_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == XR_ACT_TANH) return tanhf(z);
  return z;
}
__device__ __forceinline__ float act_grad(float z, float a, int act) {
  if (act == XR_ACT_PRELU) return z > 0.f ? 1.f : a;
  if (act == XR_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == XR_ACT_TANH) { const float th = tanhf(z); return 1.f - th * th; }
  return 1.f;
}

__device__ __forceinline__ void load_coef8(const float* p, int idx, float (&v)[8], float dflt) {
  if (p != nullptr) {
    ld8(p + idx, v);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = dflt;
  }
}

// block-level reduction of NV per-thread 8-vectors over the rpb row-lanes, then ONE atomic per (vector, channel).
// Threads sharing a channel chunk sit cpr lanes apart: when cpr is a power of two <= 64 the in-wave part is a
// shuffle-xor ladder (no LDS, no barrier); the four per-wave results meet in LDS once.
template <int NV>
__device__ __forceinline__ void block_reduce_atomic(float (&acc)[NV][8], float* dst, int G, int g, int C, int cpr, int cch,
                                                    int rsub, int rpb, bool active, float* lds) {
  const int t = threadIdx.x;
  const bool pow2 = (cpr & (cpr - 1)) == 0 && cpr <= 64;
  if (pow2) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = acc[v][e];
        for (int o = cpr; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
        acc[v][e] = x;
      }
    // lanes 0..cpr-1 of each wave now hold the wave's sums for chunk (lane % cpr); waves cover different rows
    const int lane = t & 63, wave = t >> 6;
    const int nw = NT / 64;
    if (cpr == 64) {
      // every wave owns whole rows: combine the 4 waves through LDS [nw][NV][C]
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[(wave * NV + v) * C + lane * 8 + e] = acc[v][e];
    } else if (lane < cpr) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[(wave * NV + v) * C + lane * 8 + e] = acc[v][e];
    }
    __syncthreads();
    for (int i = t; i < NV * C; i += NT) {
      const int v = i / C, c = i - v * C;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < nw; ++w) s += lds[(w * NV + v) * C + c];
      atomicAdd(dst + ((size_t)v * G + g) * C + c, s);
    }
    return;
  }
  // generic path: lds [rpb][C] floats, one vector at a time
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) lds[rsub * C + cch * 8 + e] = acc[v][e];
    }
    __syncthreads();
    for (int c = t; c < C; c += NT) {
      float s = 0.f;
      for (int r = 0; r < rpb; ++r) s += lds[r * C + c];
      atomicAdd(dst + ((size_t)v * G + g) * C + c, s);
    }
  }
}

// PIV: the sums are taken over (x - pivot) with pivot[g][c] = x[g][row 0][c] -- the same value in every block of the group, so the
// partial sums stay additive -- and the pivot is stored for xr_norm_finalize_pivot, which rebuilds mean = pivot + S1 / n and
// var = S2 / n - (S1 / n)^2.  E[x^2] - mean^2 on raw fp32 sums loses (mean / std)^2 * 6e-8 of the variance (all of it once
// |mean| reaches a few thousand std); shifted by a sample of the data itself the two terms are O(var) and nothing cancels.
template <typename T, bool PIV>
__global__ __launch_bounds__(NT) void group_stats_kernel(const T* __restrict__ x, float* __restrict__ sums, float* __restrict__ pivot,
                                                         Geo geo) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const bool active = t < geo.active;
  const int cch = t % geo.cpr, rsub = t / geo.cpr;
  const int g = blockIdx.y;
  const int r_begin = blockIdx.x * geo.rows_per_block;
  int r_end = r_begin + geo.rows_per_block;
  if (r_end > geo.rows) r_end = geo.rows;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  if (active) {
    const T* base = x + ((size_t)g * geo.rows) * geo.C + cch * 8;
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pv[e] = 0.f;
    if constexpr (PIV) {
      ld8(base, pv);
      if (blockIdx.x == 0 && rsub == 0) {
        float* pd = pivot + (size_t)g * geo.C + cch * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) pd[e] = pv[e];
      }
    }
    int r = r_begin + rsub;
    for (; r + 3 * geo.rpb < r_end; r += 4 * geo.rpb) {  // 4 independent 16-B loads in flight per lane
      float v0[8], v1[8], v2[8], v3[8];
      ld8(base + (size_t)r * geo.C, v0);
      ld8(base + (size_t)(r + geo.rpb) * geo.C, v1);
      ld8(base + (size_t)(r + 2 * geo.rpb) * geo.C, v2);
      ld8(base + (size_t)(r + 3 * geo.rpb) * geo.C, v3);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (PIV) { v0[e] -= pv[e]; v1[e] -= pv[e]; v2[e] -= pv[e]; v3[e] -= pv[e]; }
        acc[0][e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        acc[1][e] += (v0[e] * v0[e] + v1[e] * v1[e]) + (v2[e] * v2[e] + v3[e] * v3[e]);
      }
    }
    for (; r < r_end; r += geo.rpb) {
      float v[8];
      ld8(base + (size_t)r * geo.C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (PIV) v[e] -= pv[e];
        acc[0][e] += v[e];
        acc[1][e] += v[e] * v[e];
      }
    }
  }
  block_reduce_atomic<2>(acc, sums, geo.G, g, geo.C, geo.cpr, cch, rsub, geo.rpb, active, lds);
}

__global__ void norm_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd,
                                     float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ rmean,
                                     float* __restrict__ rvar, int G, int rows, int C, float eps, float momentum,
                                     const float* __restrict__ pivot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int c = i % C;
  const float n = (float)rows;
  const float m1 = sums[i] / n;                       // mean of (x - pivot)
  const float mu = (pivot ? pivot[i] : 0.f) + m1;
  float var = sums[(size_t)G * C + i] / n - m1 * m1;
  var = var > 0.f ? var : 0.f;
  const float is = rsqrtf(var + eps);
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  if (mean) mean[i] = mu;
  if (invstd) invstd[i] = is;
  scale[i] = ga * is;
  shift[i] = be - mu * ga * is;
  if (rmean != nullptr && G == 1) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
    const float unb = rows > 1 ? var * n / (n - 1.f) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// BatchNorm with partial sums: sums is [2][fold][C] (per-image or pseudo-group partials of ONE statistics group); fold them and
// finalize in the same launch -- (replaces xr_reduce_groups + xr_norm_finalize)
__global__ __launch_bounds__(256) void norm_finalize_fold_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ mean,
                                                                 float* __restrict__ invstd, float* __restrict__ scale,
                                                                 float* __restrict__ shift, float* __restrict__ rmean,
                                                                 float* __restrict__ rvar, int fold, int rows, int C, float eps,
                                                                 float momentum, const float* __restrict__ pivot) {
  // 256 threads = 8 channels x 32 partial lanes: the fold is a latency chain, so it is kept short (fold / 32 loads per lane)
  __shared__ float part[2][32][8];
  const int cl = threadIdx.x & 7, fl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float s0 = 0.f, s1 = 0.f;
  float p0 = 0.f;
  if (c < C) {
    if (pivot != nullptr) {
      // every partial f was taken relative to its OWN pivot p_f (a sample of its rows; equal row counts): re-base to p_0 --
      // sum (x - p0) = S1_f + n_f d, sum (x - p0)^2 = S2_f + 2 d S1_f + n_f d^2 with d = p_f - p0 of the order of one std
      p0 = pivot[c];
      const float nf = (float)rows / (float)fold;
#pragma unroll 4
      for (int f = fl; f < fold; f += 32) {
        const float a = sums[(size_t)f * C + c], b = sums[((size_t)fold + f) * C + c];
        const float d = pivot[(size_t)f * C + c] - p0;
        s0 += a + nf * d;
        s1 += b + d * (2.f * a + nf * d);
      }
    } else {
#pragma unroll 4
      for (int f = fl; f < fold; f += 32) {   // (unrolled: per-image partials, fold = N, keep eight loads in flight)
        s0 += sums[(size_t)f * C + c];
        s1 += sums[((size_t)fold + f) * C + c];
      }
    }
  }
  part[0][fl][cl] = s0;
  part[1][fl][cl] = s1;
  __syncthreads();
  if (fl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 32; ++k) {
    s0 += part[0][k][cl];
    s1 += part[1][k][cl];
  }
  const float n = (float)rows;
  const float m1 = s0 / n;
  const float mu = p0 + m1;
  float var = s1 / n - m1 * m1;
  var = var > 0.f ? var : 0.f;
  const float is = rsqrtf(var + eps);
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  if (mean) mean[c] = mu;
  if (invstd) invstd[c] = is;
  scale[c] = ga * is;
  shift[c] = be - mu * ga * is;
  if (rmean != nullptr) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
    const float unb = rows > 1 ? var * n / (n - 1.f) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float* scale, float* shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = rsqrtf(rvar[c] + eps);
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  scale[c] = ga * is;
  shift[c] = be - rmean[c] * ga * is;
}

struct AffP {
  const void* x; const float* scale; const float* shift; const void* res; const float* alpha; int act;
  void* y; const void* dy; float* red; const float* coef; void* dx; void* dres;
  int coef_per_group;  // scale/shift are [G][C] (1) or [C] (0)
  const void* dx_add;  // optional: added to dx (gradient arriving through an identity branch of the same input)
  // fused second reduction of the backward apply (xr_affine_act_bwd_apply_red): red2[0][g][c] += sum dx, red2[1][g][c] +=
  // sum dx * y2 over the rows of group g; with bcast the coefficients are per channel and shared by all groups
  const void* y2; float* red2; int bcast;
  float* pivot;        // xr_affine_act_stats_pivot: [G][C], the statistics are taken relative to it (and it is written)
  // xr_affine_act_bwd_apply_sub: dx_add is COMPACT, [N][H / s][W / s][C] -- the gradient of a sub-sampled identity branch
  // (MaxPool2d(1, s), model_irse.py:53); it is added at the pixels with h % s == 0 and w % s == 0 only.  0 = dense dx_add.
  int sub_s, sub_H, sub_W;
};

template <typename T> __device__ __forceinline__ float as_stored(float v);   // the value a later pass reads back
template <> __device__ __forceinline__ float as_stored<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float as_stored<float>(float v) { return v; }

// STATS: also sum the (stored) outputs and their squares per (group, channel) into p.red[2][G][C] -- the statistics pass
// of a BatchNorm that reads y next (xr_affine_act_stats)
template <typename T, bool STATS>
__global__ __launch_bounds__(NT) void affine_act_kernel(AffP p, Geo geo) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const bool active = t < geo.active;
  if (!STATS && !active) return;
  const int cch = t % geo.cpr, rsub = t / geo.cpr;
  const int g = blockIdx.y;
  const int r_begin = blockIdx.x * geo.rows_per_block;
  int r_end = r_begin + geo.rows_per_block;
  if (r_end > geo.rows) r_end = geo.rows;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  if (active) {
    const int ci = (p.coef_per_group ? g * geo.C : 0) + cch * 8;
    float sc[8], sh[8], al[8];
    load_coef8(p.scale, ci, sc, 1.f);
    load_coef8(p.shift, ci, sh, 0.f);
    load_coef8(p.alpha, cch * 8, al, 0.f);
    const size_t gbase = ((size_t)g * geo.rows) * geo.C + cch * 8;
    const T* x = reinterpret_cast<const T*>(p.x) + gbase;
    const T* res = p.res ? reinterpret_cast<const T*>(p.res) + gbase : nullptr;
    T* y = reinterpret_cast<T*>(p.y) + gbase;
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pv[e] = 0.f;
    if constexpr (STATS) {
      if (p.pivot != nullptr) {    // pivot = the group's own output at row 0 (every block recomputes the same value)
        float v[8], rv[8];
        ld8(x, v);
        if (res) ld8(res, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = v[e] * sc[e] + sh[e];
          if (res) z += rv[e];
          pv[e] = as_stored<T>(act_fwd(z, al[e], p.act));
        }
        if (blockIdx.x == 0 && rsub == 0) {
          float* pd = p.pivot + (size_t)g * geo.C + cch * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) pd[e] = pv[e];
        }
      }
    }
    for (int r = r_begin + rsub; r < r_end; r += geo.rpb) {
      float v[8], rv[8], o[8];
      ld8(x + (size_t)r * geo.C, v);
      if (res) ld8(res + (size_t)r * geo.C, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float z = v[e] * sc[e] + sh[e];
        if (res) z += rv[e];
        o[e] = act_fwd(z, al[e], p.act);
        if constexpr (STATS) {
          const float q = as_stored<T>(o[e]) - pv[e];
          acc[0][e] += q;
          acc[1][e] += q * q;
        }
      }
      st8(y + (size_t)r * geo.C, o);
    }
  }
  if constexpr (STATS) block_reduce_atomic<2>(acc, p.red, geo.G, g, geo.C, geo.cpr, cch, rsub, geo.rpb, active, lds);
}

template <typename T>
__global__ __launch_bounds__(NT) void affine_act_bwd_reduce_kernel(AffP p, Geo geo) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const bool active = t < geo.active;
  const int cch = t % geo.cpr, rsub = t / geo.cpr;
  const int g = blockIdx.y;
  const int r_begin = blockIdx.x * geo.rows_per_block;
  int r_end = r_begin + geo.rows_per_block;
  if (r_end > geo.rows) r_end = geo.rows;
  float acc[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = acc[2][e] = 0.f;
  if (active) {
    const int ci = (p.coef_per_group ? g * geo.C : 0) + cch * 8;
    float sc[8], sh[8], al[8];
    load_coef8(p.scale, ci, sc, 1.f);
    load_coef8(p.shift, ci, sh, 0.f);
    load_coef8(p.alpha, cch * 8, al, 0.f);
    const size_t gbase = ((size_t)g * geo.rows) * geo.C + cch * 8;
    const T* x = reinterpret_cast<const T*>(p.x) + gbase;
    const T* res = p.res ? reinterpret_cast<const T*>(p.res) + gbase : nullptr;
    const T* dy = reinterpret_cast<const T*>(p.dy) + gbase;
    for (int r = r_begin + rsub; r < r_end; r += geo.rpb) {
      float v[8], rv[8], d[8];
      ld8(x + (size_t)r * geo.C, v);
      ld8(dy + (size_t)r * geo.C, d);
      if (res) ld8(res + (size_t)r * geo.C, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float z = v[e] * sc[e] + sh[e];
        if (res) z += rv[e];
        const float dz = d[e] * act_grad(z, al[e], p.act);
        acc[0][e] += dz;
        acc[1][e] += dz * v[e];
        if (p.act == XR_ACT_PRELU && z <= 0.f) acc[2][e] += d[e] * z;
      }
    }
  }
  block_reduce_atomic<3>(acc, p.red, geo.G, g, geo.C, geo.cpr, cch, rsub, geo.rpb, active, lds);
}

__global__ void norm_bwd_coeffs_kernel(const float* __restrict__ red, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* __restrict__ coef, float* dgamma, float* dbeta, float* dalpha, int G, int rows,
                                       int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float n = (float)rows;
  const float ga = gamma ? gamma[c] : 1.f;
  float dg = 0.f, db = 0.f, da = 0.f;
  const size_t GC = (size_t)G * C;
  // blockIdx.y owns a run of groups (InstanceNorm: G = batch size; one thread walking all of them made this the third
  // most expensive kernel of the FHN step); the per-channel sums of the runs meet by atomics
  const int gpb = (G + gridDim.y - 1) / gridDim.y;
  const int g_beg = blockIdx.y * gpb, g_end = (g_beg + gpb < G) ? g_beg + gpb : G;
  for (int g = g_beg; g < g_end; ++g) {
    const size_t i = (size_t)g * C + c;
    const float s_dz = red[i], s_dzx = red[GC + i];
    const float mu = mean[i], is = invstd[i];
    const float s_dzxhat = is * (s_dzx - mu * s_dz);
    const float m1 = s_dz / n, m2 = s_dzxhat / n;
    const float A = ga * is;
    const float B = -ga * is * is * m2;
    coef[i] = A;
    coef[GC + i] = B;
    coef[2 * GC + i] = -A * m1 - B * mu;
    dg += s_dzxhat;
    db += s_dz;
    da += red[2 * GC + i];
  }
  if (gridDim.y > 1) {
    if (dgamma) atomicAdd(dgamma + c, dg);
    if (dbeta) atomicAdd(dbeta + c, db);
    if (dalpha) atomicAdd(dalpha + c, da);
  } else {
    if (dgamma) dgamma[c] += dg;
    if (dbeta) dbeta[c] += db;
    if (dalpha) dalpha[c] += da;
  }
}

// BatchNorm backward coefficients from PARTIAL reductions: red is [3][fold][C] (pseudo-group partials of the one statistics
// group); fold + coefficients in one launch (replaces xr_reduce_groups + xr_norm_bwd_coeffs); 8 channels x 32 partial lanes
__global__ __launch_bounds__(256) void norm_bwd_coeffs_fold_kernel(const float* __restrict__ red, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   float* __restrict__ coef, float* dgamma, float* dbeta,
                                                                   float* dalpha, int fold, int rows, int C) {
  __shared__ float part[3][32][8];
  const int cl = threadIdx.x & 7, fl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int f = fl; f < fold; f += 32) {
      s0 += red[(size_t)f * C + c];
      s1 += red[((size_t)fold + f) * C + c];
      s2 += red[((size_t)2 * fold + f) * C + c];
    }
  part[0][fl][cl] = s0;
  part[1][fl][cl] = s1;
  part[2][fl][cl] = s2;
  __syncthreads();
  if (fl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 32; ++k) {
    s0 += part[0][k][cl];
    s1 += part[1][k][cl];
    s2 += part[2][k][cl];
  }
  const float n = (float)rows;
  const float ga = gamma ? gamma[c] : 1.f;
  const float mu = mean[c], is = invstd[c];
  const float s_dzxhat = is * (s1 - mu * s0);
  const float m1 = s0 / n, m2 = s_dzxhat / n;
  const float A = ga * is;
  const float B = -ga * is * is * m2;
  coef[c] = A;
  coef[C + c] = B;
  coef[2 * C + c] = -A * m1 - B * mu;
  if (dgamma) dgamma[c] += s_dzxhat;
  if (dbeta) dbeta[c] += s0;
  if (dalpha) dalpha[c] += s2;
}

// sums red[v][g][c] over g into out[c] (used for PReLU-only dalpha and bias gradients)
// blockIdx.y = vector v: out[v][c] (+)= sum_g red[v][g][c]; 256 threads = 8 channels x 32 group lanes -- the fold is a
// latency chain of G / lanes dependent loads, so it gets many lanes rather than wide rows (the data is a few hundred KB in L2)
__global__ void reduce_groups_kernel(const float* __restrict__ red, float* __restrict__ out, int G, int C, int accumulate) {
  __shared__ float part[32][8];
  const int cl = threadIdx.x & 7, gl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const float* base = red + (size_t)blockIdx.y * G * C;
  float s = 0.f;
  if (c < C)
    for (int g = gl; g < G; g += 32) s += base[(size_t)g * C + c];
  part[gl][cl] = s;
  __syncthreads();
  if (gl == 0 && c < C) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) tot += part[k][cl];
    float* o = out + (size_t)blockIdx.y * C + c;
    *o = accumulate ? *o + tot : tot;
  }
}

// RED2: the rows of a group are one image; besides writing dx the kernel sums (dx, dx * y2) per (image, channel) into
// p.red2[2][G][C] -- the backward reduction of the block tail that produced this norm's input (xr_affine_act_bwd_apply_red)
template <typename T, bool RED2>
__global__ __launch_bounds__(NT) void affine_act_bwd_apply_kernel(AffP p, Geo geo) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const bool active = t < geo.active;
  if (!RED2 && !active) return;
  const int cch = t % geo.cpr, rsub = t / geo.cpr;
  const int g = blockIdx.y;
  const int r_begin = blockIdx.x * geo.rows_per_block;
  int r_end = r_begin + geo.rows_per_block;
  if (r_end > geo.rows) r_end = geo.rows;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  if (active) {
    const int gc = p.bcast ? 0 : g * geo.C;
    const int ci = (p.coef_per_group ? gc : 0) + cch * 8;
    float sc[8], sh[8], al[8], cA[8], cB[8], cC[8];
    load_coef8(p.scale, ci, sc, 1.f);
    load_coef8(p.shift, ci, sh, 0.f);
    load_coef8(p.alpha, cch * 8, al, 0.f);
    const size_t GC = p.bcast ? (size_t)geo.C : (size_t)geo.G * geo.C;
    if (p.coef) {
      ld8(p.coef + (size_t)gc + cch * 8, cA);
      ld8(p.coef + GC + (size_t)gc + cch * 8, cB);
      ld8(p.coef + 2 * GC + (size_t)gc + cch * 8, cC);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { cA[e] = sc[e]; cB[e] = 0.f; cC[e] = 0.f; }
    }
    const size_t gbase = ((size_t)g * geo.rows) * geo.C + cch * 8;
    const T* x = reinterpret_cast<const T*>(p.x) + gbase;
    const T* res = p.res ? reinterpret_cast<const T*>(p.res) + gbase : nullptr;
    const T* dy = reinterpret_cast<const T*>(p.dy) + gbase;
    T* dx = p.dx ? reinterpret_cast<T*>(p.dx) + gbase : nullptr;
    T* dres = p.dres ? reinterpret_cast<T*>(p.dres) + gbase : nullptr;
    const T* dxa = p.dx_add ? reinterpret_cast<const T*>(p.dx_add) + gbase : nullptr;
    const T* y2 = RED2 ? reinterpret_cast<const T*>(p.y2) + gbase : nullptr;
    for (int r = r_begin + rsub; r < r_end; r += geo.rpb) {
      float v[8], rv[8], d[8], o[8], dzv[8], ex[8], w[8];
      ld8(x + (size_t)r * geo.C, v);
      ld8(dy + (size_t)r * geo.C, d);
      if (res) ld8(res + (size_t)r * geo.C, rv);
      bool addx = dxa != nullptr;
      if (addx) {
        if (p.sub_s > 0) {      // compact strided add: pixel (n, h, w) of the full grid takes dx_add[n][h / s][w / s] when both divide
          const int pix = g * geo.rows + r;
          const int wq = pix % p.sub_W, hn = pix / p.sub_W, hq = hn % p.sub_H, n = hn / p.sub_H;
          addx = (hq % p.sub_s == 0) && (wq % p.sub_s == 0);
          if (addx)
            ld8(reinterpret_cast<const T*>(p.dx_add) + ((size_t)(n * (p.sub_H / p.sub_s) + hq / p.sub_s) * (p.sub_W / p.sub_s) + wq / p.sub_s) * geo.C + cch * 8, ex);
        } else {
          ld8(dxa + (size_t)r * geo.C, ex);
        }
      }
      if constexpr (RED2) ld8(y2 + (size_t)r * geo.C, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float z = v[e] * sc[e] + sh[e];
        if (res) z += rv[e];
        const float dz = d[e] * act_grad(z, al[e], p.act);
        dzv[e] = dz;
        o[e] = cA[e] * dz + cB[e] * v[e] + cC[e];
        if (addx) o[e] += ex[e];
        if constexpr (RED2) {
          const float q = as_stored<T>(o[e]);
          acc[0][e] += q;
          acc[1][e] += q * w[e];
        }
      }
      if (dx) st8(dx + (size_t)r * geo.C, o);
      if (dres) st8(dres + (size_t)r * geo.C, dzv);
    }
  }
  if constexpr (RED2) block_reduce_atomic<2>(acc, p.red2, geo.G, g, geo.C, geo.cpr, cch, rsub, geo.rpb, active, lds);
}

// --------------------------------------------------------------------------------------------- SE excitation
__global__ __launch_bounds__(NT) void se_excite_fwd_kernel(const float* __restrict__ pooled_sum, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, float* __restrict__ hidden,
                                                           float* __restrict__ s, int C, int Cr, float inv_hw) {
  extern __shared__ float lds[];  // pooled[C] + hid[Cr]
  float* pooled = lds;
  float* hid = lds + C;
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < C; c += NT) pooled[c] = pooled_sum[(size_t)n * C + c] * inv_hw;
  __syncthreads();
  for (int j = wave; j < Cr; j += NT / 64) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w1[(size_t)j * C + c] * pooled[c];
    a = wave_sum(a);
    if (lane == 0) {
      a = a > 0.f ? a : 0.f;
      hid[j] = a;
      hidden[(size_t)n * Cr + j] = a;
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += NT) {
    float a = 0.f;
    for (int j = 0; j < Cr; ++j) a += w2[(size_t)c * Cr + j] * hid[j];
    s[(size_t)n * C + c] = 1.f / (1.f + __expf(-a));
  }
}

__global__ __launch_bounds__(NT) void se_excite_bwd_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                           const float* __restrict__ hidden, const float* __restrict__ s,
                                                           const float* __restrict__ ds, float* __restrict__ dpre2,
                                                           float* __restrict__ dhid, float* __restrict__ dpooled, int C,
                                                           int Cr, float inv_hw) {
  extern __shared__ float lds[];  // d2[C] + dh[Cr]
  float* d2 = lds;
  float* dh = lds + C;
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < C; c += NT) {
    const float sv = s[(size_t)n * C + c];
    const float v = ds[(size_t)n * C + c] * sv * (1.f - sv);
    d2[c] = v;
    dpre2[(size_t)n * C + c] = v;
  }
  __syncthreads();
  for (int j = wave; j < Cr; j += NT / 64) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w2[(size_t)c * Cr + j] * d2[c];
    a = wave_sum(a);
    if (lane == 0) {
      a = hidden[(size_t)n * Cr + j] > 0.f ? a : 0.f;
      dh[j] = a;
      dhid[(size_t)n * Cr + j] = a;
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += NT) {
    float a = 0.f;
    for (int j = 0; j < Cr; ++j) a += w1[(size_t)j * C + c] * dh[j];
    dpooled[(size_t)n * C + c] = a * inv_hw;
  }
}

// squeeze widths the all-at-once dot products below handle (a power of two between 4 and 64: NT / Cr lanes per output fold
// inside one wave); anything else takes the one-output-per-wave loops
__device__ __forceinline__ bool se_fast(int Cr) { return Cr >= 4 && Cr <= 64 && (Cr & (Cr - 1)) == 0; }

// ---- fused BatchNorm -> SE -> (+shortcut) tail of bottleneck_IR_SE (model_irse.py:76-91) ---------------------------
// With r = a*y + b (BatchNorm as a per-channel affine) the SE squeeze is pooled_r = a*mean_hw(y) + b, so r is never
// materialised: out = y*(a*s) + (b*s) + shortcut.  Forward kernel: one block per image.
__global__ __launch_bounds__(NT) void bnse_fwd_kernel(const float* __restrict__ sum_y, const float* __restrict__ a,
                                                      const float* __restrict__ b, const float* __restrict__ w1,
                                                      const float* __restrict__ w2, float* __restrict__ pooled_r_sum,
                                                      float* __restrict__ hidden, float* __restrict__ s,
                                                      float* __restrict__ coefA, float* __restrict__ coefB, int C, int Cr,
                                                      float hw) {
  extern __shared__ float lds[];  // pooled[C] + hid[Cr]
  float* pooled = lds;
  float* hid = lds + C;
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < C; c += NT) {
    const float ps = a[c] * sum_y[(size_t)n * C + c] + hw * b[c];
    pooled_r_sum[(size_t)n * C + c] = ps;
    pooled[c] = ps / hw;
  }
  __syncthreads();
  if (se_fast(Cr)) {
    // all Cr dot products at once: NT / Cr adjacent lanes per output walk c (coalesced rows of w1), shuffle-xor fold -- one
    // round trip to memory instead of Cr / 4 dependent ones per wave (these kernels sit on the critical path of every unit)
    const int G = NT / Cr, j = t / G, cl = t - j * G;
    float v = 0.f;
#pragma unroll 4
    for (int c = cl; c < C; c += G) v += w1[(size_t)j * C + c] * pooled[c];
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (cl == 0) {
      v = v > 0.f ? v : 0.f;
      hid[j] = v;
      hidden[(size_t)n * Cr + j] = v;
    }
  } else {
    for (int j = wave; j < Cr; j += NT / 64) {
      float v = 0.f;
      for (int c = lane; c < C; c += 64) v += w1[(size_t)j * C + c] * pooled[c];
      v = wave_sum(v);
      if (lane == 0) {
        v = v > 0.f ? v : 0.f;
        hid[j] = v;
        hidden[(size_t)n * Cr + j] = v;
      }
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += NT) {
    float v = 0.f;
#pragma unroll 8
    for (int j = 0; j < Cr; ++j) v += w2[(size_t)c * Cr + j] * hid[j];
    const float sv = 1.f / (1.f + __expf(-v));
    s[(size_t)n * C + c] = sv;
    coefA[(size_t)n * C + c] = a[c] * sv;
    coefB[(size_t)n * C + c] = b[c] * sv;
  }
}

// backward, per image: ds = sum_hw dout*r = a*S2 + b*S1 -> excitation backward -> dp (gradient every pixel of r gets
// through the squeeze path)
__global__ __launch_bounds__(NT) void bnse_bwd_excite_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ w1, const float* __restrict__ w2,
                                                             const float* __restrict__ hidden, const float* __restrict__ s,
                                                             float* __restrict__ dpre2, float* __restrict__ dhid,
                                                             float* __restrict__ dp, int C, int Cr, float hw) {
  extern __shared__ float lds[];  // d2[C] + dh[Cr] + part[4 * Cr]
  float* d2 = lds;
  float* dh = lds + C;
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < C; c += NT) {
    const size_t i = (size_t)n * C + c;
    const float ds = a[c] * S2[i] + b[c] * S1[i];
    const float sv = s[i];
    const float v = ds * sv * (1.f - sv);
    d2[c] = v;
    dpre2[i] = v;
  }
  __syncthreads();
  if (se_fast(Cr)) {
    // w2 is [C][Cr]: adjacent lanes walk j (coalesced rows), NT / Cr lane groups walk c; fold the groups of a wave by
    // shuffle-xor (same j sits Cr lanes apart), the four waves through LDS
    float* part = dh + Cr;   // [NT / 64][Cr]
    const int G = NT / Cr, j = t % Cr, cl = t / Cr;
    float v = 0.f;
#pragma unroll 4
    for (int c = cl; c < C; c += G) v += w2[(size_t)c * Cr + j] * d2[c];
    for (int o = Cr; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (lane < Cr) part[wave * Cr + lane] = v;
    __syncthreads();
    if (t < Cr) {
      float u = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) u += part[w * Cr + t];
      u = hidden[(size_t)n * Cr + t] > 0.f ? u : 0.f;
      dh[t] = u;
      dhid[(size_t)n * Cr + t] = u;
    }
  } else {
    for (int j = wave; j < Cr; j += NT / 64) {
      float v = 0.f;
      for (int c = lane; c < C; c += 64) v += w2[(size_t)c * Cr + j] * d2[c];
      v = wave_sum(v);
      if (lane == 0) {
        v = hidden[(size_t)n * Cr + j] > 0.f ? v : 0.f;
        dh[j] = v;
        dhid[(size_t)n * Cr + j] = v;
      }
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += NT) {
    float v = 0.f;
#pragma unroll 8
    for (int j = 0; j < Cr; ++j) v += w1[(size_t)j * C + c] * dh[j];
    dp[(size_t)n * C + c] = v / hw;
  }
}

// backward, per channel: BatchNorm sums of d_r = dout*s + dp folded from the per-image sums, then the coefficients of
// dy = cA[n,c]*dout + cB[c]*y + cC[n,c].  One wave per channel (lanes stride over n).
__global__ __launch_bounds__(NT) void bnse_bwd_coeffs_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                             const float* __restrict__ s, const float* __restrict__ dp,
                                                             const float* __restrict__ sum_y, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ a_eval, float* __restrict__ coef,
                                                             float* dgamma, float* dbeta, int N, int C, float hw, int train) {
  // 256 threads = 8 channels x 32 image lanes (32-B row segments, short dependent-load chains: N / 32 iterations)
  __shared__ float part[2][32][8];
  __shared__ float cf[3][8];
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const bool live = c < C;
  float t1 = 0.f, t2 = 0.f;
  if (live) {
    // eight images per pass with all forty loads issued before the first use: this launch is a latency chain on the critical
    // path of every unit (a rolled loop made it eight dependent round trips)
    for (int n0 = lane; n0 < N; n0 += 32 * 8) {
      float vs[8], v1[8], v2[8], vd[8], vy[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int n = n0 + 32 * k;
        const size_t i = (size_t)(n < N ? n : n0) * C + c;
        vs[k] = n < N ? s[i] : 0.f;
        v1[k] = S1[i]; v2[k] = S2[i]; vd[k] = n < N ? dp[i] : 0.f; vy[k] = sum_y[i];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        t1 += vs[k] * v1[k] + hw * vd[k];
        t2 += vs[k] * v2[k] + vd[k] * vy[k];
      }
    }
  }
  part[0][lane][cl] = t1;
  part[1][lane][cl] = t2;
  __syncthreads();
  float t2hat = 0.f;
  if (lane == 0 && live) {
    t1 = 0.f; t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      t1 += part[0][k][cl];
      t2 += part[1][k][cl];
    }
    const float ga = gamma ? gamma[c] : 1.f;
    const float mu = mean[c], is = invstd[c];
    t2hat = is * (t2 - mu * t1);
    float Abn, Bbn, Cbn;
    if (train) {
      const float ntot = (float)N * hw;
      const float m1 = t1 / ntot, m2 = t2hat / ntot;
      Abn = ga * is;
      Bbn = -ga * is * is * m2;
      Cbn = -Abn * m1 - Bbn * mu;
    } else {
      Abn = a_eval[c];
      Bbn = 0.f;
      Cbn = 0.f;
    }
    cf[0][cl] = Abn; cf[1][cl] = Bbn; cf[2][cl] = Cbn;
  }
  __syncthreads();
  if (live) {
    const float Abn = cf[0][cl], Bbn = cf[1][cl], Cbn = cf[2][cl];
    const size_t NC = (size_t)N * C;
    for (int n = lane; n < N; n += 32) {
      const size_t i = (size_t)n * C + c;
      coef[i] = Abn * s[i];
      coef[NC + i] = Bbn;
      coef[2 * NC + i] = Abn * dp[i] + Cbn;
    }
  }
  if (lane == 0 && live) {
    if (dgamma) dgamma[c] += t2hat;
    if (dbeta) dbeta[c] += t1;
  }
}

// out[i][j] += scale * sum_n A[n][i] * B[n][j]   (tiny outer-product accumulation): blockIdx.y slices the n range,
// 8 independent partial sums per thread keep loads in flight, one atomic per (output, slice)
__global__ void small_atb_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int N,
                                 int I, int J, float scale, int nper) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= I * J) return;
  const int i = idx / J, j = idx - i * J;
  const int n0 = blockIdx.y * nper;
  const int n1 = n0 + nper < N ? n0 + nper : N;
  float a = 0.f;
#pragma unroll 8
  for (int n = n0; n < n1; ++n) a += A[(size_t)n * I + i] * B[(size_t)n * J + j];
  atomicAdd(out + idx, a * scale);
}

template <typename T>
static int launch_aff(void (*kern)(AffP, Geo), AffP& p, const Geo& geo, size_t smem, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(kern, geo_grid(geo), dim3(NT), smem, st, p, geo);
  XR_CHECK_LAUNCH(name);
  return XR_OK;
}

static int check_geo(const char* name, int dtype, int G, int rows, int C) {
  XR_CHECK_ARG(dtype == XR_BF16 || dtype == XR_F32, "%s: bad dtype %d", name, dtype);
  XR_CHECK_ARG(G > 0 && rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "%s: bad geometry G=%d rows=%d C=%d", name, G, rows, C);
  XR_CHECK_ARG(G <= 65535, "%s: G=%d too large", name, G);
  return XR_OK;
}

}  // namespace

static int group_stats_run(int dtype, const void* x, float* sums, float* pivot, int G, int rows, int C, void* stream) {
  if (int e = check_geo("xr_group_stats", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && sums, "xr_group_stats: null pointer");
  Geo geo = make_geo(G, rows, C, g_tune[8], true);
  const size_t smem = (size_t)(geo.rpb > 8 ? geo.rpb : 8) * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XR_BF16) {
    if (pivot) hipLaunchKernelGGL((group_stats_kernel<bf16_t, true>), geo_grid(geo), dim3(NT), smem, st, (const bf16_t*)x, sums, pivot, geo);
    else hipLaunchKernelGGL((group_stats_kernel<bf16_t, false>), geo_grid(geo), dim3(NT), smem, st, (const bf16_t*)x, sums, pivot, geo);
  } else {
    if (pivot) hipLaunchKernelGGL((group_stats_kernel<float, true>), geo_grid(geo), dim3(NT), smem, st, (const float*)x, sums, pivot, geo);
    else hipLaunchKernelGGL((group_stats_kernel<float, false>), geo_grid(geo), dim3(NT), smem, st, (const float*)x, sums, pivot, geo);
  }
  XR_CHECK_LAUNCH("xr_group_stats");
  return XR_OK;
}

extern "C" int xr_group_stats(int dtype, const void* x, float* sums, int G, int rows, int C, void* stream) {
  return group_stats_run(dtype, x, sums, nullptr, G, rows, C, stream);
}

extern "C" int xr_group_stats_pivot(int dtype, const void* x, float* sums, float* pivot, int G, int rows, int C, void* stream) {
  XR_CHECK_ARG(pivot, "xr_group_stats_pivot: null pivot");
  return group_stats_run(dtype, x, sums, pivot, G, rows, C, stream);
}

extern "C" int xr_norm_finalize(const float* sums, const float* gamma, const float* beta, float* mean, float* invstd,
                                float* scale, float* shift, float* running_mean, float* running_var, int G, int rows, int C,
                                float eps, float momentum, int fold, void* stream) {
  return xr_norm_finalize_pivot(sums, nullptr, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, G, rows, C, eps,
                                momentum, fold, stream);
}

extern "C" int xr_norm_finalize_pivot(const float* sums, const float* pivot, const float* gamma, const float* beta, float* mean,
                                      float* invstd, float* scale, float* shift, float* running_mean, float* running_var, int G,
                                      int rows, int C, float eps, float momentum, int fold, void* stream) {
  XR_CHECK_ARG(sums && scale && shift, "xr_norm_finalize: null pointer");
  XR_CHECK_ARG(pivot == nullptr || fold <= 1 || rows % fold == 0, "xr_norm_finalize_pivot: pivoted partials must cover equal row counts");
  XR_CHECK_ARG(G > 0 && rows > 0 && C > 0, "xr_norm_finalize: bad geometry");
  XR_CHECK_ARG(fold <= 1 || G == 1, "xr_norm_finalize: partial sums (fold > 1) belong to one statistics group (G == 1)");
  if (fold > 1) {
    XR_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "xr_norm_finalize: running stats must come in pairs");
    hipLaunchKernelGGL(norm_finalize_fold_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, mean,
                       invstd, scale, shift, running_mean, running_var, fold, rows, C, eps, momentum, pivot);
    XR_CHECK_LAUNCH("xr_norm_finalize");
    return XR_OK;
  }
  XR_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "xr_norm_finalize: running stats must come in pairs");
  XR_CHECK_ARG(running_mean == nullptr || G == 1, "xr_norm_finalize: running statistics need G == 1");
  const int n = G * C;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, mean,
                     invstd, scale, shift, running_mean, running_var, G, rows, C, eps, momentum, pivot);
  XR_CHECK_LAUNCH("xr_norm_finalize");
  return XR_OK;
}

extern "C" int xr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float* scale, float* shift, int C, float eps, void* stream) {
  XR_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "xr_bn_eval_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, scale, shift, C, eps);
  XR_CHECK_LAUNCH("xr_bn_eval_coeffs");
  return XR_OK;
}

extern "C" int xr_affine_act(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                             const float* alpha, int act, void* y, int G, int rows, int C, int coef_per_group, void* stream) {
  if (int e = check_geo("xr_affine_act", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && y, "xr_affine_act: null pointer");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act: PReLU needs alpha");
  AffP p{x, scale, shift, res, alpha, act, y, nullptr, nullptr, nullptr, nullptr, nullptr, coef_per_group, nullptr};
  Geo geo = make_geo(G, rows, C, 4096);
  if (dtype == XR_BF16) return launch_aff<bf16_t>(affine_act_kernel<bf16_t, false>, p, geo, 0, (hipStream_t)stream, "xr_affine_act");
  return launch_aff<float>(affine_act_kernel<float, false>, p, geo, 0, (hipStream_t)stream, "xr_affine_act");
}

extern "C" int xr_affine_act_stats(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                   const float* alpha, int act, void* y, float* stats, int G, int rows, int C,
                                   int coef_per_group, void* stream) {
  return xr_affine_act_stats_pivot(dtype, x, scale, shift, res, alpha, act, y, stats, nullptr, G, rows, C, coef_per_group, stream);
}

extern "C" int xr_affine_act_stats_pivot(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                         const float* alpha, int act, void* y, float* stats, float* pivot, int G, int rows, int C,
                                         int coef_per_group, void* stream) {
  if (int e = check_geo("xr_affine_act_stats", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && y && stats, "xr_affine_act_stats: null pointer");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act_stats: PReLU needs alpha");
  AffP p{x, scale, shift, res, alpha, act, y, nullptr, stats, nullptr, nullptr, nullptr, coef_per_group, nullptr};
  p.pivot = pivot;
  Geo geo = make_geo(G, rows, C, g_tune[8], true);
  const size_t smem = (size_t)(geo.rpb > 8 ? geo.rpb : 8) * C * sizeof(float);
  if (dtype == XR_BF16)
    return launch_aff<bf16_t>(affine_act_kernel<bf16_t, true>, p, geo, smem, (hipStream_t)stream, "xr_affine_act_stats");
  return launch_aff<float>(affine_act_kernel<float, true>, p, geo, smem, (hipStream_t)stream, "xr_affine_act_stats");
}

extern "C" int xr_affine_act_bwd_reduce(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                        const float* alpha, int act, const void* dy, float* red, int G, int rows, int C,
                                        int coef_per_group, void* stream) {
  if (int e = check_geo("xr_affine_act_bwd_reduce", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && dy && red, "xr_affine_act_bwd_reduce: null pointer");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act_bwd_reduce: PReLU needs alpha");
  AffP p{x, scale, shift, res, alpha, act, nullptr, dy, red, nullptr, nullptr, nullptr, coef_per_group, nullptr};
  Geo geo = make_geo(G, rows, C, g_tune[9], true);
  const size_t smem = (size_t)(geo.rpb > 12 ? geo.rpb : 12) * C * sizeof(float);
  if (dtype == XR_BF16)
    return launch_aff<bf16_t>(affine_act_bwd_reduce_kernel<bf16_t>, p, geo, smem, (hipStream_t)stream, "xr_affine_act_bwd_reduce");
  return launch_aff<float>(affine_act_bwd_reduce_kernel<float>, p, geo, smem, (hipStream_t)stream, "xr_affine_act_bwd_reduce");
}

extern "C" int xr_norm_bwd_coeffs(const float* red, const float* gamma, const float* mean, const float* invstd, float* coef,
                                  float* dgamma, float* dbeta, float* dalpha, int G, int rows, int C, int fold, void* stream) {
  XR_CHECK_ARG(red && mean && invstd && coef && G > 0 && rows > 0 && C > 0, "xr_norm_bwd_coeffs: bad arguments");
  XR_CHECK_ARG(fold <= 1 || G == 1, "xr_norm_bwd_coeffs: partial reductions (fold > 1) belong to one statistics group (G == 1)");
  if (fold > 1) {
    hipLaunchKernelGGL(norm_bwd_coeffs_fold_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, red, gamma, mean, invstd,
                       coef, dgamma, dbeta, dalpha, fold, rows, C);
    XR_CHECK_LAUNCH("xr_norm_bwd_coeffs");
    return XR_OK;
  }
  const int gy = (G >= 8 && !XR_DET()) ? (G < 64 ? G : 64) : 1;   // (y blocks meet in dgamma / dbeta / dalpha by atomics)
  hipLaunchKernelGGL(norm_bwd_coeffs_kernel, dim3(cdiv(C, 128), gy), dim3(128), 0, (hipStream_t)stream, red, gamma, mean, invstd,
                     coef, dgamma, dbeta, dalpha, G, rows, C);
  XR_CHECK_LAUNCH("xr_norm_bwd_coeffs");
  return XR_OK;
}

extern "C" int xr_reduce_groups(const float* red, float* out, int NV, int G, int C, int accumulate, void* stream) {
  XR_CHECK_ARG(red && out && NV > 0 && G > 0 && C > 0, "xr_reduce_groups: bad arguments");
  hipLaunchKernelGGL(reduce_groups_kernel, dim3(cdiv(C, 8), NV), dim3(256), 0, (hipStream_t)stream, red, out, G, C, accumulate);
  XR_CHECK_LAUNCH("xr_reduce_groups");
  return XR_OK;
}

extern "C" int xr_affine_act_bwd_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                       const float* alpha, int act, const void* dy, const float* coef, void* dx, void* dres,
                                       int G, int rows, int C, int coef_per_group, const void* dx_add, void* stream) {
  if (int e = check_geo("xr_affine_act_bwd_apply", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && dy && (dx || dres), "xr_affine_act_bwd_apply: null pointer");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act_bwd_apply: PReLU needs alpha");
  AffP p{x, scale, shift, res, alpha, act, nullptr, dy, nullptr, coef, dx, dres, coef_per_group, dx_add};
  Geo geo = make_geo(G, rows, C, 4096);
  if (dtype == XR_BF16)
    return launch_aff<bf16_t>(affine_act_bwd_apply_kernel<bf16_t, false>, p, geo, 0, (hipStream_t)stream, "xr_affine_act_bwd_apply");
  return launch_aff<float>(affine_act_bwd_apply_kernel<float, false>, p, geo, 0, (hipStream_t)stream, "xr_affine_act_bwd_apply");
}

extern "C" int xr_affine_act_bwd_apply_sub(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                           const float* alpha, int act, const void* dy, const float* coef, void* dx, int N, int H, int W,
                                           int C, const void* dx_add_sub, int sub_stride, const void* y2, float* red2, void* stream) {
  if (int e = check_geo("xr_affine_act_bwd_apply_sub", dtype, N, H * W, C)) return e;
  XR_CHECK_ARG(x && dy && dx && dx_add_sub, "xr_affine_act_bwd_apply_sub: null pointer");
  XR_CHECK_ARG(N > 0 && H > 0 && W > 0 && sub_stride >= 2 && H % sub_stride == 0 && W % sub_stride == 0,
               "xr_affine_act_bwd_apply_sub: H and W must be multiples of the stride");
  XR_CHECK_ARG((y2 == nullptr) == (red2 == nullptr), "xr_affine_act_bwd_apply_sub: y2 and red2 come together");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act_bwd_apply_sub: PReLU needs alpha");
  hipStream_t st = (hipStream_t)stream;
  if (red2 != nullptr) {   // chained with the previous unit's tail: one group per image, shared coefficients (xr_affine_act_bwd_apply_red)
    AffP p{x, scale, shift, res, alpha, act, nullptr, dy, nullptr, coef, dx, nullptr, 1, dx_add_sub, y2, red2, 1};
    p.sub_s = sub_stride; p.sub_H = H; p.sub_W = W;
    Geo geo = make_geo(N, H * W, C, 2 * g_tune[9], true);
    const size_t smem = (size_t)(geo.rpb > 8 ? geo.rpb : 8) * C * sizeof(float);
    if (dtype == XR_BF16) return launch_aff<bf16_t>(affine_act_bwd_apply_kernel<bf16_t, true>, p, geo, smem, st, "xr_affine_act_bwd_apply_sub");
    return launch_aff<float>(affine_act_bwd_apply_kernel<float, true>, p, geo, smem, st, "xr_affine_act_bwd_apply_sub");
  }
  AffP p{x, scale, shift, res, alpha, act, nullptr, dy, nullptr, coef, dx, nullptr, 1, dx_add_sub};
  p.sub_s = sub_stride; p.sub_H = H; p.sub_W = W;
  Geo geo = make_geo(1, N * H * W, C, 4096);
  if (dtype == XR_BF16) return launch_aff<bf16_t>(affine_act_bwd_apply_kernel<bf16_t, false>, p, geo, 0, st, "xr_affine_act_bwd_apply_sub");
  return launch_aff<float>(affine_act_bwd_apply_kernel<float, false>, p, geo, 0, st, "xr_affine_act_bwd_apply_sub");
}

extern "C" int xr_affine_act_bwd_apply_red(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                                           const float* alpha, int act, const void* dy, const float* coef, void* dx, void* dres,
                                           int G, int rows, int C, const void* dx_add, const void* y2, float* red2,
                                           void* stream) {
  if (int e = check_geo("xr_affine_act_bwd_apply_red", dtype, G, rows, C)) return e;
  XR_CHECK_ARG(x && dy && dx && y2 && red2, "xr_affine_act_bwd_apply_red: null pointer");
  XR_CHECK_ARG(act != XR_ACT_PRELU || alpha, "xr_affine_act_bwd_apply_red: PReLU needs alpha");
  AffP p{x, scale, shift, res, alpha, act, nullptr, dy, nullptr, coef, dx, dres, 1, dx_add, y2, red2, 1};
  Geo geo = make_geo(G, rows, C, 2 * g_tune[9], true);   // five streams per row: wants twice the blocks of the plain reduce (tools/norm_bench.py)
  const size_t smem = (size_t)(geo.rpb > 8 ? geo.rpb : 8) * C * sizeof(float);
  if (dtype == XR_BF16)
    return launch_aff<bf16_t>(affine_act_bwd_apply_kernel<bf16_t, true>, p, geo, smem, (hipStream_t)stream,
                              "xr_affine_act_bwd_apply_red");
  return launch_aff<float>(affine_act_bwd_apply_kernel<float, true>, p, geo, smem, (hipStream_t)stream, "xr_affine_act_bwd_apply_red");
}

extern "C" int xr_se_excite_fwd(const float* pooled_sum, const float* w1, const float* w2, float* hidden, float* s, int N,
                                int C, int Cr, float inv_hw, void* stream) {
  XR_CHECK_ARG(pooled_sum && w1 && w2 && hidden && s && N > 0 && C > 0 && Cr > 0 && C <= 4096, "xr_se_excite_fwd: bad arguments");
  hipLaunchKernelGGL(se_excite_fwd_kernel, dim3(N), dim3(NT), (C + Cr) * sizeof(float), (hipStream_t)stream, pooled_sum, w1, w2,
                     hidden, s, C, Cr, inv_hw);
  XR_CHECK_LAUNCH("xr_se_excite_fwd");
  return XR_OK;
}

extern "C" int xr_se_excite_bwd(const float* w1, const float* w2, const float* hidden, const float* s, const float* ds,
                                float* dpre2, float* dhid, float* dpooled, int N, int C, int Cr, float inv_hw, void* stream) {
  XR_CHECK_ARG(w1 && w2 && hidden && s && ds && dpre2 && dhid && dpooled && N > 0 && C > 0 && Cr > 0 && C <= 4096,
               "xr_se_excite_bwd: bad arguments");
  hipLaunchKernelGGL(se_excite_bwd_kernel, dim3(N), dim3(NT), (C + Cr) * sizeof(float), (hipStream_t)stream, w1, w2, hidden, s,
                     ds, dpre2, dhid, dpooled, C, Cr, inv_hw);
  XR_CHECK_LAUNCH("xr_se_excite_bwd");
  return XR_OK;
}

extern "C" int xr_bnse_fwd(const float* sum_y, const float* a, const float* b, const float* w1, const float* w2,
                           float* pooled_r_sum, float* hidden, float* s, float* coefA, float* coefB, int N, int C, int Cr, int HW,
                           void* stream) {
  XR_CHECK_ARG(sum_y && a && b && w1 && w2 && pooled_r_sum && hidden && s && coefA && coefB && N > 0 && C > 0 && Cr > 0 &&
                   C <= 4096 && HW > 0,
               "xr_bnse_fwd: bad arguments");
  hipLaunchKernelGGL(bnse_fwd_kernel, dim3(N), dim3(NT), (C + Cr) * sizeof(float), (hipStream_t)stream, sum_y, a, b, w1, w2,
                     pooled_r_sum, hidden, s, coefA, coefB, C, Cr, (float)HW);
  XR_CHECK_LAUNCH("xr_bnse_fwd");
  return XR_OK;
}

extern "C" int xr_bnse_bwd(const float* S1, const float* S2, const float* sum_y, const float* a, const float* b,
                           const float* w1, const float* w2, const float* hidden, const float* s, const float* gamma,
                           const float* mean, const float* invstd, float* dpre2, float* dhid, float* dp, float* coef,
                           float* dgamma, float* dbeta, int N, int C, int Cr, int HW, int train, void* stream) {
  XR_CHECK_ARG(S1 && S2 && sum_y && a && b && w1 && w2 && hidden && s && mean && invstd && dpre2 && dhid && dp && coef &&
                   N > 0 && C > 0 && Cr > 0 && C <= 4096 && HW > 0,
               "xr_bnse_bwd: bad arguments");
  hipLaunchKernelGGL(bnse_bwd_excite_kernel, dim3(N), dim3(NT), (C + 5 * Cr) * sizeof(float), (hipStream_t)stream, S1, S2, a, b, w1,
                     w2, hidden, s, dpre2, dhid, dp, C, Cr, (float)HW);
  XR_CHECK_LAUNCH("xr_bnse_bwd(excite)");
  hipLaunchKernelGGL(bnse_bwd_coeffs_kernel, dim3(cdiv(C, 8)), dim3(NT), 0, (hipStream_t)stream, S1, S2, s, dp, sum_y,
                     gamma, mean, invstd, a, coef, dgamma, dbeta, N, C, (float)HW, train);
  XR_CHECK_LAUNCH("xr_bnse_bwd(coeffs)");
  return XR_OK;
}

extern "C" int xr_small_atb(const float* A, const float* B, float* out, int N, int I, int J, float scale, int accumulate,
                            void* stream) {
  XR_CHECK_ARG(A && B && out && N > 0 && I > 0 && J > 0, "xr_small_atb: bad arguments");
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, (size_t)I * J * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      xr_set_error("xr_small_atb: memset failed");
      return XR_E_LAUNCH;
    }
  }
  int slices = (N >= 64 && !XR_DET()) ? 16 : 1;
  const int nper = cdiv(N, slices);
  slices = cdiv(N, nper);
  hipLaunchKernelGGL(small_atb_kernel, dim3(cdiv((long long)I * J, 256), slices), dim3(256), 0, (hipStream_t)stream, A, B, out,
                     N, I, J, scale, nper);
  XR_CHECK_LAUNCH("xr_small_atb");
  return XR_OK;
}
