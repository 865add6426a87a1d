// xr_common.h -- shared device/host helpers for the gfx950 (CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include "xrface.h"

// Launch-configuration selectors (xr_tune): they pick WHICH kernel variant / tile a launcher uses, never what it computes.
// Relaxed atomics: launchers are called concurrently from autograd worker threads, xr_tune is a test / A-B hook that the
// product path calls only at library load (XR_TUNE) -- a reader sees either the old or the new selector, never a torn one.
struct XrTune {
  std::atomic<int> v;
  XrTune(int x = 0) : v(x) {}
  operator int() const { return v.load(std::memory_order_relaxed); }
  XrTune& operator=(int x) { v.store(x, std::memory_order_relaxed); return *this; }
};
extern XrTune g_tune[20];
#define XR_DET() (g_tune[16] != 0)   // deterministic reductions (xr_set_deterministic): no order-dependent fp32 atomics

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

void xr_set_error(const char* fmt, ...);

#define XR_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      xr_set_error(__VA_ARGS__);         \
      return XR_E_INVALID;               \
    }                                    \
  } while (0)

#define XR_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      xr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return XR_E_LAUNCH;                                                  \
    }                                                                      \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
  return __builtin_bit_cast(unsigned short, b);
}
// pack two floats -> two bf16 in one dword (lo = a, hi = b): ONE v_cvt_pk_bf16_f32 (the element-wise form costs two conversions
// plus three bit operations -- a fifth of the VALU work of the fused direct-convolution epilogues, which are issue-bound)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ unsigned pack2bf(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
// the two bf16 of a dword as floats (lo, hi)
__device__ __forceinline__ f32x2_t unpack2bf(unsigned w) { return (f32x2_t){__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)}; }

// split fp32 into hi + lo bf16 (x ~= hi + lo, |err| <~ 2^-17 |x|)
__device__ __forceinline__ void split_bf(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

template <typename T> struct XrT;
template <> struct XrT<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct XrT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// load / store 8 consecutive elements as fp32 (16 B for bf16, 32 B for f32); pointers 16-B aligned
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]); u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
