// Shared host/device helpers of lib3dgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tdg.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

// ---------------------------------------------------------------------------- errors
void tdg_set_error(const char* fmt, ...);
void tdg_note_kernel(const char* name);
void tdg_timing_start(const char* name, double flops, hipStream_t s);   // no-ops unless tdg_timing_begin() is active
void tdg_timing_stop(hipStream_t s);

#define TDG_CHECK_ARG(cond, ...)   \
  do {                             \
    if (!(cond)) {                 \
      tdg_set_error(__VA_ARGS__);  \
      return TDG_EINVAL;           \
    }                              \
  } while (0)

#define TDG_HIP_LAUNCH_CHECK(name)                                          \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      tdg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TDG_EHIP;                                                      \
    }                                                                       \
  } while (0)

static inline int tdg_dtype_size(int dtype) { return dtype == TDG_BF16 ? 2 : 4; }
static inline int tdg_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline long long tdg_round_up(long long a, long long b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------- fast division
// q = n / d for 0 <= n < 2^31 with a precomputed (mul, shift): q = (umulhi(n, mul) + n) >> shift
struct FastDiv {
  uint32_t mul, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shift;
}

// ---------------------------------------------------------------------------- dtype helpers (device)
template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float apply_act(float v, int act, float leak) {
  switch (act) {
    case TDG_ACT_RELU: return fmaxf(v, 0.f);
    case TDG_ACT_LRELU: return fmaxf(leak * v, v);
    case TDG_ACT_TANH: return tanhf(v);
    case TDG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// Activation chosen at compile time inside an epilogue loop: a kernel dispatches ONCE on the runtime code
// (dispatch_act) instead of walking a branch tree per accumulator element.  ACT = TDG_ACT_GENERIC keeps the runtime
// switch (tanh / sigmoid: one layer per net).
#define TDG_ACT_GENERIC 99
template <int V>
struct IntC { static constexpr int value = V; };
template <int ACT>
__device__ __forceinline__ float apply_act_c(float v, int act, float leak) {
  if constexpr (ACT == TDG_ACT_NONE) return v;
  else if constexpr (ACT == TDG_ACT_RELU) return fmaxf(v, 0.f);
  else if constexpr (ACT == TDG_ACT_LRELU) return fmaxf(leak * v, v);
  else return apply_act(v, act, leak);
}
template <class F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
  if (act == TDG_ACT_LRELU) f(IntC<TDG_ACT_LRELU>{});
  else if (act == TDG_ACT_RELU) f(IntC<TDG_ACT_RELU>{});
  else if (act == TDG_ACT_NONE) f(IntC<TDG_ACT_NONE>{});
  else f(IntC<TDG_ACT_GENERIC>{});
}
// mask_factor with the mode folded into one value: factor = m > 0 ? 1 : mask_low(mode, leak)
__device__ __forceinline__ float mask_low(int mode, float leak) {
  return mode == TDG_MASK_LRELU ? leak : (mode == TDG_MASK_RELU ? 0.f : 1.f);
}

// derivative factor from the mask source (post-activation for lrelu: sign is preserved;
// pre-activation for relu-after-BN).  TF MaximumGrad: slope `leak` at x <= 0.
__device__ __forceinline__ float mask_factor(float m, int mode, float leak) {
  if (mode == TDG_MASK_LRELU) return m > 0.f ? 1.f : leak;
  if (mode == TDG_MASK_RELU) return m > 0.f ? 1.f : 0.f;
  return 1.f;
}

// d act(pre) / d pre evaluated from the pre-activation value (batch-norm backward)
__device__ __forceinline__ float act_deriv_from_pre(float pre, int act, float leak) {
  switch (act) {
    case TDG_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case TDG_ACT_LRELU: return pre > 0.f ? 1.f : leak;
    case TDG_ACT_TANH: { const float t = tanhf(pre); return 1.f - t * t; }
    case TDG_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-pre)); return s * (1.f - s); }
    default: return 1.f;
  }
}

// ---------------------------------------------------------------------------- XCD-aware block remap
// Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous chunk of
// logical tile ids so that tiles sharing operand panels hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, idx = orig >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
