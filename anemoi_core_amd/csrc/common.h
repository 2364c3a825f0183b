// Shared device/host helpers for libanemoi_hip.so (gfx950 / CDNA4 only: wave64, MFMA, DPP).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "anemoi_hip.h"

namespace anemoi {

// ---------------------------------------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define ANEMOI_REQUIRE(cond, ...)             \
  do {                                        \
    if (!(cond)) {                            \
      ::anemoi::set_error(__VA_ARGS__);       \
      return ANEMOI_E_INVALID;                \
    }                                         \
  } while (0)

// ---------------------------------------------------------------------------------------------- dtypes
struct bf16_t {
  uint16_t x;
};
struct f16_t {
  _Float16 x;
};

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16_t v) { return __uint_as_float(((uint32_t)v.x) << 16); }
__device__ __forceinline__ float to_float(f16_t v) { return (float)v.x; }

template <typename T>
__device__ __forceinline__ T from_float(float v);
template <>
__device__ __forceinline__ float from_float<float>(float v) {
  return v;
}
template <>
__device__ __forceinline__ bf16_t from_float<bf16_t>(float v) {
  // hardware round-to-nearest-even conversion (v_cvt_pk_bf16_f32 on gfx950), same rule as torch's float -> bfloat16
  bf16_t r;
  r.x = __builtin_bit_cast(uint16_t, (__bf16)v);
  return r;
}
template <>
__device__ __forceinline__ f16_t from_float<f16_t>(float v) {
  f16_t r;
  r.x = (_Float16)v;
  return r;
}

// Vector of N elements of T moved with the widest possible instructions (N*sizeof(T) in {2,4,8,16,32,64}).
template <typename T, int N>
struct alignas(sizeof(T) * N >= 16 ? 16 : sizeof(T) * N) Vec {
  T v[N];
};

template <typename T, int N>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float (&out)[N]) {
  Vec<T, N> tmp = *reinterpret_cast<const Vec<T, N>*>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = to_float(tmp.v[i]);
}

template <typename T, int N>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const float (&in)[N]) {
  Vec<T, N> tmp;
#pragma unroll
  for (int i = 0; i < N; ++i) tmp.v[i] = from_float<T>(in[i]);
  *reinterpret_cast<Vec<T, N>*>(p) = tmp;
}

// ---------------------------------------------------------------------------------------------- cross-lane
// Butterfly all-reduce (sum) over aligned groups of G lanes (G power of two, <= 64) of a wave64.
// G <= 16 stays inside a DPP row: quad_perm for xor 1/2, row_half_mirror / row_mirror for xor 4/8
// (valid because the lanes being mirrored already hold group-uniform partial sums).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

template <int G>
__device__ __forceinline__ float group_sum(float x) {
  if constexpr (G >= 2) x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
  if constexpr (G >= 4) x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
  if constexpr (G >= 8) x += dpp_f<0x141>(x);  // row_half_mirror
  if constexpr (G >= 16) x += dpp_f<0x140>(x); // row_mirror
  if constexpr (G >= 32) x += __shfl_xor(x, 16, 64);
  if constexpr (G >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

__device__ __forceinline__ float wave_sum(float x) { return group_sum<64>(x); }

// Attention dropout (reference layers/conv.py:145: dropout on the softmax weights, training mode).  The keep / drop decision of
// (CSC edge, head) is a pure function of (seed, edge, head) - a counter-based generator (splitmix64 finaliser of seed + counter,
// top 24 bits as a uniform in [0, 1)) - so the backward pass REPLAYS the forward's mask instead of storing it.
// Returns 1 / (1 - p) for a kept weight, 0 for a dropped one.
__host__ __device__ __forceinline__ float attn_dropout_scale(uint64_t seed, int edge, int head, float p, float inv_keep) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)(uint32_t)edge) << 16 | (uint64_t)(uint32_t)head) + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 resolution of the surrounding arithmetic):
// one v_rcp, one v_exp and five FMAs instead of libm's branchy erff in the GEMM epilogue.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

// GELU for the 16-bit GEMM epilogue (21 M evaluations per MLP launch, all on the VALU while the matrix cores idle):
//   gelu(x) = max(x, 0) - |x| * Phi(-|x|),   Phi(-a) = 2^Q(a),  Q = degree-6 fit of log2(0.5 erfc(a / sqrt 2)) on [0, 6]
// (beyond a = 6 the term is < 6e-9 and a is clamped).  One v_exp_f32 and 9 full-rate ops instead of v_rcp + v_exp + 14:
// max |error| 6.4e-6 over all x, RELATIVE error of the negative branch 1e-4 (the exponent form keeps tiny outputs
// accurate), i.e. 0.4 % of bf16 results differ from the erf form by one ulp - fewer than with the A&S erf above.
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fminf(fabsf(x), 6.0f);
  float q = fmaf(2.301719668e-05f, a, -6.095573365e-04f);
  q = fmaf(q, a, 7.168568210e-03f);
  q = fmaf(q, a, -5.103366076e-02f);
  q = fmaf(q, a, -4.616288390e-01f);
  q = fmaf(q, a, -1.149844046e+00f);
  q = fmaf(q, a, -1.000145551e+00f);
  return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(q), fmaxf(x, 0.0f));
}

// The same GELU on two values with the polynomial in packed fp32 FMAs (v_pk_fma_f32: half the instructions of two scalar
// chains; bit-identical results - the operations and their order are those of gelu_fast).
__device__ __forceinline__ void gelu_fast2(float& x0, float& x1) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const f2 a = {fminf(fabsf(x0), 6.0f), fminf(fabsf(x1), 6.0f)};
  auto k = [](float c) { return f2{c, c}; };
  f2 q = __builtin_elementwise_fma(k(2.301719668e-05f), a, k(-6.095573365e-04f));
  q = __builtin_elementwise_fma(q, a, k(7.168568210e-03f));
  q = __builtin_elementwise_fma(q, a, k(-5.103366076e-02f));
  q = __builtin_elementwise_fma(q, a, k(-4.616288390e-01f));
  q = __builtin_elementwise_fma(q, a, k(-1.149844046e+00f));
  q = __builtin_elementwise_fma(q, a, k(-1.000145551e+00f));
  x0 = fmaf(-fabsf(x0), __builtin_amdgcn_exp2f(q[0]), fmaxf(x0, 0.0f));
  x1 = fmaf(-fabsf(x1), __builtin_amdgcn_exp2f(q[1]), fmaxf(x1, 0.0f));
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute (raised dynamic-LDS limit) is a per-DEVICE setting: one flag per device id, so that a second GPU
// used from the same process gets the limit as well.
// An integer tuning knob from the environment: unset, non-numeric or out-of-range values fall back to the default (a grid of
// 0 workgroups from ANEMOI_ATTN_BLOCKS_PER_CU=0 used to fail with an opaque launch error).
inline int env_int(const char* e, int dflt, int lo, int hi) {
  if (e == nullptr || *e == 0) return dflt;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if (end == e || *end != 0 || v < lo || v > hi) return dflt;
  return (int)v;
}

// Switches that exist for timing experiments only - some of them make a kernel skip part of its work, i.e. return WRONG results - are
// compiled into the library only with -DANEMOI_EXPERIMENTS (`python -m anemoi_core_amd.build --experiments` -> lib/libanemoi_hip_exp.so,
// together with csrc/experiments/*.hip and the in-kernel timeline instantiations).  In the product library they are constants: no
// environment variable can change what a kernel computes, and `nm` / `strings` of the .so show none of their names (tests/test_abi_cpu.py).
#ifdef ANEMOI_EXPERIMENTS
constexpr bool kExperiments = true;
#define ANEMOI_EXPERIMENT_ENV(name, dflt, lo, hi) ::anemoi::env_int(getenv(name), dflt, lo, hi)
#else
constexpr bool kExperiments = false;
#define ANEMOI_EXPERIMENT_ENV(name, dflt, lo, hi) (dflt)
#endif

struct PerDeviceOnce {
  std::once_flag flag[64];
  // Runs f once per device and returns only after it HAS run (a second thread waits instead of launching ahead of the
  // raised LDS limit); device ids beyond the table simply run f every time (hipFuncSetAttribute is idempotent).
  template <typename F>
  void run(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    if (d < 0 || d >= 64) {
      f();
      return;
    }
    std::call_once(flag[d], f);
  }
};

}  // namespace anemoi
