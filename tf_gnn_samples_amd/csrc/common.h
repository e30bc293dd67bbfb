// Shared device/host helpers for librelgnn (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/relgnn.h"

#define RELGNN_WAVE 64

namespace relgnn {

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Launch-error check without synchronising: hipGetLastError only reports launch failures.
static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RELGNN_OK : RELGNN_EHIP;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Grid size helper for flat elementwise kernels: cap at 256 CUs x 8 blocks and grid-stride.
static inline unsigned flat_grid(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return static_cast<unsigned>(g);
}

// ---- activations (utils/utils.py:36-58) -----------------------------------------------
// tanh/relu/leaky_relu(0.2)/elu/selu/gelu(erf).  Evaluated in fp32 like TF's CPU kernels.
template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if constexpr (ACT == RELGNN_ACT_LINEAR) return x;
  if constexpr (ACT == RELGNN_ACT_TANH) return tanhf(x);
  if constexpr (ACT == RELGNN_ACT_RELU) return x > 0.f ? x : 0.f;
  if constexpr (ACT == RELGNN_ACT_LEAKY_RELU) return x > 0.f ? x : 0.2f * x;
  if constexpr (ACT == RELGNN_ACT_ELU) return x > 0.f ? x : expf(x) - 1.f;  // TF: exp(x) - 1, not expm1
  if constexpr (ACT == RELGNN_ACT_SELU) {
    const float scale = 1.0507009873554804934193349852946f;
    const float scale_alpha = 1.7580993408473768599402175208123f;  // scale * alpha
    return x > 0.f ? scale * x : scale_alpha * (expf(x) - 1.f);
  }
  if constexpr (ACT == RELGNN_ACT_GELU) {
    // x * 0.5 * (1 + erf(x / sqrt(2)))   -- utils/utils.py:53-55
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));  // x / sqrt(2) as a multiply (<= 1 ulp apart)
    return x * cdf;
  }
  return x;
}

// derivative w.r.t. the pre-activation x
template <int ACT>
__device__ __forceinline__ float act_grad(float x) {
  if constexpr (ACT == RELGNN_ACT_LINEAR) return 1.f;
  if constexpr (ACT == RELGNN_ACT_TANH) {
    float t = tanhf(x);
    return 1.f - t * t;
  }
  if constexpr (ACT == RELGNN_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if constexpr (ACT == RELGNN_ACT_LEAKY_RELU) return x > 0.f ? 1.f : 0.2f;
  if constexpr (ACT == RELGNN_ACT_ELU) return x > 0.f ? 1.f : expf(x);
  if constexpr (ACT == RELGNN_ACT_SELU) {
    const float scale = 1.0507009873554804934193349852946f;
    const float scale_alpha = 1.7580993408473768599402175208123f;
    return x > 0.f ? scale : scale_alpha * expf(x);
  }
  if constexpr (ACT == RELGNN_ACT_GELU) {
    const float inv_sqrt2 = 0.70710678118654752440f;
    const float inv_sqrt_2pi = 0.39894228040143267794f;
    float cdf = 0.5f * (1.0f + erff(x * inv_sqrt2));
    float pdf = inv_sqrt_2pi * expf(-0.5f * x * x);
    return cdf + x * pdf;
  }
  return 1.f;
}

// ---- per-MESSAGE activations ---------------------------------------------------------------------------------------
// The edge kernels evaluate the activation once per message and feature (C2 shape: 4.7e8 evaluations per launch).  With
// the library erff / expf / tanhf (~35-50 VALU slots per GELU) those kernels are ALU-bound: the Edge-MLP0 forward ran
// 681 us against 249 us for the same gather with ReLU.  These variants use the hardware's v_exp_f32 / v_rcp_f32
// (1 ulp each) and a branch-free two-piece erf (|error| <= 1e-7 absolute): the same error class as the 1-2 ulp of the
// library calls they replace, far inside the 1e-5 parity tolerance.
// Per-NODE epilogues keep act_fwd / act_grad above.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }  // x <= 0 here
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float tanh_fast(float x) {
  const float e = exp_fast(-2.f * fabsf(x));  // (0, 1]
  return copysignf((1.f - e) * rcp_fast(1.f + e), x);
}

// erf(z), branch-free: |z| <= 1: z + z p(z^2);  else 1 - 2^(-q(|z|)) (q fitted to -log2 erfc on [1, 4.2], where fp32
// erf saturates).  Coefficients: weighted least squares on Chebyshev nodes (scripts/fit_fast_erf.py); evaluated in fp32
// the absolute error is <= 9e-8 on either side.
__device__ __forceinline__ float erf_fast(float z) {
  const float az = fabsf(z);
  const float s = z * z;
  float p = fmaf(s, -0.0005489283939823508f, 0.004878316540271044f);
  p = fmaf(s, p, -0.02667193114757538f);
  p = fmaf(s, p, 0.11278452724218369f);
  p = fmaf(s, p, -0.3761201798915863f);
  p = fmaf(s, p, 0.1283789724111557f);
  const float small = fmaf(az, p, az);
  const float a = fminf(az, 4.2f);
  float q = fmaf(a, -0.00026959332171827555f, 0.004258748143911362f);   // -log2 erfc(a): log2(e) folded in
  q = fmaf(a, q, -0.031527601182460785f);
  q = fmaf(a, q, 0.1487920582294464f);
  q = fmaf(a, q, 0.9206075072288513f);
  q = fmaf(a, q, 1.6260673999786377f);
  q = fmaf(a, q, 0.00048813220928423107f);
  const float large = 1.f - __builtin_amdgcn_exp2f(-q);
  return copysignf(az <= 1.f ? small : large, z);
}

template <int ACT>
__device__ __forceinline__ float act_fwd_fast(float x) {
  if constexpr (ACT == RELGNN_ACT_TANH) return tanh_fast(x);
  else if constexpr (ACT == RELGNN_ACT_ELU) return x > 0.f ? x : exp_fast(fminf(x, 0.f)) - 1.f;
  else if constexpr (ACT == RELGNN_ACT_SELU) {
    const float scale = 1.0507009873554804934193349852946f;
    const float scale_alpha = 1.7580993408473768599402175208123f;
    return x > 0.f ? scale * x : scale_alpha * (exp_fast(fminf(x, 0.f)) - 1.f);
  } else if constexpr (ACT == RELGNN_ACT_GELU) {
    const float hx = 0.5f * x;
    return fmaf(hx, erf_fast(x * 0.70710678118654752440f), hx);
  } else return act_fwd<ACT>(x);
}

template <int ACT>
__device__ __forceinline__ float act_grad_fast(float x) {
  if constexpr (ACT == RELGNN_ACT_TANH) {
    const float t = tanh_fast(x);
    return 1.f - t * t;
  } else if constexpr (ACT == RELGNN_ACT_ELU) return x > 0.f ? 1.f : exp_fast(fminf(x, 0.f));
  else if constexpr (ACT == RELGNN_ACT_SELU) {
    const float scale = 1.0507009873554804934193349852946f;
    const float scale_alpha = 1.7580993408473768599402175208123f;
    return x > 0.f ? scale : scale_alpha * exp_fast(fminf(x, 0.f));
  } else if constexpr (ACT == RELGNN_ACT_GELU) {
    const float cdf = fmaf(0.5f, erf_fast(x * 0.70710678118654752440f), 0.5f);
    return fmaf(x * 0.39894228040143267794f, __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x), cdf);  // log2(e)/2
  } else return act_grad<ACT>(x);
}

// Dispatch a runtime activation id to a compile-time template argument.
#define RELGNN_DISPATCH_ACT(act, ACT_CONST, ...)                                       \
  switch (act) {                                                                       \
    case RELGNN_ACT_LINEAR: { constexpr int ACT_CONST = RELGNN_ACT_LINEAR; __VA_ARGS__; break; }         \
    case RELGNN_ACT_TANH: { constexpr int ACT_CONST = RELGNN_ACT_TANH; __VA_ARGS__; break; }             \
    case RELGNN_ACT_RELU: { constexpr int ACT_CONST = RELGNN_ACT_RELU; __VA_ARGS__; break; }             \
    case RELGNN_ACT_LEAKY_RELU: { constexpr int ACT_CONST = RELGNN_ACT_LEAKY_RELU; __VA_ARGS__; break; } \
    case RELGNN_ACT_ELU: { constexpr int ACT_CONST = RELGNN_ACT_ELU; __VA_ARGS__; break; }               \
    case RELGNN_ACT_SELU: { constexpr int ACT_CONST = RELGNN_ACT_SELU; __VA_ARGS__; break; }             \
    case RELGNN_ACT_GELU: { constexpr int ACT_CONST = RELGNN_ACT_GELU; __VA_ARGS__; break; }             \
    default: return RELGNN_EINVAL;                                                     \
  }

// XCD-aware logical block id: hardware places block b on XCD b % 8 (observed, speed only).
// Give every XCD one contiguous range of logical blocks so that neighbouring segments
// (same graph of the disjoint-union batch -> same source rows) share one 4 MiB L2.
// Launch with grid = 8 * ceil(n_logical / 8); returns -1 for the padding blocks.
__device__ __forceinline__ int64_t xcd_logical_block(int64_t n_logical) {
  const int64_t per_xcd = (n_logical + 7) >> 3;
  const int64_t b = blockIdx.x;
  const int64_t lb = (b & 7) * per_xcd + (b >> 3);
  return lb < n_logical ? lb : -1;
}

}  // namespace relgnn
