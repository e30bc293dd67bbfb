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
