// tf.contrib.layers.layer_norm on [rows, D] node states (gnns/gnn_film.py:120, rgin.py:139, gnn_edge_mlp.py:120,
// models/sparse_graph_model.py:192-193): moments over the last axis, biased variance, variance_epsilon 1e-12,
//     y = (x - mean) * rsqrt(var + eps) * gamma + beta.
// G = min(64, D/4 rounded up to a power of two) lanes own one row (float4 per lane, NCH chunks for D > 256); a row is
// read once, mean and variance are formed from the registers (two-pass: sum, then sum of squared deviations), so the
// forward is one read + one write of the tensor and the backward one read of g and x + one write of dx.
// d gamma / d beta: every lane group walks rows with a grid stride and keeps its columns' partial sums in registers;
// the [groups, 2D] partials are column-summed by the caller (relgnn_column_sum): no atomics.
// Bound: HBM, 8D bytes per row forward, 12D backward.
#include "common.h"

using namespace relgnn;

namespace {

template <int G>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int o = G >> 1; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

template <int G, int NCH>
__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const float4* __restrict__ X, int64_t ldx4, int32_t D4,
                                                             int64_t rows, const float4* __restrict__ gamma,
                                                             const float4* __restrict__ beta, float eps,
                                                             float4* __restrict__ Y, int64_t ldy4,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  const bool valid = r < rows;
  const int64_t rr = valid ? r : rows - 1;
  const float inv_d = 1.0f / (float)(4 * D4);
  float4 x[NCH];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = gl + G * c;
    x[c] = col < D4 ? X[rr * ldx4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
  }
  const float mean = group_sum<G>(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (gl + G * c < D4) {
      const float a = x[c].x - mean, b = x[c].y - mean, cc = x[c].z - mean, d = x[c].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(group_sum<G>(q) * inv_d + eps);
  if (!valid) return;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = gl + G * c;
    if (col < D4) {
      const float4 g = gamma[col], b = beta[col];
      float4 y;
      y.x = (x[c].x - mean) * rstd * g.x + b.x;
      y.y = (x[c].y - mean) * rstd * g.y + b.y;
      y.z = (x[c].z - mean) * rstd * g.z + b.z;
      y.w = (x[c].w - mean) * rstd * g.w + b.w;
      Y[r * ldy4 + col] = y;
    }
  }
  if (gl == 0) {
    mean_out[r] = mean;
    rstd_out[r] = rstd;
  }
}

// dx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat)),  gy = g * gamma,  xhat = (x - mean) * rstd
// partial[group, 0:D] += g * xhat (d gamma), partial[group, D:2D] += g (d beta)
template <int G, int NCH>
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float4* __restrict__ X, int64_t ldx4,
                                                             const float4* __restrict__ Gy, int64_t ldg4, int32_t D4,
                                                             int64_t rows, const float4* __restrict__ gamma,
                                                             const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, float4* __restrict__ dX,
                                                             int64_t ldd4, float4* __restrict__ partial,
                                                             int64_t num_groups) {
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  const float inv_d = 1.0f / (float)(4 * D4);
  float4 gam[NCH], dgam[NCH], dbet[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = gl + G * c;
    gam[c] = col < D4 ? gamma[col] : make_float4(0.f, 0.f, 0.f, 0.f);
    dgam[c] = dbet[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // all groups of a wave run the same number of iterations (shuffles need every lane); rows beyond the end are
  // clamped for the loads and masked for the stores / partial sums
  const int64_t iters = (rows + num_groups - 1) / num_groups;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t r = grp + it * num_groups;
    const bool valid = grp < num_groups && r < rows;
    const int64_t rr = valid ? r : 0;
    const float mean = mean_in[rr], rstd = rstd_in[rr];
    float4 xh[NCH], gy[NCH], g[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = gl + G * c;
      const bool on = col < D4;
      const float4 x = on ? X[rr * ldx4 + col] : make_float4(mean, mean, mean, mean);
      g[c] = on ? Gy[rr * ldg4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
      xh[c] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
      gy[c] = make_float4(g[c].x * gam[c].x, g[c].y * gam[c].y, g[c].z * gam[c].z, g[c].w * gam[c].w);
      s1 += (gy[c].x + gy[c].y) + (gy[c].z + gy[c].w);
      s2 += (gy[c].x * xh[c].x + gy[c].y * xh[c].y) + (gy[c].z * xh[c].z + gy[c].w * xh[c].w);
    }
    const float c1 = group_sum<G>(s1) * inv_d;
    const float c2 = group_sum<G>(s2) * inv_d;
    if (valid) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = gl + G * c;
        if (col < D4) {
          float4 d;
          d.x = rstd * (gy[c].x - c1 - xh[c].x * c2);
          d.y = rstd * (gy[c].y - c1 - xh[c].y * c2);
          d.z = rstd * (gy[c].z - c1 - xh[c].z * c2);
          d.w = rstd * (gy[c].w - c1 - xh[c].w * c2);
          dX[r * ldd4 + col] = d;
          dgam[c].x += g[c].x * xh[c].x; dgam[c].y += g[c].y * xh[c].y;
          dgam[c].z += g[c].z * xh[c].z; dgam[c].w += g[c].w * xh[c].w;
          dbet[c].x += g[c].x; dbet[c].y += g[c].y; dbet[c].z += g[c].z; dbet[c].w += g[c].w;
        }
      }
    }
  }
  if (grp < num_groups) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = gl + G * c;
      if (col < D4) {
        partial[grp * (2 * D4) + col] = dgam[c];
        partial[grp * (2 * D4) + D4 + col] = dbet[c];
      }
    }
  }
}

struct LnGeo { int G, NCH; };
inline bool ln_geo(int D, LnGeo* g) {
  if (D <= 0 || D % 4 != 0 || D > 1024) return false;
  const int D4 = D / 4;
  if (D4 <= 8) *g = {8, 1};
  else if (D4 <= 16) *g = {16, 1};
  else if (D4 <= 32) *g = {32, 1};
  else if (D4 <= 64) *g = {64, 1};
  else if (D4 <= 128) *g = {64, 2};
  else *g = {64, 4};
  return true;
}

#define LN_DISPATCH(geo, GG, NN, ...)                                      \
  if (geo.G == 8) { constexpr int GG = 8, NN = 1; __VA_ARGS__; }            \
  else if (geo.G == 16) { constexpr int GG = 16, NN = 1; __VA_ARGS__; }     \
  else if (geo.G == 32) { constexpr int GG = 32, NN = 1; __VA_ARGS__; }     \
  else if (geo.NCH == 1) { constexpr int GG = 64, NN = 1; __VA_ARGS__; }    \
  else if (geo.NCH == 2) { constexpr int GG = 64, NN = 2; __VA_ARGS__; }    \
  else { constexpr int GG = 64, NN = 4; __VA_ARGS__; }

}  // namespace

extern "C" {

int64_t relgnn_layer_norm_groups(int64_t rows, int32_t D) {
  LnGeo geo;
  if (rows <= 0 || !ln_geo(D, &geo)) return 0;
  const int64_t per_block = 256 / geo.G;
  const int64_t want = 2048 * per_block / 4;            // ~2048 blocks worth of lane groups at G = 64
  const int64_t groups = rows < want ? rows : want;
  return (groups + per_block - 1) / per_block * per_block;   // whole blocks
}

int relgnn_layer_norm_fwd(const float* X, int64_t ldx, int64_t rows, int32_t D, const float* gamma, const float* beta,
                          float eps, float* Y, int64_t ldy, float* mean, float* rstd, void* stream) {
  LnGeo geo;
  if (rows < 0 || ldx < D || ldy < D) return RELGNN_EINVAL;
  if (!ln_geo(D, &geo) || ldx % 4 != 0 || ldy % 4 != 0 || !aligned16(X) || !aligned16(Y) || !aligned16(gamma) ||
      !aligned16(beta))
    return RELGNN_EUNSUPPORTED;
  if (rows == 0) return RELGNN_OK;
  if (!X || !Y || !gamma || !beta || !mean || !rstd) return RELGNN_EINVAL;
  const unsigned grid = (unsigned)((rows * geo.G + 255) / 256);
  LN_DISPATCH(geo, GG, NN, (layer_norm_fwd_kernel<GG, NN><<<grid, 256, 0, as_stream(stream)>>>(
                                (const float4*)X, ldx / 4, D / 4, rows, (const float4*)gamma, (const float4*)beta, eps,
                                (float4*)Y, ldy / 4, mean, rstd)));
  return launch_status();
}

int relgnn_layer_norm_bwd(const float* X, int64_t ldx, const float* gY, int64_t ldg, int64_t rows, int32_t D,
                          const float* gamma, const float* mean, const float* rstd, float* dX, int64_t ldd,
                          float* partial, int64_t num_groups, void* stream) {
  LnGeo geo;
  if (rows < 0 || ldx < D || ldg < D || ldd < D) return RELGNN_EINVAL;
  if (!ln_geo(D, &geo) || ldx % 4 != 0 || ldg % 4 != 0 || ldd % 4 != 0 || !aligned16(X) || !aligned16(gY) ||
      !aligned16(dX) || !aligned16(gamma) || !aligned16(partial))
    return RELGNN_EUNSUPPORTED;
  if (rows == 0) return RELGNN_OK;
  if (!X || !gY || !gamma || !mean || !rstd || !dX || !partial) return RELGNN_EINVAL;
  if (num_groups != relgnn_layer_norm_groups(rows, D)) return RELGNN_EINVAL;
  const unsigned grid = (unsigned)((num_groups * geo.G + 255) / 256);
  LN_DISPATCH(geo, GG, NN, (layer_norm_bwd_kernel<GG, NN><<<grid, 256, 0, as_stream(stream)>>>(
                                (const float4*)X, ldx / 4, (const float4*)gY, ldg / 4, D / 4, rows,
                                (const float4*)gamma, mean, rstd, (float4*)dX, ldd / 4, (float4*)partial, num_groups)));
  return launch_status();
}

}  // extern "C"
