// RGAT attention-logit tables (gnns/rgat.py:103-115).
//
// The reference concatenates [T_l[src] || T_l[tgt]] per message ([E, K, 2*Dh]) and contracts it with the attention
// parameters a_l (reshaped (K, 2*Dh), rgat.py:110-111).  The logit is linear in the two endpoint rows, so it splits
// into two per-(node, type) tables
//     s_src[v*L+l, k] = < T[v*L+l, head k], a_l[k, 0:Dh]   >
//     s_tgt[v*L+l, k] = < T[v*L+l, head k], a_l[k, Dh:2Dh] >
// that the softmax kernels (rgat_fast.hip) read with 4*K bytes per message.  This file computes both tables in one
// pass over T (fwd) and folds their gradients back in one pass (bwd):
//     gT[r, head k] += gs_src[r,k] * a_l[k, 0:Dh] + gs_tgt[r,k] * a_l[k, Dh:2Dh]          (in place)
//     ga_l[k, 0:Dh]  = sum_v gs_src[v*L+l, k] * T[v*L+l, head k]   (and the tgt half alike)
// G = D/4 lanes own one row (float4 per lane), Dh/4 consecutive lanes own one head; reductions inside a head are
// xor-shuffles.  Bound: HBM, 4*D bytes per row per pass.
#include "common.h"

using namespace relgnn;

namespace {

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int G>
__global__ __launch_bounds__(256) void rgat_scores_fwd_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ att, int32_t Dh4, int32_t K, int32_t L,
    int64_t rows, float* __restrict__ s_src, float* __restrict__ s_tgt) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  const bool valid = r < rows;
  const int64_t rr = valid ? r : rows - 1;
  const int l = (int)(rr % L);
  const int k = gl / Dh4, j = gl % Dh4;
  const float4 t = T[rr * ldt4 + gl];
  const float4* a = att + (int64_t)l * (2 * G) + (int64_t)k * (2 * Dh4);
  float ps = dot4(t, a[j]);
  float pt = dot4(t, a[Dh4 + j]);
  for (int o = Dh4 >> 1; o > 0; o >>= 1) {
    ps += __shfl_xor(ps, o, 64);
    pt += __shfl_xor(pt, o, 64);
  }
  if (valid && j == 0) {
    s_src[rr * K + k] = ps;
    s_tgt[rr * K + k] = pt;
  }
}

template <int G>
__global__ __launch_bounds__(256) void rgat_scores_bwd_rows_kernel(
    const float4* __restrict__ att, int32_t Dh4, int32_t K, int32_t L, int64_t rows, const float* __restrict__ gs_src,
    const float* __restrict__ gs_tgt, float4* __restrict__ gT, int64_t ldg4) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (r >= rows) return;
  const int gl = threadIdx.x % G;
  const int l = (int)(r % L);
  const int k = gl / Dh4, j = gl % Dh4;
  const float4* a = att + (int64_t)l * (2 * G) + (int64_t)k * (2 * Dh4);
  const float gs = gs_src[r * K + k], gt = gs_tgt[r * K + k];
  const float4 as = a[j], at = a[Dh4 + j];
  float4 g = gT[r * ldg4 + gl];
  g.x += gs * as.x + gt * at.x;
  g.y += gs * as.y + gt * at.y;
  g.z += gs * as.z + gt * at.z;
  g.w += gs * as.w + gt * at.w;
  gT[r * ldg4 + gl] = g;
}

// partial[grp, l, :] (a_l layout (K, 2*Dh)) = sum over the nodes of lane group `grp` of gs[v*L+l, k] * T[v*L+l, head k]
template <int G>
__global__ __launch_bounds__(256) void rgat_scores_bwd_att_kernel(
    const float4* __restrict__ T, int64_t ldt4, int32_t Dh4, int32_t K, int32_t L, int64_t num_nodes,
    int64_t nodes_per_group, int64_t num_groups, const float* __restrict__ gs_src, const float* __restrict__ gs_tgt,
    float4* __restrict__ partial) {
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (grp >= num_groups) return;
  const int gl = threadIdx.x % G;
  const int k = gl / Dh4, j = gl % Dh4;
  const int64_t v0 = grp * nodes_per_group;
  const int64_t v1 = min(v0 + nodes_per_group, num_nodes);
  for (int l = 0; l < L; ++l) {
    float4 as = make_float4(0.f, 0.f, 0.f, 0.f), at = as;
    for (int64_t v = v0; v < v1; ++v) {
      const int64_t r = v * L + l;
      const float4 t = T[r * ldt4 + gl];
      const float gs = gs_src[r * K + k], gt = gs_tgt[r * K + k];
      as.x += gs * t.x; as.y += gs * t.y; as.z += gs * t.z; as.w += gs * t.w;
      at.x += gt * t.x; at.y += gt * t.y; at.z += gt * t.z; at.w += gt * t.w;
    }
    float4* p = partial + (grp * L + l) * (2 * G) + (int64_t)k * (2 * Dh4);
    p[j] = as;
    p[Dh4 + j] = at;
  }
}

inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// lanes per row: D/4 in {8,16,32,64}; heads split the row evenly into power-of-two lane groups
inline bool geometry_ok(int32_t D, int32_t K) {
  if (D <= 0 || K <= 0 || D % 4 != 0) return false;
  const int G = D / 4;
  if (!(G == 8 || G == 16 || G == 32 || G == 64)) return false;
  if (G % K != 0) return false;
  return pow2(G / K);
}

#define SCORES_DISPATCH_G(G_, ...)                        \
  if (G_ == 8) { constexpr int GG = 8; __VA_ARGS__; }      \
  else if (G_ == 16) { constexpr int GG = 16; __VA_ARGS__; } \
  else if (G_ == 32) { constexpr int GG = 32; __VA_ARGS__; } \
  else { constexpr int GG = 64; __VA_ARGS__; }

}  // namespace

extern "C" {

int64_t relgnn_rgat_scores_groups(int64_t num_nodes) {
  if (num_nodes <= 0) return 0;
  const int64_t target = 4096;                       // lane groups (>= 1024 waves) that share the node range
  const int64_t per = (num_nodes + target - 1) / target;
  return (num_nodes + per - 1) / per;
}

int relgnn_rgat_scores_fwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* att,
                           int32_t num_edge_types, int64_t num_nodes, float* s_src, float* s_tgt, void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || ldt < D) return RELGNN_EINVAL;
  if (!geometry_ok(D, num_heads) || ldt % 4 != 0 || !aligned16(T) || !aligned16(att)) return RELGNN_EUNSUPPORTED;
  const int64_t rows = num_nodes * num_edge_types;
  if (rows == 0) return RELGNN_OK;
  if (!T || !att || !s_src || !s_tgt) return RELGNN_EINVAL;
  const int G = D / 4;
  const int64_t threads = rows * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  SCORES_DISPATCH_G(G, rgat_scores_fwd_kernel<GG><<<grid, 256, 0, as_stream(stream)>>>(
      (const float4*)T, ldt / 4, (const float4*)att, G / num_heads, num_heads, num_edge_types, rows, s_src, s_tgt));
  return launch_status();
}

int relgnn_rgat_scores_bwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* att,
                           int32_t num_edge_types, int64_t num_nodes, const float* gs_src, const float* gs_tgt,
                           float* gT, int64_t ldg, float* att_partial, int64_t num_groups, void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || ldt < D || ldg < D) return RELGNN_EINVAL;
  if (!geometry_ok(D, num_heads) || ldt % 4 != 0 || ldg % 4 != 0 || !aligned16(T) || !aligned16(att) || !aligned16(gT) ||
      !aligned16(att_partial))
    return RELGNN_EUNSUPPORTED;
  const int64_t rows = num_nodes * num_edge_types;
  if (rows == 0) return RELGNN_OK;
  if (!T || !att || !gs_src || !gs_tgt) return RELGNN_EINVAL;
  const int G = D / 4;
  hipStream_t st = as_stream(stream);
  if (gT) {
    const unsigned grid = (unsigned)((rows * G + 255) / 256);
    SCORES_DISPATCH_G(G, rgat_scores_bwd_rows_kernel<GG><<<grid, 256, 0, st>>>(
        (const float4*)att, G / num_heads, num_heads, num_edge_types, rows, gs_src, gs_tgt, (float4*)gT, ldg / 4));
  }
  if (att_partial) {
    if (num_groups != relgnn_rgat_scores_groups(num_nodes)) return RELGNN_EINVAL;
    const int64_t per = (num_nodes + num_groups - 1) / num_groups;
    const unsigned grid = (unsigned)((num_groups * G + 255) / 256);
    SCORES_DISPATCH_G(G, rgat_scores_bwd_att_kernel<GG><<<grid, 256, 0, st>>>(
        (const float4*)T, ldt / 4, G / num_heads, num_heads, num_edge_types, num_nodes, per, num_groups, gs_src, gs_tgt,
        (float4*)att_partial));
  }
  return launch_status();
}

}  // extern "C"
