// Dynamic per-target convolution of sparse_rgdcn_layer (gnns/rgdcn.py:126-160), node side.
//
// The K x K kernel of a message depends on the TARGET node (and edge type, channel) only and sum / mean / sqrt_n are
// linear, so the per-edge einsum('vi,vij->vj') of the reference (rgdcn.py:146, on an [E, K, K] gather per channel and
// type) is applied ONCE per (target, type, channel) to the already aggregated source states:
//
//   out[v, c, j] = act( f_mode( sum_l sum_i  A[v, l, c, i] * act( P[v, l, c, i, j] ) ) )
//
//   A = seg_reduce of the raw source states into the (target, type) buckets   [V, L, C*K]   (relgnn_seg_reduce_fwd)
//   P = pre-activation dynamic weights, Dense_{l,c}(h_v or h_{v,c})            (node-side GEMM), addressed as
//       P[v*sv + l*sl + c*sc + i*K + j] so that both GEMM layouts (full state: [V, L, C, K*K]; per channel:
//       [C, V, L, K*K]) are read in place.
//
// One wave per node, lanes across the D = C*K outputs (c, j): P is streamed once (the only large operand: L*C*K*K
// floats per node), A rows come from L1 as 16-lane broadcasts.  Bound: HBM, 4*L*C*K*K bytes per node forward,
// 3x that backward (read P, write dP).
#include "common.h"

using namespace relgnn;

namespace {

template <int ACT>
__global__ __launch_bounds__(256) void rgdcn_apply_fwd_kernel(
    const float* __restrict__ A, const float* __restrict__ P, int64_t sv, int64_t sl, int64_t sc, int32_t V, int32_t L,
    int32_t C, int32_t K, int32_t mode, const int32_t* __restrict__ rowptr_t, int32_t out_act, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int D = C * K;
  float factor = 1.f;
  if (mode != RELGNN_AGG_SUM) {
    const float n = (float)max(rowptr_t[(v + 1) * L] - rowptr_t[v * L], 1);
    factor = mode == RELGNN_AGG_MEAN ? n : sqrtf(n);
  }
  for (int o = lane; o < D; o += 64) {
    const int c = o / K, j = o - c * K;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const float* a = A + ((int64_t)v * L + l) * D + c * K;
      const float* p = P + v * sv + l * sl + c * sc + j;
      for (int i = 0; i < K; ++i) acc += a[i] * act_fwd<ACT>(p[i * K]);
    }
    if (mode != RELGNN_AGG_SUM) acc /= factor;
    float y = acc;
    switch (out_act) {
      case RELGNN_ACT_TANH: y = act_fwd<RELGNN_ACT_TANH>(acc); break;
      case RELGNN_ACT_RELU: y = act_fwd<RELGNN_ACT_RELU>(acc); break;
      case RELGNN_ACT_LEAKY_RELU: y = act_fwd<RELGNN_ACT_LEAKY_RELU>(acc); break;
      case RELGNN_ACT_ELU: y = act_fwd<RELGNN_ACT_ELU>(acc); break;
      case RELGNN_ACT_SELU: y = act_fwd<RELGNN_ACT_SELU>(acc); break;
      case RELGNN_ACT_GELU: y = act_fwd<RELGNN_ACT_GELU>(acc); break;
      default: break;
    }
    out[v * D + o] = y;
  }
}

// G = d loss / d (f_mode(sum)) already (the caller folds the output activation's derivative and the mean / sqrt_n
// factor in):  gA[v,l,c,i] = sum_j G[v,c,j] act(P[..i,j]);   gP[v,l,c,i,j] = G[v,c,j] * A[v,l,c,i] * act'(P[..i,j]).
// The sum over j runs across the K lanes of a channel group (K a power of two <= 64: xor-shuffles).
template <int ACT>
__global__ __launch_bounds__(256) void rgdcn_apply_bwd_kernel(
    const float* __restrict__ A, const float* __restrict__ P, int64_t sv, int64_t sl, int64_t sc, int32_t V, int32_t L,
    int32_t C, int32_t K, const float* __restrict__ G, float* __restrict__ gA, float* __restrict__ gP) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int D = C * K;
  const int rounds = (D + 63) / 64;                 // wave-uniform trip count: shuffles need every lane
  for (int r = 0; r < rounds; ++r) {
    const int o = r * 64 + lane;
    const bool on = o < D;
    const int oc = on ? o : D - 1;
    const int c = oc / K, j = oc - c * K;
    const float g = on ? G[v * D + oc] : 0.f;
    for (int l = 0; l < L; ++l) {
      const float* a = A + ((int64_t)v * L + l) * D + c * K;
      const int64_t base = v * sv + l * sl + c * sc + j;
      for (int i = 0; i < K; ++i) {
        const float p = P[base + i * K];
        float part = g * act_fwd<ACT>(p);
        for (int off = 1; off < K; off <<= 1) part += __shfl_xor(part, off);
        if (on) {
          gP[base + i * K] = g * a[i] * act_grad<ACT>(p);
          if (j == 0) gA[((int64_t)v * L + l) * D + c * K + i] = part;
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int relgnn_rgdcn_apply_fwd(int32_t mode, int32_t weight_act, int32_t out_act, const float* A, const float* P,
                           int64_t p_node_stride, int64_t p_type_stride, int64_t p_channel_stride, int32_t num_nodes,
                           int32_t num_edge_types, int32_t num_channels, int32_t channel_dim, const int32_t* rowptr_t,
                           float* out, void* stream) {
  if (mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_SQRT_N) return mode == RELGNN_AGG_MAX ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (out_act < RELGNN_ACT_LINEAR || out_act > RELGNN_ACT_GELU || num_nodes < 0 || num_edge_types <= 0 || num_channels <= 0 ||
      channel_dim <= 0)
    return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!A || !P || !out || (mode != RELGNN_AGG_SUM && !rowptr_t)) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)((num_nodes + 3) / 4);
  RELGNN_DISPATCH_ACT(weight_act, WA,
                      (rgdcn_apply_fwd_kernel<WA><<<grid, 256, 0, st>>>(A, P, p_node_stride, p_type_stride, p_channel_stride,
                                                                       num_nodes, num_edge_types, num_channels, channel_dim,
                                                                       mode, rowptr_t, out_act, out)));
  return launch_status();
}

int relgnn_rgdcn_apply_bwd(int32_t weight_act, const float* A, const float* P, int64_t p_node_stride, int64_t p_type_stride,
                           int64_t p_channel_stride, int32_t num_nodes, int32_t num_edge_types, int32_t num_channels,
                           int32_t channel_dim, const float* G, float* gA, float* gP, void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || num_channels <= 0 || channel_dim <= 0) return RELGNN_EINVAL;
  if (channel_dim > 64 || (channel_dim & (channel_dim - 1)) != 0) return RELGNN_EUNSUPPORTED;
  if (num_nodes == 0) return RELGNN_OK;
  if (!A || !P || !G || !gA || !gP) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)((num_nodes + 3) / 4);
  RELGNN_DISPATCH_ACT(weight_act, WA,
                      (rgdcn_apply_bwd_kernel<WA><<<grid, 256, 0, st>>>(A, P, p_node_stride, p_type_stride, p_channel_stride,
                                                                       num_nodes, num_edge_types, num_channels, channel_dim,
                                                                       G, gA, gP)));
  return launch_status();
}

}  // extern "C"
