// Fused elementwise halves of the Keras GRUCell used by sparse_ggnn_layer (gnns/ggnn.py:92 through
// utils/utils.py:15-16; TF 1.13 semantics: reset_after=False, recurrent_activation = hard_sigmoid, gate order z, r, h):
//   z = hs(xk_z + rec_z), r = hs(xk_r + rec_r), rh = r * h            (gates)
//   hh = act(xk_h + q),  q = rh @ U_h (GEMM, caller);  out = z * h + (1 - z) * hh        (output)
// The un-fused formulation costs ~14 elementwise launches forward and ~25 backward per cell; on the QM9-sized
// batches of config C3 (50k nodes, D=128) the cell is launch-bound.  Node-side work, not the gather/segment path.
#include "common.h"

using namespace relgnn;

namespace {

__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
// derivative of hard_sigmoid evaluated from its OUTPUT y (0 < y < 1 on the linear piece)
__device__ __forceinline__ float hard_sigmoid_grad_from_out(float y) { return (y > 0.f && y < 1.f) ? 0.2f : 0.f; }

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case RELGNN_ACT_TANH: return act_fwd<RELGNN_ACT_TANH>(x);
    case RELGNN_ACT_RELU: return act_fwd<RELGNN_ACT_RELU>(x);
    case RELGNN_ACT_LEAKY_RELU: return act_fwd<RELGNN_ACT_LEAKY_RELU>(x);
    case RELGNN_ACT_ELU: return act_fwd<RELGNN_ACT_ELU>(x);
    case RELGNN_ACT_SELU: return act_fwd<RELGNN_ACT_SELU>(x);
    default: return x;
  }
}
__device__ __forceinline__ float act_grad_from_out(int act, float y) {
  switch (act) {
    case RELGNN_ACT_TANH: return 1.f - y * y;
    case RELGNN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RELGNN_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.2f;
    case RELGNN_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    case RELGNN_ACT_SELU: return y > 0.f ? 1.0507009873554804934193349852946f : y + 1.7580993408473768599402175208123f;
    default: return 1.f;
  }
}

__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float* __restrict__ xk, const float* __restrict__ rec,
                                                            const float* __restrict__ h, int64_t V, int32_t u,
                                                            float* __restrict__ z, float* __restrict__ r,
                                                            float* __restrict__ rh) {
  const int64_t n = V * u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / u;
    const int c = (int)(i - v * u);
    const float zz = hard_sigmoid(xk[v * 3 * u + c] + rec[v * 2 * u + c]);
    const float rr = hard_sigmoid(xk[v * 3 * u + u + c] + rec[v * 2 * u + u + c]);
    z[i] = zz;
    r[i] = rr;
    rh[i] = rr * h[i];
  }
}

__global__ __launch_bounds__(256) void gru_out_fwd_kernel(const float* __restrict__ xk, const float* __restrict__ q,
                                                          const float* __restrict__ z, const float* __restrict__ h,
                                                          int64_t V, int32_t u, int32_t act, float* __restrict__ hh,
                                                          float* __restrict__ out) {
  const int64_t n = V * u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / u;
    const int c = (int)(i - v * u);
    const float cand = act_apply(act, xk[v * 3 * u + 2 * u + c] + q[i]);
    const float zz = z[i];
    hh[i] = cand;
    out[i] = zz * h[i] + (1.0f - zz) * cand;
  }
}

// gxk[:, 2u:3u] = g_pre_h = gout * (1 - z) * act'(hh);  gz = gout * (h - hh);  gh = gout * z
__global__ __launch_bounds__(256) void gru_out_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ z,
                                                          const float* __restrict__ h, const float* __restrict__ hh,
                                                          int64_t V, int32_t u, int32_t act, float* __restrict__ gxk,
                                                          float* __restrict__ gq, float* __restrict__ gz,
                                                          float* __restrict__ gh) {
  const int64_t n = V * u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / u;
    const int c = (int)(i - v * u);
    const float g = gout[i], zz = z[i], cand = hh[i];
    const float gpre = g * (1.0f - zz) * act_grad_from_out(act, cand);
    gxk[v * 3 * u + 2 * u + c] = gpre;
    gq[i] = gpre;
    gz[i] = g * (h[i] - cand);
    gh[i] = g * zz;
  }
}

// gxk[:, 0:u] = gz * hs'(z);  gxk[:, u:2u] = (grh * h) * hs'(r);  gh += grh * r
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ grh, const float* __restrict__ gz,
                                                            const float* __restrict__ z, const float* __restrict__ r,
                                                            const float* __restrict__ h, int64_t V, int32_t u,
                                                            float* __restrict__ gxk, float* __restrict__ gh) {
  const int64_t n = V * u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / u;
    const int c = (int)(i - v * u);
    const float rr = r[i], g = grh[i];
    gxk[v * 3 * u + c] = gz[i] * hard_sigmoid_grad_from_out(z[i]);
    gxk[v * 3 * u + u + c] = (g * h[i]) * hard_sigmoid_grad_from_out(rr);
    gh[i] = gh[i] + g * rr;
  }
}

inline bool act_ok(int act) { return act >= RELGNN_ACT_LINEAR && act <= RELGNN_ACT_SELU; }

}  // namespace

extern "C" {

int relgnn_gru_gates_fwd(const float* xk, const float* rec, const float* h, int64_t num_nodes, int32_t units, float* z,
                         float* r, float* rh, void* stream) {
  if (num_nodes < 0 || units <= 0) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!xk || !rec || !h || !z || !r || !rh) return RELGNN_EINVAL;
  gru_gates_fwd_kernel<<<flat_grid(num_nodes * units, 256), 256, 0, as_stream(stream)>>>(xk, rec, h, num_nodes, units, z, r, rh);
  return launch_status();
}

int relgnn_gru_out_fwd(const float* xk, const float* q, const float* z, const float* h, int64_t num_nodes, int32_t units,
                       int32_t act, float* hh, float* out, void* stream) {
  if (num_nodes < 0 || units <= 0) return RELGNN_EINVAL;
  if (!act_ok(act)) return act == RELGNN_ACT_GELU ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!xk || !q || !z || !h || !hh || !out) return RELGNN_EINVAL;
  gru_out_fwd_kernel<<<flat_grid(num_nodes * units, 256), 256, 0, as_stream(stream)>>>(xk, q, z, h, num_nodes, units, act, hh, out);
  return launch_status();
}

int relgnn_gru_out_bwd(const float* gout, const float* z, const float* h, const float* hh, int64_t num_nodes,
                       int32_t units, int32_t act, float* gxk, float* gq, float* gz, float* gh, void* stream) {
  if (num_nodes < 0 || units <= 0) return RELGNN_EINVAL;
  if (!act_ok(act)) return act == RELGNN_ACT_GELU ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!gout || !z || !h || !hh || !gxk || !gq || !gz || !gh) return RELGNN_EINVAL;
  gru_out_bwd_kernel<<<flat_grid(num_nodes * units, 256), 256, 0, as_stream(stream)>>>(gout, z, h, hh, num_nodes, units, act,
                                                                                      gxk, gq, gz, gh);
  return launch_status();
}

int relgnn_gru_gates_bwd(const float* grh, const float* gz, const float* z, const float* r, const float* h,
                         int64_t num_nodes, int32_t units, float* gxk, float* gh, void* stream) {
  if (num_nodes < 0 || units <= 0) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!grh || !gz || !z || !r || !h || !gxk || !gh) return RELGNN_EINVAL;
  gru_gates_bwd_kernel<<<flat_grid(num_nodes * units, 256), 256, 0, as_stream(stream)>>>(grh, gz, z, r, h, num_nodes, units, gxk, gh);
  return launch_status();
}

}  // extern "C"
