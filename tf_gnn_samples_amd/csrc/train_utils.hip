// Training-step plumbing AROUND the hot path, fused so that the step is not
// launch-bound once the gather/segment kernels and GEMMs are fast:
//   * multi-tensor per-variable clip_by_norm + TF-style Adam   models/sparse_graph_model.py:227-260
//   * PPI output head loss + micro-F1 counts in one pass        tasks/ppi_task.py:181-191, utils/utils.py:61-74
// Deterministic (no float atomics): fixed-shape tree reductions only.
#include "common.h"

using namespace relgnn;

namespace {

struct MtArgs {
  float* p[RELGNN_MT_MAX];
  const float* g[RELGNN_MT_MAX];
  float* m[RELGNN_MT_MAX];
  float* v[RELGNN_MT_MAX];
  long long n[RELGNN_MT_MAX];
};

struct NormArgs {
  const float* g[RELGNN_MT_MAX];
  long long n[RELGNN_MT_MAX];
};

__device__ __forceinline__ double block_sum_1024(double x, double* red) {
  // wave reduce (64 lanes) then across the 16 waves of a 1024-thread block, fixed order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = x;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  __syncthreads();
  return t;  // valid in thread 0
}

// norms[t] = sqrt(sum g^2): grid (kNormChunks, tensors) of partial sums in double, then one small block per tensor
// adds the kNormChunks partials in a fixed order (deterministic; a single block per tensor took 35 us for the
// 196 k-element kernels of C2, this pair takes ~10 us)
constexpr int kNormChunks = 32;

__global__ __launch_bounds__(256) void mt_l2norm_partial_kernel(NormArgs a, double* __restrict__ partial) {
  __shared__ double red[4];
  const int t = blockIdx.y;
  const float* g = a.g[t];
  const long long n = a.n[t];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double x = g[i];
    acc += x * x;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)t * kNormChunks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void mt_l2norm_final_kernel(const double* __restrict__ partial, float* __restrict__ norms) {
  const int t = blockIdx.x;
  double x = threadIdx.x < kNormChunks ? partial[(long long)t * kNormChunks + threadIdx.x] : 0.0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  if (threadIdx.x == 0) norms[t] = (float)sqrt(x);
}

// grid (chunks, tensors): g' = g * clip / max(||g||, clip)  (tf.clip_by_norm), then TF1 Adam:
//   m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr_t * m / (sqrt(v) + eps),  lr_t = lr sqrt(1-b2^t)/(1-b1^t)
// state[0] = number of Adam steps taken so far (as a float: exact up to 2^24), state[1] = lr_t of the step being taken.
// One thread: t += 1; lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t).  Keeps the step count ON THE DEVICE so that a captured
// hipGraph of the training step advances it on every replay (a host scalar would be frozen into the graph).
__global__ void adam_step_size_kernel(float* __restrict__ state, float lr, float b1, float b2) {
  const float t = state[0] + 1.f;
  state[0] = t;
  state[1] = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
}

__global__ __launch_bounds__(256) void mt_adam_clip_kernel(MtArgs a, const float* __restrict__ norms, float clip,
                                                           float lr_t, const float* __restrict__ d_lr_t, float b1, float b2,
                                                           float eps) {
  if (d_lr_t) lr_t = d_lr_t[0];
  const int t = blockIdx.y;
  const long long n = a.n[t];
  const long long chunk = 4096;
  const long long beg = (long long)blockIdx.x * chunk;
  if (beg >= n) return;
  const long long end = min(n, beg + chunk);
  float scale = 1.f;
  if (clip > 0.f) {
    const float nrm = norms[t];
    scale = clip / fmaxf(nrm, clip);
  }
  float* p = a.p[t];
  const float* g = a.g[t];
  float* m = a.m[t];
  float* v = a.v[t];
  for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// ---- PPI head: sigmoid cross-entropy sum + micro-F1 counts -------------------------------------
// stats = {sum of losses, true_pos, false_pos, false_neg, micro-F1, mean_scale * sum of losses}
struct CeAcc {
  double loss;
  int tp, fp, fn;
};

__device__ __forceinline__ void ce_element(float x, float z, CeAcc& a) {
  // tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|))
  const float e = expf(-fabsf(x));
  a.loss += (double)(fmaxf(x, 0.f) - x * z + log1pf(e));
  // round(sigmoid(x)) with round-half-even == (1 / (1 + exp(-x)) > 0.5).  For x <= 0, exp(-x) >= 1 makes the quotient
  // <= 0.5 in every rounding; for x > 0, exp(-x) is the e above — one exponential serves the loss and the prediction.
  const bool pred = x > 0.f && (1.f / (1.f + e)) > 0.5f;
  const int zi = (int)z;
  a.tp += (pred && zi != 0) ? 1 : 0;
  a.fp += (pred && zi != 1) ? 1 : 0;
  a.fn += (!pred && zi != 0) ? 1 : 0;
}

__global__ __launch_bounds__(1024) void sigmoid_ce_stats_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ labels, long long n,
                                                                double* __restrict__ partial) {
  __shared__ double red[16];
  CeAcc a = {0.0, 0, 0, 0};
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  // 16-byte loads over the aligned body (both arrays are whole allocations), scalar tail
  const bool vec = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(labels)) & 15) == 0;
  const long long n4 = vec ? n / 4 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(logits);
  const float4* z4 = reinterpret_cast<const float4*>(labels);
  for (long long i = tid; i < n4; i += nth) {
    const float4 x = x4[i], z = z4[i];
    ce_element(x.x, z.x, a); ce_element(x.y, z.y, a); ce_element(x.z, z.z, a); ce_element(x.w, z.w, a);
  }
  for (long long i = 4 * n4 + tid; i < n; i += nth) ce_element(logits[i], labels[i], a);
  double s;
  s = block_sum_1024(a.loss, red);       if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = s;
  s = block_sum_1024((double)a.tp, red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = s;
  s = block_sum_1024((double)a.fp, red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 2] = s;
  s = block_sum_1024((double)a.fn, red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 3] = s;
}

// 256 threads: thread b holds the four partial sums of block b; fixed-shape tree reduction
__global__ __launch_bounds__(256) void sigmoid_ce_stats_final_kernel(const double* __restrict__ partial, int nblk,
                                                                     float mean_scale, float* __restrict__ stats) {
  __shared__ double red[4][4];
  __shared__ double tot[4];
  const int b = threadIdx.x;
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = b < nblk ? partial[b * 4 + k] : 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
  }
  const int wave = b >> 6, lane = b & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
  }
  __syncthreads();
  if (b < 4) {
    const double s = (red[0][b] + red[1][b]) + (red[2][b] + red[3][b]);
    stats[b] = (float)s;
    tot[b] = s;
  }
  __syncthreads();
  if (b == 0) {
    // utils/utils.py:70-74: int64 counts, float64 true division, cast to float32
    const double precision = tot[1] / (tot[1] + tot[2]);
    const double recall = tot[1] / (tot[1] + tot[3]);
    stats[4] = (float)((2.0 * precision * recall) / (precision + recall));
    stats[5] = (float)tot[0] * mean_scale;      // float32 product, as tf.reduce_sum(...) / num_nodes would round it
  }
}

// glogits = gs * (sigmoid(x) - z),  gs = g_mean[0] * mean_scale + g_total[0]  (either pointer may be null)
__global__ __launch_bounds__(256) void sigmoid_ce_bwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ labels, long long n,
                                                             const float* __restrict__ g_mean, float mean_scale,
                                                             const float* __restrict__ g_total, float* __restrict__ gl) {
  float gs = 0.f;
  if (g_mean) gs = g_mean[0] * mean_scale;
  if (g_total) gs = g_mean ? gs + g_total[0] : g_total[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = logits[i];
    gl[i] = gs * (1.f / (1.f + expf(-x)) - labels[i]);
  }
}

// the same gradient into rows of ldg >= cols floats, the columns behind `cols` written as zeros: the operand of an input-gradient
// product whose reduction length (cols = 121 labels) is then a multiple of 16 (relgnn_limb_gemm_xf32 over ldg columns)
__global__ __launch_bounds__(256) void sigmoid_ce_bwd_padded_kernel(const float* __restrict__ logits,
                                                                    const float* __restrict__ labels, long long rows, int cols,
                                                                    int ldg, const float* __restrict__ g_mean, float mean_scale,
                                                                    const float* __restrict__ g_total, float* __restrict__ gl) {
  float gs = 0.f;
  if (g_mean) gs = g_mean[0] * mean_scale;
  if (g_total) gs = g_mean ? gs + g_total[0] : g_total[0];
  const long long n = rows * ldg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / ldg;
    const int c = (int)(i - r * ldg);
    float v = 0.f;
    if (c < cols) {
      const float x = logits[r * cols + c];
      v = gs * (1.f / (1.f + expf(-x)) - labels[r * cols + c]);
    }
    gl[i] = v;
  }
}

constexpr int kStatsBlocks = 256;

}  // namespace

extern "C" {

size_t relgnn_mt_l2norm_workspace_bytes(void) { return (size_t)RELGNN_MT_MAX * kNormChunks * sizeof(double); }

int relgnn_mt_l2norm(const float* const* h_grads, const int64_t* h_sizes, int32_t n, float* norms, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (n < 0 || n > RELGNN_MT_MAX) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!h_grads || !h_sizes || !norms) return RELGNN_EINVAL;
  if (!workspace || workspace_bytes < relgnn_mt_l2norm_workspace_bytes()) return RELGNN_ENOSPC;
  NormArgs a;
  for (int i = 0; i < n; ++i) {
    if (h_sizes[i] < 0 || (h_sizes[i] > 0 && !h_grads[i])) return RELGNN_EINVAL;
    a.g[i] = h_grads[i];
    a.n[i] = h_sizes[i];
  }
  double* partial = static_cast<double*>(workspace);
  mt_l2norm_partial_kernel<<<dim3(kNormChunks, (unsigned)n), 256, 0, as_stream(stream)>>>(a, partial);
  mt_l2norm_final_kernel<<<n, 64, 0, as_stream(stream)>>>(partial, norms);
  return launch_status();
}

int relgnn_adam_step_size(float* d_state, float lr, float beta1, float beta2, void* stream) {
  if (!d_state) return RELGNN_EINVAL;
  adam_step_size_kernel<<<1, 1, 0, as_stream(stream)>>>(d_state, lr, beta1, beta2);
  return launch_status();
}

static int mt_adam_clip_impl(float* const* h_params, const float* const* h_grads, float* const* h_m, float* const* h_v,
                             const int64_t* h_sizes, int32_t n, const float* norms, float clip, float lr_t,
                             const float* d_lr_t, float beta1, float beta2, float eps, void* stream);

int relgnn_mt_adam_clip(float* const* h_params, const float* const* h_grads, float* const* h_m, float* const* h_v,
                        const int64_t* h_sizes, int32_t n, const float* norms, float clip, float lr_t, float beta1,
                        float beta2, float eps, void* stream) {
  return mt_adam_clip_impl(h_params, h_grads, h_m, h_v, h_sizes, n, norms, clip, lr_t, nullptr, beta1, beta2, eps, stream);
}

int relgnn_mt_adam_clip_devlr(float* const* h_params, const float* const* h_grads, float* const* h_m, float* const* h_v,
                              const int64_t* h_sizes, int32_t n, const float* norms, float clip, const float* d_lr_t,
                              float beta1, float beta2, float eps, void* stream) {
  if (!d_lr_t) return RELGNN_EINVAL;
  return mt_adam_clip_impl(h_params, h_grads, h_m, h_v, h_sizes, n, norms, clip, 0.f, d_lr_t, beta1, beta2, eps, stream);
}

static int mt_adam_clip_impl(float* const* h_params, const float* const* h_grads, float* const* h_m, float* const* h_v,
                             const int64_t* h_sizes, int32_t n, const float* norms, float clip, float lr_t,
                             const float* d_lr_t, float beta1, float beta2, float eps, void* stream) {
  if (n < 0 || n > RELGNN_MT_MAX) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!h_params || !h_grads || !h_m || !h_v || !h_sizes || (clip > 0.f && !norms)) return RELGNN_EINVAL;
  MtArgs a;
  long long maxn = 0;
  for (int i = 0; i < n; ++i) {
    if (h_sizes[i] < 0) return RELGNN_EINVAL;
    a.p[i] = h_params[i]; a.g[i] = h_grads[i]; a.m[i] = h_m[i]; a.v[i] = h_v[i]; a.n[i] = h_sizes[i];
    if (h_sizes[i] > maxn) maxn = h_sizes[i];
  }
  if (maxn == 0) return RELGNN_OK;
  dim3 grid((unsigned)((maxn + 4095) / 4096), (unsigned)n);
  mt_adam_clip_kernel<<<grid, 256, 0, as_stream(stream)>>>(a, norms, clip, lr_t, d_lr_t, beta1, beta2, eps);
  return launch_status();
}

size_t relgnn_sigmoid_ce_stats_workspace_bytes(void) { return (size_t)kStatsBlocks * 4 * sizeof(double); }

int relgnn_sigmoid_ce_stats(const float* logits, const float* labels, int64_t n, float mean_scale, float* stats,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || !stats) return RELGNN_EINVAL;
  if (!workspace || workspace_bytes < relgnn_sigmoid_ce_stats_workspace_bytes()) return RELGNN_ENOSPC;
  if (n > 0 && (!logits || !labels)) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  int nblk = (int)((n + 1023) / 1024);
  if (nblk < 1) nblk = 1;
  if (nblk > kStatsBlocks) nblk = kStatsBlocks;
  sigmoid_ce_stats_kernel<<<nblk, 1024, 0, st>>>(logits, labels, n, static_cast<double*>(workspace));
  sigmoid_ce_stats_final_kernel<<<1, 256, 0, st>>>(static_cast<const double*>(workspace), nblk, mean_scale, stats);
  return launch_status();
}

int relgnn_sigmoid_ce_bwd(const float* logits, const float* labels, int64_t n, const float* g_mean, float mean_scale,
                          const float* g_total, float* glogits, void* stream) {
  if (n < 0) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!logits || !labels || (!g_mean && !g_total) || !glogits) return RELGNN_EINVAL;
  sigmoid_ce_bwd_kernel<<<flat_grid(n, 256), 256, 0, as_stream(stream)>>>(logits, labels, n, g_mean, mean_scale, g_total,
                                                                          glogits);
  return launch_status();
}

int relgnn_sigmoid_ce_bwd_padded(const float* logits, const float* labels, int64_t rows, int32_t cols, const float* g_mean,
                                 float mean_scale, const float* g_total, float* glogits, int32_t ldg, void* stream) {
  if (rows < 0 || cols < 0 || ldg < cols) return RELGNN_EINVAL;
  if (rows == 0 || ldg == 0) return RELGNN_OK;
  if (!logits || !labels || (!g_mean && !g_total) || !glogits) return RELGNN_EINVAL;
  sigmoid_ce_bwd_padded_kernel<<<flat_grid(rows * ldg, 256), 256, 0, as_stream(stream)>>>(logits, labels, rows, cols, ldg, g_mean,
                                                                                          mean_scale, g_total, glogits);
  return launch_status();
}

}  // extern "C"
