// Plain library GEMMs of the node-side Dense layers (Keras Dense, models/sparse_graph_model.py:165-172,194-200; the
// per-edge-type transforms of gnns/rgcn.py:70-74 evaluated node-side) through hipBLASLt with a SOLUTION CACHE.
//
// Why not just torch.mm: every batch of a shuffled epoch has its own node count V, so every GEMM of every step is a shape
// the library has not seen, and its solution lookup (hipblasLtMatmulAlgoGetHeuristic, run by torch on every call) costs
// ~70 us of HOST time per new shape — measured 74 us vs 18 us for a repeated shape (scripts/exp_matmul_host_cost.py);
// with ~21 GEMM calls per training step that was 1.5 of the 2.5 ms the host needs to enqueue one C2 step, and the host,
// not the GPU, bounded the step.  A solution picked for one V is as good for the next V: this file asks the heuristic
// once per (layout, N, K, batch, V rounded to 4096 rows) and reuses the answer.  Row-major operands are handed to the
// column-major library as the transposed problem (C^T = B^T A^T), no copies.
#include "common.h"

#include <hipblaslt/hipblaslt.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

using namespace relgnn;

namespace {

struct Key {
  int32_t layout, n, k, batch, m_bucket, bias, beta1, act;
  bool operator==(const Key& o) const {
    return layout == o.layout && n == o.n && k == o.k && batch == o.batch && m_bucket == o.m_bucket && bias == o.bias &&
           beta1 == o.beta1 && act == o.act;
  }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 1469598103934665603ull;
    for (int32_t v : {k.layout, k.n, k.k, k.batch, k.m_bucket, k.bias, k.beta1, k.act}) h = (h ^ (uint32_t)v) * 1099511628211ull;
    return (size_t)h;
  }
};

struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  uint64_t pref_ws = 0;
  std::unordered_map<Key, hipblasLtMatmulAlgo_t, KeyHash> algos;
};
State& state() {
  static State s;
  return s;
}

struct Layout {
  hipblasLtMatrixLayout_t l = nullptr;
  ~Layout() { if (l) hipblasLtMatrixLayoutDestroy(l); }
  bool make(uint64_t rows, uint64_t cols, int64_t ld, int32_t batch, int64_t stride) {
    if (hipblasLtMatrixLayoutCreate(&l, HIP_R_32F, rows, cols, ld) != HIPBLAS_STATUS_SUCCESS) return false;
    if (batch > 1) {
      if (hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch)) != HIPBLAS_STATUS_SUCCESS) return false;
      if (hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &stride, sizeof(stride)) != HIPBLAS_STATUS_SUCCESS) return false;
    }
    return true;
  }
};
struct Desc {
  hipblasLtMatmulDesc_t d = nullptr;
  ~Desc() { if (d) hipblasLtMatmulDescDestroy(d); }
};

}  // namespace

extern "C" {

int relgnn_blaslt_gemm_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                           float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t batch, int64_t stride_a,
                           int64_t stride_b, int64_t stride_c, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  if (layout < RELGNN_GEMM_NN || layout > RELGNN_GEMM_TN || M < 0 || N < 0 || K < 0 || batch < 1 || workspace_bytes < 0)
    return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C || K == 0) return RELGNN_EINVAL;
  if (bias && (batch > 1 || accumulate)) return RELGNN_EINVAL;
  if (act != RELGNN_ACT_LINEAR && act != RELGNN_ACT_RELU) return RELGNN_EUNSUPPORTED;   // the library's epilogues: none / ReLU
  if (act != RELGNN_ACT_LINEAR && (batch > 1 || accumulate)) return RELGNN_EINVAL;
  State& s = state();
  std::lock_guard<std::mutex> lock(s.mu);
  if (!s.handle && hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) return RELGNN_EHIP;

  // row-major C[M, N] = op(A) op(B)   ==   column-major C^T[N, M] = op'(B) op'(A)
  //   NN: B is [K, N] row-major = column-major N x K (ld ldb), not transposed;  A is [M, K] = column-major K x M (ld lda)
  //   NT: B is [N, K] row-major = column-major K x N: transposed;               A as in NN
  //   TN: A is [K, M] row-major = column-major M x K: transposed;               B as in NN
  const hipblasOperation_t op_first = layout == RELGNN_GEMM_NT ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  const hipblasOperation_t op_second = layout == RELGNN_GEMM_TN ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  Desc desc;
  if (hipblasLtMatmulDescCreate(&desc.d, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return RELGNN_EHIP;
  int32_t opa = (int32_t)op_first, opb = (int32_t)op_second;
  hipblasLtMatmulDescSetAttribute(desc.d, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa));
  hipblasLtMatmulDescSetAttribute(desc.d, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb));
  if (bias || act == RELGNN_ACT_RELU) {
    hipblasLtEpilogue_t ep = bias ? (act == RELGNN_ACT_RELU ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                  : HIPBLASLT_EPILOGUE_RELU;
    hipblasLtMatmulDescSetAttribute(desc.d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
    if (bias) hipblasLtMatmulDescSetAttribute(desc.d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  }
  Layout first, second, out;
  const bool ok = (layout == RELGNN_GEMM_NT ? first.make(K, N, ldb, batch, stride_b) : first.make(N, K, ldb, batch, stride_b)) &&
                  (layout == RELGNN_GEMM_TN ? second.make(M, K, lda, batch, stride_a) : second.make(K, M, lda, batch, stride_a)) &&
                  out.make(N, M, ldc, batch, stride_c);
  if (!ok) return RELGNN_EHIP;

  const float alpha = 1.f, beta = accumulate ? 1.f : 0.f;
  hipStream_t st_ = as_stream(stream);
  // (in the weight-gradient layout K is the node dimension — per-batch, like M in the other two: bucketed as well)
  // The per-batch dimension (M; K in the weight-gradient layout) enters the key in buckets of 4096 (256) rows.  Coarser
  // classes were measured and are worse: with ONE class for the 28 k .. 40 k nodes of the C2 batches (round(log2 M)) the
  // pick for the first batch's M (MT256x144x32) served every batch, where the heuristic asked per 4096-row bucket also
  // returns MT256x112x32 / MT256x32x64 / MT256x160x32 — the library GEMMs of 50 steps took 71.2 ms instead of 60.6 ms
  // (+0.21 ms per step; the macro tile interacts with M through the number of tile rows per CU).
  static const int m_shift = [] { const char* e = getenv("RELGNN_GEMM_BUCKET_SHIFT"); return e ? atoi(e) : 12; }();
  const Key key{layout, N, layout == RELGNN_GEMM_TN ? (K >> (m_shift > 4 ? m_shift - 4 : 0)) : K, batch, M >> m_shift, bias ? 1 : 0,
                accumulate ? 1 : 0, act};
  auto it = s.algos.find(key);
  bool first_use = false;
  if (it == s.algos.end()) {
    if (!s.pref && hipblasLtMatmulPreferenceCreate(&s.pref) != HIPBLAS_STATUS_SUCCESS) return RELGNN_EHIP;
    const uint64_t ws = (uint64_t)workspace_bytes;
    if (ws != s.pref_ws) {
      hipblasLtMatmulPreferenceSetAttribute(s.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
      s.pref_ws = ws;
    }
    // First use of a shape class: the library's heuristic returns its candidates best-guess first, and that first guess is
    // what is cached by default.  RELGNN_GEMM_TUNE=n (n >= 2) MEASURES the first n candidates instead — each run on the
    // caller's operands, timed with events on the caller's stream, the fastest kept for the class.  Measured on the C2
    // shapes the first guess is within 2-5 % of the best except for the 121-column head (53 -> 32 us); a class costs ~4 ms
    // to measure, which a long training run amortises and a 100-step benchmark does not (2.91 vs 2.63 ms per step with the
    // measurements inside the run), hence opt-in.  (With coarse round(log2) size classes, so that every class is measured in
    // the warm-up steps, n = 8 gave 2.32 ms against 2.37 ms for the first guess of the same coarse classes — and the first
    // guess per 4096-row bucket gives that 2.30-2.32 ms without measuring anything.)  The winner of a close race, and with
    // it a split product's summation order, would also depend on timing noise.  Never done for accumulating calls (the
    // timing runs would add into C) nor while the stream is being captured into a graph.
    static const int want = [] { const char* e = getenv("RELGNN_GEMM_TUNE"); return e ? atoi(e) : 1; }();
    constexpr int kMax = 16;
    hipblasLtMatmulHeuristicResult_t res[kMax];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st_, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const int ask = (accumulate || capturing || want < 2) ? 1 : (want > kMax ? kMax : want);
    int found = 0;
    if (hipblasLtMatmulAlgoGetHeuristic(s.handle, desc.d, first.l, second.l, out.l, out.l, s.pref, ask, res, &found) !=
            HIPBLAS_STATUS_SUCCESS || found < 1)
      return RELGNN_EUNSUPPORTED;
    int best = 0;
    if (found > 1) {
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float best_ms = 0.f;
        bool have = false;
        for (int i = 0; i < found; ++i) {
          bool ok_run = true;
          for (int r = 0; r < 4 && ok_run; ++r) {          // run 0 warms the kernel up, runs 1-3 are timed together
            if (r == 1) hipEventRecord(e0, st_);
            ok_run = hipblasLtMatmul(s.handle, desc.d, &alpha, B, first.l, A, second.l, &beta, C, out.l, C, out.l, &res[i].algo,
                                     workspace, (size_t)workspace_bytes, st_) == HIPBLAS_STATUS_SUCCESS;
          }
          hipEventRecord(e1, st_);
          float ms = 0.f;
          if (hipEventSynchronize(e1) != hipSuccess || !ok_run || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
          if (getenv("RELGNN_GEMM_TUNE_LOG")) fprintf(stderr, "tune layout %d M %d N %d K %d batch %d cand %d: %.1f us\n", layout, M, N, K, batch, i, ms / 3 * 1e3f);
          if (!have || ms < best_ms) { best_ms = ms; best = i; have = true; }
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
      }
    }
    it = s.algos.emplace(key, res[best].algo).first;
    first_use = true;
  }
  static const bool miss_log = getenv("RELGNN_GEMM_MISS_LOG") != nullptr;
  if (miss_log && first_use) hipStreamSynchronize(st_);
  const auto t_miss = std::chrono::steady_clock::now();
  hipblasStatus_t st = hipblasLtMatmul(s.handle, desc.d, &alpha, B, first.l, A, second.l, &beta, C, out.l, C, out.l, &it->second,
                                       workspace, (size_t)workspace_bytes, st_);
  if (miss_log && first_use) {          // (diagnostic: what the FIRST launch of a newly picked solution costs — its code object may load now)
    hipStreamSynchronize(st_);
    fprintf(stderr, "relgnn_blaslt_gemm_f32: new class layout %d M %d N %d K %d batch %d: first launch %.1f ms\n", layout, M, N, K, batch,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_miss).count());
  }
  if (st != HIPBLAS_STATUS_SUCCESS) {
    // a cached solution that does not take this V: ask again for exactly this shape (and keep that answer)
    s.algos.erase(it);
    hipblasLtMatmulHeuristicResult_t res[1];
    int found = 0;
    if (!s.pref || hipblasLtMatmulAlgoGetHeuristic(s.handle, desc.d, first.l, second.l, out.l, out.l, s.pref, 1, res, &found) !=
                       HIPBLAS_STATUS_SUCCESS || found < 1)
      return RELGNN_EUNSUPPORTED;
    st = hipblasLtMatmul(s.handle, desc.d, &alpha, B, first.l, A, second.l, &beta, C, out.l, C, out.l, &res[0].algo, workspace,
                         (size_t)workspace_bytes, st_);
    if (st != HIPBLAS_STATUS_SUCCESS) return RELGNN_EHIP;
    s.algos.emplace(key, res[0].algo);
  }
  return RELGNN_OK;
}

}  // extern "C"
