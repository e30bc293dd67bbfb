/*
 * Sequential segment reductions — restatement of TensorFlow-CPU's UnsortedSegment{Sum,Max}
 * functor as used through utils/utils.py:23-33 of the reference: ONE thread walks the
 * messages j = 0..M-1 in order and folds data[j, :] into out[segment_ids[j], :]
 * [TF-internal: tensorflow/core/kernels/segment_reduction_ops.cc, UnsortedSegmentFunctor<CPUDevice>;
 * negative ids are skipped, outputs are initialised to 0 (sum) / lowest() (max)].
 *
 * Test infrastructure (oracle/__init__.py).  Built by oracle/Makefile into oracle/_build/.
 */
#include <float.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_SEG(NAME, T, INIT, FOLD)                                                    \
  void NAME(const T* data, const int32_t* ids, int64_t M, int64_t D, int64_t S, T* out) { \
    for (int64_t i = 0; i < S * D; ++i) out[i] = (INIT);                                   \
    for (int64_t j = 0; j < M; ++j) {                                                      \
      int64_t s = ids[j];                                                                  \
      if (s < 0) continue; /* TF drops negative segment ids */                             \
      T* o = out + s * D;                                                                  \
      const T* x = data + j * D;                                                           \
      for (int64_t d = 0; d < D; ++d) { FOLD; }                                            \
    }                                                                                      \
  }

DEFINE_SEG(oracle_seg_sum_f32, float, 0.0f, o[d] = o[d] + x[d])
DEFINE_SEG(oracle_seg_sum_f64, double, 0.0, o[d] = o[d] + x[d])
DEFINE_SEG(oracle_seg_max_f32, float, -FLT_MAX, o[d] = (x[d] > o[d]) ? x[d] : o[d])
DEFINE_SEG(oracle_seg_max_f64, double, -DBL_MAX, o[d] = (x[d] > o[d]) ? x[d] : o[d])

/* Fused reference op chain of one RGCN-style message pass, for the CPU baseline timing:
 * out[tgt[j], :] += scale[j] * msgs[j, :]   (gnns/rgcn.py:100-112), still sequential in j. */
void oracle_scaled_seg_sum_f32(const float* msgs, const float* scale, const int32_t* ids, int64_t M,
                               int64_t D, int64_t S, float* out) {
  memset(out, 0, sizeof(float) * (size_t)(S * D));
  for (int64_t j = 0; j < M; ++j) {
    float* o = out + (int64_t)ids[j] * D;
    const float* x = msgs + j * D;
    const float sc = scale ? scale[j] : 1.0f;
    for (int64_t d = 0; d < D; ++d) {
      float m = sc * x[d];
      o[d] = o[d] + m;
    }
  }
}

/* The same chain with the gather folded in, for BASELINE-size parity cases (a [M, D] message tensor of 2 GB is
 * slow to materialise in NumPy): out[tgt[j], :] += scale[j] * table[row[j], :], sequential in j, product and
 * add rounded separately.  table row = the message value the reference would have computed for edge j
 * (gnns/rgcn.py:87-104 with the per-edge MatMul evaluated once per (edge type, source node)). */
void oracle_gather_scaled_seg_sum_f32(const float* table, const int64_t* row, const float* scale, const int32_t* ids,
                                      int64_t M, int64_t D, int64_t S, float* out) {
  memset(out, 0, sizeof(float) * (size_t)(S * D));
  for (int64_t j = 0; j < M; ++j) {
    float* o = out + (int64_t)ids[j] * D;
    const float* x = table + row[j] * D;
    const float sc = scale ? scale[j] : 1.0f;
    for (int64_t d = 0; d < D; ++d) {
      float m = sc * x[d];
      o[d] = o[d] + m;
    }
  }
}
