// Fused gather + per-message scale + segment reduce: THE hot kernel of the path.
//
// Replaces, in one pass over a (target, edge type)-bucketed CSR:
//   tf.nn.embedding_lookup            gnns/rgcn.py:87-89
//   1/(num_incoming + 1e-7) multiply  gnns/rgcn.py:100-104
//   tf.concat(messages_per_type)      gnns/rgcn.py:108
//   tf.unsorted_segment_{sum,mean,sqrt_n,max}   gnns/rgcn.py:109-112 (utils/utils.py:23-33)
// and the same call-site family in ggnn.py / rgin.py / gnn_film.py / gnn_edge_mlp.py.
//
// Mapping (CDNA4, wave = 64):
//   * lanes run across the FEATURE dimension (float4 per lane: one 1 KiB coalesced
//     global_load_dwordx4 per wave for a 256-float row), so one wave owns one output row
//     and accumulates the segment's messages SEQUENTIALLY in registers, in the
//     reference's message order (type-major, then edge order).  No atomics, no LDS, and
//     the fp32 summation order of TF-CPU's UnsortedSegmentSum is preserved.
//   * the gathered row index is wave-uniform: it is fetched with one coalesced load per
//     64 messages and broadcast with v_readlane into an SGPR, so every row load is
//     `scalar base + lane offset` (no per-lane 64-bit address arithmetic).
//   * UNROLL independent row loads are in flight per wave before the first add.
//   * for D <= 128 a wave is split into 64/GROUP lane groups, one segment each.
//   * blockIdx is remapped so that each XCD (own 4 MiB L2) walks a contiguous range of
//     targets: a batch is a disjoint union of graphs, so neighbouring targets gather
//     from the same ~2 MiB slab of source rows.
//
// Bound: HBM / Infinity-Cache bandwidth.  Algorithmic bytes per launch:
//   M*(4*D + 8) + S_out*4*D (+ 4*(S_out*stride+1) for rowptr)      
#include "common.h"

#include <stdlib.h>

using namespace relgnn;

namespace {

constexpr int kUnroll = 8;

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case RELGNN_ACT_TANH: return act_fwd<RELGNN_ACT_TANH>(x);
    case RELGNN_ACT_RELU: return act_fwd<RELGNN_ACT_RELU>(x);
    case RELGNN_ACT_LEAKY_RELU: return act_fwd<RELGNN_ACT_LEAKY_RELU>(x);
    case RELGNN_ACT_ELU: return act_fwd<RELGNN_ACT_ELU>(x);
    case RELGNN_ACT_SELU: return act_fwd<RELGNN_ACT_SELU>(x);
    case RELGNN_ACT_GELU: return act_fwd<RELGNN_ACT_GELU>(x);
    default: return x;
  }
}

// per-MESSAGE activation (Edge-MLP messages): the v_exp / v_rcp based variants of common.h
__device__ __forceinline__ float msg_act_apply(int act, float x) {
  switch (act) {
    case RELGNN_ACT_TANH: return act_fwd_fast<RELGNN_ACT_TANH>(x);
    case RELGNN_ACT_RELU: return act_fwd_fast<RELGNN_ACT_RELU>(x);
    case RELGNN_ACT_LEAKY_RELU: return act_fwd_fast<RELGNN_ACT_LEAKY_RELU>(x);
    case RELGNN_ACT_ELU: return act_fwd_fast<RELGNN_ACT_ELU>(x);
    case RELGNN_ACT_SELU: return act_fwd_fast<RELGNN_ACT_SELU>(x);
    case RELGNN_ACT_GELU: return act_fwd_fast<RELGNN_ACT_GELU>(x);
    default: return x;
  }
}

// maximum over the 64 lanes of a wave on the DPP network (six VALU instructions, no LDS round trips: __shfl_xor compiles to
// ds_bpermute, six dependent ~100-cycle LDS accesses at the end of every wave of the gather: measured +5 us per C2 launch).
// quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every 16-lane row holds its maximum in all lanes; row_bcast15 into
// rows 1 and 3, row_bcast31 into rows 2 and 3: lane 63 holds the wave's.  (Masked-out lanes read `old` = 0: neutral for an unsigned maximum.)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <bool IS_MAX, bool MSGACT = false>
__device__ __forceinline__ void combine(float4& acc, float w, const float4& v, int msg_act = RELGNN_ACT_LINEAR) {
  // product and add are rounded separately (file is built with -ffp-contract=off):
  // messages = scale * gathered_row; acc = acc + messages, as the reference's op chain.
  float4 m = make_float4(w * v.x, w * v.y, w * v.z, w * v.w);
  if constexpr (MSGACT) {  // activation applied to every message BEFORE the reduction (separate instantiation:
                           // the default kernel must not pay for the extra argument / branch — measured +42 %)
    m.x = msg_act_apply(msg_act, m.x); m.y = msg_act_apply(msg_act, m.y);
    m.z = msg_act_apply(msg_act, m.z); m.w = msg_act_apply(msg_act, m.w);
  }
  if constexpr (IS_MAX) {
    acc.x = fmaxf(acc.x, m.x); acc.y = fmaxf(acc.y, m.y);
    acc.z = fmaxf(acc.z, m.z); acc.w = fmaxf(acc.w, m.w);
  } else {
    acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
  }
}

__device__ __forceinline__ void combine64(double (&acc)[4], float w, const float4& v) {
  const double wd = (double)w;           // float * float is exact in float64: one rounding (2^-53) per message
  acc[0] = fma(wd, (double)v.x, acc[0]); acc[1] = fma(wd, (double)v.y, acc[1]);
  acc[2] = fma(wd, (double)v.z, acc[2]); acc[3] = fma(wd, (double)v.w, acc[3]);
}

__device__ __forceinline__ float4 finalize(int mode, int act, float4 a, int n) {
  if (mode == RELGNN_AGG_MEAN) {
    float d = (float)max(n, 1);
    a.x /= d; a.y /= d; a.z /= d; a.w /= d;
  } else if (mode == RELGNN_AGG_SQRT_N) {
    float d = sqrtf((float)max(n, 1));
    a.x /= d; a.y /= d; a.z /= d; a.w /= d;
  }
  if (act != RELGNN_ACT_LINEAR) {
    a.x = act_apply(act, a.x); a.y = act_apply(act, a.y);
    a.z = act_apply(act, a.z); a.w = act_apply(act, a.w);
  }
  return a;
}

// ---------------------------------------------------------------------------------------
// One wave per output row.  NCH = float4 chunks per lane (row width up to NCH*256 floats).
// ---------------------------------------------------------------------------------------
// ACC64 (sum-like modes only): the bucket is accumulated in float64 — w * x is exact there and every add rounds at 2^-53 —
// and rounded to float32 ONCE at the end.  Used where the bucket sums feed a GEMM (aggregate-then-transform: the reference
// never forms these per-(target, type) sums, so there is no summation order to reproduce; what matters is how little
// rounding they add in front of the K = L*D dot products).  The kernel is memory-bound: the f64 adds hide under the gather.
template <int NCH, bool IS_MAX, bool HAS_W, int UNROLL = kUnroll, bool NT = false, bool XCD = true, bool MSGACT = false,
          bool ACC64 = false>
__global__ __launch_bounds__(256) void seg_reduce_wave_kernel(
    const float4* __restrict__ X, int64_t ldx4, int32_t D4, const int32_t* __restrict__ rowptr,
    int64_t num_segments, int32_t stride, const int32_t* __restrict__ col,
    const float* __restrict__ w, int32_t mode, int32_t act, float4* __restrict__ out, int64_t ldo4,
    int64_t n_logical_blocks, int32_t col_block0, int32_t msg_act, float* __restrict__ rowmax) {
  const int64_t lb = XCD ? xcd_logical_block(n_logical_blocks)
                         : ((int64_t)blockIdx.x < n_logical_blocks ? (int64_t)blockIdx.x : -1);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t s = lb * 4 + (threadIdx.x >> 6);
  if (s >= num_segments) return;

  // wave-uniform segment bounds -> SGPRs
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[s * stride]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[(s + 1) * stride]);

  const int c0 = (col_block0 + (int)blockIdx.y * NCH) * 64 + lane;  // this lane's first float4 column
  static_assert(!ACC64 || (!IS_MAX && !MSGACT), "float64 accumulation: plain sums only");
  float4 acc[NCH];
  double accd[ACC64 ? NCH : 1][4];
  bool on[NCH];
  uint32_t cc[NCH];  // column used for LOADS: clamped into the row so that lanes past the
                     // row end read a valid (ignored) address instead of branching
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float init = IS_MAX ? -FLT_MAX : 0.f;
    acc[c] = make_float4(init, init, init, init);
    if constexpr (ACC64) accd[c][0] = accd[c][1] = accd[c][2] = accd[c][3] = 0.0;
    on[c] = (c0 + 64 * c) < D4;
    cc[c] = (uint32_t)min(c0 + 64 * c, D4 - 1);
  }
  const uint32_t ld = (uint32_t)ldx4;  // host guarantees num_rows_x * ldx4 < 2^32

  for (int p = beg; p < end; p += 64) {
    const int n = min(64, end - p);
    // one coalesced index (and weight) load per 64 messages
    // index / weight streams are read once: keep them out of the way of the gathered rows (NT)
    const int my_col = (lane < n) ? (NT ? __builtin_nontemporal_load(col + p + lane) : col[p + lane]) : 0;
    float my_w = 1.f;
    if constexpr (HAS_W) my_w = (lane < n) ? (NT ? __builtin_nontemporal_load(w + p + lane) : w[p + lane]) : 0.f;

    int k = 0;
    for (; k + UNROLL <= n; k += UNROLL) {
      float4 v[UNROLL][NCH];
      float ww[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k + u);
        ww[u] = HAS_W ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), k + u)) : 1.f;
        const float4* row = X + (size_t)(r * ld);  // scalar (SGPR) row base
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[u][c] = row[cc[c]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if constexpr (ACC64) combine64(accd[c], ww[u], v[u][c]);
          else combine<IS_MAX, MSGACT>(acc[c], ww[u], v[u][c], msg_act);
        }
    }
    for (; k < n; ++k) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k);
      const float wk = HAS_W ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), k)) : 1.f;
      const float4* row = X + (size_t)(r * ld);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float4 v = row[cc[c]];
        if constexpr (ACC64) combine64(accd[c], wk, v);
        else combine<IS_MAX, MSGACT>(acc[c], wk, v, msg_act);
      }
    }
  }
  if constexpr (ACC64) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      acc[c] = make_float4((float)accd[c][0], (float)accd[c][1], (float)accd[c][2], (float)accd[c][3]);
  }

  float4* orow = out + s * ldo4 + c0;
  // the largest FINITE magnitude of the row as it is written (rowmax: relgnn_seg_reduce_fwd_rowmax), as a bit pattern (magnitudes
  // order like unsigned integers; inf / NaN patterns are the largest of all).  inf / NaN elements do not count: a scale derived
  // from the finite ones keeps those representable, and the non-finite element spoils its row of the product, as it does in fp32.
  // Cheap path: one AND per value and unsigned maxima; only a row that holds a non-finite value (its maximum says so) is walked
  // again with the filter.
  uint32_t mxb = 0u, mxf = 0u;
  auto absb = [](float x) { return __float_as_uint(x) & 0x7FFFFFFFu; };
  auto finb = [](float x) { const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu; return u < 0x7F800000u ? u : 0u; };
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) {
      const float4 r = finalize(mode, act, acc[c], end - beg);
      mxb = max(max(mxb, max(absb(r.x), absb(r.y))), max(absb(r.z), absb(r.w)));
      if (__builtin_expect(mxb >= 0x7F800000u, 0))
        mxf = max(max(mxf, max(finb(r.x), finb(r.y))), max(finb(r.z), finb(r.w)));
      else
        mxf = mxb;
      if constexpr (NT) {  // streamed output: do not displace gathered rows from L2
        float* o = reinterpret_cast<float*>(orow + 64 * c);
        __builtin_nontemporal_store(r.x, o); __builtin_nontemporal_store(r.y, o + 1);
        __builtin_nontemporal_store(r.z, o + 2); __builtin_nontemporal_store(r.w, o + 3);
      } else {
        orow[64 * c] = r;
      }
    }
  if (rowmax) {                        // (wave-uniform; the wave holds the whole row: one column block)
    const uint32_t m = wave_max_u32(mxf);
    if (lane == 0) rowmax[s] = __uint_as_float(m);
  }
}

// ---------------------------------------------------------------------------------------
// 64/GROUP segments per wave (rows of at most GROUP float4 = GROUP*4 floats).
// ---------------------------------------------------------------------------------------
template <int GROUP, bool IS_MAX, bool HAS_W, bool MSGACT = false>
__global__ __launch_bounds__(256) void seg_reduce_group_kernel(
    const float4* __restrict__ X, int64_t ldx4, int32_t D4, const int32_t* __restrict__ rowptr,
    int64_t num_segments, int32_t stride, const int32_t* __restrict__ col,
    const float* __restrict__ w, int32_t mode, int32_t act, float4* __restrict__ out, int64_t ldo4,
    int64_t n_logical_blocks, int32_t msg_act) {
  constexpr int SEGS_PER_WAVE = 64 / GROUP;
  constexpr int U = 4;
  const int64_t lb = xcd_logical_block(n_logical_blocks);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int g = lane / GROUP, gl = lane % GROUP;
  const int64_t s = (lb * 4 + (threadIdx.x >> 6)) * SEGS_PER_WAVE + g;
  const bool valid = s < num_segments;
  const int beg = valid ? rowptr[s * stride] : 0;
  const int end = valid ? rowptr[(s + 1) * stride] : 0;
  const int len = end - beg;
  // wave-uniform trip count: longest segment of the wave
  int maxlen = len;
#pragma unroll
  for (int off = 32; off >= GROUP; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);

  const bool on = gl < D4;
  const float init = IS_MAX ? -FLT_MAX : 0.f;
  float4 acc = make_float4(init, init, init, init);
  // loads are unconditional (clamped column, row 0 for padding messages) and the combine is
  // predicated with a select: no divergent branches inside the wave
  const float4* Xl = X + min(gl, D4 - 1);
  const uint32_t ld = (uint32_t)ldx4;

  for (int p0 = 0; p0 < maxlen; p0 += GROUP) {
    const int idx = p0 + gl;
    const int my_col = (idx < len) ? col[beg + idx] : 0;
    float my_w = 1.f;
    if constexpr (HAS_W) my_w = (idx < len) ? w[beg + idx] : 0.f;
    const int nn = min(GROUP, maxlen - p0);
    for (int k = 0; k < nn; k += U) {
      float4 v[U];
      float ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t r = (uint32_t)__shfl(my_col, (k + u) & (GROUP - 1), GROUP);
        ww[u] = HAS_W ? __shfl(my_w, (k + u) & (GROUP - 1), GROUP) : 1.f;
        v[u] = Xl[(size_t)(r * ld)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float4 t = acc;
        combine<IS_MAX, MSGACT>(t, ww[u], v[u], msg_act);
        if (p0 + k + u < len) acc = t;
      }
    }
  }
  if (valid && on) out[s * ldo4 + gl] = finalize(mode, act, acc, len);
}

// ---------------------------------------------------------------------------------------
// Scalar-lane fallback: any D / ld / alignment.  One wave per output row, lanes stride D.
// ---------------------------------------------------------------------------------------
template <bool IS_MAX>
__global__ __launch_bounds__(256) void seg_reduce_scalar_kernel(
    const float* __restrict__ X, int64_t ldx, int32_t D, const int32_t* __restrict__ rowptr,
    int64_t num_segments, int32_t stride, const int32_t* __restrict__ col,
    const float* __restrict__ w, int32_t mode, int32_t act, float* __restrict__ out, int64_t ldo, int32_t msg_act) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= num_segments) return;
  const int beg = rowptr[s * stride], end = rowptr[(s + 1) * stride];
  for (int d = lane; d < D; d += 64) {
    float acc = IS_MAX ? -FLT_MAX : 0.f;
    for (int p = beg; p < end; ++p) {
      float m = msg_act_apply(msg_act, (w ? w[p] : 1.f) * X[(int64_t)col[p] * ldx + d]);
      acc = IS_MAX ? fmaxf(acc, m) : acc + m;
    }
    float nrm = (float)max(end - beg, 1);
    if (mode == RELGNN_AGG_MEAN) acc /= nrm;
    if (mode == RELGNN_AGG_SQRT_N) acc /= sqrtf(nrm);
    out[s * ldo + d] = act_apply(act, acc);
  }
}

// ---- unsorted_segment_max backward helpers (not on the default path; generic lanes) ----
// gsel[s,d] = gout[s,d] / #{p in s : w[p]*X[col[p],d] == out[s,d]}
__global__ __launch_bounds__(256) void seg_max_count_kernel(
    const float* __restrict__ X, int64_t ldx, int32_t D, const int32_t* __restrict__ rowptr,
    int64_t num_segments, int32_t stride, const int32_t* __restrict__ col,
    const float* __restrict__ w, const float* __restrict__ out, const float* __restrict__ gout,
    int64_t ldo, float* __restrict__ gsel) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= num_segments) return;
  const int beg = rowptr[s * stride], end = rowptr[(s + 1) * stride];
  for (int d = lane; d < D; d += 64) {
    const float o = out[s * ldo + d];
    float cnt = 0.f;
    for (int p = beg; p < end; ++p) {
      float m = (w ? w[p] : 1.f) * X[(int64_t)col[p] * ldx + d];
      cnt += (m == o) ? 1.f : 0.f;
    }
    gsel[s * ldo + d] = cnt > 0.f ? gout[s * ldo + d] / cnt : 0.f;
  }
}

// transposed plan: gX[r,d] = sum_q [w_b[q]*X[r,d] == out[seg_b[q],d]] * w_b[q] * gsel[seg_b[q],d]
__global__ __launch_bounds__(256) void seg_max_bwd_kernel(
    const float* __restrict__ X, int64_t ldx, int32_t D, const int32_t* __restrict__ rowptr_b,
    int64_t num_rows, int32_t stride_b, const int32_t* __restrict__ seg_b,
    const float* __restrict__ w_b, const float* __restrict__ out, const float* __restrict__ gsel,
    int64_t ldo, float* __restrict__ gX, int64_t ldgx) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= num_rows) return;
  const int beg = rowptr_b[r * stride_b], end = rowptr_b[(r + 1) * stride_b];
  for (int d = lane; d < D; d += 64) {
    const float x = X[r * ldx + d];
    float acc = 0.f;
    for (int q = beg; q < end; ++q) {
      const float wq = w_b ? w_b[q] : 1.f;
      const int64_t s = seg_b[q];
      if (wq * x == out[s * ldo + d]) acc += wq * gsel[s * ldo + d];
    }
    gX[r * ldgx + d] = acc;
  }
}

// ---- the same two passes with float4 lanes (rows of D % 4 == 0 floats, 16-byte aligned) ---------------------------------
// G lanes per output row (G = 8 / 16 / 32 for D <= 32 / 64 / 128, else 64 with a chunk loop), 64 / G rows per wave, four
// entries' indices and rows in flight per lane group before the first compare: the structure of seg_reduce_group_kernel.
// Same arithmetic as the scalar kernels above (the product w * x is compared with the stored maximum bit for bit).
__device__ __forceinline__ float4 eq4(float4 a, float4 b) {
  return make_float4(a.x == b.x ? 1.f : 0.f, a.y == b.y ? 1.f : 0.f, a.z == b.z ? 1.f : 0.f, a.w == b.w ? 1.f : 0.f);
}

template <int G>
__global__ __launch_bounds__(256) void seg_max_count_vec_kernel(
    const float4* __restrict__ X, int64_t ldx4, int32_t D4, const int32_t* __restrict__ rowptr, int64_t num_segments,
    int32_t stride, const int32_t* __restrict__ col, const float* __restrict__ w, const float4* __restrict__ out,
    const float4* __restrict__ gout, int64_t ldo4, float4* __restrict__ gsel, int64_t nlb) {
  constexpr int U = 4;
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63, g = lane / G, gl = lane % G;
  const int64_t s = (lb * 4 + (threadIdx.x >> 6)) * (64 / G) + g;
  if (s >= num_segments) return;
  const int beg = rowptr[s * stride], end = rowptr[(s + 1) * stride];
  for (int c = gl; c < D4; c += G) {
    const float4 o = out[s * ldo4 + c];
    float4 cnt = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = beg; p < end; p += U) {
      int r[U];
      float ww[U];
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = min(p + u, end - 1);
        r[u] = col[idx];
        ww[u] = w ? w[idx] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = X[(int64_t)r[u] * ldx4 + c];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u < end) {
          const float4 e = eq4(make_float4(ww[u] * v[u].x, ww[u] * v[u].y, ww[u] * v[u].z, ww[u] * v[u].w), o);
          cnt.x += e.x; cnt.y += e.y; cnt.z += e.z; cnt.w += e.w;
        }
    }
    const float4 gg = gout[s * ldo4 + c];
    gsel[s * ldo4 + c] = make_float4(cnt.x > 0.f ? gg.x / cnt.x : 0.f, cnt.y > 0.f ? gg.y / cnt.y : 0.f,
                                     cnt.z > 0.f ? gg.z / cnt.z : 0.f, cnt.w > 0.f ? gg.w / cnt.w : 0.f);
  }
}

template <int G>
__global__ __launch_bounds__(256) void seg_max_bwd_vec_kernel(
    const float4* __restrict__ X, int64_t ldx4, int32_t D4, const int32_t* __restrict__ rowptr_b, int64_t num_rows,
    int32_t stride_b, const int32_t* __restrict__ seg_b, const float* __restrict__ w_b, const float4* __restrict__ out,
    const float4* __restrict__ gsel, int64_t ldo4, float4* __restrict__ gX, int64_t ldgx4, int64_t nlb) {
  constexpr int U = 4;
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63, g = lane / G, gl = lane % G;
  const int64_t r = (lb * 4 + (threadIdx.x >> 6)) * (64 / G) + g;
  if (r >= num_rows) return;
  const int beg = rowptr_b[r * stride_b], end = rowptr_b[(r + 1) * stride_b];
  for (int c = gl; c < D4; c += G) {
    const float4 x = X[r * ldx4 + c];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = beg; q < end; q += U) {
      int sg[U];
      float wq[U];
      float4 o[U], gs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = min(q + u, end - 1);
        sg[u] = seg_b[idx];
        wq[u] = w_b ? w_b[idx] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        o[u] = out[(int64_t)sg[u] * ldo4 + c];
        gs[u] = gsel[(int64_t)sg[u] * ldo4 + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (q + u < end) {
          const float4 e = eq4(make_float4(wq[u] * x.x, wq[u] * x.y, wq[u] * x.z, wq[u] * x.w), o[u]);
          // (select, not multiply: 0 * inf from an overflowed gradient must not turn the other entries into NaN)
          acc.x += e.x != 0.f ? wq[u] * gs[u].x : 0.f; acc.y += e.y != 0.f ? wq[u] * gs[u].y : 0.f;
          acc.z += e.z != 0.f ? wq[u] * gs[u].z : 0.f; acc.w += e.w != 0.f ? wq[u] * gs[u].w : 0.f;
        }
    }
    gX[r * ldgx4 + c] = acc;
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_from_output_kernel(const float* __restrict__ y,
                                                                  const float* __restrict__ gout,
                                                                  int64_t n, float* __restrict__ gin) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float yy = y[i], g = gout[i];
    float d = 1.f;
    if constexpr (ACT == RELGNN_ACT_TANH) d = 1.f - yy * yy;
    if constexpr (ACT == RELGNN_ACT_RELU) d = yy > 0.f ? 1.f : 0.f;
    if constexpr (ACT == RELGNN_ACT_LEAKY_RELU) d = yy > 0.f ? 1.f : 0.2f;
    if constexpr (ACT == RELGNN_ACT_ELU) d = yy > 0.f ? 1.f : yy + 1.f;
    if constexpr (ACT == RELGNN_ACT_SELU) {
      const float scale = 1.0507009873554804934193349852946f;
      const float scale_alpha = 1.7580993408473768599402175208123f;
      d = yy > 0.f ? scale : yy + scale_alpha;
    }
    gin[i] = g * d;
  }
}

// Tuning knob (experiments only): RELGNN_SEG_VARIANT bit0 = unroll 16, bit1 = non-temporal streams,
// bit2 = no XCD swizzle.  Read once.
inline int seg_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RELGNN_SEG_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
}

// gX[m, :] = w[m] * act'(w[m] * X[m, :]) * gagg[tgt[m], :]   (gradient of  sum_m act(w_m X_m)  w.r.t. the
// materialised message tensor X; gagg already carries the mean / sqrt_n factor of the target)
__global__ __launch_bounds__(256) void msg_act_bwd_kernel(int32_t act, const float4* __restrict__ X, int32_t D4,
                                                          const float* __restrict__ w, const int32_t* __restrict__ tgt,
                                                          const float4* __restrict__ gagg, int64_t M,
                                                          float4* __restrict__ gX) {
  const int64_t total = M * D4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D4;
    const int c = (int)(i - m * D4);
    const float wm = w ? w[m] : 1.f;
    const float4 x = X[i];
    const float4 g = gagg[(int64_t)tgt[m] * D4 + c];
    float4 r;
    switch (act) {
#define RELGNN_CASE(A)                                                                                             \
  case A:                                                                                                         \
    r = make_float4(wm * act_grad_fast<A>(wm * x.x) * g.x, wm * act_grad_fast<A>(wm * x.y) * g.y,                   \
                    wm * act_grad_fast<A>(wm * x.z) * g.z, wm * act_grad_fast<A>(wm * x.w) * g.w);              \
    break;
      RELGNN_CASE(RELGNN_ACT_TANH) RELGNN_CASE(RELGNN_ACT_RELU) RELGNN_CASE(RELGNN_ACT_LEAKY_RELU)
      RELGNN_CASE(RELGNN_ACT_ELU) RELGNN_CASE(RELGNN_ACT_SELU) RELGNN_CASE(RELGNN_ACT_GELU)
#undef RELGNN_CASE
      default: r = make_float4(wm * g.x, wm * g.y, wm * g.z, wm * g.w); break;
    }
    gX[i] = r;
  }
}

// set by relgnn_seg_reduce_fwd_rowmax around its call (the wave kernels write one magnitude per output row there)
thread_local float* tl_rowmax = nullptr;

template <int NCH, bool IS_MAX>
int launch_wave(bool has_w, const float* X, int64_t ldx, int32_t D, const int32_t* rowptr,
                int64_t S, int32_t stride, const int32_t* col, const float* w, int32_t mode,
                int32_t act, float* out, int64_t ldo, hipStream_t st, int32_t msg_act, bool acc64 = false) {
  const int D4 = D / 4;
  const int64_t nlb = (S + 3) / 4;
  const int col_blocks = (D4 + 64 * NCH - 1) / (64 * NCH);
  dim3 grid((unsigned)(((nlb + 7) / 8) * 8), (unsigned)col_blocks);
  if constexpr (!IS_MAX) {
    if (acc64) {
      if (msg_act != RELGNN_ACT_LINEAR) return RELGNN_EUNSUPPORTED;
      if (has_w)
        seg_reduce_wave_kernel<NCH, false, true, kUnroll, false, true, false, true><<<grid, 256, 0, st>>>(
            reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
            reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
      else
        seg_reduce_wave_kernel<NCH, false, false, kUnroll, false, true, false, true><<<grid, 256, 0, st>>>(
            reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
            reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
      return launch_status();
    }
  }
  if constexpr (NCH == 1 && !IS_MAX) {
    const int var = seg_variant();
    if (var != 0 && has_w) {
#define RELGNN_VARIANT_CASE(ID, U, N, X_)                                                              \
  case ID:                                                                                             \
    seg_reduce_wave_kernel<1, false, true, U, N, X_><<<grid, 256, 0, st>>>(                            \
        reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,         \
        reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);                                     \
    return launch_status();
      switch (var) {
        RELGNN_VARIANT_CASE(1, 16, false, true)
        RELGNN_VARIANT_CASE(2, 8, true, true)
        RELGNN_VARIANT_CASE(3, 16, true, true)
        RELGNN_VARIANT_CASE(4, 8, false, false)
        RELGNN_VARIANT_CASE(5, 16, false, false)
        RELGNN_VARIANT_CASE(6, 8, true, false)
        RELGNN_VARIANT_CASE(7, 16, true, false)
        RELGNN_VARIANT_CASE(8, 4, false, true)
        default: break;
      }
#undef RELGNN_VARIANT_CASE
    }
  }
  if (msg_act != RELGNN_ACT_LINEAR) {
    if (has_w)
      seg_reduce_wave_kernel<NCH, IS_MAX, true, kUnroll, false, true, true><<<grid, 256, 0, st>>>(
          reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
          reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
    else
      seg_reduce_wave_kernel<NCH, IS_MAX, false, kUnroll, false, true, true><<<grid, 256, 0, st>>>(
          reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
          reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
    return launch_status();
  }
  if (has_w)
    seg_reduce_wave_kernel<NCH, IS_MAX, true><<<grid, 256, 0, st>>>(
        reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
        reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
  else
    seg_reduce_wave_kernel<NCH, IS_MAX, false><<<grid, 256, 0, st>>>(
        reinterpret_cast<const float4*>(X), ldx / 4, D4, rowptr, S, stride, col, w, mode, act,
        reinterpret_cast<float4*>(out), ldo / 4, nlb, 0, msg_act, tl_rowmax);
  return launch_status();
}

template <int GROUP, bool IS_MAX>
int launch_group(bool has_w, const float* X, int64_t ldx, int32_t D, const int32_t* rowptr,
                 int64_t S, int32_t stride, const int32_t* col, const float* w, int32_t mode,
                 int32_t act, float* out, int64_t ldo, hipStream_t st, int32_t msg_act) {
  constexpr int SEGS_PER_BLOCK = 4 * (64 / GROUP);
  const int64_t nlb = (S + SEGS_PER_BLOCK - 1) / SEGS_PER_BLOCK;
  dim3 grid((unsigned)(((nlb + 7) / 8) * 8));
  if (msg_act != RELGNN_ACT_LINEAR) {
    if (has_w)
      seg_reduce_group_kernel<GROUP, IS_MAX, true, true><<<grid, 256, 0, st>>>(
          reinterpret_cast<const float4*>(X), ldx / 4, D / 4, rowptr, S, stride, col, w, mode, act,
          reinterpret_cast<float4*>(out), ldo / 4, nlb, msg_act);
    else
      seg_reduce_group_kernel<GROUP, IS_MAX, false, true><<<grid, 256, 0, st>>>(
          reinterpret_cast<const float4*>(X), ldx / 4, D / 4, rowptr, S, stride, col, w, mode, act,
          reinterpret_cast<float4*>(out), ldo / 4, nlb, msg_act);
    return launch_status();
  }
  if (has_w)
    seg_reduce_group_kernel<GROUP, IS_MAX, true><<<grid, 256, 0, st>>>(
        reinterpret_cast<const float4*>(X), ldx / 4, D / 4, rowptr, S, stride, col, w, mode, act,
        reinterpret_cast<float4*>(out), ldo / 4, nlb, msg_act);
  else
    seg_reduce_group_kernel<GROUP, IS_MAX, false><<<grid, 256, 0, st>>>(
        reinterpret_cast<const float4*>(X), ldx / 4, D / 4, rowptr, S, stride, col, w, mode, act,
        reinterpret_cast<float4*>(out), ldo / 4, nlb, msg_act);
  return launch_status();
}

template <bool IS_MAX>
int dispatch_fwd(const float* X, int64_t num_rows_x, int64_t ldx, int32_t D, const int32_t* rowptr, int64_t S,
                 int32_t stride, const int32_t* col, const float* w, int32_t mode, int32_t act,
                 float* out, int64_t ldo, hipStream_t st, int32_t msg_act, bool acc64 = false) {
  const bool vec_ok = (D % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(X) &&
                      aligned16(out) && (num_rows_x * (ldx / 4) < ((int64_t)1 << 32));
  if (acc64) {        // float64 bucket sums: the one-wave-per-row kernels (rows of 132 .. 1024 floats), plain sums
    if (IS_MAX || !vec_ok || D / 4 <= 32 || D / 4 > 256) return RELGNN_EUNSUPPORTED;
    const bool has_w = w != nullptr;
    if (D / 4 <= 64) return launch_wave<1, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act, true);
    if (D / 4 <= 128) return launch_wave<2, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act, true);
    return launch_wave<4, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act, true);
  }
  if (!vec_ok) {
    seg_reduce_scalar_kernel<IS_MAX><<<(unsigned)((S + 3) / 4), 256, 0, st>>>(
        X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, msg_act);
    return launch_status();
  }
  const bool has_w = w != nullptr;
  const int D4 = D / 4;
  if (D4 <= 8) return launch_group<8, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
  if (D4 <= 16) return launch_group<16, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
  if (D4 <= 32) return launch_group<32, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
  if (D4 <= 64) return launch_wave<1, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
  if (D4 <= 128) return launch_wave<2, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
  return launch_wave<4, IS_MAX>(has_w, X, ldx, D, rowptr, S, stride, col, w, mode, act, out, ldo, st, msg_act);
}

}  // namespace

extern "C" {

static int seg_reduce_any(int32_t mode, int32_t msg_act, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                          const int32_t* rowptr, int64_t num_segments, int32_t seg_stride, const int32_t* col,
                          const float* w, int32_t act, float* out, int64_t ldo, void* stream, bool acc64 = false);

int relgnn_seg_reduce_acc64_fwd(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                                const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                const int32_t* col, const float* w, int32_t act, float* out, int64_t ldo,
                                void* stream) {
  if (mode == RELGNN_AGG_MAX) return RELGNN_EINVAL;
  return seg_reduce_any(mode, RELGNN_ACT_LINEAR, X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w, act,
                        out, ldo, stream, true);
}

int relgnn_seg_reduce_fwd(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                          const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                          const int32_t* col, const float* w, int32_t act, float* out, int64_t ldo,
                          void* stream) {
  return seg_reduce_any(mode, RELGNN_ACT_LINEAR, X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w, act,
                        out, ldo, stream);
}

int relgnn_seg_reduce_fwd_rowmax(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                                 const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                 const int32_t* col, const float* w, int32_t act, float* out, int64_t ldo,
                                 float* rowmax, void* stream) {
  if (!rowmax) return RELGNN_EINVAL;
  // the one-wave-per-row kernels with the whole row in one wave: 128 < D <= 1024, 16-byte aligned rows
  if (mode == RELGNN_AGG_MAX || D % 4 != 0 || D <= 128 || D > 1024 || ldx % 4 != 0 || ldo % 4 != 0 || !aligned16(X) || !aligned16(out) ||
      num_rows_x * (ldx / 4) >= ((int64_t)1 << 32))
    return RELGNN_EUNSUPPORTED;
  tl_rowmax = rowmax;
  const int rc = seg_reduce_any(mode, RELGNN_ACT_LINEAR, X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w, act,
                                out, ldo, stream);
  tl_rowmax = nullptr;
  return rc;
}

int relgnn_seg_reduce_msgact_fwd(int32_t mode, int32_t msg_act, const float* X, int64_t num_rows_x, int64_t ldx,
                                 int32_t D, const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                 const int32_t* col, const float* w, float* out, int64_t ldo, void* stream) {
  if (msg_act < RELGNN_ACT_LINEAR || msg_act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  return seg_reduce_any(mode, msg_act, X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w,
                        RELGNN_ACT_LINEAR, out, ldo, stream);
}

static int seg_reduce_any(int32_t mode, int32_t msg_act, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                          const int32_t* rowptr, int64_t num_segments, int32_t seg_stride, const int32_t* col,
                          const float* w, int32_t act, float* out, int64_t ldo, void* stream, bool acc64) {
  if (mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_MAX) return RELGNN_EINVAL;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (D < 0 || num_segments < 0 || seg_stride <= 0 || num_rows_x < 0 || ldx < D || ldo < D)
    return RELGNN_EINVAL;
  if (num_segments == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out) return RELGNN_EINVAL;
  // X / col may be null only when there is not a single message; the kernel never touches them then.
  if (num_segments > (int64_t)INT32_MAX * 4) return RELGNN_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (mode == RELGNN_AGG_MAX)
    return dispatch_fwd<true>(X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w, mode, act, out, ldo, st, msg_act);
  return dispatch_fwd<false>(X, num_rows_x, ldx, D, rowptr, num_segments, seg_stride, col, w, mode, act, out, ldo, st, msg_act,
                             acc64);
}

int relgnn_msg_act_bwd(int32_t act, const float* X, int32_t D, const float* w, const int32_t* tgt, const float* gagg,
                       int64_t num_messages, float* gX, void* stream) {
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU || D < 0 || num_messages < 0) return RELGNN_EINVAL;
  if (num_messages == 0 || D == 0) return RELGNN_OK;
  if (!X || !tgt || !gagg || !gX) return RELGNN_EINVAL;
  if (D % 4 != 0 || !aligned16(X) || !aligned16(gagg) || !aligned16(gX)) return RELGNN_EUNSUPPORTED;
  msg_act_bwd_kernel<<<flat_grid(num_messages * (D / 4), 256), 256, 0, as_stream(stream)>>>(
      act, (const float4*)X, D / 4, w, tgt, (const float4*)gagg, num_messages, (float4*)gX);
  return launch_status();
}

int relgnn_seg_max_count(const float* X, int64_t ldx, int32_t D, const int32_t* rowptr,
                         int64_t num_segments, int32_t seg_stride, const int32_t* col,
                         const float* w, const float* out, const float* gout, int64_t ldo,
                         float* gsel, void* stream) {
  if (D < 0 || num_segments < 0 || seg_stride <= 0) return RELGNN_EINVAL;
  if (num_segments == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out || !gout || !gsel) return RELGNN_EINVAL;
  if (D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(X) && aligned16(out) && aligned16(gout) && aligned16(gsel)) {
    const int D4 = D / 4;
    hipStream_t st = as_stream(stream);
#define RELGNN_MAXCNT(GG)                                                                                              \
  {                                                                                                                    \
    const int64_t nlb = (num_segments + 4 * (64 / GG) - 1) / (4 * (64 / GG));                                          \
    seg_max_count_vec_kernel<GG><<<(unsigned)(((nlb + 7) / 8) * 8), 256, 0, st>>>(                                      \
        (const float4*)X, ldx / 4, D4, rowptr, num_segments, seg_stride, col, w, (const float4*)out, (const float4*)gout, \
        ldo / 4, (float4*)gsel, nlb);                                                                                  \
  }
    if (D4 <= 8) RELGNN_MAXCNT(8) else if (D4 <= 16) RELGNN_MAXCNT(16) else if (D4 <= 32) RELGNN_MAXCNT(32) else RELGNN_MAXCNT(64)
#undef RELGNN_MAXCNT
    return launch_status();
  }
  seg_max_count_kernel<<<(unsigned)((num_segments + 3) / 4), 256, 0, as_stream(stream)>>>(
      X, ldx, D, rowptr, num_segments, seg_stride, col, w, out, gout, ldo, gsel);
  return launch_status();
}

int relgnn_seg_max_bwd(const float* X, int64_t ldx, int32_t D, const int32_t* rowptr_b,
                       int64_t num_rows_x, int32_t seg_stride_b, const int32_t* seg_b,
                       const float* w_b, const float* out, const float* gsel, int64_t ldo,
                       float* gX, int64_t ldgx, void* stream) {
  if (D < 0 || num_rows_x < 0 || seg_stride_b <= 0) return RELGNN_EINVAL;
  if (num_rows_x == 0 || D == 0) return RELGNN_OK;
  if (!X || !rowptr_b || !gX) return RELGNN_EINVAL;
  if (D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldgx % 4 == 0 && aligned16(X) && aligned16(out) && aligned16(gsel) &&
      aligned16(gX)) {
    const int D4 = D / 4;
    hipStream_t st = as_stream(stream);
#define RELGNN_MAXBWD(GG)                                                                                              \
  {                                                                                                                    \
    const int64_t nlb = (num_rows_x + 4 * (64 / GG) - 1) / (4 * (64 / GG));                                            \
    seg_max_bwd_vec_kernel<GG><<<(unsigned)(((nlb + 7) / 8) * 8), 256, 0, st>>>(                                        \
        (const float4*)X, ldx / 4, D4, rowptr_b, num_rows_x, seg_stride_b, seg_b, w_b, (const float4*)out,              \
        (const float4*)gsel, ldo / 4, (float4*)gX, ldgx / 4, nlb);                                                     \
  }
    if (D4 <= 8) RELGNN_MAXBWD(8) else if (D4 <= 16) RELGNN_MAXBWD(16) else if (D4 <= 32) RELGNN_MAXBWD(32) else RELGNN_MAXBWD(64)
#undef RELGNN_MAXBWD
    return launch_status();
  }
  seg_max_bwd_kernel<<<(unsigned)((num_rows_x + 3) / 4), 256, 0, as_stream(stream)>>>(
      X, ldx, D, rowptr_b, num_rows_x, seg_stride_b, seg_b, w_b, out, gsel, ldo, gX, ldgx);
  return launch_status();
}

int relgnn_act_bwd_from_output(int32_t act, const float* y, const float* gout, int64_t n, float* gin,
                               void* stream) {
  if (n < 0) return RELGNN_EINVAL;
  if (act == RELGNN_ACT_GELU) return RELGNN_EUNSUPPORTED;  // not monotone: needs the pre-activation
  if (n == 0) return RELGNN_OK;
  if (!y || !gout || !gin) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  RELGNN_DISPATCH_ACT(act, A,
                      (act_bwd_from_output_kernel<A><<<flat_grid(n, 256), 256, 0, st>>>(y, gout, n, gin)));
  return launch_status();
}

}  // extern "C"
