// Fused edge-wise message kernels over the (target, type)-bucketed CSR:
//   FiLM   gnns/gnn_film.py:92-116      msg = act( gamma[v,l] * (w * T[src,l]) + beta[v,l] )
//   PAIR   gnns/gnn_edge_mlp.py:91-112, gnns/rgin.py:110-129, gnns/rgcn.py:91-104
//                                        msg = act( w * (P[src,l] + Q[v,l]) )
// Each replaces the reference's chain  embedding_lookup (x2) -> elementwise -> concat ->
// unsorted_segment_*  by ONE pass: a lane group owns one target node, walks its per-type
// sub-segments, loads the per-(target,type) row(s) once per sub-segment and the gathered source
// row once per message, and accumulates sequentially in registers (reference message order,
// no atomics).  Backward = one by-target pass (gradients of the per-(target,type) rows) plus one
// by-(source,type) pass (gradients of the gathered rows), both atomics-free as well.
//
// Lanes run across features (float4 per lane).  G lanes per node: 64 (D > 128) or 32/16/8,
// NCH float4 chunks per lane (D <= 1024).  Bound: HBM / Infinity-Cache bandwidth;
// algorithmic bytes: M*(4D+8) + S_nonempty*rowbytes + V*4D.
#include <type_traits>

#include "common.h"

using namespace relgnn;

namespace {

constexpr int KIND_FILM = 0;
constexpr int KIND_PAIR = 1;
constexpr int PU = 4;  // messages in flight per lane group

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

// The activation id is a wave-uniform RUNTIME value (one uniform branch per message) so that the kernel count stays at
// geometry x kind instead of x 7 activations.  Evaluated once per message and feature, so the v_exp / v_rcp based
// variants of common.h: with the library erff the GELU kernels were ALU-bound (Edge-MLP0 forward 681 us vs 249 us with ReLU).
template <int ACT>
__device__ __forceinline__ float4 act4_t(float4 x) {
  return make_float4(act_fwd_fast<ACT>(x.x), act_fwd_fast<ACT>(x.y), act_fwd_fast<ACT>(x.z), act_fwd_fast<ACT>(x.w));
}
template <int ACT>
__device__ __forceinline__ float4 actg4_t(float4 x) {
  return make_float4(act_grad_fast<ACT>(x.x), act_grad_fast<ACT>(x.y), act_grad_fast<ACT>(x.z), act_grad_fast<ACT>(x.w));
}
// Runs f(integral_constant<ACT>) for the runtime activation id: ONE wave-uniform switch per batch of messages, the
// batch's loop nest is then straight-line code of that activation (a switch per message and float4 put seven inlined
// activations between any two loads of the unrolled loop).
template <class F>
__device__ __forceinline__ void with_act(int act, F&& f) {
  switch (act) {
    case RELGNN_ACT_TANH: f(std::integral_constant<int, RELGNN_ACT_TANH>{}); break;
    case RELGNN_ACT_RELU: f(std::integral_constant<int, RELGNN_ACT_RELU>{}); break;
    case RELGNN_ACT_LEAKY_RELU: f(std::integral_constant<int, RELGNN_ACT_LEAKY_RELU>{}); break;
    case RELGNN_ACT_ELU: f(std::integral_constant<int, RELGNN_ACT_ELU>{}); break;
    case RELGNN_ACT_SELU: f(std::integral_constant<int, RELGNN_ACT_SELU>{}); break;
    case RELGNN_ACT_GELU: f(std::integral_constant<int, RELGNN_ACT_GELU>{}); break;
    default: f(std::integral_constant<int, RELGNN_ACT_LINEAR>{}); break;
  }
}
// per-call form (one switch per float4): the lane-group by-target backward keeps it — hoisting the switch there costs
// 12 VGPRs (96 -> 108), i.e. one resident wave per SIMD, and 20 % of the kernel at the C5 shape (675 -> 810 us)
__device__ __forceinline__ float4 actg4(int act, float4 x) {
  float4 r;
  with_act(act, [&](auto a_tag) { r = actg4_t<decltype(a_tag)::value>(x); });
  return r;
}
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// pre-activation of one message.  rowA/rowB: gamma/beta (FiLM) or q/unused (PAIR)
template <int KIND>
__device__ __forceinline__ float4 pre_act(float w, float4 t, float4 rowA, float4 rowB) {
  if constexpr (KIND == KIND_FILM) return rowA * (w * t) + rowB;  // gnn_film.py:100,108
  else return w * (t + rowA);                                      // gnn_edge_mlp.py:102-108
}

struct GroupGeom {
  int g, gl;          // lane group within the wave, lane within the group
  int64_t node;       // node / row owned by this group
  bool valid;
};

template <int G>
__device__ __forceinline__ GroupGeom geom(int64_t n_rows, int64_t n_logical_blocks) {
  GroupGeom r;
  const int64_t lb = xcd_logical_block(n_logical_blocks);
  const int lane = threadIdx.x & 63;
  r.g = lane / G;
  r.gl = lane % G;
  r.node = lb < 0 ? n_rows : (lb * 4 + (threadIdx.x >> 6)) * (64 / G) + r.g;
  r.valid = r.node < n_rows;
  return r;
}

// The L + 1 bucket boundaries of a node (and, with compact tables, the rows of its buckets) are fetched by the group's lanes
// with ONE coalesced load each and handed around with shuffles; the walk then visits the NON-EMPTY types only (a bit mask
// from the boundaries).  Walking all L types with a dependent rowptr load (and a brow load) per type made every node a chain
// of ~L + 2 x (non-empty types) serial L2 round trips — on VarMisuse-shaped graphs (23 types, ~7 non-empty per node) the
// kernels were latency-bound at 45 % of the HBM rate.  Needs L < G (else the per-type loads of the plain walk).
template <int G>
struct NodeBuckets {
  int rp;            // lane gl holds rowptr[v*L + gl] (gl <= L)
  int br;            // lane gl holds the row of bucket (v, gl) (gl < L)
  uint32_t mask;     // bit l set: bucket (v, l) is non-empty
  __device__ __forceinline__ int begin(int l) const { return __shfl(rp, l, G); }
  __device__ __forceinline__ int end(int l) const { return __shfl(rp, l + 1, G); }
  __device__ __forceinline__ int row(int l) const { return __shfl(br, l, G); }
};
template <int G>
__device__ __forceinline__ NodeBuckets<G> load_buckets(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ brow,
                                                       int64_t v, int L, int gl, int g) {
  NodeBuckets<G> nb;
  nb.rp = gl <= L ? rowptr[v * L + gl] : 0;
  nb.br = gl < L ? (brow ? brow[v * L + gl] : (int)(v * L + gl)) : -1;
  const int nxt = __shfl_down(nb.rp, 1, G);
  const uint64_t ball = __ballot(gl < L && nxt > nb.rp);
  nb.mask = (uint32_t)((ball >> (g * G)) & ((G >= 32) ? 0xffffffffull : ((1ull << G) - 1)));
  return nb;
}

// -----------------------------------------------------------------------------------------
// forward:  out[v] = finalize( AGG_l AGG_{p in (v,l)} act(pre_act(w[p], T[col[p]], rows(v,l))) )
// -----------------------------------------------------------------------------------------
template <int G, int NCH, int KIND, bool IS_MAX>
__global__ __launch_bounds__(256) void edge_fwd_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr, int32_t V, int32_t L, const int32_t* __restrict__ col,
    const float* __restrict__ w, int32_t mode, int32_t act, float4* __restrict__ out, int64_t ldo4, int64_t nlb,
    const int32_t* __restrict__ brow) {
  const GroupGeom gg = geom<G>(V, nlb);
  if (!gg.valid) return;
  const int64_t v = gg.node;
  bool on[NCH];
  int cc[NCH];
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = gg.gl + G * c < D4;
    cc[c] = min(gg.gl + G * c, D4 - 1);
    acc[c] = f4(IS_MAX ? -FLT_MAX : 0.f);
  }
  int seg_b, seg_e;
  auto bucket = [&](int b, int e, int64_t arow_id) {
    float4 ra[NCH], rb[NCH];
    const float4* arow = A + arow_id * lda4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      ra[c] = arow[cc[c]];
      rb[c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
    }
    for (int p = b; p < e; p += PU) {
      int r[PU];
      float ww[PU];
      float4 t[PU][NCH];
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int idx = min(p + u, e - 1);
        r[u] = col[idx];
        ww[u] = w ? w[idx] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < PU; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) t[u][c] = T[(int64_t)r[u] * ldt4 + cc[c]];
      with_act(act, [&](auto a_tag) {
        constexpr int ACT = decltype(a_tag)::value;
#pragma unroll
        for (int u = 0; u < PU; ++u)
          if (p + u < e) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const float4 m = act4_t<ACT>(pre_act<KIND>(ww[u], t[u][c], ra[c], rb[c]));
              acc[c] = IS_MAX ? max4(acc[c], m) : acc[c] + m;
            }
          }
      });
    }
  };
  if (G < 64 && L < G) {
    const NodeBuckets<G> nb = load_buckets<G>(rowptr, brow, v, L, gg.gl, gg.g);
    seg_b = nb.begin(0);
    seg_e = nb.end(L - 1);
    for (uint32_t m = nb.mask; m; m &= m - 1) {          // ascending type order = the reference's message order
      const int l = __builtin_ctz(m);
      bucket(nb.begin(l), nb.end(l), nb.row(l));
    }
  } else {
    seg_b = rowptr[v * L];
    seg_e = rowptr[(v + 1) * L];
    int b = seg_b;
    for (int l = 0; l < L; ++l) {
      const int e = rowptr[v * L + l + 1];
      if (b < e) bucket(b, e, brow ? (int64_t)brow[v * L + l] : v * L + l);
      b = e;
    }
  }
  const float n = (float)max(seg_e - seg_b, 1);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) {
      float4 a = acc[c];
      if (mode == RELGNN_AGG_MEAN) a = make_float4(a.x / n, a.y / n, a.z / n, a.w / n);
      if (mode == RELGNN_AGG_SQRT_N) { const float s = sqrtf(n); a = make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
      out[v * ldo4 + gg.gl + G * c] = a;
    }
}

// -----------------------------------------------------------------------------------------
// backward A (by-target): gradient of the per-(target,type) rows.
//   g_p = gagg[v] * act'(pre_p)
//   FiLM: gA[(v,l)] = [ sum_p g_p * (w_p T[col p]) | sum_p g_p ]       (d gamma | d beta)
//   PAIR: gA[(v,l)] =   sum_p g_p * w_p                                  (d q)
// Every (v,l) row is written (zeros for empty sub-segments).
// -----------------------------------------------------------------------------------------
template <int G, int NCH, int KIND>
__global__ __launch_bounds__(256) void edge_bwd_rows_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr, int32_t V, int32_t L, const int32_t* __restrict__ col,
    const float* __restrict__ w, const float4* __restrict__ gagg, int64_t ldg4, float4* __restrict__ gA,
    int64_t ldga4, int32_t act, int64_t nlb, const int32_t* __restrict__ brow, float4* __restrict__ dmsg) {
  const GroupGeom gg = geom<G>(V, nlb);
  if (!gg.valid) return;
  const int64_t v = gg.node;
  bool on[NCH];
  int cc[NCH];
  float4 g[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = gg.gl + G * c < D4;
    cc[c] = min(gg.gl + G * c, D4 - 1);
    g[c] = gagg[v * ldg4 + cc[c]];
  }
  // one bucket (v, l) = messages [b, e): its gradient row goes to `orow` (-1: an empty bucket of a compact table owns no row)
  auto bucket = [&](int b, int e, int64_t arow_id, int64_t orow) {
    float4 s1[NCH], s2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) s1[c] = s2[c] = f4(0.f);
    if (b < e) {
      float4 ra[NCH], rb[NCH];
      const float4* arow = A + arow_id * lda4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        ra[c] = arow[cc[c]];
        rb[c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
      }
      for (int p = b; p < e; p += PU) {
        int r[PU];
        float ww[PU];
        float4 t[PU][NCH];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
          const int idx = min(p + u, e - 1);
          r[u] = col[idx];
          ww[u] = w ? w[idx] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
          for (int c = 0; c < NCH; ++c) t[u][c] = T[(int64_t)r[u] * ldt4 + cc[c]];
#pragma unroll
        for (int u = 0; u < PU; ++u)
          if (p + u < e) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const float4 gp = g[c] * actg4(act, pre_act<KIND>(ww[u], t[u][c], ra[c], rb[c]));
              if constexpr (KIND == KIND_FILM) {
                s1[c] = s1[c] + gp * (ww[u] * t[u][c]);
                s2[c] = s2[c] + gp;
              } else {
                s1[c] = s1[c] + ww[u] * gp;
              }
              // gradient w.r.t. the gathered row of THIS message, for the by-source reduction that follows
              if (dmsg && on[c])
                dmsg[(int64_t)(p + u) * D4 + gg.gl + G * c] = (KIND == KIND_FILM) ? ww[u] * (ra[c] * gp) : ww[u] * gp;
            }
          }
      }
    }
    if (orow >= 0) {
      float4* grow = gA + orow * ldga4;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (on[c]) {
          grow[gg.gl + G * c] = s1[c];
          if constexpr (KIND == KIND_FILM) grow[D4 + gg.gl + G * c] = s2[c];
        }
    }
  };
  if (G < 64 && L < G) {
    const NodeBuckets<G> nb = load_buckets<G>(rowptr, brow, v, L, gg.gl, gg.g);
    if (brow) {                                            // compact tables: only non-empty buckets own a row
      for (uint32_t m = nb.mask; m; m &= m - 1) {
        const int l = __builtin_ctz(m);
        const int row = nb.row(l);
        bucket(nb.begin(l), nb.end(l), row, row);
      }
    } else {                                               // dense tables: every (v, l) row is written (zeros when empty)
      for (int l = 0; l < L; ++l) bucket(nb.begin(l), nb.end(l), v * L + l, v * L + l);
    }
  } else {
    int b = rowptr[v * L];
    for (int l = 0; l < L; ++l) {
      const int e = rowptr[v * L + l + 1];
      // compact row tables (brow): only non-empty buckets own a row
      const int64_t row = brow ? (b < e ? (int64_t)brow[v * L + l] : -1) : v * L + l;
      bucket(b, e, row, row);
      b = e;
    }
  }
}

// -----------------------------------------------------------------------------------------
// backward B (by-(source,type) rows r of T): gradient of the gathered rows.
//   gT[r] = sum_q dmsg_q,  g_q = gagg[tgt_b[q]] * act'(pre_q)
//   FiLM: dmsg = w_q * gamma[frow_b[q]] * g_q ;  PAIR: dmsg = w_q * g_q
// -----------------------------------------------------------------------------------------
template <int G, int NCH, int KIND>
__global__ __launch_bounds__(256) void edge_bwd_msgs_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr_b, int64_t n_rows, const int32_t* __restrict__ tgt_b,
    const int32_t* __restrict__ frow_b, const float* __restrict__ w_b, const float4* __restrict__ gagg,
    int64_t ldg4, float4* __restrict__ gT, int64_t ldgt4, int32_t act, int64_t nlb, const int32_t* __restrict__ trow) {
  const GroupGeom gg = geom<G>(n_rows, nlb);
  if (!gg.valid) return;
  const int64_t r = gg.node;
  const int b = rowptr_b[r], e = rowptr_b[r + 1];
  // compact row tables (trow): bucket r owns row trow[r] of T / gT, empty buckets own none
  const int64_t tr = trow ? (b < e ? (int64_t)trow[r] : -1) : r;
  if (tr < 0) return;
  bool on[NCH];
  int cc[NCH];
  float4 t[NCH], acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = gg.gl + G * c < D4;
    cc[c] = min(gg.gl + G * c, D4 - 1);
    t[c] = T[tr * ldt4 + cc[c]];
    acc[c] = f4(0.f);
  }
  for (int q = b; q < e; q += PU) {
    int fr[PU], tg[PU];
    float ww[PU];
    float4 ra[PU][NCH], rb[PU][NCH], g[PU][NCH];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = min(q + u, e - 1);
      fr[u] = frow_b[idx];
      tg[u] = tgt_b[idx];
      ww[u] = w_b ? w_b[idx] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < PU; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4* arow = A + (int64_t)fr[u] * lda4;
        ra[u][c] = arow[cc[c]];
        rb[u][c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
        g[u][c] = gagg[(int64_t)tg[u] * ldg4 + cc[c]];
      }
    with_act(act, [&](auto a_tag) {
      constexpr int ACT = decltype(a_tag)::value;
#pragma unroll
      for (int u = 0; u < PU; ++u)
        if (q + u < e) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const float4 gp = g[u][c] * actg4_t<ACT>(pre_act<KIND>(ww[u], t[c], ra[u][c], rb[u][c]));
            if constexpr (KIND == KIND_FILM) acc[c] = acc[c] + ww[u] * (ra[u][c] * gp);
            else acc[c] = acc[c] + ww[u] * gp;
          }
        }
    });
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) gT[tr * ldgt4 + gg.gl + G * c] = acc[c];
}

// =========================================================================================
// Wave-per-node variants (D > 128, L <= 63): same structure as seg_reduce_wave_kernel — the bucket
// bounds of the node live in ONE vector register (lane j holds rowptr[v*L + j]) and are broadcast with
// v_readlane; row indices of a 64-message chunk are fetched with one coalesced load and broadcast into
// SGPRs, so every gathered-row load is scalar base + lane offset, WU loads in flight.
// =========================================================================================
constexpr int WU = 8;

__device__ __forceinline__ float rlf(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int NCH, int KIND, bool IS_MAX>
__global__ __launch_bounds__(256) void edge_fwd_wave_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr, int32_t V, int32_t L, const int32_t* __restrict__ col,
    const float* __restrict__ w, int32_t mode, int32_t act, float4* __restrict__ out, int64_t ldo4, int64_t nlb,
    const int32_t* __restrict__ brow) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t v = lb * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int my_b = rowptr[v * L + min(lane, L)];
  bool on[NCH];
  uint32_t cc[NCH];
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = lane + 64 * c < D4;
    cc[c] = (uint32_t)min(lane + 64 * c, D4 - 1);
    acc[c] = f4(IS_MAX ? -FLT_MAX : 0.f);
  }
  const uint32_t ld = (uint32_t)ldt4;
  const int seg_b = __builtin_amdgcn_readlane(my_b, 0);
  int b = seg_b;
  for (int l = 0; l < L; ++l) {
    const int e = __builtin_amdgcn_readlane(my_b, l + 1);
    if (b < e) {
      float4 ra[NCH], rb[NCH];
      const float4* arow = A + (brow ? (int64_t)brow[v * L + l] : v * L + l) * lda4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        ra[c] = arow[cc[c]];
        rb[c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
      }
      for (int p = b; p < e; p += 64) {
        const int n = min(64, e - p);
        const int my_col = (lane < n) ? col[p + lane] : 0;
        const float my_w = (w && lane < n) ? w[p + lane] : 1.f;
        for (int k = 0; k < n; k += WU) {
          const int rem = n - k;  // wave-uniform
          float4 t[WU][NCH];
          float ww[WU];
#pragma unroll
          for (int u = 0; u < WU; ++u) {
            const int ku = k + min(u, rem - 1);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, ku);
            ww[u] = rlf(my_w, ku);
            const float4* row = T + (size_t)(r * ld);
            if (u < rem) {
#pragma unroll
              for (int c = 0; c < NCH; ++c) t[u][c] = row[cc[c]];
            }
          }
          with_act(act, [&](auto a_tag) {
            constexpr int ACT = decltype(a_tag)::value;
#pragma unroll
            for (int u = 0; u < WU; ++u)
              if (u < rem) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                  const float4 m = act4_t<ACT>(pre_act<KIND>(ww[u], t[u][c], ra[c], rb[c]));
                  acc[c] = IS_MAX ? max4(acc[c], m) : acc[c] + m;
                }
              }
          });
        }
      }
    }
    b = e;
  }
  const float n = (float)max(b - seg_b, 1);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) {
      float4 a = acc[c];
      if (mode == RELGNN_AGG_MEAN) a = make_float4(a.x / n, a.y / n, a.z / n, a.w / n);
      if (mode == RELGNN_AGG_SQRT_N) { const float s = sqrtf(n); a = make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
      out[v * ldo4 + lane + 64 * c] = a;
    }
}

template <int NCH, int KIND>
__global__ __launch_bounds__(256) void edge_bwd_rows_wave_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr, int32_t V, int32_t L, const int32_t* __restrict__ col,
    const float* __restrict__ w, const float4* __restrict__ gagg, int64_t ldg4, float4* __restrict__ gA,
    int64_t ldga4, int32_t act, int64_t nlb, const int32_t* __restrict__ brow, float4* __restrict__ dmsg,
    unsigned long long* __restrict__ smask) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t v = lb * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int my_b = rowptr[v * L + min(lane, L)];
  bool on[NCH];
  uint32_t cc[NCH];
  float4 g[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = lane + 64 * c < D4;
    cc[c] = (uint32_t)min(lane + 64 * c, D4 - 1);
    g[c] = gagg[v * ldg4 + cc[c]];
  }
  const uint32_t ld = (uint32_t)ldt4;
  int b = __builtin_amdgcn_readlane(my_b, 0);
  for (int l = 0; l < L; ++l) {
    const int e = __builtin_amdgcn_readlane(my_b, l + 1);
    float4 s1[NCH], s2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) s1[c] = s2[c] = f4(0.f);
    if (b < e) {
      float4 ra[NCH], rb[NCH];
      const float4* arow = A + (brow ? (int64_t)brow[v * L + l] : v * L + l) * lda4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        ra[c] = arow[cc[c]];
        rb[c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
      }
      for (int p = b; p < e; p += 64) {
        const int n = min(64, e - p);
        const int my_col = (lane < n) ? col[p + lane] : 0;
        const float my_w = (w && lane < n) ? w[p + lane] : 1.f;
        for (int k = 0; k < n; k += WU) {
          const int rem = n - k;
          float4 t[WU][NCH];
          float ww[WU];
#pragma unroll
          for (int u = 0; u < WU; ++u) {
            const int ku = k + min(u, rem - 1);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, ku);
            ww[u] = rlf(my_w, ku);
            const float4* row = T + (size_t)(r * ld);
            if (u < rem) {
#pragma unroll
              for (int c = 0; c < NCH; ++c) t[u][c] = row[cc[c]];
            }
          }
          with_act(act, [&](auto a_tag) {
            constexpr int ACT = decltype(a_tag)::value;
#pragma unroll
            for (int u = 0; u < WU; ++u)
              if (u < rem) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                  const float4 pre = pre_act<KIND>(ww[u], t[u][c], ra[c], rb[c]);
                  const float4 gp = g[c] * actg4_t<ACT>(pre);
                  if constexpr (KIND == KIND_FILM) {
                    s1[c] = s1[c] + gp * (ww[u] * t[u][c]);
                    s2[c] = s2[c] + gp;
                  } else {
                    s1[c] = s1[c] + ww[u] * gp;
                  }
                  if (dmsg && on[c])
                    dmsg[(int64_t)(p + k + u) * D4 + lane + 64 * c] = (KIND == KIND_FILM) ? ww[u] * (ra[c] * gp) : ww[u] * gp;
                  if constexpr (NCH == 1) {
                    // sign of the pre-activation, one bit per feature: word j = ballot over the lanes of component j.
                    // All the by-source pass needs of this message when the activation is piecewise linear.
                    if (smask) {
                      const unsigned long long b0 = __ballot(pre.x > 0.f), b1 = __ballot(pre.y > 0.f);
                      const unsigned long long b2 = __ballot(pre.z > 0.f), b3 = __ballot(pre.w > 0.f);
                      const unsigned long long word = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
                      if (lane < 4) smask[(int64_t)(p + k + u) * 4 + lane] = word;
                    }
                  }
                }
              }
          });
        }
      }
    }
    const int64_t orow = brow ? (b < e ? (int64_t)brow[v * L + l] : -1) : v * L + l;
    if (orow >= 0) {
      float4* grow = gA + orow * ldga4;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (on[c]) {
          grow[lane + 64 * c] = s1[c];
          if constexpr (KIND == KIND_FILM) grow[D4 + lane + 64 * c] = s2[c];
        }
    }
    b = e;
  }
}

// by-(source,type) rows: one wave per row r of T; per message the bucket row A[frow] and the target's gagg row
// are gathered (scalar bases), MU messages (2-3 row loads each) in flight.
template <int NCH, int KIND, int MU>
__global__ __launch_bounds__(256) void edge_bwd_msgs_wave_kernel(
    const float4* __restrict__ T, int64_t ldt4, const float4* __restrict__ A, int64_t lda4, int32_t D4,
    const int32_t* __restrict__ rowptr_b, int64_t n_rows, const int32_t* __restrict__ tgt_b,
    const int32_t* __restrict__ frow_b, const float* __restrict__ w_b, const float4* __restrict__ gagg,
    int64_t ldg4, float4* __restrict__ gT, int64_t ldgt4, int32_t act, int64_t nlb, const int32_t* __restrict__ trow) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t r = lb * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int b = __builtin_amdgcn_readfirstlane(rowptr_b[r]);
  const int e = __builtin_amdgcn_readfirstlane(rowptr_b[r + 1]);
  const int64_t tr = trow ? (b < e ? (int64_t)trow[r] : -1) : r;
  if (tr < 0) return;
  bool on[NCH];
  uint32_t cc[NCH];
  float4 t[NCH], acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = lane + 64 * c < D4;
    cc[c] = (uint32_t)min(lane + 64 * c, D4 - 1);
    t[c] = T[tr * ldt4 + cc[c]];
    acc[c] = f4(0.f);
  }
  const uint32_t lda = (uint32_t)lda4, ldg = (uint32_t)ldg4;
  for (int q = b; q < e; q += 64) {
    const int n = min(64, e - q);
    const int my_fr = (lane < n) ? frow_b[q + lane] : 0;
    const int my_tg = (lane < n) ? tgt_b[q + lane] : 0;
    const float my_w = (w_b && lane < n) ? w_b[q + lane] : 1.f;
    for (int k = 0; k < n; k += MU) {
      const int rem = n - k;
      float4 ra[MU][NCH], rb[MU][NCH], g[MU][NCH];
      float ww[MU];
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int ku = k + min(u, rem - 1);
        const uint32_t fr = (uint32_t)__builtin_amdgcn_readlane(my_fr, ku);
        const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane(my_tg, ku);
        ww[u] = rlf(my_w, ku);
        const float4* arow = A + (size_t)(fr * lda);
        const float4* grow = gagg + (size_t)(tg * ldg);
        if (u < rem) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            ra[u][c] = arow[cc[c]];
            rb[u][c] = (KIND == KIND_FILM) ? arow[D4 + cc[c]] : f4(0.f);
            g[u][c] = grow[cc[c]];
          }
        }
      }
      with_act(act, [&](auto a_tag) {
        constexpr int ACT = decltype(a_tag)::value;
#pragma unroll
        for (int u = 0; u < MU; ++u)
          if (u < rem) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const float4 gp = g[u][c] * actg4_t<ACT>(pre_act<KIND>(ww[u], t[c], ra[u][c], rb[u][c]));
              if constexpr (KIND == KIND_FILM) acc[c] = acc[c] + ww[u] * (ra[u][c] * gp);
              else acc[c] = acc[c] + ww[u] * gp;
            }
          }
      });
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) gT[tr * ldgt4 + lane + 64 * c] = acc[c];
}

// by-(source,type) rows for PIECEWISE-LINEAR activations (linear / ReLU / leaky ReLU), D <= 256: the derivative of a
// message is one bit per feature, written by the by-target pass (smask, by-target order; pos_b[q] = by-target position of
// by-source message q).  Per message this pass gathers gamma (FiLM; nothing for PAIR), the target's gradient row and 32
// bytes of mask — not beta, not the pre-activation: 2 KiB instead of 3 KiB per message for FiLM at D = 256 (the
// recomputing pass moves 3.2 GB per launch at the C2 shape and runs at the memory system's 6 TB/s).
template <int KIND>
__global__ __launch_bounds__(256) void edge_bwd_msgs_mask_wave_kernel(
    const float4* __restrict__ A, int64_t lda4, int32_t D4, const int32_t* __restrict__ rowptr_b, int64_t n_rows,
    const int32_t* __restrict__ tgt_b, const int32_t* __restrict__ frow_b, const float* __restrict__ w_b,
    const int32_t* __restrict__ pos_b, const unsigned long long* __restrict__ smask, float neg_slope,
    const float4* __restrict__ gagg, int64_t ldg4, float4* __restrict__ gT, int64_t ldgt4, int64_t nlb) {
  constexpr int MU = 8;
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t r = lb * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int b = __builtin_amdgcn_readfirstlane(rowptr_b[r]);
  const int e = __builtin_amdgcn_readfirstlane(rowptr_b[r + 1]);
  const bool on = lane < D4;
  const uint32_t cc = (uint32_t)min(lane, D4 - 1);
  float4 acc = f4(0.f);
  const uint32_t lda = (uint32_t)lda4, ldg = (uint32_t)ldg4;
  for (int q = b; q < e; q += 64) {
    const int n = min(64, e - q);
    const int my_fr = (KIND == KIND_FILM && lane < n) ? frow_b[q + lane] : 0;
    const int my_tg = (lane < n) ? tgt_b[q + lane] : 0;
    const int my_pos = (lane < n) ? pos_b[q + lane] : 0;
    const float my_w = (w_b && lane < n) ? w_b[q + lane] : 1.f;
    for (int k = 0; k < n; k += MU) {
      const int rem = n - k;
      float4 ra[MU], g[MU];
      unsigned long long m[MU][4];      // wave-uniform addresses: scalar loads into SGPRs
      float ww[MU];
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int ku = k + min(u, rem - 1);
        const uint32_t fr = (uint32_t)__builtin_amdgcn_readlane(my_fr, ku);
        const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane(my_tg, ku);
        const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane(my_pos, ku);
        ww[u] = rlf(my_w, ku);
        if (u < rem) {
          ra[u] = (KIND == KIND_FILM) ? A[(size_t)(fr * lda) + cc] : f4(1.f);
          g[u] = gagg[(size_t)(tg * ldg) + cc];
          const unsigned long long* mw = smask + (size_t)ps * 4;   // the same 32 bytes for every lane
          m[u][0] = mw[0]; m[u][1] = mw[1]; m[u][2] = mw[2]; m[u][3] = mw[3];
        }
      }
#pragma unroll
      for (int u = 0; u < MU; ++u)
        if (u < rem) {
          float4 d;
          d.x = ((m[u][0] >> lane) & 1ull) ? 1.f : neg_slope;
          d.y = ((m[u][1] >> lane) & 1ull) ? 1.f : neg_slope;
          d.z = ((m[u][2] >> lane) & 1ull) ? 1.f : neg_slope;
          d.w = ((m[u][3] >> lane) & 1ull) ? 1.f : neg_slope;
          const float4 gp = g[u] * d;
          if constexpr (KIND == KIND_FILM) acc = acc + ww[u] * (ra[u] * gp);
          else acc = acc + ww[u] * gp;
        }
    }
  }
  if (on) gT[r * ldgt4 + lane] = acc;
}

// -----------------------------------------------------------------------------------------
// materialise per-message hidden states in the ORIGINAL type-major message order:
//   hidden[m] = act( P[row_src[m]] + Q[row_tgt[m]] )     (rows = node*L + type)
// (first Dense of an edge MLP on [h_u || h_v], utils/utils.py:120-126; the per-type dense layers
//  that follow are genuinely per-edge GEMMs on contiguous [E_l, D] blocks.)
// and its gradient w.r.t. the pre-activation: gpre[m] = ghidden[m] * act'(P[..] + Q[..]).
// -----------------------------------------------------------------------------------------
template <bool GRAD>
__global__ __launch_bounds__(256) void pair_materialize_kernel(int32_t act, 
    const float4* __restrict__ P, int64_t ldp4, const float4* __restrict__ Q, int64_t ldq4, int32_t D4,
    const int32_t* __restrict__ row_src, const int32_t* __restrict__ row_tgt, int64_t M,
    const float4* __restrict__ ghidden, float4* __restrict__ out, int64_t ldo4) {
  const int64_t total = M * D4;
  with_act(act, [&](auto a_tag) {
    constexpr int ACT = decltype(a_tag)::value;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t m = i / D4;
      const int c = (int)(i - m * D4);
      float4 pre = P[(int64_t)row_src[m] * ldp4 + c];
      if (Q) pre = pre + Q[(int64_t)row_tgt[m] * ldq4 + c];
      if constexpr (GRAD) out[m * ldo4 + c] = ghidden[m * ldo4 + c] * actg4_t<ACT>(pre);
      else out[m * ldo4 + c] = act4_t<ACT>(pre);
    }
  });
}

// ---- launch helpers ---------------------------------------------------------------------------
struct Geo { int G, NCH; };
inline bool pick_geo(int D, Geo* g) {
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
inline int64_t logical_blocks(int64_t rows, int G) { return (rows + 4 * (64 / G) - 1) / (4 * (64 / G)); }
inline unsigned padded_grid(int64_t nlb) { return (unsigned)(((nlb + 7) / 8) * 8); }

#define RELGNN_DISPATCH_GEO(geo, GG, NN, ...)                                        \
  if (geo.G == 8) { constexpr int GG = 8, NN = 1; __VA_ARGS__; }                      \
  else if (geo.G == 16) { constexpr int GG = 16, NN = 1; __VA_ARGS__; }               \
  else if (geo.G == 32) { constexpr int GG = 32, NN = 1; __VA_ARGS__; }               \
  else if (geo.NCH == 1) { constexpr int GG = 64, NN = 1; __VA_ARGS__; }              \
  else if (geo.NCH == 2) { constexpr int GG = 64, NN = 2; __VA_ARGS__; }              \
  else { constexpr int GG = 64, NN = 4; __VA_ARGS__; }

inline bool vec_ok(const void* p, int64_t ld) { return aligned16(p) && ld % 4 == 0; }

template <int KIND>
int launch_fwd(int32_t mode, int32_t act, const float* T, int64_t ldt, const float* A, int64_t lda,
               int32_t D, const int32_t* rowptr, int32_t V, int32_t L, const int32_t* col,
               const float* w, float* out, int64_t ldo, hipStream_t st, const int32_t* brow = nullptr) {
  Geo geo;
  if (!pick_geo(D, &geo) || !vec_ok(T, ldt) || !vec_ok(A, lda) || !vec_ok(out, ldo)) return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(V, geo.G);
  const unsigned grid = padded_grid(nlb);
  const bool is_max = mode == RELGNN_AGG_MAX;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (geo.G == 64 && L <= 63 && (int64_t)V * L * (ldt / 4) < ((int64_t)1 << 32)) {
    const int64_t wnlb = ((int64_t)V + 3) / 4;
    const unsigned wgrid = padded_grid(wnlb);
#define EDGE_FWD_WAVE(NN, MX)                                                                                      \
  edge_fwd_wave_kernel<NN, KIND, MX><<<wgrid, 256, 0, st>>>((const float4*)T, ldt / 4, (const float4*)A, lda / 4,   \
                                                             D / 4, rowptr, V, L, col, w, mode, act, (float4*)out,  \
                                                             ldo / 4, wnlb, brow)
    if (geo.NCH == 1) { if (is_max) EDGE_FWD_WAVE(1, true); else EDGE_FWD_WAVE(1, false); }
    else if (geo.NCH == 2) { if (is_max) EDGE_FWD_WAVE(2, true); else EDGE_FWD_WAVE(2, false); }
    else { if (is_max) EDGE_FWD_WAVE(4, true); else EDGE_FWD_WAVE(4, false); }
#undef EDGE_FWD_WAVE
    return launch_status();
  }
  RELGNN_DISPATCH_GEO(geo, GG, NN, {
    if (is_max)
      edge_fwd_kernel<GG, NN, KIND, true><<<grid, 256, 0, st>>>(
          (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr, V, L, col, w, mode, act, (float4*)out, ldo / 4, nlb, brow);
    else
      edge_fwd_kernel<GG, NN, KIND, false><<<grid, 256, 0, st>>>(
          (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr, V, L, col, w, mode, act, (float4*)out, ldo / 4, nlb, brow);
  });
  return launch_status();
}

template <int KIND>
int launch_bwd_rows(int32_t act, const float* T, int64_t ldt, const float* A, int64_t lda, int32_t D,
                    const int32_t* rowptr, int32_t V, int32_t L, const int32_t* col, const float* w,
                    const float* gagg, int64_t ldg, float* gA, int64_t ldga, hipStream_t st,
                    const int32_t* brow = nullptr, float* dmsg = nullptr, unsigned long long* smask = nullptr) {
  Geo geo;
  if (!pick_geo(D, &geo) || !vec_ok(T, ldt) || !vec_ok(A, lda) || !vec_ok(gagg, ldg) || !vec_ok(gA, ldga) ||
      !aligned16(dmsg))
    return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(V, geo.G);
  const unsigned grid = padded_grid(nlb);
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (geo.G == 64 && L <= 63 && (int64_t)V * L * (ldt / 4) < ((int64_t)1 << 32)) {
    const int64_t wnlb = ((int64_t)V + 3) / 4;
#define EDGE_ROWS_WAVE(NN)                                                                                          \
  edge_bwd_rows_wave_kernel<NN, KIND><<<padded_grid(wnlb), 256, 0, st>>>(                                            \
      (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr, V, L, col, w, (const float4*)gagg, ldg / 4, \
      (float4*)gA, ldga / 4, act, wnlb, brow, (float4*)dmsg, smask)
    if (smask && geo.NCH != 1) return RELGNN_EUNSUPPORTED;
    if (geo.NCH == 1) EDGE_ROWS_WAVE(1); else if (geo.NCH == 2) EDGE_ROWS_WAVE(2); else EDGE_ROWS_WAVE(4);
#undef EDGE_ROWS_WAVE
    return launch_status();
  }
  if (smask) return RELGNN_EUNSUPPORTED;      // sign masks: wave kernels with one float4 per lane only (128 < D <= 256)
  RELGNN_DISPATCH_GEO(geo, GG, NN, {
    edge_bwd_rows_kernel<GG, NN, KIND><<<grid, 256, 0, st>>>(
        (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr, V, L, col, w, (const float4*)gagg, ldg / 4,
        (float4*)gA, ldga / 4, act, nlb, brow, (float4*)dmsg);
  });
  return launch_status();
}

template <int KIND>
int launch_bwd_msgs(int32_t act, const float* T, int64_t ldt, const float* A, int64_t lda, int32_t D,
                    const int32_t* rowptr_b, int64_t n_rows, const int32_t* tgt_b, const int32_t* frow_b,
                    const float* w_b, const float* gagg, int64_t ldg, float* gT, int64_t ldgt, hipStream_t st,
                    const int32_t* trow = nullptr) {
  Geo geo;
  if (!pick_geo(D, &geo) || !vec_ok(T, ldt) || !vec_ok(A, lda) || !vec_ok(gagg, ldg) || !vec_ok(gT, ldgt))
    return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(n_rows, geo.G);
  const unsigned grid = padded_grid(nlb);
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (geo.G == 64 && n_rows * (lda / 4) < ((int64_t)1 << 32) && n_rows * (ldg / 4) < ((int64_t)1 << 32)) {
    const int64_t wnlb = (n_rows + 3) / 4;
    // messages in flight per wave: RELGNN_EDGE_MSGS_MU = 4 (default) | 8.  Measured at the C2 shape: 8 does not pay —
    // the extra registers cost as many resident waves as the unroll adds loads (FiLM 7.51 vs 7.52 ms per step,
    // pair messages 9.02 vs 8.77 ms).
    static const int mu_env = [] { const char* e = getenv("RELGNN_EDGE_MSGS_MU"); return e ? atoi(e) : 0; }();
    const int mu = mu_env == 8 ? 8 : 4;
#define EDGE_MSGS_WAVE(NN)                                                                                          \
  do {                                                                                                              \
    if (mu == 8)                                                                                                    \
      edge_bwd_msgs_wave_kernel<NN, KIND, 8><<<padded_grid(wnlb), 256, 0, st>>>(                                     \
          (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr_b, n_rows, tgt_b, frow_b, w_b,         \
          (const float4*)gagg, ldg / 4, (float4*)gT, ldgt / 4, act, wnlb, trow);                                     \
    else                                                                                                            \
      edge_bwd_msgs_wave_kernel<NN, KIND, 4><<<padded_grid(wnlb), 256, 0, st>>>(                                     \
          (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr_b, n_rows, tgt_b, frow_b, w_b,         \
          (const float4*)gagg, ldg / 4, (float4*)gT, ldgt / 4, act, wnlb, trow);                                     \
  } while (0)
    if (geo.NCH == 1) EDGE_MSGS_WAVE(1); else if (geo.NCH == 2) EDGE_MSGS_WAVE(2); else EDGE_MSGS_WAVE(4);
#undef EDGE_MSGS_WAVE
    return launch_status();
  }
  RELGNN_DISPATCH_GEO(geo, GG, NN, {
    edge_bwd_msgs_kernel<GG, NN, KIND><<<grid, 256, 0, st>>>(
        (const float4*)T, ldt / 4, (const float4*)A, lda / 4, D / 4, rowptr_b, n_rows, tgt_b, frow_b, w_b,
        (const float4*)gagg, ldg / 4, (float4*)gT, ldgt / 4, act, nlb, trow);
  });
  return launch_status();
}

inline bool bad_common(int32_t D, int32_t V, int32_t L) { return D < 0 || V < 0 || L <= 0; }

}  // namespace

extern "C" {

int relgnn_film_fwd(int32_t mode, int32_t act, const float* T, int64_t ldt, const float* film,
                    int64_t ldf, int32_t D, const int32_t* rowptr, int32_t num_nodes,
                    int32_t num_edge_types, const int32_t* col, const float* w, float* out, int64_t ldo,
                    const int32_t* bucket_row, void* stream) {
  if (bad_common(D, num_nodes, num_edge_types) || mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_MAX || ldf < 2 * D)
    return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out) return RELGNN_EINVAL;
  return launch_fwd<KIND_FILM>(mode, act, T, ldt, film, ldf, D, rowptr, num_nodes, num_edge_types, col, w, out, ldo, as_stream(stream), bucket_row);
}

int relgnn_film_bwd_film(int32_t act, const float* T, int64_t ldt, const float* film, int64_t ldf,
                         int32_t D, const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types,
                         const int32_t* col, const float* w, const float* gagg, int64_t ldg, float* gfilm,
                         int64_t ldgf, const int32_t* bucket_row, float* dmsg, void* sign_mask, void* stream) {
  if (bad_common(D, num_nodes, num_edge_types) || ldf < 2 * D || ldgf < 2 * D) return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !gagg || !gfilm) return RELGNN_EINVAL;
  if (sign_mask && (!aligned16(sign_mask) || bucket_row)) return RELGNN_EUNSUPPORTED;
  return launch_bwd_rows<KIND_FILM>(act, T, ldt, film, ldf, D, rowptr, num_nodes, num_edge_types, col, w, gagg, ldg, gfilm, ldgf, as_stream(stream), bucket_row, dmsg,
                                    static_cast<unsigned long long*>(sign_mask));
}

int relgnn_film_bwd_msg_masked(int32_t act, const float* film, int64_t ldf, int32_t D, const int32_t* rowptr_b,
                               int64_t num_rows_t, const int32_t* tgt_b, const int32_t* frow_b, const float* w_b,
                               const int32_t* pos_b, const void* sign_mask, const float* gagg, int64_t ldg, float* gT,
                               int64_t ldgt, void* stream) {
  if (D < 0 || num_rows_t < 0 || ldf < 2 * D) return RELGNN_EINVAL;
  if (num_rows_t == 0 || D == 0) return RELGNN_OK;
  if (!rowptr_b || !film || !gT || !gagg || !tgt_b || !frow_b || !pos_b || !sign_mask) return RELGNN_EINVAL;
  float neg_slope;
  if (act == RELGNN_ACT_LINEAR) neg_slope = 1.f;
  else if (act == RELGNN_ACT_RELU) neg_slope = 0.f;
  else if (act == RELGNN_ACT_LEAKY_RELU) neg_slope = 0.2f;
  else return RELGNN_EUNSUPPORTED;
  if (D % 4 != 0 || D > 256 || !vec_ok(film, ldf) || !vec_ok(gagg, ldg) || !vec_ok(gT, ldgt) || !aligned16(sign_mask) ||
      num_rows_t * (ldf / 4) >= ((int64_t)1 << 32) || num_rows_t * (ldg / 4) >= ((int64_t)1 << 32))
    return RELGNN_EUNSUPPORTED;
  const int64_t wnlb = (num_rows_t + 3) / 4;
  edge_bwd_msgs_mask_wave_kernel<KIND_FILM><<<padded_grid(wnlb), 256, 0, as_stream(stream)>>>(
      (const float4*)film, ldf / 4, D / 4, rowptr_b, num_rows_t, tgt_b, frow_b, w_b, pos_b,
      static_cast<const unsigned long long*>(sign_mask), neg_slope, (const float4*)gagg, ldg / 4, (float4*)gT, ldgt / 4, wnlb);
  return launch_status();
}

int relgnn_film_bwd_msg(int32_t act, const float* T, int64_t ldt, const float* film, int64_t ldf,
                        int32_t D, const int32_t* rowptr_b, int64_t num_rows_t, const int32_t* tgt_b,
                        const int32_t* frow_b, const float* w_b, const float* gagg, int64_t ldg, float* gT,
                        int64_t ldgt, const int32_t* bucket_row_b, void* stream) {
  if (D < 0 || num_rows_t < 0 || ldf < 2 * D) return RELGNN_EINVAL;
  if (num_rows_t == 0 || D == 0) return RELGNN_OK;
  if (!rowptr_b || !T || !gT) return RELGNN_EINVAL;
  return launch_bwd_msgs<KIND_FILM>(act, T, ldt, film, ldf, D, rowptr_b, num_rows_t, tgt_b, frow_b, w_b, gagg, ldg, gT, ldgt, as_stream(stream), bucket_row_b);
}

int relgnn_pair_fwd(int32_t mode, int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq,
                    int32_t D, const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types,
                    const int32_t* col, const float* w, float* out, int64_t ldo, void* stream) {
  if (bad_common(D, num_nodes, num_edge_types) || mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_MAX) return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out) return RELGNN_EINVAL;
  return launch_fwd<KIND_PAIR>(mode, act, P, ldp, Q, ldq, D, rowptr, num_nodes, num_edge_types, col, w, out, ldo, as_stream(stream));
}

int relgnn_pair_bwd_q(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq, int32_t D,
                      const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types, const int32_t* col,
                      const float* w, const float* gagg, int64_t ldg, float* gQ, int64_t ldgq, float* dmsg,
                      void* stream) {
  if (bad_common(D, num_nodes, num_edge_types)) return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !gagg || !gQ) return RELGNN_EINVAL;
  return launch_bwd_rows<KIND_PAIR>(act, P, ldp, Q, ldq, D, rowptr, num_nodes, num_edge_types, col, w, gagg, ldg, gQ, ldgq, as_stream(stream), nullptr, dmsg);
}

int relgnn_pair_bwd_p(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq, int32_t D,
                      const int32_t* rowptr_b, int64_t num_rows_p, const int32_t* tgt_b, const int32_t* frow_b,
                      const float* w_b, const float* gagg, int64_t ldg, float* gP, int64_t ldgp, void* stream) {
  if (D < 0 || num_rows_p < 0) return RELGNN_EINVAL;
  if (num_rows_p == 0 || D == 0) return RELGNN_OK;
  if (!rowptr_b || !P || !gP) return RELGNN_EINVAL;
  return launch_bwd_msgs<KIND_PAIR>(act, P, ldp, Q, ldq, D, rowptr_b, num_rows_p, tgt_b, frow_b, w_b, gagg, ldg, gP, ldgp, as_stream(stream));
}

int relgnn_pair_materialize(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq, int32_t D,
                            const int32_t* row_src, const int32_t* row_tgt, int64_t num_messages,
                            const float* ghidden, float* out, int64_t ldo, void* stream) {
  if (D < 0 || num_messages < 0) return RELGNN_EINVAL;
  if (num_messages == 0 || D == 0) return RELGNN_OK;
  if (!P || !row_src || !out || (Q && !row_tgt)) return RELGNN_EINVAL;
  if (D % 4 != 0 || !vec_ok(P, ldp) || (Q && !vec_ok(Q, ldq)) || !vec_ok(out, ldo) || (ghidden && !aligned16(ghidden)))
    return RELGNN_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  const unsigned grid = flat_grid(num_messages * (D / 4), 256);
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (ghidden)
    pair_materialize_kernel<true><<<grid, 256, 0, st>>>(act, (const float4*)P, ldp / 4, (const float4*)Q, ldq / 4, D / 4, row_src,
                                                        row_tgt, num_messages, (const float4*)ghidden, (float4*)out, ldo / 4);
  else
    pair_materialize_kernel<false><<<grid, 256, 0, st>>>(act, (const float4*)P, ldp / 4, (const float4*)Q, ldq / 4, D / 4, row_src,
                                                         row_tgt, num_messages, nullptr, (float4*)out, ldo / 4);
  return launch_status();
}

}  // extern "C"
