// RGAT: segmented softmax attention over ALL incoming messages of a target node.
// Replaces gnns/rgat.py:98-136 of the reference:
//   two embedding_lookups of the transformed states, reshape/concat/einsum + leaky_relu (rgat.py:98-115),
//   per head: dpu_utils unsorted_segment_log_softmax (5 TF ops) + exp (rgat.py:126-130),
//   multiply + unsorted_segment_sum (rgat.py:131-136), concat over heads (rgat.py:138)
// by one by-target kernel.  The attention logit decomposes into per-node scalars
//   a_l[k] . [T_l[u]_k || T_l[v]_k] = s_src[u*L+l, k] + s_tgt[v*L+l, k]
// which the caller computes node-side (a [V*L, K] table each); the kernel needs only K floats per
// message for the two softmax passes and gathers the 4*D-byte source row once, in the third pass.
//
//   e[p,k]   = leaky_relu_slope(s_src[col[p],k] + s_tgt[v*L+l(p),k])
//   a[p,k]   = exp( (e - max_p e) - log(sum_p exp(e - max_p e)) )      (the reference's formula)
//   out[v,k] = sum_p a[p,k] * T[col[p], head k]                        (sequential in message order)
//
// Lanes run across features (float4 per lane, a float4 never straddles two heads: Dh % 4 == 0).
#include "common.h"

using namespace relgnn;

namespace {

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float lrelu(float z, float slope) { return z > 0.f ? z : slope * z; }

template <int G>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, G);
  return x;
}

struct Geom {
  int gl;
  int64_t node;
  bool valid;
};
template <int G>
__device__ __forceinline__ Geom geom(int64_t n_rows, int64_t nlb) {
  Geom r;
  const int64_t lb = xcd_logical_block(nlb);
  const int lane = threadIdx.x & 63;
  r.gl = lane % G;
  r.node = lb < 0 ? n_rows : (lb * 4 + (threadIdx.x >> 6)) * (64 / G) + lane / G;
  r.valid = r.node < n_rows;
  return r;
}

template <int G, int NCH>
__global__ __launch_bounds__(256) void rgat_fwd_kernel(
    const float4* __restrict__ T, int64_t ldt4, int32_t D4, int32_t K, int32_t Dh4,
    const float* __restrict__ s_src, const float* __restrict__ s_tgt, const int32_t* __restrict__ rowptr,
    int32_t V, int32_t L, const int32_t* __restrict__ col, float slope, float4* __restrict__ out,
    int64_t ldo4, float* __restrict__ alpha, int64_t nlb) {
  const Geom gg = geom<G>(V, nlb);
  if (!gg.valid) return;
  const int64_t v = gg.node;
  bool on[NCH], leader[NCH];
  int cc[NCH], head[NCH];
  float mx[NCH], sm[NCH], lse[NCH];
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ci = gg.gl + G * c;
    on[c] = ci < D4;
    cc[c] = min(ci, D4 - 1);
    head[c] = cc[c] / Dh4;
    leader[c] = on[c] && (cc[c] % Dh4 == 0);
    mx[c] = -FLT_MAX;
    sm[c] = 0.f;
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int seg_b = rowptr[v * L];
  // pass 1: per-head maximum of the logits (unsorted_segment_max)
  int b = seg_b;
  for (int l = 0; l < L; ++l) {
    const int e = rowptr[v * L + l + 1];
    if (b < e) {
      float st[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) st[c] = s_tgt[(v * L + l) * K + head[c]];
      for (int p = b; p < e; ++p) {
        const int64_t r = col[p];
#pragma unroll
        for (int c = 0; c < NCH; ++c) mx[c] = fmaxf(mx[c], lrelu(s_src[r * K + head[c]] + st[c], slope));
      }
    }
    b = e;
  }
  // pass 2: sum of exp(e - max), sequential in message order (unsorted_segment_sum)
  b = seg_b;
  for (int l = 0; l < L; ++l) {
    const int e = rowptr[v * L + l + 1];
    if (b < e) {
      float st[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) st[c] = s_tgt[(v * L + l) * K + head[c]];
      for (int p = b; p < e; ++p) {
        const int64_t r = col[p];
#pragma unroll
        for (int c = 0; c < NCH; ++c) sm[c] = sm[c] + expf(lrelu(s_src[r * K + head[c]] + st[c], slope) - mx[c]);
      }
    }
    b = e;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) lse[c] = logf(sm[c]);
  // pass 3: attention-weighted sum of the gathered rows
  b = seg_b;
  for (int l = 0; l < L; ++l) {
    const int e = rowptr[v * L + l + 1];
    if (b < e) {
      float st[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) st[c] = s_tgt[(v * L + l) * K + head[c]];
      for (int p = b; p < e; p += 2) {
        const int p1 = min(p + 1, e - 1);
        const int64_t r0 = col[p], r1 = col[p1];
        float4 t0[NCH], t1[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          t0[c] = T[r0 * ldt4 + cc[c]];
          t1[c] = T[r1 * ldt4 + cc[c]];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const float a0 = expf((lrelu(s_src[r0 * K + head[c]] + st[c], slope) - mx[c]) - lse[c]);
          acc[c].x += a0 * t0[c].x; acc[c].y += a0 * t0[c].y; acc[c].z += a0 * t0[c].z; acc[c].w += a0 * t0[c].w;
          if (alpha && leader[c]) alpha[(int64_t)p * K + head[c]] = a0;
          if (p + 1 < e) {
            const float a1 = expf((lrelu(s_src[r1 * K + head[c]] + st[c], slope) - mx[c]) - lse[c]);
            acc[c].x += a1 * t1[c].x; acc[c].y += a1 * t1[c].y; acc[c].z += a1 * t1[c].z; acc[c].w += a1 * t1[c].w;
            if (alpha && leader[c]) alpha[(int64_t)(p + 1) * K + head[c]] = a1;
          }
        }
      }
    }
    b = e;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) out[v * ldo4 + gg.gl + G * c] = acc[c];
}

// per-head reduction of a per-lane partial: every lane ends up with the total of ITS head(s).
// POW2 fast path: one chunk per lane and lanes-per-head a power of two dividing G.
template <int G, int NCH, bool POW2>
__device__ __forceinline__ void head_reduce(const float (&part)[NCH], const int (&head)[NCH], const bool (&on)[NCH],
                                            int K, int Dh4, float (&res)[NCH]) {
  if constexpr (POW2) {
    float x = on[0] ? part[0] : 0.f;
    for (int off = Dh4 >> 1; off >= 1; off >>= 1) x += __shfl_xor(x, off, G);
    res[0] = x;
  } else {
    for (int k = 0; k < K; ++k) {
      float x = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) x += (on[c] && head[c] == k) ? part[c] : 0.f;
      x = group_sum<G>(x);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (head[c] == k) res[c] = x;
    }
  }
}

// backward A (by target): dz[p,k] and gs_tgt[(v,l),k]
template <int G, int NCH, bool POW2>
__global__ __launch_bounds__(256) void rgat_bwd_logits_kernel(
    const float4* __restrict__ T, int64_t ldt4, int32_t D4, int32_t K, int32_t Dh4,
    const float* __restrict__ s_src, const float* __restrict__ s_tgt, const int32_t* __restrict__ rowptr,
    int32_t V, int32_t L, const int32_t* __restrict__ col, float slope, const float* __restrict__ alpha,
    const float4* __restrict__ out, const float4* __restrict__ gout, int64_t ldo4, float* __restrict__ dz,
    float* __restrict__ gs_tgt, int64_t nlb) {
  const Geom gg = geom<G>(V, nlb);
  if (!gg.valid) return;
  const int64_t v = gg.node;
  bool on[NCH], leader[NCH];
  int cc[NCH], head[NCH];
  float4 go[NCH];
  float part[NCH], cdot[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ci = gg.gl + G * c;
    on[c] = ci < D4;
    cc[c] = min(ci, D4 - 1);
    head[c] = cc[c] / Dh4;
    leader[c] = on[c] && (cc[c] % Dh4 == 0);
    go[c] = gout[v * ldo4 + cc[c]];
    part[c] = dot4(go[c], out[v * ldo4 + cc[c]]);
  }
  head_reduce<G, NCH, POW2>(part, head, on, K, Dh4, cdot);  // <gout_vk, out_vk> = sum_p a_p * dalpha_p
  int b = rowptr[v * L];
  for (int l = 0; l < L; ++l) {
    const int e = rowptr[v * L + l + 1];
    float st[NCH], gst[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      st[c] = s_tgt[(v * L + l) * K + head[c]];
      gst[c] = 0.f;
    }
    for (int p = b; p < e; ++p) {
      const int64_t r = col[p];
      float dal[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) part[c] = dot4(go[c], T[r * ldt4 + cc[c]]);
      head_reduce<G, NCH, POW2>(part, head, on, K, Dh4, dal);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float z = s_src[r * K + head[c]] + st[c];
        const float a = alpha[(int64_t)p * K + head[c]];
        const float d = a * (dal[c] - cdot[c]) * (z > 0.f ? 1.f : slope);
        gst[c] += d;
        if (leader[c]) dz[(int64_t)p * K + head[c]] = d;
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (leader[c]) gs_tgt[(v * L + l) * K + head[c]] = gst[c];
    b = e;
  }
}

// backward B (by (source,type) rows r of T): gT[r, head k] = sum_q alpha[pos_b[q],k] * gout[tgt_b[q], head k];
// gs_src[r,k] = sum_q dz[pos_b[q],k]
template <int G, int NCH>
__global__ __launch_bounds__(256) void rgat_bwd_msg_kernel(
    int32_t D4, int32_t K, int32_t Dh4, const int32_t* __restrict__ rowptr_b, int64_t n_rows,
    const int32_t* __restrict__ tgt_b, const int32_t* __restrict__ pos_b, const float* __restrict__ alpha,
    const float* __restrict__ dz, const float4* __restrict__ gout, int64_t ldo4, float4* __restrict__ gT,
    int64_t ldgt4, float* __restrict__ gs_src, int64_t nlb) {
  const Geom gg = geom<G>(n_rows, nlb);
  if (!gg.valid) return;
  const int64_t r = gg.node;
  bool on[NCH];
  int cc[NCH], head[NCH];
  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ci = gg.gl + G * c;
    on[c] = ci < D4;
    cc[c] = min(ci, D4 - 1);
    head[c] = cc[c] / Dh4;
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float gs = 0.f;
  const int b = rowptr_b[r], e = rowptr_b[r + 1];
  for (int q = b; q < e; ++q) {
    const int64_t tg = tgt_b[q], pp = pos_b[q];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float a = alpha[pp * K + head[c]];
      const float4 g = gout[tg * ldo4 + cc[c]];
      acc[c].x += a * g.x; acc[c].y += a * g.y; acc[c].z += a * g.z; acc[c].w += a * g.w;
    }
    if (gg.gl < K) gs += dz[pp * K + gg.gl];
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) gT[r * ldgt4 + gg.gl + G * c] = acc[c];
  if (gg.gl < K) gs_src[r * K + gg.gl] = gs;
}

struct Geo { int G, NCH; };
inline bool pick_geo(int D, int K, Geo* g) {
  if (D <= 0 || K <= 0 || D % K != 0 || (D / K) % 4 != 0 || D > 1024) return false;
  const int D4 = D / 4;
  if (D4 <= 8) *g = {8, 1};
  else if (D4 <= 16) *g = {16, 1};
  else if (D4 <= 32) *g = {32, 1};
  else if (D4 <= 64) *g = {64, 1};
  else if (D4 <= 128) *g = {64, 2};
  else *g = {64, 4};
  return K <= g->G;
}
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline int64_t logical_blocks(int64_t rows, int G) { return (rows + 4 * (64 / G) - 1) / (4 * (64 / G)); }
inline unsigned padded_grid(int64_t nlb) { return (unsigned)(((nlb + 7) / 8) * 8); }
inline bool vec_ok(const void* p, int64_t ld) { return aligned16(p) && ld % 4 == 0; }

#define RGAT_DISPATCH_GEO(geo, GG, NN, ...)                                           \
  if (geo.G == 8) { constexpr int GG = 8, NN = 1; __VA_ARGS__; }                      \
  else if (geo.G == 16) { constexpr int GG = 16, NN = 1; __VA_ARGS__; }               \
  else if (geo.G == 32) { constexpr int GG = 32, NN = 1; __VA_ARGS__; }               \
  else if (geo.NCH == 1) { constexpr int GG = 64, NN = 1; __VA_ARGS__; }              \
  else if (geo.NCH == 2) { constexpr int GG = 64, NN = 2; __VA_ARGS__; }              \
  else { constexpr int GG = 64, NN = 4; __VA_ARGS__; }

}  // namespace

extern "C" {

int relgnn_rgat_fwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* s_src,
                    const float* s_tgt, const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types,
                    const int32_t* col, float slope, float* out, int64_t ldo, float* alpha, void* stream) {
  if (D < 0 || num_nodes < 0 || num_edge_types <= 0 || num_heads <= 0) return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out || !s_src || !s_tgt) return RELGNN_EINVAL;
  Geo geo;
  if (!pick_geo(D, num_heads, &geo) || !vec_ok(T, ldt) || !vec_ok(out, ldo)) return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(num_nodes, geo.G);
  RGAT_DISPATCH_GEO(geo, GG, NN, {
    rgat_fwd_kernel<GG, NN><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
        (const float4*)T, ldt / 4, D / 4, num_heads, D / num_heads / 4, s_src, s_tgt, rowptr, num_nodes, num_edge_types,
        col, slope, (float4*)out, ldo / 4, alpha, nlb);
  });
  return launch_status();
}

int relgnn_rgat_bwd_logits(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* s_src,
                           const float* s_tgt, const int32_t* rowptr, int32_t num_nodes,
                           int32_t num_edge_types, const int32_t* col, float slope, const float* alpha,
                           const float* out, const float* gout, int64_t ldo, float* dz, float* gs_tgt,
                           void* stream) {
  if (D < 0 || num_nodes < 0 || num_edge_types <= 0 || num_heads <= 0) return RELGNN_EINVAL;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out || !gout || !s_src || !s_tgt || !gs_tgt) return RELGNN_EINVAL;
  Geo geo;
  if (!pick_geo(D, num_heads, &geo) || !vec_ok(T, ldt) || !vec_ok(out, ldo) || !aligned16(gout)) return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(num_nodes, geo.G);
  const int Dh4 = D / num_heads / 4;
  const bool pow2 = geo.NCH == 1 && is_pow2(Dh4) && (D / 4) <= geo.G && geo.G % Dh4 == 0 && (D / 4) % Dh4 == 0;
  RGAT_DISPATCH_GEO(geo, GG, NN, {
    if (pow2 && NN == 1)
      rgat_bwd_logits_kernel<GG, 1, true><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
          (const float4*)T, ldt / 4, D / 4, num_heads, Dh4, s_src, s_tgt, rowptr, num_nodes, num_edge_types, col, slope,
          alpha, (const float4*)out, (const float4*)gout, ldo / 4, dz, gs_tgt, nlb);
    else
      rgat_bwd_logits_kernel<GG, NN, false><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
          (const float4*)T, ldt / 4, D / 4, num_heads, Dh4, s_src, s_tgt, rowptr, num_nodes, num_edge_types, col, slope,
          alpha, (const float4*)out, (const float4*)gout, ldo / 4, dz, gs_tgt, nlb);
  });
  return launch_status();
}

int relgnn_rgat_bwd_msg(int32_t D, int32_t num_heads, const int32_t* rowptr_b, int64_t num_rows_t,
                        const int32_t* tgt_b, const int32_t* pos_b, const float* alpha, const float* dz,
                        const float* gout, int64_t ldo, float* gT, int64_t ldgt, float* gs_src, void* stream) {
  if (D < 0 || num_rows_t < 0 || num_heads <= 0) return RELGNN_EINVAL;
  if (num_rows_t == 0 || D == 0) return RELGNN_OK;
  if (!rowptr_b || !gT || !gs_src) return RELGNN_EINVAL;
  Geo geo;
  if (!pick_geo(D, num_heads, &geo) || !vec_ok(gout, ldo) || !vec_ok(gT, ldgt)) return RELGNN_EUNSUPPORTED;
  const int64_t nlb = logical_blocks(num_rows_t, geo.G);
  RGAT_DISPATCH_GEO(geo, GG, NN, {
    rgat_bwd_msg_kernel<GG, NN><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
        D / 4, num_heads, D / num_heads / 4, rowptr_b, num_rows_t, tgt_b, pos_b, alpha, dz, (const float4*)gout, ldo / 4,
        (float4*)gT, ldgt / 4, gs_src, nlb);
  });
  return launch_status();
}

}  // extern "C"
