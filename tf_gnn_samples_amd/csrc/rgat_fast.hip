// RGAT fast path (wave-uniform addressing, same loop structure as seg_reduce.hip):
//   relgnn_rgat_alpha      segmented softmax over all incoming messages of a target; LANES RUN ACROSS MESSAGES
//                          (K floats per message: the two softmax passes never touch the 4*D-byte rows)
//   relgnn_headw_reduce    out[s, head h] = sum_p W[wpos[p], h] * X[col[p], head h]   (per-head weighted gather-reduce;
//                          forward pass 3 with W = alpha, and the gradient w.r.t. T on the transposed buckets)
//   relgnn_rgat_dz         dz[p,h] = a[p,h] * (<gout_vh, T[col p]_h> - <gout_vh, out_vh>) * lrelu'(z[p,h])
// Replaces gnns/rgat.py:98-136 like rgat.hip does; these kernels are used when K in {1,2,4,8} and Dh % 4 == 0.
#include "common.h"

using namespace relgnn;

namespace {

constexpr int kU = 8;

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  return x;
}
// wave total in every lane: four DPP adds inside each 16-lane row, then the four row totals through SGPRs
__device__ __forceinline__ float wave_total_dpp(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float lrelu(float z, float slope) { return z > 0.f ? z : slope * z; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- alpha: one wave per target, lanes across messages -----------------------------------------
template <int K>
__global__ __launch_bounds__(256) void rgat_alpha_kernel(const float* __restrict__ s_src, const float* __restrict__ s_tgt,
                                                         const int32_t* __restrict__ rowptr, int32_t V, int32_t L,
                                                         const int32_t* __restrict__ col, float slope,
                                                         float* __restrict__ alpha, int64_t nlb) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t v = lb * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[v * L]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[(v + 1) * L]);
  if (beg == end) return;

  auto logits = [&](int idx, float (&e)[K]) {
    // edge type of sorted position idx: number of bucket boundaries <= idx (buckets of v are consecutive)
    int l = 0;
    for (int j = 1; j < L; ++j) l += (idx >= rowptr[v * L + j]) ? 1 : 0;
    const int64_t r = col[idx];
    const float* ss = s_src + r * K;
    const float* st = s_tgt + (v * L + l) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) e[k] = lrelu(ss[k] + st[k], slope);
  };

  float mx[K], sm[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { mx[k] = -FLT_MAX; sm[k] = 0.f; }
  const bool single = (end - beg) <= 64;
  float e0[K];
  if (single) {  // the common case: all logits of the target stay in registers
    const int idx = beg + lane;
    const bool ok = idx < end;
    if (ok) logits(idx, e0);
#pragma unroll
    for (int k = 0; k < K; ++k) mx[k] = wave_max(ok ? e0[k] : -FLT_MAX);
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k] = wave_sum(ok ? expf(e0[k] - mx[k]) : 0.f);
    if (ok) {
#pragma unroll
      for (int k = 0; k < K; ++k) alpha[(int64_t)idx * K + k] = expf((e0[k] - mx[k]) - logf(sm[k]));
    }
    return;
  }
  for (int p = beg; p < end; p += 64) {
    const int idx = p + lane;
    const bool ok = idx < end;
    float e[K];
    if (ok) logits(idx, e);
#pragma unroll
    for (int k = 0; k < K; ++k) mx[k] = fmaxf(mx[k], wave_max(ok ? e[k] : -FLT_MAX));
  }
  for (int p = beg; p < end; p += 64) {
    const int idx = p + lane;
    const bool ok = idx < end;
    float e[K];
    if (ok) logits(idx, e);
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k] += wave_sum(ok ? expf(e[k] - mx[k]) : 0.f);
  }
  for (int p = beg; p < end; p += 64) {
    const int idx = p + lane;
    if (idx < end) {
      float e[K];
      logits(idx, e);
#pragma unroll
      for (int k = 0; k < K; ++k) alpha[(int64_t)idx * K + k] = expf((e[k] - mx[k]) - logf(sm[k]));
    }
  }
}

// ---- per-head weighted gather-reduce --------------------------------------------------------------
template <int NCH, int K, bool HAS_Z>
__global__ __launch_bounds__(256) void headw_reduce_kernel(
    const float4* __restrict__ X, int64_t ldx4, int32_t D4, int32_t Dh4, const int32_t* __restrict__ rowptr,
    int64_t num_segments, int32_t stride, const int32_t* __restrict__ col, const float* __restrict__ W,
    const int32_t* __restrict__ wpos, float4* __restrict__ out, int64_t ldo4, int64_t nlb,
    const float* __restrict__ Z, float* __restrict__ zsum) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t s = lb * 4 + (threadIdx.x >> 6);
  if (s >= num_segments) return;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[s * stride]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[(s + 1) * stride]);
  // Z (optional): a second [*, K] per-message table indexed like W; zsum[s, :] = its sum over the segment's messages.
  // Rides along because the RGAT backward needs exactly that sum of dz over the same by-source buckets (rgat.py:103-110,
  // gradient of the per-source score table): as a separate gather-reduce it cost 60 us per layer at the C2 shape.
  float zs[K];
#pragma unroll
  for (int k = 0; k < K; ++k) zs[k] = 0.f;
  float4 acc[NCH];
  bool on[NCH];
  uint32_t cc[NCH];
  int head[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    on[c] = lane + 64 * c < D4;
    cc[c] = (uint32_t)min(lane + 64 * c, D4 - 1);
    head[c] = (int)cc[c] / Dh4;
  }
  const uint32_t ld = (uint32_t)ldx4;
  for (int p = beg; p < end; p += 64) {
    const int n = min(64, end - p);
    const bool ok = lane < n;
    const int my_col = ok ? col[p + lane] : 0;
    const int64_t wrow = ok ? (wpos ? (int64_t)wpos[p + lane] : (int64_t)(p + lane)) : 0;
    float my_w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) my_w[k] = ok ? W[wrow * K + k] : 0.f;
    if constexpr (HAS_Z) {
#pragma unroll
      for (int k = 0; k < K; ++k) zs[k] += ok ? Z[wrow * K + k] : 0.f;
    }
    int k0 = 0;
    for (; k0 + kU <= n; k0 += kU) {
      float4 v[kU][NCH];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k0 + u);
        const float4* row = X + (size_t)(r * ld);
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[u][c] = row[cc[c]];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        float wl[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) wl[c] = 0.f;
#pragma unroll
        for (int h = 0; h < K; ++h) {
          const float sh = rl(my_w[h], k0 + u);
#pragma unroll
          for (int c = 0; c < NCH; ++c) wl[c] = (head[c] == h) ? sh : wl[c];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          acc[c].x += wl[c] * v[u][c].x; acc[c].y += wl[c] * v[u][c].y;
          acc[c].z += wl[c] * v[u][c].z; acc[c].w += wl[c] * v[u][c].w;
        }
      }
    }
    for (; k0 < n; ++k0) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k0);
      const float4* row = X + (size_t)(r * ld);
      float wl[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) wl[c] = 0.f;
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const float sh = rl(my_w[h], k0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) wl[c] = (head[c] == h) ? sh : wl[c];
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 t = row[cc[c]];
        acc[c].x += wl[c] * t.x; acc[c].y += wl[c] * t.y; acc[c].z += wl[c] * t.z; acc[c].w += wl[c] * t.w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (on[c]) out[s * ldo4 + lane + 64 * c] = acc[c];
  if constexpr (HAS_Z) {
    float tot = 0.f;      // lane k keeps the total of column k
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float t = wave_total_dpp(zs[k]);
      tot = (lane == k) ? t : tot;
    }
    if (lane < K) zsum[s * K + lane] = tot;
  }
}

// ---- dz: one wave per target, lanes across features ---------------------------------------------
// requires NCH == 1 and Dh4 a power of two (lanes of a head are one aligned group): the per-message
// per-head dot product is a log2(Dh4)-step butterfly (DPP lane selects inside a 16-lane row: 349 -> 236 us at C4;
// a transposed reduction that takes Dh4 messages at a time and needs Dh4-1 shuffles per Dh4 messages was tried and
// was slower, 477 us: 16 float4 rows + 16 partials per lane cost more occupancy than the shuffles it saves).
template <int K>
__global__ __launch_bounds__(256) void rgat_dz_kernel(
    const float4* __restrict__ T, int64_t ldt4, int32_t D4, int32_t Dh4, const float* __restrict__ s_src,
    const float* __restrict__ s_tgt, const int32_t* __restrict__ rowptr, int32_t V, int32_t L,
    const int32_t* __restrict__ col, float slope, const float* __restrict__ alpha, const float4* __restrict__ out,
    const float4* __restrict__ gout, int64_t ldo4, float* __restrict__ dz, int64_t nlb, float* __restrict__ gs_tgt) {
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t v = lb * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[v * L]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[(v + 1) * L]);
  // gs_tgt (optional, L * K <= 64): gs_tgt[(v, l), k] = sum of dz[p, k] over the messages of bucket (v, l) — the gradient
  // of the per-(target, type) score table (rgat.py:103-110).  dz is in registers here, lane j keeps the running total of
  // (l, k) = (j / K, j % K); as a separate gather-reduce over dz it cost 70 us per layer at the C2 shape.
  float bucket_total = 0.f;
  if (beg == end) {
    if (gs_tgt && lane < L * K) gs_tgt[(int64_t)v * L * K + lane] = 0.f;
    return;
  }
  const bool on = lane < D4;
  const uint32_t cc = (uint32_t)min(lane, D4 - 1);
  const int head = (int)cc / Dh4;
  const float4 go = on ? gout[v * ldo4 + cc] : make_float4(0.f, 0.f, 0.f, 0.f);
  // all-reduce over the Dh4 aligned lanes of a head.  Up to 16 lanes (one DPP row) it is four full-rate VALU adds
  // with DPP lane selects (xor 1, xor 2, half-row mirror, row mirror) instead of LDS-crossbar shuffles.
  auto head_sum = [&](float x) {
    if (Dh4 >= 2) x += dpp_f32<0xB1>(x);    // quad_perm [1,0,3,2]
    if (Dh4 >= 4) x += dpp_f32<0x4E>(x);    // quad_perm [2,3,0,1]
    if (Dh4 >= 8) x += dpp_f32<0x141>(x);   // row_half_mirror: the other quad of the 8-lane half
    if (Dh4 >= 16) x += dpp_f32<0x140>(x);  // row_mirror: the other half of the 16-lane row
    if (Dh4 >= 32) x += __shfl_xor(x, 16);
    if (Dh4 >= 64) x += __shfl_xor(x, 32);
    return x;  // total of this lane's head
  };
  const float cdot = head_sum(on ? dot4(go, out[v * ldo4 + cc]) : 0.f);
  const uint32_t ld = (uint32_t)ldt4;
  // The per-head values of a message live in the lanes of that head; the per-message factors (alpha, lrelu') live one
  // message per lane.  The transposition goes through LDS: the first lane of every head stores its value at
  // [message][head] (one ds_write with K active lanes per message), after the chunk every lane loads its message's K
  // values (K readlane + K select per message before: 241 us per launch at the C2 shape).
  __shared__ float stage_all[4][64 * K];
  float* stage = stage_all[threadIdx.x >> 6];
  const bool leader = on && ((int)cc % Dh4) == 0;
  for (int p = beg; p < end; p += 64) {
    const int n = min(64, end - p);
    const int idx = p + lane;
    const bool ok = lane < n;
    const int my_col = ok ? col[idx] : 0;
    // this lane's message: logits derivative factors and alpha, K heads
    float my_a[K], my_d[K], my_dz[K];
    int l = 0;
    if (ok) {
      for (int j = 1; j < L; ++j) l += (idx >= rowptr[v * L + j]) ? 1 : 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float z = s_src[(int64_t)my_col * K + k] + s_tgt[(v * L + l) * K + k];
        my_d[k] = z > 0.f ? 1.f : slope;
        my_a[k] = alpha[(int64_t)idx * K + k];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) my_dz[k] = 0.f;
    int k0 = 0;
    for (; k0 + kU <= n; k0 += kU) {
      float4 t[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k0 + u);
        t[u] = T[(size_t)(r * ld) + cc];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float dal = head_sum(on ? dot4(go, t[u]) : 0.f) - cdot;   // (dalpha - <gout,out>) of my head
        if (leader) stage[(k0 + u) * K + head] = dal;                   // message-major, read back one message per lane
      }
    }
    for (; k0 < n; ++k0) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, k0);
      const float4 t = T[(size_t)(r * ld) + cc];
      const float dal = head_sum(on ? dot4(go, t) : 0.f) - cdot;
      if (leader) stage[k0 * K + head] = dal;
    }
    // same wave, LDS operations complete in issue order: lane i now reads the K values the head leaders wrote for message i
    if (ok) {
#pragma unroll
      for (int k = 0; k < K; ++k) my_dz[k] = stage[lane * K + k];
    }
    float val[K];
#pragma unroll
    for (int k = 0; k < K; ++k) val[k] = ok ? my_a[k] * my_dz[k] * my_d[k] : 0.f;
    if (ok) {
#pragma unroll
      for (int k = 0; k < K; ++k) dz[(int64_t)idx * K + k] = val[k];
    }
    if (gs_tgt) {
      // the chunk's messages are in bucket order: types l_lo .. l_hi, each a contiguous lane range
      const int l_lo = __builtin_amdgcn_readlane(l, 0), l_hi = __builtin_amdgcn_readlane(l, n - 1);
      for (int t = l_lo; t <= l_hi; ++t) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float tot = wave_total_dpp((ok && l == t) ? val[k] : 0.f);
          bucket_total = (lane == t * K + k) ? bucket_total + tot : bucket_total;
        }
      }
    }
  }
  if (gs_tgt && lane < L * K) gs_tgt[(int64_t)v * L * K + lane] = bucket_total;
}

inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline bool vec_ok(const void* p, int64_t ld) { return aligned16(p) && ld % 4 == 0; }
inline unsigned padded_grid(int64_t nlb) { return (unsigned)(((nlb + 7) / 8) * 8); }

#define RGAT_DISPATCH_K(K_, KK, ...)                      \
  switch (K_) {                                           \
    case 1: { constexpr int KK = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int KK = 2; __VA_ARGS__; break; } \
    case 4: { constexpr int KK = 4; __VA_ARGS__; break; } \
    case 8: { constexpr int KK = 8; __VA_ARGS__; break; } \
    default: return RELGNN_EUNSUPPORTED;                  \
  }

}  // namespace

extern "C" {

int relgnn_rgat_alpha(const float* s_src, const float* s_tgt, int32_t num_heads, const int32_t* rowptr,
                      int32_t num_nodes, int32_t num_edge_types, const int32_t* col, float slope, float* alpha,
                      void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || num_heads <= 0) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!s_src || !s_tgt || !rowptr || !alpha) return RELGNN_EINVAL;
  const int64_t nlb = ((int64_t)num_nodes + 3) / 4;
  RGAT_DISPATCH_K(num_heads, KK, (rgat_alpha_kernel<KK><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
                                     s_src, s_tgt, rowptr, num_nodes, num_edge_types, col, slope, alpha, nlb)));
  return launch_status();
}

int relgnn_headw_reduce(const float* X, int64_t num_rows_x, int64_t ldx, int32_t D, int32_t num_heads,
                        const int32_t* rowptr, int64_t num_segments, int32_t seg_stride, const int32_t* col,
                        const float* W, const int32_t* wpos, float* out, int64_t ldo, const float* Z, float* zsum,
                        void* stream) {
  if (D < 0 || num_segments < 0 || seg_stride <= 0 || num_heads <= 0 || num_rows_x < 0) return RELGNN_EINVAL;
  if (num_segments == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out || !W || ((Z == nullptr) != (zsum == nullptr))) return RELGNN_EINVAL;
  if (D % num_heads != 0 || (D / num_heads) % 4 != 0 || D > 1024 || !vec_ok(X, ldx) || !vec_ok(out, ldo) ||
      num_rows_x * (ldx / 4) >= ((int64_t)1 << 32))
    return RELGNN_EUNSUPPORTED;
  const int D4 = D / 4, Dh4 = D / num_heads / 4;
  const int64_t nlb = (num_segments + 3) / 4;
  const int nch = D4 <= 64 ? 1 : (D4 <= 128 ? 2 : 4);
  hipStream_t st = as_stream(stream);
#define HEADW_LAUNCH(NN, KK)                                                                                       \
  do {                                                                                                             \
    if (Z)                                                                                                         \
      headw_reduce_kernel<NN, KK, true><<<padded_grid(nlb), 256, 0, st>>>(                                          \
          (const float4*)X, ldx / 4, D4, Dh4, rowptr, num_segments, seg_stride, col, W, wpos, (float4*)out, ldo / 4, nlb, Z, zsum); \
    else                                                                                                           \
      headw_reduce_kernel<NN, KK, false><<<padded_grid(nlb), 256, 0, st>>>(                                         \
          (const float4*)X, ldx / 4, D4, Dh4, rowptr, num_segments, seg_stride, col, W, wpos, (float4*)out, ldo / 4, nlb, Z, zsum); \
  } while (0)
  RGAT_DISPATCH_K(num_heads, KK, {
    if (nch == 1) HEADW_LAUNCH(1, KK);
    else if (nch == 2) HEADW_LAUNCH(2, KK);
    else HEADW_LAUNCH(4, KK);
  });
#undef HEADW_LAUNCH
  return launch_status();
}

int relgnn_rgat_dz(const float* T, int64_t num_rows_t, int64_t ldt, int32_t D, int32_t num_heads, const float* s_src,
                   const float* s_tgt, const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types,
                   const int32_t* col, float slope, const float* alpha, const float* out, const float* gout,
                   int64_t ldo, float* dz, float* gs_tgt, void* stream) {
  if (D < 0 || num_nodes < 0 || num_edge_types <= 0 || num_heads <= 0) return RELGNN_EINVAL;
  if (gs_tgt && (int64_t)num_edge_types * num_heads > 64) return RELGNN_EUNSUPPORTED;
  if (num_nodes == 0 || D == 0) return RELGNN_OK;
  if (!rowptr || !out || !gout || !s_src || !s_tgt || !alpha || !dz) return RELGNN_EINVAL;
  const int D4 = D / 4;
  if (D % 4 != 0 || D % num_heads != 0 || (D / num_heads) % 4 != 0) return RELGNN_EUNSUPPORTED;
  const int Dh4 = D / num_heads / 4;
  if (D4 > 64 || !is_pow2(Dh4) || !vec_ok(T, ldt) || !vec_ok(out, ldo) || !aligned16(gout) ||
      num_rows_t * (ldt / 4) >= ((int64_t)1 << 32))
    return RELGNN_EUNSUPPORTED;
  const int64_t nlb = ((int64_t)num_nodes + 3) / 4;
  RGAT_DISPATCH_K(num_heads, KK, (rgat_dz_kernel<KK><<<padded_grid(nlb), 256, 0, as_stream(stream)>>>(
                                     (const float4*)T, ldt / 4, D4, Dh4, s_src, s_tgt, rowptr, num_nodes, num_edge_types,
                                     col, slope, alpha, (const float4*)out, (const float4*)gout, ldo / 4, dz, nlb, gs_tgt)));
  return launch_status();
}

}  // extern "C"
