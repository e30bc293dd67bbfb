// Fused aggregate -> transform for the dense-weight relational layers (the north-star "per-edge-type linear transform
// fused as an MFMA GEMM"):
//
//     out[s, :] = act( f_mode( sum_l  ( sum_{p in bucket (s,l)} w[p] * X[col[p], :] ) @ W_l ) )
//
// i.e. gnns/rgcn.py:84-114 (and ggnn.py:76-89) with the per-edge-type Dense moved BEHIND the aggregation (sum / mean /
// sqrt_n are linear: sum_e w_e (h_e W_l) == (sum_e w_e h_e) W_l up to fp32 rounding), evaluated on the matrix cores in
// exact fp32 (v_mfma_f32_32x32x2_f32).  The [V, L*D] table of transformed states the unfused path writes and gathers
// from never exists: messages are gathered from the 3x smaller node-state table (one graph's slab fits the 4 MiB L2 of
// its XCD), and the matrix work runs UNDER the gather instead of after it.
//
// One workgroup = 32 output rows, 8 waves, WAVE-SPECIALISED:
//   waves 4-7 (producers)  fold the (s, l) buckets of 8 rows each exactly like seg_reduce_wave_kernel (lanes across the
//                          features, row indices broadcast into SGPRs, 16 row loads in flight per wave) and park the 32
//                          aggregated rows of edge type l in LDS buffer l & 1 (row stride Din + 4 floats: conflict-free
//                          ds_read_b128 for the MFMA A operand); optionally they also stream the rows to agg_out
//                          (the operand of the weight gradient dW_l = A_l^T @ dOut);
//   waves 0-3 (consumers)  multiply the 32 x Din tile of type l-1 with W_{l-1} [Din, Dout]: wave c owns output columns
//                          [c*Dout/4, (c+1)*Dout/4); B operands come straight from global memory (L2-resident weights)
//                          in a pre-packed MFMA order, one dwordx4 per lane = the operands of 4 MFMAs, prefetched one
//                          k-tile ahead.
// One barrier per edge type; L + 1 phases per workgroup; two workgroups per CU run at different phases.
// The backward pass dX = sum_l (sum_{p in (source,l)} w_p dOut[tgt_p]) @ W_l^T is THE SAME kernel on the by-source
// buckets with the transposed weights packed.
//
// Packed weights: Wp[l][kt][j4][nt][lane][e] = W_l[kt*32 + (lane>>5)*16 + j4*4 + e][nt*32 + (lane&31)]
#include "common.h"

#include <stdlib.h>

using namespace relgnn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 32;      // output rows per workgroup
constexpr int GU = 16;        // gathered rows in flight per producer wave

template <int DIN, int TN>
__global__ __launch_bounds__(512) void agg_transform_kernel(
    const float4* __restrict__ X, int64_t ldx4, const int32_t* __restrict__ rowptr, int64_t S, int32_t L,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float4* __restrict__ Wp, int32_t mode, int32_t act,
    float* __restrict__ out, int64_t ldo, float* __restrict__ agg_out, int64_t ld_agg, int64_t n_logical, int32_t ablate) {
  constexpr int ASTRIDE = DIN + 4;
  constexpr int D4 = DIN / 4;
  constexpr int NCH = (D4 + 63) / 64;
  constexpr int KT = DIN / 32;
  constexpr int NT = 4 * TN;                 // 32-column tiles of the output
  __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * ASTRIDE];
  const int64_t lb = xcd_logical_block(n_logical);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v0 = lb * ROWS;
  const bool consumer = wave < 4;
  const uint32_t ld = (uint32_t)ldx4;

  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  for (int t = 0; t <= L; ++t) {
    if (!consumer && t < L && !(ablate & 2)) {
      // ---- producers: aggregated rows of edge type t -> LDS buffer t & 1 -------------------------------------
      float* buf = lds + (t & 1) * ROWS * ASTRIDE;
      const int p = wave - 4;
      for (int ii = 0; ii < ROWS / 4; ++ii) {
        const int i = p * (ROWS / 4) + ii;
        const int64_t v = v0 + i;
        float4 a[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < S) {
          const int beg = __builtin_amdgcn_readfirstlane(rowptr[v * L + t]);
          const int end = __builtin_amdgcn_readfirstlane(rowptr[v * L + t + 1]);
          for (int q = beg; q < end; q += 64) {
            const int n = min(64, end - q);
            const int my_col = (lane < n) ? col[q + lane] : 0;
            const float my_w = (w && lane < n) ? w[q + lane] : 1.f;
            for (int k = 0; k < n; k += GU) {
              const int rem = n - k;
              float4 x[GU][NCH];
              float ww[GU];
#pragma unroll
              for (int u = 0; u < GU; ++u) {
                const int ku = k + min(u, rem - 1);          // padding slots re-read the last row (never added)
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, ku);
                ww[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), ku));
                const float4* row = X + (size_t)(r * ld);
#pragma unroll
                for (int c = 0; c < NCH; ++c) x[u][c] = row[min(lane + 64 * c, D4 - 1)];
              }
#pragma unroll
              for (int u = 0; u < GU; ++u)
                if (u < rem) {
#pragma unroll
                  for (int c = 0; c < NCH; ++c) {
                    a[c].x += ww[u] * x[u][c].x; a[c].y += ww[u] * x[u][c].y;
                    a[c].z += ww[u] * x[u][c].z; a[c].w += ww[u] * x[u][c].w;
                  }
                }
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int k4 = lane + 64 * c;
          if (k4 < D4) {
            *reinterpret_cast<float4*>(buf + i * ASTRIDE + 4 * k4) = a[c];
            if (agg_out && v < S) *reinterpret_cast<float4*>(agg_out + v * ld_agg + (int64_t)t * DIN + 4 * k4) = a[c];
          }
        }
      }
    }
    if (consumer && t >= 1 && !(ablate & 1)) {
      // ---- consumers: acc[32 x Dout/4] += A_{t-1}[32 x DIN] @ W_{t-1} ----------------------------------------
      const int l = t - 1;
      const float* arow = lds + (l & 1) * ROWS * ASTRIDE + (lane & 31) * ASTRIDE + (lane >> 5) * 16;
      const float4* wl = Wp + (size_t)l * KT * 4 * NT * 64 + (size_t)(wave * TN) * 64 + lane;
      // operands of k-tile kt, group j4, column tile b:  wl[((kt*4 + j4)*NT + b) * 64]
      float4 bq[4][TN];
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int b = 0; b < TN; ++b) bq[j4][b] = wl[(size_t)(j4 * NT + b) * 64];
      for (int kt = 0; kt < KT; ++kt) {
        float4 bn[4][TN];
        if (kt + 1 < KT) {                                   // next k-tile's B operands: in flight during this one's MFMAs
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int b = 0; b < TN; ++b) bn[j4][b] = wl[(size_t)(((kt + 1) * 4 + j4) * NT + b) * 64];
        }
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 av = *reinterpret_cast<const float4*>(arow + kt * 32 + j4 * 4);
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bq[j4][b].x, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bq[j4][b].y, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bq[j4][b].z, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bq[j4][b].w, acc[b], 0, 0, 0);
          }
        }
        if (kt + 1 < KT) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int b = 0; b < TN; ++b) bq[j4][b] = bn[j4][b];
        }
      }
    }
    __syncthreads();
  }

  if (!consumer) return;
  // ---- epilogue: mean / sqrt_n factor (messages of the row over all types), activation, store --------------------
  // C layout of v_mfma_f32_32x32x2: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int cidx = (wave * TN + b) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t v = v0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (v < S) {
        float x = acc[b][r];
        if (mode != RELGNN_AGG_SUM) {
          const float n = (float)max(rowptr[(v + 1) * L] - rowptr[v * L], 1);
          x = (mode == RELGNN_AGG_MEAN) ? x / n : x / sqrtf(n);
        }
        switch (act) {
          case RELGNN_ACT_TANH: x = act_fwd<RELGNN_ACT_TANH>(x); break;
          case RELGNN_ACT_RELU: x = act_fwd<RELGNN_ACT_RELU>(x); break;
          case RELGNN_ACT_LEAKY_RELU: x = act_fwd<RELGNN_ACT_LEAKY_RELU>(x); break;
          case RELGNN_ACT_ELU: x = act_fwd<RELGNN_ACT_ELU>(x); break;
          case RELGNN_ACT_SELU: x = act_fwd<RELGNN_ACT_SELU>(x); break;
          case RELGNN_ACT_GELU: x = act_fwd<RELGNN_ACT_GELU>(x); break;
          default: break;
        }
        out[v * ldo + cidx] = x;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ring variant: ONE persistent workgroup per CU, 16 waves = 4 consumers + 12 producers, and a ring of three 32-row LDS
// tiles between them.  Producers run up to two tiles ahead of the consumers (flag counters in LDS, no workgroup-wide
// barrier in the loop), so a phase with little to gather (self-loop type) or a long one (a hub) no longer stalls the other
// side: the kernel runs at max(sum of gather time, sum of MFMA time) instead of the sum of per-phase maxima.
//   tile T of a workgroup = (row block b_begin + T / L, edge type T % L), slot T % 3
//   prod_done[slot] counts producer waves that finished writing the slot (monotonic: tile T is ready at 12 * (T/3 + 1))
//   cons_done[slot] counts consumer waves that finished reading it  (slot free for tile T at 4 * (T/3))
// LDS operations of one wave are executed in order, so "rows written, then counter incremented" (release) and "counter
// read, then rows read" (acquire) is all the ordering the hand-off needs.  Every wait is bounded: after ~2^22 polls (about half a second) a
// wave raises the abort flag, everybody leaves, and the launch reports RELGNN_EHIP through the error word.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RING = 3, NPROD = 12, NCONS = 4;

__device__ __forceinline__ bool wait_at_least(int* flag, int target, int* abort_flag) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1 << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return false;
    }
  }
  return true;
}

template <int DIN, int TN>
__global__ __launch_bounds__(1024) void agg_transform_ring_kernel(
    const float4* __restrict__ X, int64_t ldx4, const int32_t* __restrict__ rowptr, int64_t S, int32_t L,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float4* __restrict__ Wp, int32_t mode, int32_t act,
    float* __restrict__ out, int64_t ldo, float* __restrict__ agg_out, int64_t ld_agg, int64_t nblocks, int32_t* __restrict__ err,
    int32_t ablate) {
  constexpr int ASTRIDE = DIN + 4;
  constexpr int D4 = DIN / 4;
  constexpr int NCH = (D4 + 63) / 64;
  constexpr int KT = DIN / 32;
  constexpr int NT = 4 * TN;
  constexpr int TILE = ROWS * ASTRIDE;
  __shared__ __attribute__((aligned(16))) float lds[RING * TILE + 16];
  int* flags = reinterpret_cast<int*>(lds + RING * TILE);       // [0..2] prod_done, [3..5] cons_done, [6] abort
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 16) flags[threadIdx.x] = 0;
  __syncthreads();
  // contiguous range of row blocks per workgroup; workgroups of one XCD (b & 7) own neighbouring ranges
  const int64_t G = gridDim.x;
  const int64_t lw = (int64_t)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int64_t b_begin = nblocks * lw / G, b_end = nblocks * (lw + 1) / G;
  const int ntiles = (int)(b_end - b_begin) * L;
  const uint32_t ld = (uint32_t)ldx4;
  int* prod_done = flags;
  int* cons_done = flags + RING;
  int* abort_flag = flags + 2 * RING;

  if (wave >= NCONS) {
    // ================================ producers ================================
    const int p = wave - NCONS;
    for (int T = 0; T < ntiles; ++T) {
      const int slot = T % RING, l = T % L;
      const int64_t v0 = (b_begin + T / L) * ROWS;
      float* buf = lds + slot * TILE;
      if (!wait_at_least(cons_done + slot, NCONS * (T / RING), abort_flag)) break;
      for (int i = p; i < ROWS; i += NPROD) {
        const int64_t v = v0 + i;
        float4 a[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < S && !(ablate & 2)) {
          const int beg = __builtin_amdgcn_readfirstlane(rowptr[v * L + l]);
          const int end = __builtin_amdgcn_readfirstlane(rowptr[v * L + l + 1]);
          for (int q = beg; q < end; q += 64) {
            const int n = min(64, end - q);
            const int my_col = (lane < n) ? col[q + lane] : 0;
            const float my_w = (w && lane < n) ? w[q + lane] : 1.f;
            for (int k = 0; k < n; k += GU) {
              const int rem = n - k;
              float4 x[GU][NCH];
              float ww[GU];
#pragma unroll
              for (int u = 0; u < GU; ++u) {
                const int ku = k + min(u, rem - 1);
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, ku);
                ww[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), ku));
                const float4* row = X + (size_t)(r * ld);
#pragma unroll
                for (int c = 0; c < NCH; ++c) x[u][c] = row[min(lane + 64 * c, D4 - 1)];
              }
#pragma unroll
              for (int u = 0; u < GU; ++u)
                if (u < rem) {
#pragma unroll
                  for (int c = 0; c < NCH; ++c) {
                    a[c].x += ww[u] * x[u][c].x; a[c].y += ww[u] * x[u][c].y;
                    a[c].z += ww[u] * x[u][c].z; a[c].w += ww[u] * x[u][c].w;
                  }
                }
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int k4 = lane + 64 * c;
          if (k4 < D4) {
            *reinterpret_cast<float4*>(buf + i * ASTRIDE + 4 * k4) = a[c];
            if (agg_out && v < S) *reinterpret_cast<float4*>(agg_out + v * ld_agg + (int64_t)l * DIN + 4 * k4) = a[c];
          }
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(prod_done + slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    // ================================ consumers ================================
    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    for (int T = 0; T < ntiles; ++T) {
      const int slot = T % RING, l = T % L;
      const int64_t v0 = (b_begin + T / L) * ROWS;
      const float4* wl = Wp + (size_t)l * KT * 4 * NT * 64 + (size_t)(wave * TN) * 64 + lane;
      float4 bq[4][TN];
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int b = 0; b < TN; ++b) bq[j4][b] = wl[(size_t)(j4 * NT + b) * 64];      // weights: issued before the wait
      if (!wait_at_least(prod_done + slot, NPROD * (T / RING + 1), abort_flag)) break;
      const float* arow = lds + slot * TILE + (lane & 31) * ASTRIDE + (lane >> 5) * 16;
      for (int kt = 0; kt < ((ablate & 1) ? 0 : KT); ++kt) {
        float4 bn[4][TN];
        if (kt + 1 < KT) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int b = 0; b < TN; ++b) bn[j4][b] = wl[(size_t)(((kt + 1) * 4 + j4) * NT + b) * 64];
        }
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 av = *reinterpret_cast<const float4*>(arow + kt * 32 + j4 * 4);
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bq[j4][b].x, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bq[j4][b].y, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bq[j4][b].z, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bq[j4][b].w, acc[b], 0, 0, 0);
          }
        }
        if (kt + 1 < KT) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int b = 0; b < TN; ++b) bq[j4][b] = bn[j4][b];
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(cons_done + slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (l == L - 1) {
        // ---- epilogue of one 32-row block ----
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int cidx = (wave * TN + b) * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t v = v0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (v < S) {
              float x = acc[b][r];
              if (mode != RELGNN_AGG_SUM) {
                const float n = (float)max(rowptr[(v + 1) * L] - rowptr[v * L], 1);
                x = (mode == RELGNN_AGG_MEAN) ? x / n : x / sqrtf(n);
              }
              switch (act) {
                case RELGNN_ACT_TANH: x = act_fwd<RELGNN_ACT_TANH>(x); break;
                case RELGNN_ACT_RELU: x = act_fwd<RELGNN_ACT_RELU>(x); break;
                case RELGNN_ACT_LEAKY_RELU: x = act_fwd<RELGNN_ACT_LEAKY_RELU>(x); break;
                case RELGNN_ACT_ELU: x = act_fwd<RELGNN_ACT_ELU>(x); break;
                case RELGNN_ACT_SELU: x = act_fwd<RELGNN_ACT_SELU>(x); break;
                case RELGNN_ACT_GELU: x = act_fwd<RELGNN_ACT_GELU>(x); break;
                default: break;
              }
              out[v * ldo + cidx] = x;
            }
            acc[b][r] = 0.f;
          }
        }
      }
    }
  }
  if (threadIdx.x == 0 && err) {
    // (read after the loops: an aborted hand-off is reported, never silently ignored)
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicOr(err, 1);
  }
}

// Wp[l][kt][j4][nt][lane][e] = W(l)[kt*32 + (lane>>5)*16 + j4*4 + e][nt*32 + (lane&31)];  W(l)[k][n] is read at
// W + l*type_stride + k*row_stride + n*col_stride (col_stride 1, row_stride ldw: W_l as stored; transposed: swap them)
__global__ __launch_bounds__(256) void pack_agg_weights_kernel(const float* __restrict__ W, int32_t L, int32_t Din, int32_t Dout,
                                                               int64_t type_stride, int64_t row_stride, int64_t col_stride,
                                                               float* __restrict__ Wp) {
  const int NT = Dout / 32, KT = Din / 32;
  const int64_t total = (int64_t)L * Din * Dout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t x = i;
    const int e = (int)(x & 3); x >>= 2;
    const int lane = (int)(x & 63); x >>= 6;
    const int nt = (int)(x % NT); x /= NT;
    const int j4 = (int)(x & 3); x >>= 2;
    const int kt = (int)(x % KT);
    const int l = (int)(x / KT);
    const int k = kt * 32 + (lane >> 5) * 16 + j4 * 4 + e, n = nt * 32 + (lane & 31);
    Wp[i] = W[(int64_t)l * type_stride + (int64_t)k * row_stride + (int64_t)n * col_stride];
  }
}

}  // namespace

extern "C" {

int relgnn_agg_transform_supported(int32_t Din, int32_t Dout) {
  return ((Din == 128 || Din == 256) && (Dout == 128 || Dout == 256)) ? 1 : 0;
}

int relgnn_agg_transform_pack_weights(const float* W, int32_t num_edge_types, int32_t Din, int32_t Dout, int64_t type_stride,
                                      int64_t row_stride, int64_t col_stride, float* packed, void* stream) {
  if (num_edge_types <= 0 || Din <= 0 || Dout <= 0) return RELGNN_EINVAL;
  if (Din % 32 != 0 || Dout % 32 != 0) return RELGNN_EUNSUPPORTED;
  if (!W || !packed) return RELGNN_EINVAL;
  const int64_t total = (int64_t)num_edge_types * Din * Dout;
  pack_agg_weights_kernel<<<flat_grid(total, 256), 256, 0, as_stream(stream)>>>(W, num_edge_types, Din, Dout, type_stride,
                                                                               row_stride, col_stride, packed);
  return launch_status();
}

int relgnn_agg_transform_fwd(int32_t mode, int32_t act, const float* X, int64_t num_rows_x, int64_t ldx, int32_t Din,
                             const int32_t* rowptr, int64_t num_out, int32_t num_edge_types, const int32_t* col, const float* w,
                             const float* packed_weights, int32_t Dout, float* out, int64_t ldo, float* agg_out, int64_t ld_agg,
                             int32_t* err_flag, void* stream) {
  if (mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_SQRT_N) return mode == RELGNN_AGG_MAX ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU || num_out < 0 || num_edge_types <= 0 || ldx < Din || ldo < Dout ||
      num_rows_x < 0)
    return RELGNN_EINVAL;
  if (!relgnn_agg_transform_supported(Din, Dout)) return RELGNN_EUNSUPPORTED;
  if (num_out == 0) return RELGNN_OK;
  if (!X || !rowptr || !packed_weights || !out) return RELGNN_EINVAL;
  if (ldx % 4 != 0 || !aligned16(X) || !aligned16(packed_weights) || num_rows_x * (ldx / 4) >= ((int64_t)1 << 32) ||
      (agg_out && (ld_agg % 4 != 0 || !aligned16(agg_out) || ld_agg < (int64_t)num_edge_types * Din)))
    return RELGNN_EUNSUPPORTED;
  const int64_t nlb = (num_out + ROWS - 1) / ROWS;
  const unsigned grid = (unsigned)(((nlb + 7) / 8) * 8);
  hipStream_t st = as_stream(stream);
  static int ablate = -1;   // experiments only: RELGNN_AGG_ABLATE bit0 = skip the MFMA phase, bit1 = skip the gather phase
  if (ablate < 0) { const char* e = getenv("RELGNN_AGG_ABLATE"); ablate = e ? atoi(e) : 0; }
  static int variant = -1;  // RELGNN_AGG_VARIANT=phase selects the barrier-per-phase kernel (2 workgroups per CU)
  if (variant < 0) { const char* e = getenv("RELGNN_AGG_VARIANT"); variant = (e && e[0] == 'p') ? 1 : 0; }
  if (variant == 0) {
    if (!err_flag) return RELGNN_EINVAL;
    const unsigned g = (unsigned)(nlb >= 256 ? 256 : ((nlb + 7) / 8) * 8);
#define RING_LAUNCH(DIN_, TN_)                                                                                              \
  agg_transform_ring_kernel<DIN_, TN_><<<g, 1024, 0, st>>>((const float4*)X, ldx / 4, rowptr, num_out, num_edge_types, col, w, \
                                                           (const float4*)packed_weights, mode, act, out, ldo, agg_out, ld_agg, nlb, err_flag, ablate)
    if (Din == 256 && Dout == 256) RING_LAUNCH(256, 2);
    else if (Din == 256 && Dout == 128) RING_LAUNCH(256, 1);
    else if (Din == 128 && Dout == 256) RING_LAUNCH(128, 2);
    else RING_LAUNCH(128, 1);
#undef RING_LAUNCH
    return launch_status();
  }
#define AGG_LAUNCH(DIN_, TN_)                                                                                              \
  agg_transform_kernel<DIN_, TN_><<<grid, 512, 0, st>>>((const float4*)X, ldx / 4, rowptr, num_out, num_edge_types, col, w,  \
                                                        (const float4*)packed_weights, mode, act, out, ldo, agg_out, ld_agg, nlb, ablate)
  if (Din == 256 && Dout == 256) AGG_LAUNCH(256, 2);
  else if (Din == 256 && Dout == 128) AGG_LAUNCH(256, 1);
  else if (Din == 128 && Dout == 256) AGG_LAUNCH(128, 2);
  else AGG_LAUNCH(128, 1);
#undef AGG_LAUNCH
  return launch_status();
}

}  // extern "C"
