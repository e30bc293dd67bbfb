// Fused aggregate -> transform for the dense-weight relational layers (RGCN / GGNN message pass):
//
//     out[v, :] = act( f_mode( sum_l  ( sum_{p in bucket (v,l)} w[p] * H[src[p], :] ) @ W_l ) )
//
// i.e. gnns/rgcn.py:87-114 (and ggnn.py:76-89) with the per-edge-type Dense moved BEHIND the aggregation
// (sum/mean/sqrt_n are linear, so sum_e w_e (h_e W_l) == (sum_e w_e h_e) W_l up to fp32 rounding) and run on
// the matrix cores: exact-f32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain in k order).
//
// Per 32-target tile and edge type l (see the kernel comment for how the two phases are pipelined):
//   phase G  each wave folds the (v,l) buckets of 8 of the 32 targets exactly like seg_reduce_wave_kernel
//            (lanes across the Din features, row indices broadcast into SGPRs, 8 row loads in flight) and
//            parks the 32 aggregated rows in LDS (row stride Din+1 floats: conflict-free column reads);
//   phase M  the 32 x Din tile is multiplied with W_l [Din, Dout]: wave w owns output column tiles
//            w, w+4, ... (32 columns each).  A fragments come from LDS (ds_read_b32), B fragments straight
//            from global memory in a pre-packed MFMA order (one coalesced dwordx4 per lane = operands of 4
//            MFMAs): W_l is read once per workgroup by exactly one wave, so staging it in LDS buys nothing.
// Several workgroups per CU (33 KB LDS, <128 VGPRs) let one group's gather phase (memory-bound) overlap another
// group's MFMA phase (matrix-pipe-bound).  The [V, L*D] intermediate of the unfused path never exists.
//
// Packed weights (host side, once per step): Wp[l][nt][kq][lane][e] = W_l[kq*8 + e*2 + (lane>>5)][nt*32 + (lane&31)]
#include "common.h"

#include <stdlib.h>

using namespace relgnn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 32;
constexpr int GU = 8;

// One workgroup = 32 target nodes, 4 waves; per edge type: phase G (all waves gather), barrier, phase M (all waves
// MFMA), barrier.  MEASURED (C2, D=256): 286 us vs 227 us for GEMM + seg_reduce — the two phases do not overlap
// (all ~1000 workgroups are resident at once and run the same phase chip-wide: gather-only 148 us + MFMA-only
// 156 us), and a warp-specialised producer/consumer variant (8 waves, double-buffered LDS) was slower still
// (398 us) because only 4-8 gather waves per CU cannot keep enough row loads in flight.  Kept as the validated
// MFMA building block (parity-green); OFF by default until the producer side uses LDS-DMA.
template <int NCH, int TPW>
__global__ __launch_bounds__(256) void rgcn_fused_fwd_kernel(
    const float4* __restrict__ H, int64_t ldh4, int32_t Din, const int32_t* __restrict__ rowptr, int32_t V, int32_t L,
    const int32_t* __restrict__ src, const float* __restrict__ w, const float4* __restrict__ Wp, int32_t Dout,
    int32_t mode, int32_t act, float* __restrict__ out, int64_t ldo, int64_t nlb, int32_t ablate) {
  extern __shared__ float As[];  // [BM][Din + 1]
  const int64_t lb = xcd_logical_block(nlb);
  if (lb < 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v0 = lb * BM;
  const int D4 = Din / 4, astride = Din + 1;
  const int NT = Dout / 32, KQ = Din / 8;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  bool on[NCH];
  uint32_t cc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    on[c] = lane + 64 * c < D4;
    cc[c] = (uint32_t)min(lane + 64 * c, D4 - 1);
  }
  const uint32_t ld = (uint32_t)ldh4;

  for (int l = 0; l < L; ++l) {
    // ---- phase G: aggregate the (v,l) buckets of this wave's 8 targets into LDS -------------------
    for (int ii = 0; ii < ((ablate & 2) ? 0 : BM / 4); ++ii) {
      const int i = wave * (BM / 4) + ii;
      const int64_t v = v0 + i;
      float4 a[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < V) {
        const int beg = __builtin_amdgcn_readfirstlane(rowptr[v * L + l]);
        const int end = __builtin_amdgcn_readfirstlane(rowptr[v * L + l + 1]);
        for (int p = beg; p < end; p += 64) {
          const int n = min(64, end - p);
          const int my_col = (lane < n) ? src[p + lane] : 0;
          const float my_w = (w && lane < n) ? w[p + lane] : 1.f;
          for (int k = 0; k < n; k += GU) {
            const int rem = n - k;
            float4 t[GU][NCH];
            float ww[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
              const int ku = k + min(u, rem - 1);
              const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(my_col, ku);
              ww[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), ku));
              const float4* row = H + (size_t)(r * ld);
              if (u < rem) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) t[u][c] = row[cc[c]];
              }
            }
#pragma unroll
            for (int u = 0; u < GU; ++u)
              if (u < rem) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                  a[c].x += ww[u] * t[u][c].x; a[c].y += ww[u] * t[u][c].y;
                  a[c].z += ww[u] * t[u][c].z; a[c].w += ww[u] * t[u][c].w;
                }
              }
          }
        }
      }
      float* arow = As + i * astride;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (on[c]) {
          const int k4 = 4 * (lane + 64 * c);
          arow[k4] = a[c].x; arow[k4 + 1] = a[c].y; arow[k4 + 2] = a[c].z; arow[k4 + 3] = a[c].w;
        }
    }
    __syncthreads();
    // ---- phase M: acc[32 x Dout] += As[32 x Din] @ W_l ------------------------------------------------
    const int ai = lane & 31, kb = lane >> 5;
    const float* arow = As + ai * astride + kb;
    const float4* wl = Wp + (size_t)l * NT * KQ * 64;
    for (int kq = 0; kq < ((ablate & 1) ? 0 : KQ); ++kq) {
      // (a register double-buffer of the B operands was tried: 357 us vs 286 us, it costs occupancy)
      float4 b[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int nt = wave + 4 * t;
        b[t] = (nt < NT) ? wl[((size_t)nt * KQ + kq) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float a0 = arow[kq * 8 + 0], a1 = arow[kq * 8 + 2], a2 = arow[kq * 8 + 4], a3 = arow[kq * 8 + 6];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (wave + 4 * t < NT) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[t].x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[t].y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[t].z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b[t].w, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // As is overwritten by the next edge type
  }

  // ---- epilogue: mean / sqrt_n factor (total messages of the target over all types), activation, store ----
  // C layout of v_mfma_f32_32x32x2: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nt = wave + 4 * t;
    if (nt >= NT) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t v = v0 + row;
      if (v < V) {
        float x = acc[t][r];
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
        out[v * ldo + nt * 32 + (lane & 31)] = x;
      }
    }
  }
}

// Wp[l][nt][kq][lane][e] = W[l][kq*8 + e*2 + (lane>>5)][nt*32 + (lane&31)]
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ W, int32_t L, int32_t Din,
                                                           int32_t Dout, int64_t ldw, int64_t type_stride,
                                                           float* __restrict__ Wp) {
  const int NT = Dout / 32, KQ = Din / 8;
  const int64_t total = (int64_t)L * NT * KQ * 64 * 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t x = i;
    const int e = (int)(x & 3); x >>= 2;
    const int lane = (int)(x & 63); x >>= 6;
    const int kq = (int)(x % KQ); x /= KQ;
    const int nt = (int)(x % NT);
    const int l = (int)(x / NT);
    const int k = kq * 8 + e * 2 + (lane >> 5), n = nt * 32 + (lane & 31);
    Wp[i] = W[(int64_t)l * type_stride + (int64_t)k * ldw + n];
  }
}

}  // namespace

extern "C" {

// Packs L weight matrices W_l [Din, Dout] (element (k, n) of type l at W[l*type_stride + k*ldw + n]) into the MFMA
// operand order relgnn_rgcn_fused_fwd streams.  `packed` holds L*Din*Dout floats.
int relgnn_pack_type_weights(const float* W, int32_t num_edge_types, int32_t Din, int32_t Dout, int64_t ldw,
                             int64_t type_stride, float* packed, void* stream) {
  if (num_edge_types <= 0 || Din <= 0 || Dout <= 0 || ldw < Dout) return RELGNN_EINVAL;
  if (Din % 8 != 0 || Dout % 32 != 0) return RELGNN_EUNSUPPORTED;
  if (!W || !packed) return RELGNN_EINVAL;
  const int64_t total = (int64_t)num_edge_types * Din * Dout;
  pack_weights_kernel<<<flat_grid(total, 256), 256, 0, as_stream(stream)>>>(W, num_edge_types, Din, Dout, ldw,
                                                                            type_stride, packed);
  return launch_status();
}

int relgnn_rgcn_fused_fwd(int32_t mode, int32_t act, const float* H, int64_t ldh, int32_t Din, const int32_t* rowptr,
                          int32_t num_nodes, int32_t num_edge_types, const int32_t* src, const float* w,
                          const float* packed_weights, int32_t Dout, float* out, int64_t ldo, void* stream) {
  if (mode < RELGNN_AGG_SUM || mode > RELGNN_AGG_SQRT_N) return mode == RELGNN_AGG_MAX ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU || num_nodes < 0 || num_edge_types <= 0 || Din <= 0 || Dout <= 0 ||
      ldh < Din || ldo < Dout)
    return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!H || !rowptr || !packed_weights || !out) return RELGNN_EINVAL;
  if (Din % 8 != 0 || Dout % 32 != 0 || Din > 384 || Dout > 512 || ldh % 4 != 0 || !aligned16(H) ||
      !aligned16(packed_weights) || (int64_t)num_nodes * (ldh / 4) >= ((int64_t)1 << 32))
    return RELGNN_EUNSUPPORTED;
  const int64_t nlb = ((int64_t)num_nodes + BM - 1) / BM;
  const unsigned grid = (unsigned)(((nlb + 7) / 8) * 8);
  const size_t lds = (size_t)BM * (Din + 1) * sizeof(float);
  const int nch = Din / 4 <= 64 ? 1 : 2;
  const int tpw = (Dout / 32 + 3) / 4;
  hipStream_t st = as_stream(stream);
  static int ablate = -1;  // experiments only: RELGNN_FUSED_ABLATE bit0 = skip the MFMA phase, bit1 = skip the gather phase
  if (ablate < 0) { const char* e = getenv("RELGNN_FUSED_ABLATE"); ablate = e ? atoi(e) : 0; }
#define FUSED_LAUNCH(NN, TT)                                                                                         \
  rgcn_fused_fwd_kernel<NN, TT><<<grid, 256, lds, st>>>((const float4*)H, ldh / 4, Din, rowptr, num_nodes,           \
                                                         num_edge_types, src, w, (const float4*)packed_weights, Dout, \
                                                         mode, act, out, ldo, nlb, ablate)
#define FUSED_TPW(NN)                                  \
  switch (tpw) {                                       \
    case 1: FUSED_LAUNCH(NN, 1); break;                \
    case 2: FUSED_LAUNCH(NN, 2); break;                \
    case 3: FUSED_LAUNCH(NN, 3); break;                \
    default: FUSED_LAUNCH(NN, 4); break;               \
  }
  if (nch == 1) { FUSED_TPW(1) } else { FUSED_TPW(2) }
#undef FUSED_TPW
#undef FUSED_LAUNCH
  return launch_status();
}

}  // extern "C"
