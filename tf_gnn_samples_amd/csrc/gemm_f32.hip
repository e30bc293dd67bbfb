// Node-side dense layers on the matrix cores, EXACT fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bitwise an
// fmaf chain in k order; 157 TFLOP/s peak = the fp32 vector rate, MI355X_MICROARCH.md).
//
// Replaces the library GEMMs behind the Keras Dense layers of the path: the per-edge-type transforms
// H @ [W_0|..|W_{L-1}] (gnns/rgcn.py:70-74,98 evaluated node-side), the inter-layer Dense
// (models/sparse_graph_model.py:194-200) and their gradients.  Why not the library: every batch of an epoch has its
// own node count V, so a per-shape tuned solution (TunableOp) never recurs; the library's default picks run the
// [36 k, 256] x [256, 768]-class shapes of a C2 step at ~105 TFLOP/s on average, and a fixed tiling written for
// exactly these shapes (K, N multiples of 32/64, M arbitrary) does not depend on a per-shape heuristic.
//
//   C[M, N] = A[M, K] @ B[K, N]          three operand layouts (all row-major in memory):
//     NN  A [M, K],  B [K, N]                  forward            Y  = X @ W
//     NT  A [M, K],  B given as [N, K]         input gradient     dX = G @ W^T
//     TN  A given as [K, M], B [K, N]          weight gradient    dW = X^T @ G   (K = node dimension, split over
//                                                                  gridDim.z; partial products are summed by the caller)
//
// Tiling: 256 threads = 4 waves as 2 x 2; every wave owns TM x TN MFMA tiles of 32 x 32 (block tile 64*TM x 64*TN),
// BK = 32.  A k-pair of one MFMA is (j, j + 16) inside the 32-deep tile, so that a lane's operands for four
// consecutive MFMAs are 16 contiguous bytes of an LDS row (one ds_read_b128) for operands stored k-contiguous
// ("RM": [rows][32 + 4 pad], conflict-free for the b128 lane groups); k-major operands ("KM": [32][cols]) are read
// with lane-contiguous ds_read_b32.  Global -> registers -> LDS double buffering: the loads of tile t+1 are in flight
// while tile t is multiplied, one barrier per k-tile.
#include "common.h"

using namespace relgnn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;
constexpr int RM_STRIDE = BK + 4;   // floats; 144 B rows keep ds_read_b128 aligned and its 16-lane groups conflict-free

template <int ROWS>  // RM operand tile: element (x, k) at x * RM_STRIDE + k
struct RmTile { static constexpr int floats = ROWS * RM_STRIDE; };
template <int COLS>  // KM operand tile: element (x, k) at k * COLS + x
struct KmTile { static constexpr int floats = BK * COLS; };

// Workgroup rendezvous on LDS contents only: global loads issued before it stay in flight across it (__syncthreads()
// would also drain vmcnt).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- global -> registers (one k-tile of one operand; 256 threads) ---------------------------------------------------
// RM: X rows x 32 k.  float4 f = tid + 256 p: row = f / 8, c4 = f % 8.
// GUARD = false: the whole tile is inside the operand (wave-uniform decision by the caller): no per-lane branches.
template <int X, bool GUARD>
__device__ __forceinline__ void load_rm(float4 (&r)[X / 32], const float* __restrict__ base, int64_t ld, int x0, int x_lim,
                                        int k0, int k_lim, int tid) {
#pragma unroll
  for (int p = 0; p < X / 32; ++p) {
    const int f = tid + 256 * p;
    const int row = f >> 3, c4 = f & 7;
    const int x = x0 + row, k = k0 + c4 * 4;
    if (!GUARD || (x < x_lim && k < k_lim)) r[p] = *reinterpret_cast<const float4*>(base + (int64_t)x * ld + k);
    else r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int X>
__device__ __forceinline__ void store_rm(float* __restrict__ tile, const float4 (&r)[X / 32], int tid) {
#pragma unroll
  for (int p = 0; p < X / 32; ++p) {
    const int f = tid + 256 * p;
    *reinterpret_cast<float4*>(tile + (f >> 3) * RM_STRIDE + (f & 7) * 4) = r[p];
  }
}
// KM: 32 k-rows x X.  float4 f = tid + 256 p: krow = f / (X/4), c4 = f % (X/4).
template <int X, bool GUARD>
__device__ __forceinline__ void load_km(float4 (&r)[X / 32], const float* __restrict__ base, int64_t ld, int x0, int x_lim,
                                        int k0, int k_lim, int tid) {
#pragma unroll
  for (int p = 0; p < X / 32; ++p) {
    const int f = tid + 256 * p;
    const int krow = f / (X / 4), c4 = f % (X / 4);
    const int x = x0 + c4 * 4, k = k0 + krow;
    if (!GUARD || (x < x_lim && k < k_lim)) r[p] = *reinterpret_cast<const float4*>(base + (int64_t)k * ld + x);
    else r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int X>
__device__ __forceinline__ void store_km(float* __restrict__ tile, const float4 (&r)[X / 32], int tid) {
#pragma unroll
  for (int p = 0; p < X / 32; ++p) {
    const int f = tid + 256 * p;
    *reinterpret_cast<float4*>(tile + (f / (X / 4)) * X + (f % (X / 4)) * 4) = r[p];
  }
}

// operands of the four MFMAs j4*4 .. j4*4+3 for the 32-wide block starting at column/row xb of the tile
template <bool KM, int X>
__device__ __forceinline__ void fetch4(float (&o)[4], const float* __restrict__ tile, int xb, int j4, int lane) {
  const int x = xb + (lane & 31), h = lane >> 5;
  if constexpr (KM) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) o[jj] = tile[(h * 16 + j4 * 4 + jj) * X + x];
  } else {
    const float4 v = *reinterpret_cast<const float4*>(tile + x * RM_STRIDE + h * 16 + j4 * 4);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
}


// A_KM: A[m][k] stored at A + k*lda + m (TN), else A + m*lda + k.  B_KM: B[k][n] at B + k*ldb + n (NN / TN), else
// B + n*ldb + k (NT).  gridDim.z > 1: split over K in chunks of k_chunk, slice z writes C + z * M * ldc.
template <int TM, int TN, bool A_KM, bool B_KM, int ACT, bool HAS_BIAS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                       int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       int k_chunk, const float* __restrict__ bias, int64_t n_logical) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_FLOATS = A_KM ? KmTile<BM>::floats : RmTile<BM>::floats;
  constexpr int B_FLOATS = B_KM ? KmTile<BN>::floats : RmTile<BN>::floats;
  __shared__ __attribute__((aligned(16))) float lds[2 * (A_FLOATS + B_FLOATS)];
  const int64_t lb = xcd_logical_block(n_logical);
  if (lb < 0) return;
  const int tiles_n = (N + BN - 1) / BN;
  const int m0 = (int)(lb / tiles_n) * BM, n0 = (int)(lb % tiles_n) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN);
  const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
  float* Cz = C + (int64_t)blockIdx.z * M * ldc;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[BM / 32], rb[BN / 32];
  const bool interior = m0 + BM <= M && n0 + BN <= N;          // block-uniform: the common case takes no lane branches
  auto gload = [&](int k0) {
    if (interior && k0 + BK <= kend) {
      if constexpr (A_KM) load_km<BM, false>(ra, A, lda, m0, M, k0, kend, tid); else load_rm<BM, false>(ra, A, lda, m0, M, k0, kend, tid);
      if constexpr (B_KM) load_km<BN, false>(rb, B, ldb, n0, N, k0, kend, tid); else load_rm<BN, false>(rb, B, ldb, n0, N, k0, kend, tid);
    } else {
      if constexpr (A_KM) load_km<BM, true>(ra, A, lda, m0, M, k0, kend, tid); else load_rm<BM, true>(ra, A, lda, m0, M, k0, kend, tid);
      if constexpr (B_KM) load_km<BN, true>(rb, B, ldb, n0, N, k0, kend, tid); else load_rm<BN, true>(rb, B, ldb, n0, N, k0, kend, tid);
    }
  };
  auto lstore = [&](int buf) {
    float* ta = lds + buf * (A_FLOATS + B_FLOATS);
    float* tb = ta + A_FLOATS;
    if constexpr (A_KM) store_km<BM>(ta, ra, tid); else store_rm<BM>(ta, ra, tid);
    if constexpr (B_KM) store_km<BN>(tb, rb, tid); else store_rm<BN>(tb, rb, tid);
  };

  // One k-tile = four groups of 4 k-pairs (16 MFMAs per wave and group at TM = TN = 2).  The loop keeps the matrix pipe
  // fed without a serial segment: everything that is not an MFMA is issued at the head of a group and completes under that
  // group's MFMAs, and the only rendezvous per k-tile sits between groups 2 and 3:
  //   group 0 | fragments of group 1 <- LDS;  tile t+1 (in registers since the previous iteration) -> the other LDS buffer
  //   group 1 | fragments of group 2 <- LDS;  global loads of tile t+2 -> registers
  //   group 2 | fragments of group 3 <- LDS
  //   -- wait for own LDS traffic, s_barrier: tile t+1 is complete in LDS, nobody reads tile t from LDS any more --
  //   group 3 | fragments of group 0 of tile t+1 <- LDS
  // Measured by ablation on random operands: k-loop with MFMAs only 109-119 TFLOP/s (the sustained
  // clock under fp32 MFMA load, not 2.4 GHz, sets that ceiling), + LDS traffic 96-105, + global loads 87-92.
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  float fa[2][TM][4], fb[2][TN][4];
  auto frags = [&](int buf, int tile, int group) {
    const float* ta = lds + (tile & 1) * (A_FLOATS + B_FLOATS);
    const float* tb = ta + A_FLOATS;
#pragma unroll
    for (int a = 0; a < TM; ++a) fetch4<A_KM, BM>(fa[buf][a], ta, wm + 32 * a, group, lane);
#pragma unroll
    for (int b = 0; b < TN; ++b) fetch4<B_KM, BN>(fb[buf][b], tb, wn + 32 * b, group, lane);
  };
  auto mfma_group = [&](int buf) {
    __builtin_amdgcn_sched_barrier(0);            // the group's loads / stores stay above its MFMA block
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][a][jj], fb[buf][b][jj], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  if (ntiles > 0) {
    gload(kbeg);
    lstore(0);
    if (ntiles > 1) gload(kbeg + BK);
    lds_barrier();
    frags(0, 0, 0);
  }
  for (int t = 0; t < ntiles; ++t) {
    frags(1, t, 1);
    if (t + 1 < ntiles) lstore((t + 1) & 1);
    mfma_group(0);
    frags(0, t, 2);
    if (t + 2 < ntiles) gload(kbeg + (t + 2) * BK);
    mfma_group(1);
    frags(1, t, 3);
    mfma_group(0);
    lds_barrier();
    if (t + 1 < ntiles) frags(0, t + 1, 0);
    mfma_group(1);
  }

  // C layout of v_mfma_f32_32x32x2: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // Interior tiles store without lane branches (a guarded store per element made the compiler wait for memory before
  // every single store: the epilogue then took as long as the whole k-loop).
  const int colbase = n0 + wn + (lane & 31);
  const int rowbase = m0 + wm + 4 * (lane >> 5);
  float bv[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    bv[b] = 0.f;
    if constexpr (HAS_BIAS) bv[b] = bias[min(colbase + 32 * b, N - 1)];
  }
  if (interior) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        float* cp = Cz + (int64_t)(rowbase + 32 * a) * ldc + colbase + 32 * b;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          cp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = act_fwd<ACT>(acc[a][b][r] + bv[b]);
      }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = colbase + 32 * b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rowbase + 32 * a + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) Cz[(int64_t)row * ldc + col] = act_fwd<ACT>(acc[a][b][r] + bv[b]);
        }
      }
  }
}

template <int TM, int TN, bool A_KM, bool B_KM>
int launch(int act, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
           int splits, int k_chunk, const float* bias, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  dim3 grid((unsigned)(((tiles + 7) / 8) * 8), 1, (unsigned)splits);
#define RELGNN_GEMM_LAUNCH(ACT_, BIAS_)                                                                                  \
  gemm_f32_kernel<TM, TN, A_KM, B_KM, ACT_, BIAS_><<<grid, 256, 0, st>>>(A, lda, B, ldb, C, ldc, M, N, K, k_chunk, bias, tiles)
  if constexpr (!A_KM && B_KM) {          // the forward layout carries the Dense epilogue (bias, activation)
    if (bias) { RELGNN_DISPATCH_ACT(act, ACT, (RELGNN_GEMM_LAUNCH(ACT, true))); }
    else { RELGNN_DISPATCH_ACT(act, ACT, (RELGNN_GEMM_LAUNCH(ACT, false))); }
  } else {
    if (bias || act != RELGNN_ACT_LINEAR) return RELGNN_EUNSUPPORTED;
    RELGNN_GEMM_LAUNCH(RELGNN_ACT_LINEAR, false);
  }
#undef RELGNN_GEMM_LAUNCH
  return launch_status();
}

// tile choice: largest tile whose grid still fills the 256 CUs about twice (2-3 workgroups are resident per CU)
template <bool A_KM, bool B_KM>
int dispatch(int act, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
             int splits, int k_chunk, const float* bias, hipStream_t st) {
  auto tiles = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * splits; };
  // (a split-K launch is sized by the caller for 128 x 128 tiles: at most ~2 workgroups per CU in ONE wave of blocks)
  if (N % 128 == 0 && (tiles(128, 128) >= 1024 || (splits > 1 && M >= 128))) return launch<2, 2, A_KM, B_KM>(act, A, lda, B, ldb, C, ldc, M, N, K, splits, k_chunk, bias, st);
  if (tiles(128, 64) >= 768) return launch<2, 1, A_KM, B_KM>(act, A, lda, B, ldb, C, ldc, M, N, K, splits, k_chunk, bias, st);
  return launch<1, 1, A_KM, B_KM>(act, A, lda, B, ldb, C, ldc, M, N, K, splits, k_chunk, bias, st);
}

bool vec_ok(const void* p, int64_t ld) { return aligned16(p) && ld % 4 == 0; }

}  // namespace

extern "C" {

int relgnn_gemm_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                    float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t k_splits, void* stream) {
  if (layout < RELGNN_GEMM_NN || layout > RELGNN_GEMM_TN || M < 0 || N < 0 || K < 0 || k_splits < 1) return RELGNN_EINVAL;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C) return RELGNN_EINVAL;
  if (!vec_ok(A, lda) || !vec_ok(B, ldb) || !vec_ok(C, ldc) || N % 4 != 0 || ldc < N) return RELGNN_EUNSUPPORTED;
  if (layout == RELGNN_GEMM_TN ? (M % 4 != 0) : (K % 4 != 0)) return RELGNN_EUNSUPPORTED;
  if (k_splits > 1 && (bias || act != RELGNN_ACT_LINEAR)) return RELGNN_EINVAL;     // partial products carry no epilogue
  int k_chunk = K;
  if (k_splits > 1) {
    k_chunk = (int)(((int64_t)K + k_splits - 1) / k_splits);
    k_chunk = (k_chunk + BK - 1) / BK * BK;
  }
  if (k_chunk < BK) k_chunk = BK;
  hipStream_t st = as_stream(stream);
  if (K == 0) {
    for (int z = 0; z < k_splits; ++z)
      if (hipMemset2DAsync(C + (int64_t)z * M * ldc, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  switch (layout) {
    case RELGNN_GEMM_NN: return dispatch<false, true>(act, A, lda, B, ldb, C, ldc, M, N, K, k_splits, k_chunk, bias, st);
    case RELGNN_GEMM_NT: return dispatch<false, false>(act, A, lda, B, ldb, C, ldc, M, N, K, k_splits, k_chunk, bias, st);
    default: return dispatch<true, true>(act, A, lda, B, ldb, C, ldc, M, N, K, k_splits, k_chunk, bias, st);
  }
}

}  // extern "C"
