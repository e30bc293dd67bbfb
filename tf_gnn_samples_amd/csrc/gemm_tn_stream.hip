// Weight gradients of the node-side Dense layers:  dW[M, N] = X[V, M]^T @ G[V, N]   (the MatMul gradient TF derives for
// every Keras Dense of the path: models/sparse_graph_model.py:165-172,194-200, tasks/ppi_task.py:176-179, gnns/rgcn.py:70-74).
//
// Shape: the reduction runs over the NODE dimension (V ~ 3e4 .. 1e6) and the output is tiny (50..768 x 121..256), so the
// product is a stream over two tall operands with 32-64 flops per byte.  A tiled LDS GEMM is the wrong tool: split over V
// to fill the chip, each workgroup sees a K of a few hundred rows and spends its time in prologue, barriers and epilogue
// (hipBLASLt strided-batched + sum: 147 us for [36 k, 256]^T @ [36 k, 256] = 33 TFLOP/s; relgnn_gemm_f32 TN: 100 us).
//
// Here every WAVE owns one 64 x 64 output tile for one chunk of rows and needs neither LDS nor barriers: both operands are
// row-major with the reduction index as the ROW, which is exactly the operand layout of v_mfma_f32_32x32x2_f32 (lane l
// supplies element [k = l >> 5][x = l & 31]).  A lane loads TWO adjacent columns of its row (one 8-byte load per operand
// and k-pair), which feeds 2 x 2 MFMA tiles whose rows / columns interleave (tile j covers x = 2 i + j); the loads of the
// next 8 k-pairs are in flight under the MFMAs of the current ones (register ring, vmcnt-counted).  The 4 waves of a
// workgroup take 4 neighbouring column tiles of the same row chunk, so the A rows they share hit in L1.
// Partial products go to a caller-provided workspace [chunks, M, N]; a second kernel sums them in chunk order
// (deterministic, no atomics).  Exact fp32 (f32 operands, f32 accumulate).
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

using namespace relgnn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kDepth = 8;   // k-pairs in flight per wave

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Per-lane column state of one operand: the two adjacent columns x, x + 1 this lane feeds, as load offsets that are always
// inside the row plus two 0/1 factors — the k-loop has no lane branches (rows past the chunk likewise: the row pointer is
// clamped to the chunk's last row and the pair multiplied by 0 when it is consumed).
struct Cols {
  int off0, off1;        // column offsets actually loaded (clamped into [0, X))
  float on0, on1;        // 1.f where the column exists
};
__device__ __forceinline__ Cols make_cols(int x, int X) {
  Cols c;
  c.on0 = x < X ? 1.f : 0.f;
  c.on1 = x + 1 < X ? 1.f : 0.f;
  c.off0 = min(x, X - 1);
  c.off1 = min(x + 1, X - 1);
  return c;
}

// One ring slot = the two columns of one row.  The loads are issued as inline assembly and waited for with an explicit
// s_waitcnt: left to the compiler, loads that stay in flight across the loop back-edge are drained with vmcnt(0) at the loop
// head (its wait-count analysis merges the back-edge conservatively), which exposes the full memory latency once per
// iteration.  VEC: one 8-byte load (row stride even, base 8-byte aligned); otherwise two 4-byte loads.
template <bool VEC> struct Slot;
template <> struct Slot<true> { f32x2 v; };
template <> struct Slot<false> { float x, y; };

// The destination is a READ-WRITE operand ("+v"): the load lands asynchronously, so the slot must stay in ONE physical
// register from the issue to the wait — with a write-only operand the allocator is free to give the load a fresh register
// and copy it into place before the data has arrived.  A value still needed from the slot (the pair just consumed) is
// thereby copied out by the compiler before the load is issued.
__device__ __forceinline__ void issue(Slot<true>& s, const float* row, const Cols& c) {
  const float* p = row + (c.on0 != 0.f ? c.off0 : 0);      // even offset; x + 1 is inside the row's storage when x exists
  asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(s.v) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue(Slot<false>& s, const float* row, const Cols& c) {
  const float* p0 = row + c.off0;
  const float* p1 = row + c.off1;
  asm volatile("global_load_dword %0, %1, off" : "+v"(s.x) : "v"(p0) : "memory");
  asm volatile("global_load_dword %0, %1, off" : "+v"(s.y) : "v"(p1) : "memory");
}
// wait until at most N vector-memory loads are outstanding; the slots are operands so that their uses stay behind the wait
template <int N>
__device__ __forceinline__ void wait_for(Slot<true>& a, Slot<true>& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a.v), "+v"(b.v) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_for(Slot<true>& a, Slot<false>& b) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a.v), "+v"(b.x), "+v"(b.y) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_for(Slot<false>& a, Slot<true>& b) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a.x), "+v"(a.y), "+v"(b.v) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_for(Slot<false>& a, Slot<false>& b) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y) : "n"(N)); }
// Copy a landed pair OUT of its slot with explicit moves, so that the slot's old value is dead when the refill is issued
// into it: otherwise the allocator keeps the old value where it is, gives the (asynchronous) refill a fresh register and
// moves it back into place right away — before the data has arrived.
__device__ __forceinline__ void take(const Slot<true>& s, float& x, float& y) {
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(x), "=&v"(y) : "v"(s.v.x), "v"(s.v.y));
}
__device__ __forceinline__ void take(const Slot<false>& s, float& x, float& y) {
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(x), "=&v"(y) : "v"(s.x), "v"(s.y));
}

// One wave's 64 x 64 tile of one chunk.  colsum (nullable, [2, N] of this chunk): the sums of the chunk's rows of B by row parity —
// the bias gradient of the Dense layer whose kernel gradient this product is (column sums of the same operand), taken from the
// registers the MFMAs read; written by the waves of the first tile row only.
template <bool VEC_A, bool VEC_B>
__device__ __forceinline__ void tn_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int64_t K,
                                        int32_t M, int32_t N, int64_t chunk_rows, float* __restrict__ out, int chunk, int m0, int n0,
                                        float* __restrict__ colsum) {
  const int lane = threadIdx.x & 63;
  const int64_t k_begin = (int64_t)chunk * chunk_rows, k_end = min(K, k_begin + chunk_rows);
  const Cols ca = make_cols(m0 + 2 * (lane & 31), M), cb = make_cols(n0 + 2 * (lane & 31), N);
  const int rows = (int)(k_end - k_begin);                     // <= chunk_rows < 2^31
  const int half = lane >> 5;                                  // this lane's row inside a k-pair

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // per-lane row pointers, advanced by two rows per k-pair and clamped to the chunk's last row
  const float* pa = A + (k_begin + min(half, rows - 1)) * lda;
  const float* pb = B + (k_begin + min(half, rows - 1)) * ldb;
  const float* pa_last = A + (k_end - 1) * lda;
  const float* pb_last = B + (k_end - 1) * ldb;
  const int64_t step_a = 2 * lda, step_b = 2 * ldb;
  auto advance = [&]() {
    pa = pa + step_a < pa_last ? pa + step_a : pa_last;
    pb = pb + step_b < pb_last ? pb + step_b : pb_last;
  };

  float cs0 = 0.f, cs1 = 0.f;                                   // this lane's two columns of B, summed over its rows (half = parity)
  constexpr int kLoadsPerPair = (VEC_A ? 1 : 2) + (VEC_B ? 1 : 2);
  Slot<VEC_A> ra[kDepth] = {};
  Slot<VEC_B> rb[kDepth] = {};
#pragma unroll
  for (int u = 0; u < kDepth; ++u) {
    issue(ra[u], pa, ca);
    issue(rb[u], pb, cb);
    advance();
  }
  // One ring step: consume the OLDEST pair in flight (slot u; everything issued after it may stay in flight), refill its
  // slot with the pair kDepth ahead.  LEAN: the pair and its refill lie wholly inside the chunk and the tile is interior,
  // so there is nothing to clamp or mask — 2 pointer adds, 2 loads, 1 wait and 4 MFMAs per k-pair (the masked form issues
  // ~14 instructions per MFMA and kept the matrix pipe 50 % busy, profiles/r02_c_tn_stream_pmc.txt).
  auto ring_pass = [&](int r0, auto lean_tag) {
    constexpr bool LEAN = decltype(lean_tag)::value;
#pragma unroll
    for (int u = 0; u < kDepth; ++u) {
      wait_for<kLoadsPerPair * (kDepth - 1)>(ra[u], rb[u]);
      float a0, a1, b0, b1;
      take(ra[u], a0, a1);
      take(rb[u], b0, b1);
      if constexpr (!LEAN) {
        const float live = r0 + 2 * u + half < rows ? 1.f : 0.f;
        a0 *= live * ca.on0; a1 *= live * ca.on1;
        b0 *= cb.on0; b1 *= cb.on1;
        cs0 += live * b0; cs1 += live * b1;
      } else {
        cs0 += b0; cs1 += b1;
      }
      issue(ra[u], pa, ca);
      issue(rb[u], pb, cb);
      if constexpr (LEAN) { pa += step_a; pb += step_b; } else advance();
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  // rows the lean passes may cover: every pair consumed AND every pair refilled (kDepth pairs later) has both rows in the
  // chunk.  Edge tiles (columns to mask) take the masked form throughout.
  const bool interior = m0 + 64 <= M && n0 + 64 <= N;
  const int lean_rows = interior ? ((rows / 2 - kDepth) / kDepth) * (2 * kDepth) : 0;
  int r0 = 0;
  for (; r0 < lean_rows; r0 += 2 * kDepth) ring_pass(r0, std::true_type{});
  pa = pa < pa_last ? pa : pa_last;      // the lean passes advance unclamped: the next refill may be the pair past the end
  pb = pb < pb_last ? pb : pb_last;
  for (; r0 < rows; r0 += 2 * kDepth) ring_pass(r0, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the refills past the end (clamped, never used)

  // (a masked element is multiplied by 0: it is a REAL element of the matrix — clamped address — so it is finite whenever
  // the matrix is.)
  // C layout of v_mfma_f32_32x32x2: column index c = lane & 31, row index i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  // tile (a, b) holds output rows m0 + 2 i + a and columns n0 + 2 c + b.
  if (colsum && m0 == 0) {
    const int n = n0 + 2 * (lane & 31);
    if (n < N) colsum[(int64_t)half * N + n] = cs0;
    if (n + 1 < N) colsum[(int64_t)half * N + n + 1] = cs1;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + a;
      const int n = n0 + 2 * (lane & 31);
      if (m < M) {
        if (n < N) out[(int64_t)m * N + n] = acc[a][0][r];
        if (n + 1 < N) out[(int64_t)m * N + n + 1] = acc[a][1][r];
      }
    }
}

template <bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(256) void gemm_tn_stream_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                             int64_t ldb, int64_t K, int32_t M, int32_t N, int64_t chunk_rows,
                                                             float* __restrict__ partial, int32_t tiles_n, int32_t tiles) {
  // (grid: chunk-major.  A chunk's tile groups as neighbouring blocks of ONE XCD, so that they share its L2, measured the same:
  //  [50 k, 128]^T [50 k, 640] 93.6 against 94.3 us, scripts/exp_tn_xcd.py at commit "XCD-grouped TN stream grid")
  const int chunk = blockIdx.x;
  const int tile = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (tile >= tiles) return;                                   // whole wave
  tn_tile<VEC_A, VEC_B>(A, lda, B, ldb, K, M, N, chunk_rows, partial + (int64_t)chunk * M * N, chunk, (tile / tiles_n) * 64,
                        (tile % tiles_n) * 64, nullptr);
}

// Up to four products over the SAME rows in one launch (the three weight gradients of a GRU cell, gnns/ggnn.py:92: x^T gxk,
// h^T gxk[:, :2u], (r * h)^T gxk[:, 2u:]): the tiles of all of them are one list, cut into the same row chunks.
constexpr int kGroupMax = 4;
struct TnGroupArgs {
  const float* A[kGroupMax]; const float* B[kGroupMax];
  int64_t lda[kGroupMax], ldb[kGroupMax];
  float* partial[kGroupMax];              // [chunks, M, N] each
  int32_t M[kGroupMax], N[kGroupMax], tiles_n[kGroupMax], tile_end[kGroupMax];      // tile_end: running total of 64 x 64 tiles
  int32_t num, tiles;
  int64_t K, chunk_rows;
  float* colsum;                          // nullable: [chunks, 2, N[0]] partial column sums of B[0]
};

__global__ __launch_bounds__(256) void gemm_tn_stream_group_kernel(const TnGroupArgs g) {
  const int chunk = blockIdx.x;
  int tile = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (tile >= g.tiles) return;                                 // whole wave
  int p = 0;
  while (p + 1 < g.num && tile >= g.tile_end[p]) ++p;          // (wave-uniform; num <= 4)
  p = __builtin_amdgcn_readfirstlane(p);
  if (p > 0) tile -= g.tile_end[p - 1];
  const int32_t M = g.M[p], N = g.N[p];
  tn_tile<true, true>(g.A[p], g.lda[p], g.B[p], g.ldb[p], g.K, M, N, g.chunk_rows, g.partial[p] + (int64_t)chunk * M * N, chunk,
                      (tile / g.tiles_n[p]) * 64, (tile % g.tiles_n[p]) * 64,
                      p == 0 && g.colsum ? g.colsum + (int64_t)chunk * 2 * N : nullptr);
}

// the reductions of such a group in one launch: segment j sums chunks[j] slabs of mn[j] floats into C[j] (row stride ldc[j]); the
// column sums are a segment of their own (one row, 2 * chunks slabs).  Blocks of 64 outputs never straddle segments.
struct SumGroupArgs {
  const float* partial[kGroupMax + 1]; float* C[kGroupMax + 1];
  int64_t ldc[kGroupMax + 1];
  int32_t mn[kGroupMax + 1], N[kGroupMax + 1], chunks[kGroupMax + 1], block_end[kGroupMax + 1];
  int32_t num;
};

__global__ __launch_bounds__(256) void sum_partials_group_kernel(const SumGroupArgs g) {
  __shared__ float part[4][64];
  int j = 0;
  while (j + 1 < g.num && (int)blockIdx.x >= g.block_end[j]) ++j;
  const int block = (int)blockIdx.x - (j ? g.block_end[j - 1] : 0);
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = block * 64 + o;
  const int mn = g.mn[j], chunks = g.chunks[j];
  const float* partial = g.partial[j];
  float s0 = 0.f, s1 = 0.f;
  if (i < mn) {
    int c = grp;
    for (; c + 4 < chunks; c += 8) {
      s0 += partial[(int64_t)c * mn + i];
      s1 += partial[(int64_t)(c + 4) * mn + i];
    }
    if (c < chunks) s0 += partial[(int64_t)c * mn + i];
  }
  part[grp][o] = s0 + s1;
  __syncthreads();
  if (grp == 0 && i < mn) g.C[j][(int64_t)(i / g.N[j]) * g.ldc[j] + (i % g.N[j])] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
}

// C[m, n] (+)= sum over chunks.  256 threads = 64 outputs x 4 chunk groups: group g sums chunks g, g + 4, ... (independent
// loads in flight instead of one long dependent chain per output), the four group sums are added in group order through
// LDS — a fixed order, so the result is deterministic.  The output may be a row of column BLOCKS, each a matrix of its own
// (bcols columns wide, bstride floats apart, row stride ldc): the L per-type weight gradients of one [M, L * bcols] product.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int64_t mn, int32_t chunks, int32_t N,
                                                           float* __restrict__ C, int64_t ldc, int32_t accumulate, int32_t bcols,
                                                           int64_t bstride) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f;
  if (i < mn) {
    int c = g;
    for (; c + 4 < chunks; c += 8) {
      s0 += partial[(int64_t)c * mn + i];
      s1 += partial[(int64_t)(c + 4) * mn + i];
    }
    if (c < chunks) s0 += partial[(int64_t)c * mn + i];
  }
  part[g][o] = s0 + s1;
  __syncthreads();
  if (g == 0 && i < mn) {
    const float s = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    const int n = (int)(i % N);
    float* dst = C + (n / bcols) * bstride + (i / N) * ldc + (n % bcols);
    *dst = accumulate ? *dst + s : s;
  }
}

// C[m, n] = sum_z slabs[z][m][n] (slab order) + sum_{r < R} At[r][m] * Bt[r][n]: the reduction of a split-K weight gradient
// whose chunks left R < 64 rows over (V = S * c + R), folded into the pass that sums the partial products anyway
// (library strided-batched GEMM + torch.sum + a second product for the leftover rows were three launches).
__global__ __launch_bounds__(256) void sum_slabs_tail_kernel(const float* __restrict__ slabs, int32_t S, int64_t mn, int32_t N,
                                                             const float* __restrict__ At, int64_t lda,
                                                             const float* __restrict__ Bt, int64_t ldb, int32_t R,
                                                             float* __restrict__ C) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= mn) return;
  float s0 = 0.f, s1 = 0.f;
  int z = 0;
  for (; z + 1 < S; z += 2) {
    s0 += slabs[(int64_t)z * mn + i];
    s1 += slabs[(int64_t)(z + 1) * mn + i];
  }
  if (z < S) s0 += slabs[(int64_t)z * mn + i];
  float s = s0 + s1;
  const int64_t m = i / N, n = i - m * N;
  for (int r = 0; r < R; ++r) s = fmaf(At[(int64_t)r * lda + m], Bt[(int64_t)r * ldb + n], s);
  C[i] = s;
}

struct Plan {
  int32_t tiles_m, tiles_n, tiles, chunks;
  int64_t chunk_rows;
};

Plan make_plan(int32_t M, int32_t N, int64_t K) {
  Plan p;
  p.tiles_m = (M + 63) / 64;
  p.tiles_n = (N + 63) / 64;
  p.tiles = p.tiles_m * p.tiles_n;
  // at least 64 rows per chunk, chunk length a multiple of the unrolled step
  // one wave per SIMD, one round of workgroups (256 CUs x 4): measured at [36 k, 256]^T @ [36 k, 256 | 121 | 50-row]:
  // 512 waves 94 / 65 / 39 us, 768: 70 / 50 / 33, 1024: 59 / 43 / 33, 1280: 84 / 59 / 41, 2048: 64 / 49 / 41, 4096: 73 / 64 / 49
  static const int64_t waves = [] { const char* e = getenv("RELGNN_TN_WAVES"); return e ? (int64_t)atoi(e) : (int64_t)1024; }();
  // whole workgroups (four tiles of a chunk), and never one more than the round holds: 20 tiles ([128, 640]) x 52 chunks are 260
  // workgroups — four CUs run two of them and the launch takes twice as long (155 us against 94: scripts/exp_tn_xcd.py)
  const int64_t tgroups = (p.tiles + 3) / 4;
  int64_t chunks = std::max<int64_t>(1, (waves / 4) / tgroups);
  chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (K + 63) / 64));
  int64_t rows = (K + chunks - 1) / chunks;
  rows = (rows + 2 * kDepth - 1) / (2 * kDepth) * (2 * kDepth);
  p.chunk_rows = rows;
  p.chunks = (int32_t)((K + rows - 1) / rows);
  return p;
}

}  // namespace

extern "C" {

int64_t relgnn_gemm_tn_stream_workspace_bytes(int32_t M, int32_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = make_plan(M, N, K);
  return (int64_t)p.chunks * M * N * 4;
}

static int tn_stream_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                            int64_t K, int32_t bcols, int64_t bstride, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  hipStream_t st = as_stream(stream);
  if (K == 0) {
    if (accumulate) return RELGNN_OK;
    for (int32_t b = 0; b < N / bcols; ++b)
      if (hipMemset2DAsync(C + b * bstride, (size_t)ldc * 4, 0, (size_t)bcols * 4, (size_t)M, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!A || !B || !workspace) return RELGNN_EINVAL;
  if (K >= ((int64_t)1 << 40)) return RELGNN_EUNSUPPORTED;
  const Plan p = make_plan(M, N, K);
  if (workspace_bytes < (int64_t)p.chunks * M * N * 4) return RELGNN_EINVAL;
  float* partial = static_cast<float*>(workspace);
  const dim3 grid((unsigned)p.chunks, (unsigned)((p.tiles + 3) / 4));
  const bool va = (reinterpret_cast<uintptr_t>(A) % 8 == 0) && lda % 2 == 0;
  const bool vb = (reinterpret_cast<uintptr_t>(B) % 8 == 0) && ldb % 2 == 0;
#define RELGNN_TN_LAUNCH(VA, VB) \
  gemm_tn_stream_kernel<VA, VB><<<grid, 256, 0, st>>>(A, lda, B, ldb, K, M, N, p.chunk_rows, partial, p.tiles_n, p.tiles)
  if (va && vb) RELGNN_TN_LAUNCH(true, true);
  else if (va) RELGNN_TN_LAUNCH(true, false);
  else if (vb) RELGNN_TN_LAUNCH(false, true);
  else RELGNN_TN_LAUNCH(false, false);
#undef RELGNN_TN_LAUNCH
  const int64_t mn = (int64_t)M * N;
  sum_partials_kernel<<<(unsigned)((mn + 63) / 64), 256, 0, st>>>(partial, mn, p.chunks, N, C, ldc, accumulate, bcols, bstride);
  return launch_status();
}

int relgnn_gemm_tn_stream_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                              int64_t K, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0 || lda < M || ldb < N || ldc < N) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!C) return RELGNN_EINVAL;
  return tn_stream_launch(A, lda, B, ldb, C, ldc, M, N, K, N, 0, accumulate, workspace, workspace_bytes, stream);
}

int relgnn_gemm_tn_stream_blocks_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                     int64_t block_stride, int32_t M, int32_t N, int32_t block_cols, int64_t K, int32_t accumulate,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0 || block_cols <= 0 || N % block_cols || lda < M || ldb < N || ldc < block_cols || block_stride < 0)
    return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!C) return RELGNN_EINVAL;
  return tn_stream_launch(A, lda, B, ldb, C, ldc, M, N, K, block_cols, block_stride, accumulate, workspace, workspace_bytes, stream);
}

static Plan make_group_plan(int32_t num, const int32_t* M, const int32_t* N, int64_t K, int32_t* tile_end) {
  Plan p{};
  for (int i = 0; i < num; ++i) {
    p.tiles += ((M[i] + 63) / 64) * ((N[i] + 63) / 64);
    if (tile_end) tile_end[i] = p.tiles;
  }
  const int64_t tgroups = (p.tiles + 3) / 4;
  int64_t chunks = std::max<int64_t>(1, 256 / tgroups);
  chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (K + 63) / 64));
  int64_t rows = (K + chunks - 1) / chunks;
  rows = (rows + 2 * kDepth - 1) / (2 * kDepth) * (2 * kDepth);
  p.chunk_rows = rows;
  p.chunks = (int32_t)((K + rows - 1) / rows);
  return p;
}

static bool group_shapes_ok(int32_t num, const int32_t* M, const int32_t* N, int64_t K) {
  if (num < 1 || num > kGroupMax || !M || !N || K < 0) return false;
  for (int i = 0; i < num; ++i)
    if (M[i] <= 0 || N[i] <= 0) return false;
  return true;
}

int64_t relgnn_gemm_tn_stream_group_workspace_bytes(int32_t num, const int32_t* M, const int32_t* N, int64_t K, int32_t with_colsum) {
  if (!group_shapes_ok(num, M, N, K) || K == 0) return 0;
  const Plan p = make_group_plan(num, M, N, K, nullptr);
  int64_t floats = with_colsum ? (int64_t)p.chunks * 2 * N[0] : 0;
  for (int i = 0; i < num; ++i) floats += (int64_t)p.chunks * M[i] * N[i];
  return floats * 4;
}

int relgnn_gemm_tn_stream_group_f32(int32_t num, const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                    float* const* C, const int64_t* ldc, const int32_t* M, const int32_t* N, int64_t K,
                                    float* colsum0, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!group_shapes_ok(num, M, N, K) || !A || !lda || !B || !ldb || !C || !ldc) return RELGNN_EINVAL;
  for (int i = 0; i < num; ++i)
    if (!C[i] || lda[i] < M[i] || ldb[i] < N[i] || ldc[i] < N[i]) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  if (K == 0) {
    for (int i = 0; i < num; ++i)
      if (hipMemset2DAsync(C[i], (size_t)ldc[i] * 4, 0, (size_t)N[i] * 4, (size_t)M[i], st) != hipSuccess) return RELGNN_EHIP;
    if (colsum0 && hipMemsetAsync(colsum0, 0, (size_t)N[0] * 4, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!workspace || K >= ((int64_t)1 << 40)) return workspace ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  for (int i = 0; i < num; ++i) {
    if (!A[i] || !B[i]) return RELGNN_EINVAL;
    if (reinterpret_cast<uintptr_t>(A[i]) % 8 || reinterpret_cast<uintptr_t>(B[i]) % 8 || lda[i] % 2 || ldb[i] % 2 || M[i] % 64 || N[i] % 64)
      return RELGNN_EUNSUPPORTED;                               // (8-byte loads and whole tiles only: the caller launches such products one by one)
  }
  if (workspace_bytes < relgnn_gemm_tn_stream_group_workspace_bytes(num, M, N, K, colsum0 != nullptr)) return RELGNN_EINVAL;
  TnGroupArgs g{};
  SumGroupArgs sg{};
  const Plan p = make_group_plan(num, M, N, K, g.tile_end);
  float* ws = static_cast<float*>(workspace);
  int blocks = 0;
  for (int i = 0; i < num; ++i) {
    g.A[i] = A[i]; g.B[i] = B[i]; g.lda[i] = lda[i]; g.ldb[i] = ldb[i]; g.M[i] = M[i]; g.N[i] = N[i]; g.tiles_n[i] = (N[i] + 63) / 64;
    g.partial[i] = ws;
    sg.partial[i] = ws; sg.C[i] = C[i]; sg.ldc[i] = ldc[i]; sg.mn[i] = M[i] * N[i]; sg.N[i] = N[i]; sg.chunks[i] = p.chunks;
    blocks += (M[i] * N[i] + 63) / 64;
    sg.block_end[i] = blocks;
    ws += (int64_t)p.chunks * M[i] * N[i];
  }
  g.num = num; g.tiles = p.tiles; g.K = K; g.chunk_rows = p.chunk_rows;
  sg.num = num;
  if (colsum0) {
    g.colsum = ws;
    sg.partial[num] = ws; sg.C[num] = colsum0; sg.ldc[num] = N[0]; sg.mn[num] = N[0]; sg.N[num] = N[0]; sg.chunks[num] = 2 * p.chunks;
    blocks += (N[0] + 63) / 64;
    sg.block_end[num] = blocks;
    sg.num = num + 1;
  }
  gemm_tn_stream_group_kernel<<<dim3((unsigned)p.chunks, (unsigned)((p.tiles + 3) / 4)), 256, 0, st>>>(g);
  sum_partials_group_kernel<<<(unsigned)blocks, 256, 0, st>>>(sg);
  return launch_status();
}

int relgnn_sum_slabs_tail_f32(const float* slabs, int32_t num_slabs, int32_t M, int32_t N, const float* At, int64_t lda,
                              const float* Bt, int64_t ldb, int32_t R, float* C, void* stream) {
  if (num_slabs < 0 || M < 0 || N < 0 || R < 0) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!C || (num_slabs > 0 && !slabs) || (R > 0 && (!At || !Bt || lda < M || ldb < N))) return RELGNN_EINVAL;
  const int64_t mn = (int64_t)M * N;
  sum_slabs_tail_kernel<<<(unsigned)((mn + 255) / 256), 256, 0, as_stream(stream)>>>(slabs, num_slabs, mn, N, At, lda, Bt, ldb, R, C);
  return launch_status();
}

}  // extern "C"
