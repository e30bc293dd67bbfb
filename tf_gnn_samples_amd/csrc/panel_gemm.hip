// Row-panel GEMM on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bitwise an fmaf chain;
// 157 TFLOP/s peak, MI355X_MICROARCH.md) with direct-to-LDS staging (global_load_lds_dwordx4) and a 3-stage LDS ring.
//
// Replaces the node-side Dense products of the path wherever the operand shapes allow (K % 4 == 0, N % 128 == 0):
//   * the stacked per-edge-type transforms  A [V, L*D] @ [L*D, D]  of gnns/rgcn.py:96-98 (aggregate-first order) and their
//     input gradients;
//   * the per-(node, type) transforms of many-type graphs  Y[r] = H[node[r]] @ W_type(r)  (gnns/gnn_film.py:92-106,
//     gnns/rgcn.py, gnns/ggnn.py on VarMisuse-shaped batches): the row GATHER and the per-512-row-tile weight selection
//     happen in the load addresses — no [P, D] copy of gathered rows, no [tiles, Din, Dout] copy of the weights;
//   * weight gradients  dW = X^T @ G  as K-split / per-tile partial products (both operands k-major).
//
// Geometry.  A workgroup (8 waves) owns ONE panel: up to 16*TMW*WM output rows x NC = 32*WN columns, full K.  The host
// sizes the panels so that their number is a multiple of the 256 CUs (or, for typed operands, one weight per panel) and
// every panel carries the same number of 16-row units +-1: the tail that costs a 128 x 128-tiled grid up to 1/3 of its
// time at M ~ 36 k rows (564 tiles on 512 slots) does not exist.  B (a few hundred KB of weights) is re-streamed from L2
// by every panel.
//
// Wave tile: TMW x 2 MFMA tiles of 16 x 16; the W fragment is the MFMA's A operand and the X fragment its B operand, so
// that a lane ends up with four CONSECUTIVE output columns of one output row (one dwordx4 store per tile).
// k-tile = 16.  A k-group permutation (lane group g of the MFMA takes k = 4g + j in step j) makes every lane's four
// steps one 16-byte LDS read for k-contiguous operands.
//
// LDS images (floats), all lane-linear for the DMA (destination = wave-uniform base + 16 B * lane), swizzled through the
// SOURCE address:
//   k-contiguous operand ("RM": X [rows, K] or W given as [N, K]): blocks of 16 rows x 16 k = 1 KiB = one DMA instruction;
//       slot(i, g) = 4 i + (g ^ 2 (i >> 3))   -> conflict-free ds_read_b128 for the 16-lane groups of gfx950
//   k-major operand ("KM": W [K, N] or X^T): [16 k][W] floats, column n stored at n ^ 16 ((k >> 2) & 1)
//       -> the two k rows a 32-lane half reads in one ds_read_b32 hit disjoint banks
// Pipeline per k-tile: wait (own DMA of tile t+1, counted vmcnt) -> s_barrier -> issue DMA of tile t+3 into the stage tile t
// occupied -> issue the LDS reads of tile t+1's fragments -> MFMAs of tile t.  One barrier per k-tile, DMA two tiles deep,
// fragment reads one tile ahead of the matrix pipe; the waits are counted by hand (raw s_barrier: __syncthreads() would drain
// the DMA queue).
#include "common.h"
#include "lds_dma.h"

#include <stdlib.h>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 16;
constexpr int KIDX_MAX = 1024;          // k-major A with gathered k rows: the row ids of the K range live in LDS

struct PanelArgs {
  const float* A; int64_t lda; const int32_t* a_rows;
  const float* B; int64_t ldb; const int32_t* b_select; int32_t rows_per_select; int64_t b_select_stride;
  const float* bias; const float* zeros;
  float* C; int64_t ldc;
  int32_t M, N, K;
  int64_t a_bs, b_bs, c_bs;            // batch strides (grid.z)
  int32_t k_chunk;                     // K range of batch z: [z * k_chunk_rows ...) only when split_k (else whole K)
  int32_t split_k;                     // 1: the batch index splits K (A/B advance along k), C gets one slab per z
  int32_t act;
  int32_t units_base, units_rem;       // panel q covers 16-row units [q*base + min(q, rem), +base + (q < rem))
};

// WM x WN waves.  Every wave owns RW = 32*T32 + 16*T16 output rows x 32 columns: T32 tiles of v_mfma_f32_32x32x2_f32 (the
// shape that sustains ~137 TFLOP/s on this chip; 16x16x4 sustains 113: profiles/r02_c_mfma_rate.txt, r03 ablation) and, for
// row counts that are an odd multiple of 16, one 16-row strip as two v_mfma_f32_16x16x4_f32 tiles — so that a panel can be
// any multiple of 16 rows and the panels of a call can be equal to within one 16-row unit.
// A_KM: A is given k-major (A[m][k] at A + k*lda + m: the X^T of a weight gradient); B_RM: B is given as [N, K] row-major
// (k contiguous: the W of an input gradient).
// The LDS ring holds STAGES super-tiles of SUB k-tiles each; the workgroup synchronises (counted DMA wait, barrier, next DMA
// batch) once per super-tile.  Measured (scripts/bench_panel_units.py, K = 768, every CU one panel of u 16-row units): the kernel
// takes (0.41 us + 0.23 us x u) per k-tile, and that does NOT move with the ring: 4 x 1 = 3 x 2 (half the barriers) = 3 x 1 to
// within 1 % at every u (u = 9: 117.2 / 117.9 / 119.0 us; library 108.6).  The matrix pipe alone, at the clock this chip holds
// under fp32 MFMA load (136 TFLOP/s for 32x32x2, 113 for 16x16x4 in a register-only loop), needs 109 us for that shape.
// Default: 4 x 1.
template <int WM, int WN, int T32, int T16, bool A_KM, bool B_RM, int SCHED = 1, int SUB = 1, int STAGES = 4>
__global__ __launch_bounds__(512) void panel_gemm_kernel(const PanelArgs a) {
  constexpr int RW = 32 * T32 + 16 * T16;           // rows per wave
  constexpr int PR = RW * WM;                       // panel rows
  constexpr int NC = 32 * WN;                       // panel columns
  constexpr int A_FLOATS = PR * BK, B_FLOATS = NC * BK;
  constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  constexpr int PA = A_FLOATS / 256, PB = B_FLOATS / 256, P = PA + PB;     // 1 KiB DMA pieces per stage
  // Only waves 0-3 (one per SIMD) issue the DMA; their partners 4-7 go from the barrier straight into their MFMAs.  With
  // all eight waves issuing, both waves of a SIMD spend the first few hundred cycles after every barrier on address
  // arithmetic and DMA issue while the matrix pipe idles (measured: 8 % of the k-loop); with one loader per SIMD the partner
  // keeps the pipe busy meanwhile and the loader catches up when the partner has run out of MFMAs.
  constexpr int LOADERS = 4;
  constexpr int G = (P + LOADERS - 1) / LOADERS;    // pieces per loader wave and k-tile (the last ones may be duplicates)
  constexpr int WPIECES = 2 * T32 + T16;            // A pieces (16-row blocks) per wave-row group
  static_assert(!A_KM || T16 == 0, "k-major A: 32-row tiles only (the column swizzle needs 32-column multiples)");
  static_assert(WM * WN == 8, "8 waves");
  // ONE shared object (a second one makes hipcc wait vmcnt(0) before every fragment read of a DMA pipeline)
  constexpr int NSLOT = STAGES * SUB;               // k-tile slots of the ring
  __shared__ __attribute__((aligned(16))) float lds[NSLOT * STAGE_FLOATS + (A_KM ? KIDX_MAX : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int q = blockIdx.x, z = blockIdx.z;

  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  const int m0 = u0 * 16;
  const int rows_here = min(nu * 16, a.M - m0);     // valid rows of this panel
  const int n0 = blockIdx.y * NC;

  int kbeg = 0, kend = a.K;
  const float* Ab = a.A;
  const float* Bb = a.B;
  float* Cb = a.C + (int64_t)z * a.c_bs;
  if (a.split_k) {
    kbeg = z * a.k_chunk;
    kend = min(a.K, kbeg + a.k_chunk);
  } else {
    Ab += (int64_t)z * a.a_bs;
    Bb += (int64_t)z * a.b_bs;
  }
  if (a.b_select) Bb += (int64_t)a.b_select[m0 / a.rows_per_select] * a.b_select_stride;
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  // k-major A whose k rows are gathered (dW of a typed transform: A[m][k] = H[a_rows[k]][m]): the row ids of this K range
  const bool gather_k = A_KM && a.a_rows != nullptr;
  int* kidx = reinterpret_cast<int*>(lds + NSLOT * STAGE_FLOATS);
  if constexpr (A_KM) {
    if (gather_k) {
      // (independent batches: product z reduces over rows a_rows[z*K .. z*K + K); a K split: over a_rows[kbeg .. kend))
      const int32_t* ids = a.a_rows + (a.split_k ? 0 : (int64_t)z * a.K);
      for (int i = tid; i < ntiles * BK; i += 512) kidx[i] = (kbeg + i < kend) ? ids[kbeg + i] : -1;
      __syncthreads();
    }
  }

  // ---- DMA sources of this wave's pieces -----------------------------------------------------------------------------
  // k-contiguous A image: 16-row pieces in panel-row order; rows [32 t, 32 t + 32) of a wave form one 32-row block
  // (slot(i, kq) = 4 i + (kq ^ ((i >> 3) & 3)), i = row inside the block), the 16-row strip one 16-row block
  // (slot(i, kq) = 4 i + (kq ^ 2 (i >> 3))): both conflict-free for ds_read_b128 (16-lane groups of gfx950).
  const bool loader = wave < LOADERS;
  const float* src[G];
  int step[G];                                       // elements per k-tile (fits: 16 * ld)
  int kk[G];                                         // k (inside the tile) of this lane's 16 bytes: for the K tail
  int piece[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int c = min((wave & (LOADERS - 1)) + LOADERS * g, P - 1);   // wave-uniform; past the end: a duplicate of the last piece
    piece[g] = c;
    const float* s;
    int st;
    int k_in;
    if (c < PA) {
      if constexpr (A_KM) {
        const int f = 256 * c + 4 * lane;
        const int k = f / PR, mp = f % PR;
        const int m = mp ^ (16 * ((k >> 2) & 1));
        const bool ok = m < rows_here;
        // (gathered k rows: `s` is the column part only, the row is looked up per tile in issue())
        s = ok ? Ab + (gather_k ? 0 : (int64_t)(kbeg + k) * a.lda) + m0 + m : a.zeros + 4 * lane;
        st = ok ? (gather_k ? 1 : BK * (int)a.lda) : 0;
        k_in = k;
      } else {
        const int cw = c % WPIECES;                  // piece inside the wave-row group
        const int i16 = lane >> 2;                   // row inside the 16-row piece
        int kq;
        if (T16 && cw == 2 * T32) kq = (lane & 3) ^ (2 * (i16 >> 3));              // the 16-row strip
        else kq = (lane & 3) ^ ((((cw & 1) * 16 + i16) >> 3) & 3);                 // half of a 32-row block
        const int r = 16 * c + i16;
        int64_t row = -1;
        if (r < rows_here) row = a.a_rows ? (int64_t)a.a_rows[m0 + r] : (int64_t)(m0 + r);
        const bool ok = row >= 0;
        s = ok ? Ab + row * a.lda + kbeg + 4 * kq : a.zeros + 4 * lane;
        st = ok ? BK : 0;
        k_in = 4 * kq;
      }
    } else {
      const int cb = c - PA;
      if constexpr (B_RM) {                          // [NC / 32] 32-row blocks of W rows (n)
        const int i16 = lane >> 2;
        const int kq = (lane & 3) ^ ((((cb & 1) * 16 + i16) >> 3) & 3);
        s = Bb + (int64_t)(n0 + 16 * cb + i16) * a.ldb + kbeg + 4 * kq;
        st = BK;
        k_in = 4 * kq;
      } else {
        const int f = 256 * cb + 4 * lane;
        const int k = f / NC, np = f % NC;
        const int n = np ^ (16 * ((k >> 2) & 1));
        s = Bb + (int64_t)(kbeg + k) * a.ldb + n0 + n;
        st = BK * (int)a.ldb;
        k_in = k;
      }
    }
    src[g] = s; step[g] = st; kk[g] = k_in;
  }
  const float* zsrc = a.zeros + 4 * lane;

  auto issue = [&](int t, int stage) {
    if (!loader) return;
    float* dst = lds + stage * STAGE_FLOATS;
    const int krem = kend - (kbeg + t * BK);         // < 16 only for the K tail; <= 0 for the tiles that pad the last super-tile
    const bool tail = krem < BK;                     // uniform
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float* s = src[g];
      if (tail && kk[g] >= krem) s = zsrc;
      if constexpr (A_KM) {
        if (gather_k && piece[g] < PA && step[g] != 0) {       // step 0 marks a lane that reads zeros throughout
          const int row = kk[g] < krem ? kidx[t * BK + kk[g]] : -1;
          s = row >= 0 ? src[g] + (int64_t)row * a.lda : zsrc;
        } else {
          src[g] += step[g];
        }
      } else {
        src[g] += step[g];
      }
      dma16(s, dst + piece[g] * 256);
    }
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------
  // 32x32x2: lane (i = lane & 31, h = lane >> 5) supplies k = 8 h + s in step s (s = 0..7: two 16-byte reads of a
  // k-contiguous row); 16x16x4: lane (i = lane & 15, g = lane >> 4) supplies k = 4 g + j in step j (one 16-byte read).
  const int i32 = lane & 31, h32 = lane >> 5;
  const int li = lane & 15, lg = lane >> 4;
  const int rm32_0 = (4 * i32 + ((2 * h32) ^ ((i32 >> 3) & 3))) * 4;           // float offsets inside a 32-row block
  const int rm32_1 = (4 * i32 + ((2 * h32 + 1) ^ ((i32 >> 3) & 3))) * 4;
  const int rm16 = (4 * li + (lg ^ (2 * (li >> 3)))) * 4;                      // inside a 16-row block
  f32x16 acc32[T32];
  f32x4 acc16[T16 ? 2 : 1];
#pragma unroll
  for (int tm = 0; tm < T32; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[tm][r] = 0.f;
  acc16[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (T16) acc16[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frags { float x32[T32][8]; float w32[8]; float x16[T16 ? 4 : 1]; float w16[T16 ? 2 : 1][4]; };
  auto read_frags = [&](Frags& f, int stage) {
    const float* sa = lds + stage * STAGE_FLOATS;
    const float* sb = sa + A_FLOATS;
    if constexpr (B_RM) {
      const float* blk = sb + wn * 512;              // this wave's 32 W rows (n): one 32-row block
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(blk + rm32_0);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(blk + rm32_1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f.w32[j] = v0[j]; f.w32[4 + j] = v1[j]; }
      if constexpr (T16) {
        // the same 32 rows seen as two 16-row tiles: rows 16 tn + li, k = 4 lg + j -> slot (row, kq = lg) of the 32-row block
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int row = 16 * tn + li;
          const f32x4 v = *reinterpret_cast<const f32x4*>(blk + (4 * row + (lg ^ ((row >> 3) & 3))) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) f.w16[tn][j] = v[j];
        }
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const int k = 8 * h32 + s2;
        f.w32[s2] = sb[k * NC + ((wn * 32 + i32) ^ (16 * ((k >> 2) & 1)))];
      }
      if constexpr (T16) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int j = 0; j < 4; ++j) f.w16[tn][j] = sb[(4 * lg + j) * NC + ((wn * 32 + tn * 16 + li) ^ (16 * (lg & 1)))];
      }
    }
    if constexpr (A_KM) {
#pragma unroll
      for (int tm = 0; tm < T32; ++tm)
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const int k = 8 * h32 + s2;
          f.x32[tm][s2] = sa[k * PR + ((wm * RW + tm * 32 + i32) ^ (16 * ((k >> 2) & 1)))];
        }
    } else {
      const float* wa = sa + wm * WPIECES * 256;
#pragma unroll
      for (int tm = 0; tm < T32; ++tm) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wa + tm * 512 + rm32_0);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wa + tm * 512 + rm32_1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f.x32[tm][j] = v0[j]; f.x32[tm][4 + j] = v1[j]; }
      }
      if constexpr (T16) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wa + T32 * 512 + rm16);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.x16[j] = v[j];
      }
    }
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
#pragma unroll
      for (int tm = 0; tm < T32; ++tm)
        acc32[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w32[s2], f.x32[tm][s2], acc32[tm], 0, 0, 0);
      if constexpr (T16) {
        if (s2 & 1) {                                // the strip's eight 16x16x4 MFMAs, two after every second k-pair
          const int j = s2 >> 1;
          acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w16[0][j], f.x16[j], acc16[0], 0, 0, 0);
          acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w16[1][j], f.x16[j], acc16[1], 0, 0, 0);
        }
      }
    }
  };

  // ---- pipeline --------------------------------------------------------------------------------------------------
  // (my own DMA: leave `supers` super-tiles in flight, wait for everything older; every super-tile is SUB * G instructions —
  //  the tiles that pad the last one read zeros)
  auto wait_dma = [&](int supers) {
    if (!loader) return;
    if constexpr (STAGES == 4) { if (supers >= 2) { wait_vm<2 * SUB * G>(); return; } }
    if (supers >= 1) wait_vm<SUB * G>(); else wait_vm<0>();
  };
  const int nsuper = (ntiles + SUB - 1) / SUB;
  auto issue_super = [&](int S) {                    // super-tile S -> slots (S % STAGES) * SUB ..
#pragma unroll
    for (int j = 0; j < SUB; ++j) issue(S * SUB + j, (S % STAGES) * SUB + j);
  };
  Frags f0, f1;
  if (ntiles > 0) {
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i)
      if (i < nsuper) issue_super(i);
    wait_dma(min(STAGES - 2, nsuper - 1));           // super-tile 0 has landed
    __builtin_amdgcn_s_barrier();
    if (STAGES - 1 < nsuper) issue_super(STAGES - 1);
    read_frags(f0, 0);
  }
  // iteration t: fragments of tile t are in registers (or on their way: the compiler waits at first use).  Tile t+1 is in the
  // same super-tile (already landed, no synchronisation) or the first of the next one (wait, barrier, issue the one after the ring).
  auto iteration = [&](int t, Frags& cur, Frags& nxt) {       // t + 1 < ntiles
    if ((t + 1) % SUB == 0) {
      const int S = (t + 1) / SUB;                             // the super-tile whose first tile is read below
      wait_dma(min(STAGES - 2, nsuper - 1 - S));               // my pieces of super-tile S have landed
      wait_lgkm0();                                            // my reads of super-tile S-1 are done
      __builtin_amdgcn_s_barrier();                            // -> S complete for everybody, the slots of S-1 free
      if (S - 1 + STAGES < nsuper) issue_super(S - 1 + STAGES);
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frags(nxt, (t + 1) % NSLOT);
    mfmas(cur);
    // Issue order inside the tile: the matrix pipe starts at once (tile t's fragments are in registers) and the LDS reads of
    // tile t+1 go out one at a time BETWEEN MFMAs, in the issue slots the pipe leaves free.  Left alone hipcc sinks the reads
    // behind the MFMA block (their latency then lands in front of the next barrier); all reads first, the earlier form of this
    // loop, idles the pipe for the ~300 cycles it takes two waves per SIMD to issue them.  RELGNN_PANEL_SCHED picks the form.
    if constexpr (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
      }
    }
  };
  int t = 0;
  for (; t + 2 < ntiles; t += 2) {
    iteration(t, f0, f1);
    iteration(t + 1, f1, f0);
  }
  if (t + 1 < ntiles) {            // two tiles left: t, t+1
    iteration(t, f0, f1);
    mfmas(f1);
  } else if (t < ntiles) {         // one tile left
    mfmas(f0);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // 32x32 tile: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register r = 4 c + {0..3});
  // 16x16 tile: output row (lane & 15) x columns 4 (lane >> 4) + {0..3}.  One 16-byte store each.
  const int colw = n0 + wn * 32;
  auto finish = [&](f32x4 v, int col) {
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
    if (a.act != RELGNN_ACT_LINEAR) {
      v[0] = act_rt(a.act, v[0]); v[1] = act_rt(a.act, v[1]); v[2] = act_rt(a.act, v[2]); v[3] = act_rt(a.act, v[3]);
    }
    return v;
  };
#pragma unroll
  for (int tm = 0; tm < T32; ++tm) {
    const int r = wm * RW + tm * 32 + i32;
    if (r < rows_here) {
      float* crow = Cb + (int64_t)(m0 + r) * a.ldc;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = colw + 8 * c + 4 * h32;
        const f32x4 v = f32x4{acc32[tm][4 * c], acc32[tm][4 * c + 1], acc32[tm][4 * c + 2], acc32[tm][4 * c + 3]};
        *reinterpret_cast<f32x4*>(crow + col) = finish(v, col);
      }
    }
  }
  if constexpr (T16) {
    const int r = wm * RW + T32 * 32 + li;
    if (r < rows_here) {
      float* crow = Cb + (int64_t)(m0 + r) * a.ldc;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = colw + tn * 16 + 4 * lg;
        *reinterpret_cast<f32x4*>(crow + col) = finish(acc16[tn], col);
      }
    }
  }
}

template <int WM, int WN, int T32, int T16, bool A_KM, bool B_RM>
int launch_cfg(const PanelArgs& a, int num_panels, int batch, hipStream_t st) {
  constexpr int NC = 32 * WN;
  dim3 grid((unsigned)num_panels, (unsigned)(a.N / NC), (unsigned)batch);
  panel_gemm_kernel<WM, WN, T32, T16, A_KM, B_RM><<<grid, 512, 0, st>>>(a);
  return launch_status();
}

// Panel sizing for free row counts: the fewest 16-row units per panel such that the panel count is a multiple of the
// resident workgroup slots (256 CUs) — every CU then gets the same number of equal panels.
struct Sizing { int cap, panels, base, rem; };

Sizing size_panels(int M, const int* caps, int n_caps, int column_chunks, int batch) {
  const int units = (M + 15) / 16;
  const int per = 256;                              // workgroup slots filled per round (one 8-wave workgroup per CU)
  // work items per round are shared by the column chunks and the batch: rows only need per / (chunks * batch) panels
  int want = per / max(1, column_chunks * batch);
  if (want < 1) want = 1;
  Sizing best{0, 0, 0, 0};
  double best_cost = 1e30;
  for (int i = 0; i < n_caps; ++i) {
    const int cap = caps[i];                        // 16-row units per panel this configuration is compiled for
    int rounds = (units + want * cap - 1) / (want * cap);
    if (rounds < 1) rounds = 1;
    int panels = rounds * want;
    if (panels > units) panels = units;
    const int base = units / panels, rem = units % panels;
    if (base + (rem ? 1 : 0) > cap) continue;
    // cost ~ rounds x (units the kernel is compiled for): padded MFMA work per workgroup slot
    // (ties go to the larger panel: fewer workgroups re-stream the right operand)
    const double cost = (double)((panels + want - 1) / want) * cap + 1e-6 * panels;
    if (cost < best_cost) { best_cost = cost; best = Sizing{cap, panels, base, rem}; }
  }
  return best;
}

template <bool A_KM, bool B_RM>
int dispatch(PanelArgs a, int batch, int force_rows_per_panel, hipStream_t st) {
  // column geometry: 256-wide panels (8 waves across) when N allows, else 128, else 64.  Short reductions (K <= 256: 16 k-tiles
  // or fewer per panel) take the 128-wide geometry even when 256 divides N: its panels need half the registers, two workgroups
  // share a CU and one's prologue / epilogue runs under the other's k-loop.  RELGNN_PANEL_NC=64|128|256 overrides (experiments).
  static const int forced_nc = [] { const char* e = getenv("RELGNN_PANEL_NC"); return e ? atoi(e) : 0; }();
  int nc = a.N % 256 == 0 ? 256 : (a.N % 128 == 0 ? 128 : (a.N % 64 == 0 ? 64 : 0));
  const int k_len = a.split_k ? a.k_chunk : a.K;
  if (nc == 256 && k_len <= 256) nc = 128;
  if (forced_nc && a.N % forced_nc == 0 && (forced_nc == 64 || forced_nc == 128 || forced_nc == 256)) nc = forced_nc;
  if (nc == 0) return RELGNN_EUNSUPPORTED;
  const int chunks = a.N / nc;
#define RELGNN_PANEL_CASE(CAP, WM_, WN_, T32_, T16_) \
  case CAP: return launch_cfg<WM_, WN_, T32_, T16_, A_KM, B_RM>(a, panels, batch, st)
  if (force_rows_per_panel > 0) {                   // typed / per-tile operands: fixed 128-row panels
    if (force_rows_per_panel != 128 || a.M % 128 != 0) return RELGNN_EUNSUPPORTED;
    const int panels = a.M / 128;
    a.units_base = 8; a.units_rem = 0;
    if (nc == 256) return launch_cfg<1, 8, 4, 0, A_KM, B_RM>(a, panels, batch, st);
    if (nc == 128) return launch_cfg<2, 4, 2, 0, A_KM, B_RM>(a, panels, batch, st);
    return launch_cfg<4, 2, 1, 0, A_KM, B_RM>(a, panels, batch, st);
  }
  if (nc == 256) {                                  // 1 x 8 waves: panel = 16 * cap rows
    static const int caps_rm[] = {2, 3, 4, 5, 6, 7, 8, 9, 10};
    static const int caps_km[] = {2, 4, 6, 8, 10};
    const Sizing s = A_KM ? size_panels(a.M, caps_km, 5, chunks, batch) : size_panels(a.M, caps_rm, 9, chunks, batch);
    if (s.cap == 0) return RELGNN_EUNSUPPORTED;
    a.units_base = s.base; a.units_rem = s.rem;
    const int panels = s.panels;
    switch (s.cap) {
      RELGNN_PANEL_CASE(2, 1, 8, 1, 0); RELGNN_PANEL_CASE(4, 1, 8, 2, 0); RELGNN_PANEL_CASE(6, 1, 8, 3, 0);
      RELGNN_PANEL_CASE(8, 1, 8, 4, 0); RELGNN_PANEL_CASE(10, 1, 8, 5, 0);
      default: break;
    }
    if constexpr (!A_KM) {
      switch (s.cap) {
        RELGNN_PANEL_CASE(3, 1, 8, 1, 1); RELGNN_PANEL_CASE(5, 1, 8, 2, 1); RELGNN_PANEL_CASE(7, 1, 8, 3, 1);
        RELGNN_PANEL_CASE(9, 1, 8, 4, 1);
        default: break;
      }
    }
    return RELGNN_EUNSUPPORTED;
  }
  if (nc == 128) {                                  // 2 x 4 waves: panel = 32 * (units per wave) rows
    static const int caps_rm[] = {4, 6, 8, 10, 12};
    static const int caps_km[] = {4, 8, 12};
    const Sizing s = A_KM ? size_panels(a.M, caps_km, 3, chunks, batch) : size_panels(a.M, caps_rm, 5, chunks, batch);
    if (s.cap == 0) return RELGNN_EUNSUPPORTED;
    a.units_base = s.base; a.units_rem = s.rem;
    const int panels = s.panels;
    switch (s.cap) {
      RELGNN_PANEL_CASE(4, 2, 4, 1, 0); RELGNN_PANEL_CASE(8, 2, 4, 2, 0); RELGNN_PANEL_CASE(12, 2, 4, 3, 0);
      default: break;
    }
    if constexpr (!A_KM) {
      switch (s.cap) {
        RELGNN_PANEL_CASE(6, 2, 4, 1, 1); RELGNN_PANEL_CASE(10, 2, 4, 2, 1);
        default: break;
      }
    }
    return RELGNN_EUNSUPPORTED;
  }
  {                                                 // 4 x 2 waves: 64 columns
    static const int caps[] = {8, 16};
    const Sizing s = size_panels(a.M, caps, 2, chunks, batch);
    if (s.cap == 0) return RELGNN_EUNSUPPORTED;
    a.units_base = s.base; a.units_rem = s.rem;
    const int panels = s.panels;
    switch (s.cap) {
      RELGNN_PANEL_CASE(8, 4, 2, 1, 0); RELGNN_PANEL_CASE(16, 4, 2, 2, 0);
      default: break;
    }
    return RELGNN_EUNSUPPORTED;
  }
#undef RELGNN_PANEL_CASE
}

}  // namespace

extern "C" {

int relgnn_panel_gemm_zeros_floats(void) { return 256; }

int relgnn_panel_gemm_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const float* B,
                          int64_t ldb, const int32_t* b_select, int32_t rows_per_select, int64_t b_select_stride,
                          const float* bias, const float* zeros, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                          int32_t batch, int64_t a_batch_stride, int64_t b_batch_stride, int64_t c_batch_stride,
                          int32_t split_k_rows, void* stream) {
  if (layout < RELGNN_GEMM_NN || layout > RELGNN_GEMM_TN || M < 0 || N < 0 || K < 0 || batch < 1) return RELGNN_EINVAL;
  if (act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU || split_k_rows < 0) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C || !zeros) return RELGNN_EINVAL;
  if (K == 0) return RELGNN_EUNSUPPORTED;
  if (split_k_rows > 0 && (bias || act != RELGNN_ACT_LINEAR || split_k_rows % BK != 0)) return RELGNN_EINVAL;
  if (a_rows && layout == RELGNN_GEMM_TN && (split_k_rows > 0 ? split_k_rows : K) > KIDX_MAX) return RELGNN_EUNSUPPORTED;
  if (b_select && (rows_per_select <= 0 || rows_per_select % 128 != 0)) return RELGNN_EINVAL;
  if (N % 64 != 0) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || !aligned16(zeros) || (bias && !aligned16(bias)) || lda % 4 || ldb % 4 ||
      ldc % 4 || ldc < N)
    return RELGNN_EUNSUPPORTED;
  // k-contiguous operands are staged 4 k at a time; k-major ones (both operands of TN) one whole k row at a time
  if (layout != RELGNN_GEMM_TN && K % 4 != 0) return RELGNN_EUNSUPPORTED;
  if (layout == RELGNN_GEMM_TN && M % 4 != 0) return RELGNN_EUNSUPPORTED;
  if (a_batch_stride % 4 || b_batch_stride % 4 || c_batch_stride % 4 || b_select_stride % 4) return RELGNN_EUNSUPPORTED;
  if (lda > (1 << 26) || ldb > (1 << 26)) return RELGNN_EUNSUPPORTED;       // per-tile address steps are kept in 32 bits
  PanelArgs a{};
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.B = B; a.ldb = ldb; a.b_select = b_select; a.rows_per_select = rows_per_select;
  a.b_select_stride = b_select_stride; a.bias = bias; a.zeros = zeros; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.a_bs = a_batch_stride; a.b_bs = b_batch_stride; a.c_bs = c_batch_stride;
  a.split_k = split_k_rows > 0 ? 1 : 0;
  a.k_chunk = split_k_rows > 0 ? split_k_rows : K;
  a.act = act;
  hipStream_t st = as_stream(stream);
  const int force = b_select ? 128 : 0;
  switch (layout) {
    case RELGNN_GEMM_NN: return dispatch<false, false>(a, batch, force, st);
    case RELGNN_GEMM_NT: return dispatch<false, true>(a, batch, force, st);
    default: return dispatch<true, false>(a, batch, force, st);
  }
}

}  // extern "C"
