// The per-(node, type) transforms of the many-type models (gnns/gnn_film.py:92-106: h_u W_l and h_v F_l over the compact pair tables,
// and their input gradients) on the wave-role form of limb_gemm_pc.hip: C[r, :] = A[a_rows[r], :] @ B_{type(r)}^T with
//   * GATHERED rows: row r of the product reads row a_rows[r] of the node table A (< 0: a padding row, zeros),
//   * PER-TILE weights: rows come in tiles of rows_per_select rows of ONE edge type, b_select[tile] picks the limb image,
//   * K in {128, 256}, N in {128, 256}: forward [P, 128] @ [128, 128 | 256], input gradients [P, 128 | 256] @ [.., 128].
//
// limb_gemm_tile_kernel (limb_gemm.hip) keeps a 128-column chunk of the weights resident in LDS and streams 128-row panels past it:
// an N = 256 product gathers every row TWICE (one workgroup per column chunk) and runs at 0.69 PFLOP/s bf16 / 3.1-3.8 TB/s of its
// own traffic (profiles/r06_gemm_pmc.txt).  Here the weights are not staged at all — the matrix waves read their W fragments straight
// from L2 three k-tiles ahead, as in limb_gemm_pc_kernel — and all of the LDS holds rows: eight producer waves gather, split
// (limb_split.h: the same limbs) and hand 32-row x 128-k sub-slabs to eight barrier-free matrix waves through counters.
// A persistent 16-wave workgroup owns a contiguous range of PANELS of U 32-row units (a panel never straddles a select tile:
// rows_per_select % (32 U) == 0).  N = 256: U = 2, wave w owns columns 32 w .. of both units; N = 128: U = 4, wave w owns columns
// 32 (w & 3) .. of units 2 (w >> 2), 2 (w >> 2) + 1 — two accumulators per wave either way, so that a W fragment feeds twelve MFMAs
// (with one unit per wave, six: the fragments' L2 traffic became the larger stream and the N = 128 forms lost to the panel kernels).
// Same k-tile and limb-product order per accumulator as limb_gemm_tile_kernel / limb_gemm_sel_kernel: bit-identical results
// (tests/test_gpu_limb_gemm.py).
#include "common.h"
#include "handover.h"
#include "lds_dma.h"
#include "limb_split.h"

#include <type_traits>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 528;              // 32 rows x 16 B (8 k of one limb) + 16 B: consecutive pieces start in consecutive bank quads
constexpr int PLANE = 16 * PIECE;       // the 16 (k-tile, k half) pieces of one limb of a sub-slab (32 rows x 128 k)
constexpr int SLAB = 3 * PLANE;         // 3 limbs: 25 344 B
constexpr int NBUF = 6;
constexpr int CTL = 16;                 // control words: [1..6] rows filled per buffer, [8..13] matrix waves done with it
constexpr int MAXPAIRS = 512;           // panels per workgroup whose edge types fit the LDS table (M <= 8.4 M rows on 256 workgroups)

struct PctArgs {
  const float* A; int64_t lda; const int32_t* a_rows;   // node table, row ids per product row (nullptr: the rows themselves)
  const uint16_t* B; int64_t b_stride;                   // limb images [N, K], one per edge type, b_stride elements apart
  const int32_t* b_select; int32_t rows_per_select;
  const float* zeros;                                    // >= 128 zero floats: what a padding row reads
  float* C; int64_t ldc;
  int32_t M, N, K;
  int32_t pairs_base, pairs_rem, groups;
  int32_t* status;
#ifdef RELGNN_PCT_TIMING
  unsigned long long* timing;          // diagnostic build: [workgroup][wave][8] s_memtime totals
#endif
};

struct Frag { bf16x8 hi, mid, lo; };

#ifdef RELGNN_PCT_TIMING
unsigned long long* g_pct_timing = nullptr;
#define TSTAMP(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define TACC(slot, t1, t0) tacc[slot] += (t1) - (t0)
#else
#define TSTAMP(v)
#define TACC(slot, t1, t0)
#endif

template <int S2, bool N128, bool GATHER>
__global__ __launch_bounds__(1024) void limb_gemm_pct_kernel(const PctArgs a) {
  constexpr int U = N128 ? 4 : 2;                                // 32-row units per panel
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * SLAB + (CTL + MAXPAIRS) * 4];
  int* ctl = reinterpret_cast<int*>(lds + NBUF * SLAB);
  int* ptype = ctl + CTL;                                        // edge type of each of my pairs (read per pair by the matrix waves)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int pr0 = q * a.pairs_base + min(q, a.pairs_rem);
  const int npan = a.pairs_base + (q < a.pairs_rem ? 1 : 0);     // my panels of U units
  const int u0 = U * pr0;
  if (tid < CTL) ctl[tid] = 0;
  // (a vector load of b_select inside the matrix loop drains every W fragment in flight — hipcc wants the value scalar at once —
  //  and the division would sit at every k-tile: the types of my pairs go through LDS, computed once)
  for (int i = tid; i < npan; i += 1024) ptype[i] = a.b_select ? a.b_select[((u0 + U * i) * 32) / a.rows_per_select] : 0;
  __syncthreads();
  if (npan == 0) return;
  const int ntiles = a.K >> 4;                                   // 8 S2
  const int nseq = npan * U * S2;                                // sub-slabs in sequence: (panel, half slab, unit)
  bool dead = false;
  const int spin_limit = handover_limit(a.status);
#ifdef RELGNN_PCT_TIMING
  // slots — matrix waves: 0 total, 1 in polls, 2 polls that waited, 3 k-loops, 4 stores; producers: 0 total, 1 in polls, 2 polls that
  // waited, 3 waiting for the rows, 4 split + LDS writes + signal, 5 issuing loads
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TSTAMP(t_begin);
  auto tflush = [&]() {
    TSTAMP(t_end);
    tacc[0] = t_end - t_begin;
    if (a.timing && lane == 0 && q < 256)
      for (int i = 0; i < 8; ++i) a.timing[(q * 16 + wave) * 8 + i] = tacc[i];
  };
#endif
  auto poll = [&](int* p, int target) {
    if (dead) return;
    TSTAMP(tp0);
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(handover_counter(p)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) { dead = true; if (lane == 0 && a.status) atomicOr(a.status, 4 + (wave < 8 ? 0 : 4)); break; }
    }
    handover_fence();
#ifdef RELGNN_PCT_TIMING
    TSTAMP(tp1);
    tacc[1] += tp1 - tp0;
    if (spins) tacc[2] += 1;
#endif
  };

  if (wave < 8) {
    // =================================================== matrix waves ===================================================
    const int i32 = lane & 31, h32 = lane >> 5;
    const int colblk = N128 ? (wave & 3) : wave;              // my 32 output columns
    const int ua = N128 ? 2 * (wave >> 2) : 0, ub = ua + 1;   // my two units of the panel
    const int64_t wlane = (int64_t)colblk * ntiles * 1536 + 8 * lane;
    auto base_of = [&](int pi) -> const uint16_t* {           // the limb image of pair pi's edge type, at my column block and lane
      return a.B + (int64_t)__builtin_amdgcn_readfirstlane(ptype[min(pi, npan - 1)]) * a.b_stride + wlane;
    };
    Frag wr[4];
    auto wload = [&](Frag& f, const uint16_t* base, int t) {  // k-tile t of the image at `base`
      const uint16_t* p = base + (int64_t)t * 1536;
      f.hi = *reinterpret_cast<const bf16x8*>(p);
      f.mid = *reinterpret_cast<const bf16x8*>(p + 512);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
    };
    auto products = [&](f32x16 c, const Frag& w, const Frag& x) {        // limb_gemm.hip's order: small terms first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.hi, c, 0, 0, 0);
      return c;
    };
    auto xread = [&](const unsigned char* p) {
      Frag f;
      f.hi = *reinterpret_cast<const bf16x8*>(p);
      f.mid = *reinterpret_cast<const bf16x8*>(p + PLANE);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 2 * PLANE);
      return f;
    };
    const uint16_t* wcur = base_of(0);
    wload(wr[0], wcur, 0); wload(wr[1], wcur, 1); wload(wr[2], wcur, 2);
    int b0 = 0, gen0 = 0;                                     // buffer / generation of the next sub-slab in sequence
    const int xlane = h32 * PIECE + i32 * 16;
    auto buf_of = [&](int i) { const int b = b0 + i; return b >= NBUF ? b - NBUF : b; };      // i < U <= 4 < NBUF
    auto poll_buf = [&](int i) {
      const int b = b0 + i;
      if (b >= NBUF) poll(ctl + 1 + b - NBUF, 32 * (gen0 + 2)); else poll(ctl + 1 + b, 32 * (gen0 + 1));
    };
    auto release_all = [&]() {                                 // every matrix wave releases all U buffers of the half slab
      wait_lgkm0();                                            // my reads of these buffers have returned
      handover_fence();
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < U; ++i) __hip_atomic_fetch_add(ctl + 8 + buf_of(i), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      b0 += U;
      if (b0 >= NBUF) { b0 -= NBUF; ++gen0; }
    };
    // Two pairs per trip of the loop, as straight-line code: hipcc's wait insertion drains every load AND store in flight at a loop
    // header (vmcnt(0) in front of the first MFMA: the previous pair's result stores have to complete) — with 8 k-tiles per pair that
    // is a store round trip per 3 000 cycles of matrix work; inside one trip it counts exactly.
    auto pair = [&](int pi) __attribute__((always_inline)) {
      const int m0 = (u0 + U * pi) * 32;
      const uint16_t* const wnxt = base_of(pi + 1);            // (the last three k-tiles of a pair request the next pair's first three)
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      // 32 x 32 tile: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register 4 c + {0..3})
      // 32 x 32 tile: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register 4 c + {0..3}).  (Turning the
      // tile through LDS into whole-line stores measured slower with two tiles per wave: 362 vs 341 us at [737 k, 128] x [128, 256].)
      auto store_tile = [&](const f32x16& acc, int r0, int colw) {          // rows m0 + r0 .. + 31
        float* crow = a.C + (int64_t)(m0 + r0 + i32) * a.ldc;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<f32x4*>(crow + colw + 8 * c + 4 * h32) = f32x4{acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
      };
#pragma unroll
      for (int hs = 0; hs < S2; ++hs) {
        // the panel's U units of this half slab sit in buffers b0 .. b0 + U - 1; a wave reads its two units' buffers and releases all U
        // (the counters count waves that are done with a buffer)
        poll_buf(ua); poll_buf(ub);
        const unsigned char* x0b = lds + buf_of(ua) * SLAB + xlane;
        const unsigned char* x1b = lds + buf_of(ub) * SLAB + xlane;
        TSTAMP(tk0);
        Frag x0 = xread(x0b), x1 = xread(x1b);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          constexpr int NT = 8 * S2;
          const int t3 = 8 * hs + kt + 3;                     // (compile-time after unrolling: no wrap test at run time)
          if (t3 < NT) wload(wr[(kt + 3) & 3], wcur, t3); else wload(wr[(kt + 3) & 3], wnxt, t3 - NT);
          __builtin_amdgcn_s_waitcnt(0x0F79);                 // vmcnt(9): the W fragments of this k-tile have landed, three k-tiles stay in flight
          acc0 = products(acc0, wr[kt & 3], x0);
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < 8) x0 = xread(x0b + (kt + 1) * 2 * PIECE);
          __builtin_amdgcn_sched_barrier(0);
          acc1 = products(acc1, wr[kt & 3], x1);
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < 8) x1 = xread(x1b + (kt + 1) * 2 * PIECE);
          __builtin_amdgcn_sched_barrier(0);
        }
        release_all();
        TSTAMP(tk1);
        TACC(3, tk1, tk0);
      }
      TSTAMP(ts0);
      store_tile(acc0, 32 * ua, 32 * colblk);
      store_tile(acc1, 32 * ub, 32 * colblk);
      TSTAMP(ts1);
      TACC(4, ts1, ts0);
      wcur = wnxt;
    };
    int pi = 0;
    for (; pi + 1 < npan; pi += 2) { pair(pi); pair(pi + 1); }
    if (pi < npan) pair(pi);
#ifdef RELGNN_PCT_TIMING
    tflush();
#endif
    return;
  }

  // ===================================================== producer waves =====================================================
  // wave p gathers rows 4 p .. 4 p + 3 of every sub-slab: their four row ids (one 16-byte load, two sub-slabs ahead of the rows
  // they address), two 1 KiB loads (two rows x 128 k each: lane = row (lane >> 5), 4 k), the split, three 8-byte LDS writes per load.
  // The loads of the three sub-slabs behind the current one are in flight; the register sets alternate by name and the loads are
  // unconditional (a position past the end repeats the last unit; a padding row reads the zeros block).
  const int pw = wave - 8;
  const int col4 = lane & 31, rsub = lane >> 5;
  const int wr_lane = (col4 >> 1) * PIECE + (col4 & 1) * 8;   // k-tile col4 >> 2, k half (col4 >> 1) & 1, k 4 (col4 & 1) .. + 3
  const f32x4* A4 = reinterpret_cast<const f32x4*>(a.A);
  const f32x4* Z4 = reinterpret_cast<const f32x4*>(a.zeros);
  const int64_t lda4 = a.lda >> 2;
  struct Pos { int g, pi, hs, tm; };                          // sequence position -> (panel, half slab, unit of the panel)
  auto advance = [&](Pos& p) {
    ++p.g;
    if (++p.tm < U) return;
    p.tm = 0;
    if (++p.hs == S2) { p.hs = 0; ++p.pi; }
  };
  auto ahead2 = [&](Pos p) { advance(p); advance(p); return p; };
  auto first_row = [&](const Pos& p) { return (u0 + U * min(p.pi, npan - 1) + p.tm) * 32 + 4 * pw; };      // of my four rows
  auto idload = [&](const Pos& p) -> int4 {
    const int r = first_row(p);
    if constexpr (!GATHER) return make_int4(r, r + 1, r + 2, r + 3);
    else return *reinterpret_cast<const int4*>(a.a_rows + r);
  };
  auto issue = [&](const Pos& p, const int4& id, f32x4 (&v)[2]) {
    TSTAMP(ti0);
    const int ia = rsub ? id.y : id.x, ib = rsub ? id.w : id.z;           // my rows of the two loads: 4 pw + rsub, 4 pw + 2 + rsub
    const f32x4* pa = ia >= 0 ? A4 + (int64_t)ia * lda4 + p.hs * 32 : Z4;
    const f32x4* pb = ib >= 0 ? A4 + (int64_t)ib * lda4 + p.hs * 32 : Z4;
    v[0] = pa[col4];
    v[1] = pb[col4];
    TSTAMP(ti1);
    TACC(5, ti1, ti0);
  };
  auto process = [&](const Pos& p, f32x4 (&v)[2]) {
    // everything but the three sub-slabs issued last has landed (GATHER: 2 row loads + 1 id quad each)
    TSTAMP(tw0);
    if constexpr (GATHER) __builtin_amdgcn_s_waitcnt(0x0F79); else __builtin_amdgcn_s_waitcnt(0x0F76);
#ifdef RELGNN_PCT_TIMING
    { const float touch = v[0][0] + v[1][0]; asm volatile("" ::"v"(touch)); }    // (the rows really are there)
#endif
    TSTAMP(tw1);
    TACC(3, tw1, tw0);
    const int fill = p.g % NBUF, gen = p.g / NBUF;
    poll(ctl + 8 + fill, 8 * gen);                            // the buffer's previous user has been consumed by the eight matrix waves
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 x = v[j];
      uint32_t h0, m0_, l0, h1, m1, l1;
      split_pair(x[0], x[1], h0, m0_, l0);
      split_pair(x[2], x[3], h1, m1, l1);
      if (__builtin_expect(max3_abs(max3_abs(x[0], x[1], x[2]), x[3], x[3]) >= __uint_as_float(0x7F7F8000u), 0)) {
        split_pair_sat(x[0], x[1], h0, m0_, l0);
        split_pair_sat(x[2], x[3], h1, m1, l1);
      }
      unsigned char* o = lds + fill * SLAB + (4 * pw + 2 * j + rsub) * 16 + wr_lane;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + PLANE) = make_uint2(m0_, m1);
      *reinterpret_cast<uint2*>(o + 2 * PLANE) = make_uint2(l0, l1);
    }
    wait_lgkm0();
    handover_fence();
    if (lane == 0) __hip_atomic_fetch_add(ctl + 1 + fill, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef RELGNN_PCT_TIMING
    { TSTAMP(tw2); tacc[4] += tw2 - tw1; }                       // (includes the poll for the buffer: slot 1 says how much)
#endif
  };
  Pos p0{0, 0, 0, 0}, p1, p2, p3;
  f32x4 v0[2], v1[2], v2[2], v3[2];
  int4 i0, i1, i2, i3;
  p1 = p0; advance(p1); p2 = p1; advance(p2); p3 = p2; advance(p3);
  i0 = idload(p0); i1 = idload(p1);
  i2 = idload(p2); issue(p0, i0, v0);
  i3 = idload(p3); issue(p1, i1, v1);
  i0 = idload(ahead2(p2)); issue(p2, i2, v2);
  for (;;) {                                                   // step: the ids of sub-slab g + 5, the rows of g + 3, then the split of g
    if (p0.g >= nseq) break;
    i1 = idload(ahead2(p3)); issue(p3, i3, v3); process(p0, v0); p0 = p3; advance(p0);
    if (p1.g >= nseq) break;
    i2 = idload(ahead2(p0)); issue(p0, i0, v0); process(p1, v1); p1 = p0; advance(p1);
    if (p2.g >= nseq) break;
    i3 = idload(ahead2(p1)); issue(p1, i1, v1); process(p2, v2); p2 = p1; advance(p2);
    if (p3.g >= nseq) break;
    i0 = idload(ahead2(p2)); issue(p2, i2, v2); process(p3, v3); p3 = p2; advance(p3);
  }
#ifdef RELGNN_PCT_TIMING
  tflush();
#endif
}

}  // namespace

extern "C" {

// 1 iff relgnn_limb_gemm_sel_pc_xf32 takes the shape (pointer alignment aside)
int relgnn_limb_gemm_sel_pc_supported(int32_t M, int32_t N, int32_t K, int32_t rows_per_select) {
  if ((N != 128 && N != 256) || (K != 128 && K != 256)) return 0;
  const int panel = N == 128 ? 128 : 64;                       // rows per panel: U units of 32
  if (M <= 0 || M % panel != 0) return 0;
  return rows_per_select == 0 || (rows_per_select > 0 && rows_per_select % panel == 0);
}

int relgnn_limb_gemm_sel_pc_xf32(const float* A, int64_t lda, const int32_t* a_rows, const uint16_t* B_limbs, int32_t num_b,
                                 const int32_t* b_select, int32_t rows_per_select, const void* zeros, float* C, int64_t ldc,
                                 int32_t M, int32_t N, int32_t K, int32_t* status, void* stream) {
  if (M < 0 || N < 0 || K < 0 || num_b < 1) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B_limbs || !C || !zeros) return RELGNN_EINVAL;
  if (!b_select && num_b != 1) return RELGNN_EINVAL;
  if (!relgnn_limb_gemm_sel_pc_supported(M, N, K, b_select ? rows_per_select : 0)) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B_limbs) || !aligned16(C) || !aligned16(zeros) || (a_rows && !aligned16(a_rows)) || ldc % 4 || ldc < N ||
      lda % 4 || lda < K)
    return RELGNN_EUNSUPPORTED;
  PctArgs a{};
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.B = B_limbs; a.b_stride = relgnn_limb_elements(N, K); a.b_select = b_select;
  a.rows_per_select = rows_per_select; a.zeros = static_cast<const float*>(zeros); a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.status = status;
#ifdef RELGNN_PCT_TIMING
  a.timing = g_pct_timing;
#endif
  // the fewest workgroups that keep the longest range of panels (limb_gemm_pc.hip)
  const int pairs = M / (N == 128 ? 128 : 64);
  int groups = pairs < 256 ? pairs : 256;
  const int longest = (pairs + groups - 1) / groups;
  groups = (pairs + longest - 1) / longest;
  a.groups = groups; a.pairs_base = pairs / groups; a.pairs_rem = pairs % groups;
  if (longest > MAXPAIRS) return RELGNN_EUNSUPPORTED;          // (more than 8.4 M rows: the panel kernel)
  const unsigned grid = (unsigned)(8 * ((groups + 7) / 8));
  hipStream_t st = as_stream(stream);
  const bool g = a_rows != nullptr;
#define PCT_LAUNCH(S2_, N128_) \
  do { if (g) limb_gemm_pct_kernel<S2_, N128_, true><<<grid, 1024, 0, st>>>(a); \
       else limb_gemm_pct_kernel<S2_, N128_, false><<<grid, 1024, 0, st>>>(a); } while (0)
  if (N == 128) {
    if (K == 128) PCT_LAUNCH(1, true); else PCT_LAUNCH(2, true);
  } else {
    if (K == 128) PCT_LAUNCH(1, false); else PCT_LAUNCH(2, false);
  }
#undef PCT_LAUNCH
  return launch_status();
}

#ifdef RELGNN_PCT_TIMING
void relgnn_pct_timing_buffer(unsigned long long* p) { g_pct_timing = p; }
#endif

}  // extern "C"
