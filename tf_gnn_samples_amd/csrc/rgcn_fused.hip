// The aggregate-first RGCN layer (gnns/rgcn.py:84-114, sum-like aggregations) as ONE kernel: gather waves fold the (target, edge type)
// buckets of raw states and hand the sums — already split into bf16 limbs — to matrix-pipe waves through LDS:
//
//     S[v, l, :] = sum_{p in bucket (v, l)} w[p] * H[col[p], :]           (sequential fp32 fold, the order of relgnn_seg_reduce_fwd)
//     out[v, :]  = act(bias + sum_l S[v, l, :] @ W_l)                      (three bf16 limbs per operand, six products: limb_gemm.hip)
//
// i.e. relgnn_seg_reduce_fwd followed by relgnn_limb_gemm_xf32 without the [V, L*256] fp32 round trip in between — and, what pays
// more, with the gather (L2-latency bound, no LDS, no matrix pipe) and the product (matrix pipe, LDS) running on the SAME CU at
// the same time instead of one after the other.  The result is bit-identical to the two-kernel route: same fold order per bucket,
// same split (limb_split.h), same k-tile order and the same six-product order per accumulator.
//
// Geometry (D_in = D_out = 256; K = L * 256): one persistent workgroup of 16 waves per CU owns a contiguous range of 32-row units,
// taken as panels of 64 rows.  Waves 0-7 (two per SIMD) are the CONSUMERS: wave c owns output columns 32 c .. 32 c + 31 of the
// panel (two 32 x 32 accumulators), reads its W limb fragments straight from L2 into registers (the weight image is in MFMA
// operand layout: one 16-byte load per lane and limb, four k-tiles ahead) and the X limb fragments from LDS.  Waves 8-15 are the
// PRODUCERS.  The unit of hand-over is a SUB-SLAB: 64 rows x 128 columns (half an edge type's block = 8 k-tiles) as limbs, 49.5 KiB;
// three of them rotate through LDS, so the producers run up to two sub-slabs ahead of the matrix pipe.  Hand-over is by counters
// in LDS (filled / freed per buffer, monotonic), polled with s_sleep: no workgroup barrier after the first one.
//
// Producers.  A wave gathers half rows: lane = two columns (8 bytes; 512 bytes per message and wave).  Work is drawn from a queue
// (an LDS counter) in BATCHES of 8 rows of one sub-slab.  The messages of a batch's 8 buckets are laid out as one flat stream (an
// empty bucket takes one placeholder position) and walked in groups of 16 row loads, one group in flight while the previous one is
// folded; bucket boundaries are wave-uniform flags inside the stream, so a wave always has 16-32 loads in flight whatever the
// bucket lengths are (a PPI-shaped batch has one message per self-loop bucket and up to ~550 per forward-edge bucket).  The next
// batch's bucket bounds and the next 64 stream positions' (col, w) are fetched one step ahead.  At a bucket's end the two sums of a
// lane are split (limb_split.h) and written as three ds_write_b32 into the sub-slab — pieces of 32 rows x 16 B padded to 528 B, so
// that the 16 pieces a wave writes for one row fall into 16 different bank quads — and, when asked for, stored to S as fp32 (the
// weight gradient of a training step reads them: gnns/rgcn.py:96-98 backward).
#include "common.h"
#include "lds_dma.h"
#include "limb_split.h"

#include <stdlib.h>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 528;              // 32 rows x 16 B (8 k of one limb) + 16 B: consecutive pieces start in consecutive bank quads
constexpr int PLANE = 16 * PIECE;       // the 16 (k-tile, k half) pieces of one (row tile, limb)
constexpr int SLAB = 6 * PLANE;         // 2 row tiles x 3 limbs: 50 688 B
constexpr int NBUF = 3;
constexpr int UNR = 16;                 // row loads per group
constexpr int BROWS = 8;                // rows per batch
constexpr int CTL = 16;                 // control words behind the slabs
constexpr uint32_t M_LAST = 1u << 31, M_EMPTY = 1u << 30, M_INVALID = 1u << 29;
constexpr int SPIN_LIMIT = 1 << 22;     // a poll that takes this long is a bug: give up, flag it, finish with wrong numbers

struct FusedArgs {
  const float* H; int64_t ldh;           // [*, ldh] states, 256 columns read
  const int32_t* rowptr;                 // [V * L + 1]: bucket (v, l) = rowptr[v * L + l] .. rowptr[v * L + l + 1]
  const int32_t* col; const float* w;    // per message: row of H, scale (w may be null)
  const uint16_t* B;                     // limb tiles of the stacked W^T [256, L * 256] (relgnn_limb_split_multi_f32)
  const float* bias;
  float* S; int64_t lds_;                // nullable: the bucket sums [V, L * 256] fp32
  float* C; int64_t ldc;                 // [V, 256]
  int32_t V, L, act;
  int32_t units_base, units_rem, groups; // workgroup q owns the 32-row units [q * base + min(q, rem), + base + (q < rem))
  int32_t* status;                       // set to != 0 when a poll gave up (SPIN_LIMIT)
#ifdef RELGNN_FUSED_TIMING
  unsigned long long* timing;            // slots: 0 total, 1 polls, 2 polls that waited, 3 row-load wait, 4 fold, 5 prepare, 6 issue / k-loop
#endif
};

struct Frag { bf16x8 hi, mid, lo; };

#ifdef RELGNN_FUSED_TIMING
unsigned long long* g_fused_timing = nullptr;   // diagnostic build: [workgroup][wave][8] cycle totals
#define TSTAMP(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define TACC(slot, t1, t0) tacc[slot] += (t1) - (t0)
#else
#define TSTAMP(v)
#define TACC(slot, t1, t0)
#endif

__device__ __forceinline__ int lds_counter(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }

template <bool HAS_W>
__global__ __launch_bounds__(1024) void rgcn_fused_kernel(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * SLAB + CTL * 4];
  int* ctl = reinterpret_cast<int*>(lds + NBUF * SLAB);      // [0] next batch, [1..3] units filled, [4..6] consumer waves done
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  const int npan = (nu + 1) >> 1;
  const int L = a.L;
  const int nsub = 2 * L;                                    // sub-slabs per panel
  const int nseq = npan * nsub;
  const int rend = min((u0 + nu) * 32, a.V);                 // first row that is not mine
  const int ntiles = L * 16;
  if (tid < CTL) ctl[tid] = 0;
  __syncthreads();
  bool dead = false;                                          // a poll gave up: stop waiting for anything
#ifdef RELGNN_FUSED_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TSTAMP(t_begin);
  auto tflush = [&]() {
    TSTAMP(t_end);
    tacc[0] = t_end - t_begin;
    if (a.timing && lane == 0)
      for (int i = 0; i < 8; ++i) a.timing[((int64_t)q * 16 + wave) * 8 + i] = tacc[i];
  };
#endif
  auto poll = [&](int* p, int target) {
    if (dead) return;
    TSTAMP(tp0);
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(lds_counter(p)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0 && a.status) atomicOr(a.status, 1 + (wave < 8 ? 0 : 1)); break; }
    }
    compiler_fence();
#ifdef RELGNN_FUSED_TIMING
    TSTAMP(tp1);
    tacc[1] += tp1 - tp0;
    if (spins) tacc[2] += 1;
#endif
  };

  if (wave < 8) {
    // =================================================== consumers ===================================================
    const int i32 = lane & 31, h32 = lane >> 5;
    const uint16_t* Bw = a.B + (int64_t)wave * ntiles * 1536 + 8 * lane;
    Frag wr[4];
    int tl = 0;                                               // next k-tile whose W fragments are requested
    auto wload = [&](Frag& f) {
      const uint16_t* p = Bw + (int64_t)tl * 1536;
      f.hi = *reinterpret_cast<const bf16x8*>(p);
      f.mid = *reinterpret_cast<const bf16x8*>(p + 512);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
      tl = tl + 1 == ntiles ? 0 : tl + 1;
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
    wload(wr[0]); wload(wr[1]); wload(wr[2]);
    int b = 0, gen = 0;
    for (int pi = 0; pi < npan; ++pi) {
      const int m0 = (u0 + 2 * pi) * 32;
      const int rows_here = min(64, rend - m0);
      const bool two = rows_here > 32;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      for (int sub = 0; sub < nsub; ++sub) {
        poll(ctl + 1 + b, 64 * (gen + 1));
        const unsigned char* xb = lds + b * SLAB + h32 * PIECE + i32 * 16;
        TSTAMP(tk0);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          wload(wr[(kt + 3) & 3]);
          const unsigned char* p = xb + kt * 2 * PIECE;
          const Frag x0 = xread(p);
          if (two) {
            const Frag x1 = xread(p + 3 * PLANE);
            acc0 = products(acc0, wr[kt & 3], x0);
            acc1 = products(acc1, wr[kt & 3], x1);
          } else {
            acc0 = products(acc0, wr[kt & 3], x0);
          }
        }
        wait_lgkm0();                                          // my reads of this buffer have returned
        compiler_fence();
        TSTAMP(tk1);
        TACC(6, tk1, tk0);
        if (lane == 0) __hip_atomic_fetch_add(ctl + 4 + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (++b == NBUF) { b = 0; ++gen; }
      }
      // epilogue: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register 4 c + {0..3}) of its 32 x 32 tile
      const int colw = wave * 32;
      auto store_tile = [&](const f32x16& acc, int r) {
        if (r >= rows_here) return;
        float* crow = a.C + (int64_t)(m0 + r) * a.ldc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cc = colw + 8 * c + 4 * h32;
          f32x4 v = f32x4{acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
          if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + cc);
          if (a.act != RELGNN_ACT_LINEAR) {
            v[0] = act_rt(a.act, v[0]); v[1] = act_rt(a.act, v[1]); v[2] = act_rt(a.act, v[2]); v[3] = act_rt(a.act, v[3]);
          }
          *reinterpret_cast<f32x4*>(crow + cc) = v;
        }
      };
      store_tile(acc0, i32);
      if (two) store_tile(acc1, 32 + i32);
    }
#ifdef RELGNN_FUSED_TIMING
    tflush();
#endif
    return;
  }

  // ===================================================== producers =====================================================
  // Every vector-memory instruction of the loop sits in straight-line code and writes the register it will be read from: a load
  // under a condition lands in a temporary that hipcc copies into the loop-carried register at the end of the region — behind an
  // s_waitcnt vmcnt(0) that drains the whole gather pipeline (the same lesson as limb_gemm.hip's x_load).  Hence: bucket bounds
  // of the next batch are re-requested every step, (col, w) go by clamped addresses instead of predicates, register sets rotate
  // by name (the loop is unrolled by four steps) and never by copy.
  const int nbatches = nseq * (64 / BROWS);
  // ---- the batch that has been drawn from the queue; its bucket bounds are on their way (lanes 0 .. BROWS-1)
  int nb_id = nbatches, nb_g = 0, nb_bi = 0, nb_m0 = 0, nb_sub = 0;
  int rp_b = 0, rp_e = 0;
  auto grab = [&]() {
    int id = 0;
    if (lane == 0) id = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    id = __builtin_amdgcn_readfirstlane(id);
    nb_id = id;
    if (id < nbatches) {
      nb_g = id >> 3; nb_bi = id & 7;
      const int pi = nb_g / nsub;
      nb_sub = nb_g - pi * nsub;
      nb_m0 = (u0 + 2 * pi) * 32;
    }
  };
  auto reload_bounds = [&]() {
    const int row = nb_m0 + nb_bi * BROWS + lane;
    const bool ok = nb_id < nbatches && lane < BROWS && row < rend;
    const int32_t* p = ok ? a.rowptr + ((int64_t)row * L + (nb_sub >> 1)) : a.rowptr;      // (rowptr has at least two entries)
    const int b = p[0], e = p[1];
    rp_b = b;
    rp_e = ok ? e : b;
  };
  // ---- the batch whose stream positions are being laid out
  int bb_valid = 1, bb_pos = 0, bb_total = 0, bb_empty = 0, bb_r0 = 0, bb_g = 0, bb_sub = 0, bb_m0 = 0;
  int bb_P[BROWS + 1], bb_base[BROWS];
#pragma unroll
  for (int j = 0; j < BROWS; ++j) { bb_P[j] = 0; bb_base[j] = 0; }
  bb_P[BROWS] = 0;
  auto open_batch = [&]() {
    if (nb_id >= nbatches) { bb_valid = 0; return; }
    bb_g = nb_g; bb_sub = nb_sub; bb_m0 = nb_m0; bb_r0 = nb_bi * BROWS;
    const int n_l = rp_e - rp_b;
    bb_P[0] = 0; bb_empty = 0;
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      const int nj = __builtin_amdgcn_readlane(n_l, j);
      const int bj = __builtin_amdgcn_readlane(rp_b, j);
      if (nj <= 0) bb_empty |= 1 << j;
      bb_base[j] = bj - bb_P[j];
      bb_P[j + 1] = bb_P[j] + max(nj, 1);
    }
    bb_total = bb_P[BROWS];
    bb_pos = 0;
    grab();
  };

  struct Group {
    int col; float w; uint32_t meta;                  // lanes 0 .. 15: the group's stream positions
    int g, sub, m0, flags;                            // wave-uniform; flags: 1 = first group of its batch, 2 = past the end of the stream
  };
  auto prepare = [&](Group& c) {
    TSTAMP(tq0);
    if (bb_valid && bb_pos >= bb_total) open_batch();
    reload_bounds();
    c.g = bb_g; c.sub = bb_sub; c.m0 = bb_m0;
    c.flags = bb_valid ? (bb_pos == 0 ? 1 : 0) : 2;
    const int f = bb_pos + lane;
    // (sums of per-bucket increments, not selects between table entries: hipcc turns a select chain over an array into an
    //  indexed load from a scratch copy of the array)
    int j = 0, msg = bb_base[0] + f, lastpos = bb_P[1] - 1;
#pragma unroll
    for (int k = 1; k < BROWS; ++k) {
      const bool ge = f >= bb_P[k];
      j += ge ? 1 : 0;
      msg += ge ? bb_base[k] - bb_base[k - 1] : 0;
      lastpos += ge ? bb_P[k + 1] - bb_P[k] : 0;
    }
    const int r = bb_r0 + j;
    const int emp = (bb_empty >> j) & 1;
    const bool valid = bb_valid && lane < UNR && f < bb_total;
    c.meta = valid ? ((uint32_t)r | (f == lastpos ? M_LAST : 0u) | (emp ? M_EMPTY : 0u)) : M_INVALID;
    const bool real = valid && !emp;
    const int32_t* cp = real ? a.col + msg : a.rowptr;                    // (rowptr[0] == 0: a placeholder gathers row 0)
    c.col = *cp;
    if constexpr (HAS_W) {
      const float* wp = real ? a.w + msg : reinterpret_cast<const float*>(a.rowptr);
      c.w = *wp;
    } else {
      c.w = 1.f;
    }
    if (bb_valid) bb_pos += UNR;
    TSTAMP(tq1);
    TACC(5, tq1, tq0);
  };

  const f32x2* H2 = reinterpret_cast<const f32x2*>(a.H);
  const int64_t ldh2 = a.ldh >> 1;
  auto issue = [&](const Group& c, f32x2 (&v)[UNR]) {
    TSTAMP(ti0);
    const int hoff = (c.sub & 1) * 64;                                     // float2 units
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(c.col, u);
      const f32x2* row = H2 + (int64_t)r * ldh2 + hoff;                    // wave-uniform base
      v[u] = row[lane];
    }
    TSTAMP(ti1);
    TACC(6, ti1, ti0);
  };
  f32x2 acc = f32x2{0.f, 0.f};
  // lane = columns 2 lane, 2 lane + 1 of the half = k-tile lane >> 3, k half (lane >> 2) & 1, k 2 (lane & 3) of that half
  const int wr_lane = (lane >> 2) * PIECE + (lane & 3) * 4;
  auto finalize = [&](const Group& c, int r) {
    const int grow = c.m0 + r;
    if (a.S && grow < rend)
      *reinterpret_cast<f32x2*>(a.S + (int64_t)grow * a.lds_ + (c.sub >> 1) * 256 + (c.sub & 1) * 128 + 2 * lane) = acc;
    uint32_t h, m, l;
    split_pair(acc[0], acc[1], h, m, l);
    if (__builtin_expect(fmaxf(fabsf(acc[0]), fabsf(acc[1])) >= __uint_as_float(0x7F7F8000u), 0))
      split_pair_sat(acc[0], acc[1], h, m, l);
    const int fill = c.g % NBUF;
    unsigned char* p = lds + fill * SLAB + (r >> 5) * 3 * PLANE + (r & 31) * 16 + wr_lane;
    *reinterpret_cast<uint32_t*>(p) = h;
    *reinterpret_cast<uint32_t*>(p + PLANE) = m;
    *reinterpret_cast<uint32_t*>(p + 2 * PLANE) = l;
    wait_lgkm0();
    compiler_fence();
    if (lane == 0) __hip_atomic_fetch_add(ctl + 1 + fill, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    acc = f32x2{0.f, 0.f};
  };
  // One copy of finalize per fold: the sixteen positions are the cases of a switch that is re-entered behind a bucket's end.
#define RELGNN_FOLD_STEP(u)                                                                                                  \
  case u: {                                                                                                                  \
    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)c.meta, u);                                                  \
    const float wk = HAS_W ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w), u)) : 1.f;                        \
    if (!(m & (M_INVALID | M_EMPTY))) acc = acc + f32x2{wk * v[u][0], wk * v[u][1]};   /* two roundings (-ffp-contract=off) */ \
    if (m & M_LAST) { r = (int)(m & 63u); next = u + 1; break; }                                                             \
  }                                                                                                                          \
    [[fallthrough]];
  auto fold = [&](const Group& c, const f32x2 (&v)[UNR]) {
    // everything but the UNR row loads issued last (the next group's) has landed: this group's rows, and (col, w, meta) before them.
    // Stated through the builtin, which hipcc's wait insertion understands — left to itself it puts vmcnt(0) in front of the
    // switch loop, which drains the next group's loads as well.  (gfx9 encoding: vmcnt 16 = bit 14, expcnt 7, lgkmcnt 15: no wait)
    static_assert(UNR == 16, "the immediate below says vmcnt(16)");
    TSTAMP(tw0);
    __builtin_amdgcn_s_waitcnt(0x4F70);
    TSTAMP(tw1);
    TACC(3, tw1, tw0);
    if (c.flags & 1) poll(ctl + 4 + c.g % NBUF, 8 * (c.g / NBUF));       // the buffer's previous user has been consumed
    int next = 0;
    do {
      int r = -1;
      switch (next) {
        RELGNN_FOLD_STEP(0) RELGNN_FOLD_STEP(1) RELGNN_FOLD_STEP(2) RELGNN_FOLD_STEP(3)
        RELGNN_FOLD_STEP(4) RELGNN_FOLD_STEP(5) RELGNN_FOLD_STEP(6) RELGNN_FOLD_STEP(7)
        RELGNN_FOLD_STEP(8) RELGNN_FOLD_STEP(9) RELGNN_FOLD_STEP(10) RELGNN_FOLD_STEP(11)
        RELGNN_FOLD_STEP(12) RELGNN_FOLD_STEP(13) RELGNN_FOLD_STEP(14) RELGNN_FOLD_STEP(15)
        default: next = UNR;
      }
      if (r >= 0) finalize(c, r);
    } while (next < UNR);
    TSTAMP(tw2);
    TACC(4, tw2, tw1);
  };
#undef RELGNN_FOLD_STEP

  Group G0, G1, G2, G3;
  grab();
  reload_bounds();
  prepare(G0); prepare(G1); prepare(G2); prepare(G3);
  f32x2 v0[UNR], v1[UNR];
  issue(G0, v0);
  for (;;) {                      // step k: rows of group k + 1 requested, group k folded, (col, w) of group k + 4 requested
    if (G0.flags & 2) break;
    issue(G1, v1); fold(G0, v0); prepare(G0);
    if (G1.flags & 2) break;
    issue(G2, v0); fold(G1, v1); prepare(G1);
    if (G2.flags & 2) break;
    issue(G3, v1); fold(G2, v0); prepare(G2);
    if (G3.flags & 2) break;
    issue(G0, v0); fold(G3, v1); prepare(G3);
  }
#ifdef RELGNN_FUSED_TIMING
  tflush();
#endif
}

int32_t* g_status = nullptr;

}  // namespace

extern "C" {

int relgnn_rgcn_fused_fwd(const float* H, int64_t num_rows_h, int64_t ldh, const int32_t* rowptr, int32_t num_nodes,
                          int32_t num_edge_types, const int32_t* col, const float* w, const uint16_t* w_limbs, const float* bias,
                          int32_t act, float* bucket_sums, int64_t lds, float* out, int64_t ldo, int32_t d_in, int32_t d_out,
                          void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || d_in < 0 || d_out < 0 || num_rows_h < 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU)
    return RELGNN_EINVAL;
  if (num_nodes == 0 || d_out == 0) return RELGNN_OK;
  if (!H || !rowptr || !w_limbs || !out) return RELGNN_EINVAL;
  if (d_in != 256 || d_out != 256 || num_edge_types > 64) return RELGNN_EUNSUPPORTED;
  if (ldh < d_in || ldo < d_out || (bucket_sums && lds < (int64_t)num_edge_types * d_in)) return RELGNN_EINVAL;
  if (!aligned16(H) || !aligned16(out) || !aligned16(w_limbs) || (bias && !aligned16(bias)) || (bucket_sums && !aligned16(bucket_sums)) ||
      ldh % 4 || ldo % 4 || lds % 4)
    return RELGNN_EUNSUPPORTED;
  if (!g_status) {
    if (hipMalloc(reinterpret_cast<void**>(&g_status), sizeof(int32_t)) != hipSuccess) return RELGNN_EHIP;
    if (hipMemset(g_status, 0, sizeof(int32_t)) != hipSuccess) return RELGNN_EHIP;
  }
  FusedArgs a{};
  a.H = H; a.ldh = ldh; a.rowptr = rowptr; a.col = col; a.w = w; a.B = w_limbs; a.bias = bias; a.S = bucket_sums; a.lds_ = lds;
  a.C = out; a.ldc = ldo; a.V = num_nodes; a.L = num_edge_types; a.act = act; a.status = g_status;
#ifdef RELGNN_FUSED_TIMING
  a.timing = g_fused_timing;
#endif
  const int units = (num_nodes + 31) / 32;
  int groups = (units + 1) / 2;                               // at least one full panel per workgroup
  if (groups > 256) groups = 256;
  a.groups = groups; a.units_base = units / groups; a.units_rem = units % groups;
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)(8 * ((groups + 7) / 8));
  if (w) rgcn_fused_kernel<true><<<grid, 1024, 0, st>>>(a);
  else rgcn_fused_kernel<false><<<grid, 1024, 0, st>>>(a);
  return launch_status();
}

#ifdef RELGNN_FUSED_TIMING
void relgnn_rgcn_fused_timing_buffer(unsigned long long* p) { g_fused_timing = p; }
#endif

int relgnn_rgcn_fused_status(int32_t* status, int32_t reset) {
  if (!status) return RELGNN_EINVAL;
  *status = 0;
  if (!g_status) return RELGNN_OK;
  if (hipDeviceSynchronize() != hipSuccess) return RELGNN_EHIP;
  if (hipMemcpy(status, g_status, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return RELGNN_EHIP;
  if (reset && hipMemset(g_status, 0, sizeof(int32_t)) != hipSuccess) return RELGNN_EHIP;
  return RELGNN_OK;
}

}  // extern "C"
