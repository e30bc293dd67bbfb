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
// operand layout: one 16-byte load per lane and limb, three k-tiles ahead) and the X limb fragments from LDS.  Waves 8-15 are the
// PRODUCERS.  The unit of hand-over is a SUB-SLAB: one 32-row tile x one edge type's 256 columns (16 k-tiles) as limbs, 49.5 KiB;
// three of them rotate through LDS: the matrix waves work on the two row tiles of edge type l while the gather waves fill the
// first tile of type l + 1 (the gather is the longer side, so the matrix waves are done before that tile is full and the second
// one finds its buffer free).  Hand-over is by counters in LDS (filled / freed per buffer, monotonic), polled with s_sleep: no
// workgroup barrier after the first one.
//
// Producers.  A wave gathers whole rows: lane = four columns (16 bytes; 1 KiB per message and wave).  Work is drawn from a queue
// (an LDS counter) in BATCHES of 8 rows of one sub-slab.  The messages of a batch's 8 buckets are laid out as one flat stream (an
// empty bucket takes one placeholder position) and walked in groups of 16 positions = two half groups of 8 row loads, one half
// group in flight while the other is folded; bucket boundaries are wave-uniform flags inside the stream, so a wave has 8-16 KiB
// in flight whatever the bucket lengths are (a PPI-shaped batch has one message per self-loop bucket and up to ~550 per
// forward-edge bucket).  The next batch's bucket bounds and the (col, w) of the group four steps ahead are on their way
// meanwhile.  At a bucket's end the four sums of a lane are split (limb_split.h) and written as three ds_write_b64 into the
// sub-slab — pieces of 32 rows x 16 B padded to 528 B, so that the 32 pieces a wave writes for one row spread over all bank
// quads — and, when asked for, stored to S as fp32 (the weight gradient of a training step reads them: gnns/rgcn.py:96-98
// backward).
//
// First form (commit 6b8c427: half rows, 64-row x half-type sub-slabs, a switch re-entered behind every bucket end): bit-identical,
// 513 us against 164 us for the two kernels on the C2 batch; s_memtime stamps showed the gather waves 97 % busy issuing
// instructions (row-load wait 0.3 % of their time) and the matrix waves 91 % of the time in polls.
#include "common.h"
#include "handover.h"
#include "lds_dma.h"
#include "limb_split.h"

#include <stdlib.h>
#include <type_traits>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 528;              // 32 rows x 16 B (8 k of one limb) + 16 B: consecutive pieces start in consecutive bank quads
constexpr int PLANE = 32 * PIECE;       // the 32 (k-tile, k half) pieces of one limb of a sub-slab (32 rows x 256 columns)
constexpr int SLAB = 3 * PLANE;         // 3 limbs: 50 688 B
constexpr int NBUF = 3;
constexpr int GROUP = 16;               // stream positions per group
constexpr int HALF = 8;                 // row loads per half group
constexpr int BROWS = 4;                // rows per batch (8: the slowest of a tile pair's 8 batches held the other gather waves in polls 40 % of the time)
constexpr int BPS = 32 / BROWS;         // batches per sub-slab
constexpr int CTL = 16;                 // control words behind the slabs
constexpr uint32_t M_LAST = 1u << 31, M_EMPTY = 1u << 30, M_INVALID = 1u << 29;

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

__device__ __forceinline__ int lds_counter(int* p) { return handover_counter(p); }
__device__ __forceinline__ void compiler_fence() { handover_fence(); }

template <bool HAS_W>
__global__ __launch_bounds__(1024) void rgcn_fused_kernel(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * SLAB + CTL * 4];
  int* ctl = reinterpret_cast<int*>(lds + NBUF * SLAB);      // [0] next batch, [1..3] units filled, [4..6] consumer waves done
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int L = a.L;
  // My range of 32-row units: equal rows.  (Equal COST — messages + 8 L per row, boundaries by a 64-way search over rowptr at
  // kernel start — was measured: 231 us against 217 us.  At a granularity of 32 rows the message counts still differ by +-25 %,
  // and the slowest workgroups are not the ones with the most messages but the ones whose panel holds a 300-550 message bucket:
  // one gather wave folds it while the other seven run into the three-buffer limit.)
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  if (tid < CTL) ctl[tid] = 0;
  __syncthreads();
  const int nfull = nu >> 1;                                 // panels of two units; an odd unit left over is a panel of one row tile
  const int npan = (nu + 1) >> 1;
  const int nsub = 2 * L;                                    // sub-slabs per full panel: (edge type, row tile); the half panel has L
  const int nseq = nfull * nsub + (nu & 1) * L;
  const int rend = min((u0 + nu) * 32, a.V);                 // first row that is not mine
  const int ntiles = L * 16;
  bool dead = false;                                          // a poll gave up: stop waiting for anything
  const int spin_limit = handover_limit(a.status);
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
      if (++spins > spin_limit) { dead = true; if (lane == 0 && a.status) atomicOr(a.status, 1 + (wave < 8 ? 0 : 1)); break; }
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
    int b0 = 0, gen0 = 0;                                     // buffer / generation of the next sub-slab in sequence
    const int xlane = h32 * PIECE + i32 * 16;
    for (int pi = 0; pi < npan; ++pi) {
      const int m0 = (u0 + 2 * pi) * 32;
      const int rows_here = min(64, rend - m0);
      const bool two = pi < nfull;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      for (int l = 0; l < L; ++l) {
        int b1 = b0, gen1 = gen0;                              // (a half panel: one sub-slab per edge type)
        if (two) {
          if (++b1 == NBUF) { b1 = 0; ++gen1; }
        }
        poll(ctl + 1 + b0, 32 * (gen0 + 1));
        if (two) poll(ctl + 1 + b1, 32 * (gen1 + 1));
        const unsigned char* x0b = lds + b0 * SLAB + xlane;
        const unsigned char* x1b = lds + b1 * SLAB + xlane;
        TSTAMP(tk0);
        // (four k-tiles per trip = the period of the W ring; unrolling all sixteen, in a copy per panel height, made the kernel
        //  76 KB of code: more than the 64 KB instruction cache two CUs share)
#pragma nounroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const unsigned char* x0k = x0b + k4 * 8 * PIECE;
          const unsigned char* x1k = x1b + k4 * 8 * PIECE;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            wload(wr[(kt + 3) & 3]);
            const Frag x0 = xread(x0k + kt * 2 * PIECE);
            acc0 = products(acc0, wr[kt], x0);
            if (two) {
              const Frag x1 = xread(x1k + kt * 2 * PIECE);
              acc1 = products(acc1, wr[kt], x1);
            }
          }
        }
        wait_lgkm0();                                          // my reads of both buffers have returned
        compiler_fence();
        TSTAMP(tk1);
        TACC(6, tk1, tk0);
        if (lane == 0) {
          __hip_atomic_fetch_add(ctl + 4 + b0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (two) __hip_atomic_fetch_add(ctl + 4 + b1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        b0 = b1 + 1; gen0 = gen1;
        if (b0 == NBUF) { b0 = 0; ++gen0; }
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
          // (ReLU or nothing: act_rt's switch inlines tanhf / expf / erff thirty-two times — 30 KB of code that would share the
          //  instruction cache with the gather loop; the other activations take the two-kernel route)
          if (a.act == RELGNN_ACT_RELU) {
            v[0] = act_fwd<RELGNN_ACT_RELU>(v[0]); v[1] = act_fwd<RELGNN_ACT_RELU>(v[1]);
            v[2] = act_fwd<RELGNN_ACT_RELU>(v[2]); v[3] = act_fwd<RELGNN_ACT_RELU>(v[3]);
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
  // by name (the loop is unrolled by three steps) and never by copy.
  const int nbatches = nseq * BPS;
  // ---- the batch that has been drawn from the queue; its bucket bounds are on their way (lanes 0 .. BROWS-1)
  int nb_id = nbatches, nb_g = 0, nb_bi = 0, nb_m0 = 0, nb_sub = 0;
  int rp_b = 0, rp_e = 0;
  auto grab = [&]() {
    int id = 0;
    if (lane == 0) id = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    id = __builtin_amdgcn_readfirstlane(id);
    nb_id = id;
    if (id < nbatches) {
      nb_g = id / BPS; nb_bi = id % BPS;
      int pi = nb_g / nsub;
      nb_sub = nb_g - pi * nsub;                              // edge type = sub >> 1, row tile = sub & 1
      if (pi >= nfull) { pi = nfull; nb_sub = (nb_g - nfull * nsub) << 1; }      // the half panel: row tile 0 of every edge type
      nb_m0 = (u0 + 2 * pi) * 32;
    }
  };
  auto reload_bounds = [&]() {
    const int row = nb_m0 + (nb_sub & 1) * 32 + nb_bi * BROWS + lane;
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
    const int r = bb_r0 + j;                          // row inside the 32-row tile
    const int emp = (bb_empty >> j) & 1;
    const bool valid = bb_valid && lane < GROUP && f < bb_total;
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
    if (bb_valid) bb_pos += GROUP;
    TSTAMP(tq1);
    TACC(5, tq1, tq0);
  };

  const f32x4* H4 = reinterpret_cast<const f32x4*>(a.H);
  const int64_t ldh4 = a.ldh >> 2;
  auto issue = [&](const Group& c, auto half_c, f32x4 (&v)[HALF]) {
    constexpr int HF = decltype(half_c)::value;
    TSTAMP(ti0);
#pragma unroll
    for (int u = 0; u < HALF; ++u) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(c.col, HF * HALF + u);
      const f32x4* row = H4 + (int64_t)r * ldh4;                           // wave-uniform base
      v[u] = row[lane];
    }
    TSTAMP(ti1);
    TACC(6, ti1, ti0);
  };
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane = columns 4 lane .. 4 lane + 3 = k-tile lane >> 2, k half (lane >> 1) & 1, k 4 (lane & 1) .. + 3 of that half
  const int wr_lane = (lane >> 1) * PIECE + (lane & 1) * 8;
  auto finalize = [&](const Group& c, int r) {
    const int grow = c.m0 + (c.sub & 1) * 32 + r;
    if (a.S && grow < rend)
      *reinterpret_cast<f32x4*>(a.S + (int64_t)grow * a.lds_ + (c.sub >> 1) * 256 + 4 * lane) = acc;
    uint32_t h0, m0_, l0, h1, m1, l1;
    split_pair(acc[0], acc[1], h0, m0_, l0);
    split_pair(acc[2], acc[3], h1, m1, l1);
    if (__builtin_expect(max3_abs(max3_abs(acc[0], acc[1], acc[2]), acc[3], acc[3]) >= __uint_as_float(0x7F7F8000u), 0)) {
      split_pair_sat(acc[0], acc[1], h0, m0_, l0);
      split_pair_sat(acc[2], acc[3], h1, m1, l1);
    }
    const int fill = c.g % NBUF;
    unsigned char* p = lds + fill * SLAB + r * 16 + wr_lane;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0_, m1);
    *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
    wait_lgkm0();
    compiler_fence();
    if (lane == 0) __hip_atomic_fetch_add(ctl + 1 + fill, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // ONE copy of the bucket-end code per fold: the half group is walked segment by segment — a loop whose body offers every position
  // (statically indexed registers) a scalar bit test, adds the ones of the current segment and closes the bucket behind them.
  // (One copy per POSITION was ~60 KB of code over the eight fold sites, the size of the instruction cache two CUs share; a switch
  // re-entered behind every bucket end — the first form — came out of hipcc's structurizer as chains of mask arithmetic, 275
  // cycles per message; registers indexed by a scalar become a seven-deep select chain per column.)
  auto fold = [&](const Group& c, auto half_c, const f32x4 (&v)[HALF]) {
    constexpr int HF = decltype(half_c)::value;
    // everything but the HALF row loads issued last has landed: this half group's rows, and (col, w, meta) before them.  Stated
    // through the builtin, which hipcc's wait insertion understands (an inline-asm wait it does not see, and adds its own).
    // (gfx9 encoding: vmcnt 8 in bits 3:0, expcnt 7, lgkmcnt 15: no wait)
    static_assert(HALF == 8, "the immediate below says vmcnt(8)");
    TSTAMP(tw0);
    __builtin_amdgcn_s_waitcnt(0x0F78);
    TSTAMP(tw1);
    TACC(3, tw1, tw0);
    if (HF == 0 && (c.flags & 1)) poll(ctl + 4 + c.g % NBUF, 8 * (c.g / NBUF));   // the buffer's previous user has been consumed
    const uint64_t bl = __builtin_amdgcn_ballot_w64((c.meta & M_LAST) != 0u);
    const uint64_t bk = __builtin_amdgcn_ballot_w64((c.meta & (M_INVALID | M_EMPTY)) != 0u);
    uint32_t ends = (uint32_t)(bl >> (HF * HALF)) & 0xFFu;               // positions that close a bucket
    const uint32_t skip = (uint32_t)(bk >> (HF * HALF)) & 0xFFu;         // placeholders and positions past the batch
    uint32_t done = 0u;
#pragma clang loop unroll(disable)
    for (;;) {
      const int e = ends ? __builtin_ctz(ends) : HALF - 1;               // last position of the segment
      const uint32_t upto = (2u << e) - 1u;
      const uint32_t now = upto & ~done & ~skip;
      if (now == 0xFFu) {                                                  // the common case: eight messages of one bucket, no tests
#pragma unroll
        for (int u = 0; u < HALF; ++u) {
          const float wk = HAS_W ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w), HF * HALF + u)) : 1.f;
          const f32x4 t = f32x4{wk * v[u][0], wk * v[u][1], wk * v[u][2], wk * v[u][3]};
          acc = acc + t;
        }
      } else {
#pragma unroll
        for (int u = 0; u < HALF; ++u)
          if ((now >> u) & 1u) {
            const float wk = HAS_W ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w), HF * HALF + u)) : 1.f;
            const f32x4 t = f32x4{wk * v[u][0], wk * v[u][1], wk * v[u][2], wk * v[u][3]};  // two roundings (-ffp-contract=off)
            acc = acc + t;
          }
      }
      if (!ends) break;
      finalize(c, (int)((uint32_t)__builtin_amdgcn_readlane((int)c.meta, HF * HALF + e) & 31u));
      done = upto;
      ends &= ends - 1u;
      if (e == HALF - 1) break;
    }
    TSTAMP(tw2);
    TACC(4, tw2, tw1);
  };

  Group G0, G1, G2;
  grab();
  reload_bounds();
  prepare(G0); prepare(G1); prepare(G2);
  f32x4 va[HALF], vb[HALF];
  const std::integral_constant<int, 0> H0{};
  const std::integral_constant<int, 1> H1{};
  issue(G0, H0, va);
  // step k: second half of group k requested, first half folded, first half of group k + 1 requested, second half folded,
  // (col, w) of group k + 3 requested (read a step and a half later).  Three steps per trip: the group sets rotate by name.
  for (;;) {
    if (G0.flags & 2) break;
    issue(G0, H1, vb); fold(G0, H0, va); issue(G1, H0, va); fold(G0, H1, vb); prepare(G0);
    if (G1.flags & 2) break;
    issue(G1, H1, vb); fold(G1, H0, va); issue(G2, H0, va); fold(G1, H1, vb); prepare(G1);
    if (G2.flags & 2) break;
    issue(G2, H1, vb); fold(G2, H0, va); issue(G0, H0, va); fold(G2, H1, vb); prepare(G2);
  }
#ifdef RELGNN_FUSED_TIMING
  tflush();
#endif
}

}  // namespace

extern "C" {

int relgnn_rgcn_fused_fwd(const float* H, int64_t num_rows_h, int64_t ldh, const int32_t* rowptr, int32_t num_nodes,
                          int32_t num_edge_types, const int32_t* col, const float* w, const uint16_t* w_limbs, const float* bias,
                          int32_t act, float* bucket_sums, int64_t lds, float* out, int64_t ldo, int32_t d_in, int32_t d_out,
                          int32_t* status, void* stream) {
  if (num_nodes < 0 || num_edge_types <= 0 || d_in < 0 || d_out < 0 || num_rows_h < 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU)
    return RELGNN_EINVAL;
  if (num_nodes == 0 || d_out == 0) return RELGNN_OK;
  if (!H || !rowptr || !w_limbs || !out) return RELGNN_EINVAL;
  if (d_in != 256 || d_out != 256 || num_edge_types > 64 || (act != RELGNN_ACT_LINEAR && act != RELGNN_ACT_RELU))
    return RELGNN_EUNSUPPORTED;
  if (ldh < d_in || ldo < d_out || (bucket_sums && lds < (int64_t)num_edge_types * d_in)) return RELGNN_EINVAL;
  if (!aligned16(H) || !aligned16(out) || !aligned16(w_limbs) || (bias && !aligned16(bias)) || (bucket_sums && !aligned16(bucket_sums)) ||
      ldh % 4 || ldo % 4 || lds % 4)
    return RELGNN_EUNSUPPORTED;
  FusedArgs a{};
  a.H = H; a.ldh = ldh; a.rowptr = rowptr; a.col = col; a.w = w; a.B = w_limbs; a.bias = bias; a.S = bucket_sums; a.lds_ = lds;
  a.C = out; a.ldc = ldo; a.V = num_nodes; a.L = num_edge_types; a.act = act; a.status = status;
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

}  // extern "C"
