// The Keras GRUCell of sparse_ggnn_layer (gnns/ggnn.py:92 through utils/utils.py:15-16; TF 1.13: reset_after=False, recurrent
// activation hard_sigmoid, gate order z, r, h) as ONE kernel per pass, for 128 units over 128-wide inputs:
//
//     [z | r] = hs([x | h] @ [[K_z K_r]; [U_z U_r]] + b_zr)              P1: K = 256, N = 256
//     hh      = act([x | r * h] @ [K_h; U_h] + b_h)                      P2: K = 256, N = 128
//     out     = z * h + (1 - z) * hh
//
// P2's left operand needs r * h for whole rows — which P1 of the SAME rows produces: a workgroup that owns a row panel can do both,
// with r * h handed from P1's epilogue to P2's k-loop through LDS.  The composition it replaces (round 6: three limb products, the
// gate kernel and the output kernel of gru.hip: 171 us per cell on C3's 50 k-node batch, five launches) wrote and re-read
// [V, 384] + [V, 256] + [V, 128] pre-activations in between.
//
// Structure: limb_gemm_pc.hip's (this file follows it line by line where it can).  A persistent 16-wave workgroup per CU owns a
// contiguous range of 32-row units, taken as 64-row panels; eight PRODUCER waves stream the panel's rows of x and of h, split them
// into three bf16 limbs (limb_split.h) and write four sub-slabs (32 rows x 128 k each, six rotate through LDS): x tile 0, x tile 1,
// h tile 0, h tile 1.  Eight MATRIX waves, W fragments straight from L2 into registers three k-tiles ahead:
//   waves 0-3 (z):  z columns 32 w .. 32 w + 31 of both row tiles over the 16 k-tiles of P1; epilogue hs(. + b_z) -> z;
//                   then the same columns of hh over the 16 k-tiles of P2 (x sub-slabs, then r * h where h was); epilogue act, blend.
//   waves 4-7 (r):  r columns of both row tiles (P1); epilogue r = hs(. + b_r), r * h with h read from global memory (the exact
//                   fp32 values), split into limbs and written over the h sub-slabs — once all eight waves are through P1's k-loop.
// Hand-over by monotonic counters in LDS, bounded polls (handover.h): buffers filled / consumed as in limb_gemm_pc.hip, plus
// "P1's reads done" (8 per panel) and "r * h written" (4 per panel).  P2 keeps only the z waves busy (a third more matrix time
// per panel than a balanced split, which would exchange partial accumulators through LDS): at three panels per CU the kernel is
// bound by its pipeline's fill, not by the matrix pipe.
// The products are the three-limb products of limb_gemm.hip (same limbs, same six-product order per k-tile); x @ K + h @ U is ONE
// accumulation over k = [x | h] here, started from the bias, where the composition added separately rounded sums: the results differ
// from it in the last bits (tests: float64 and the oracle, not the composition's bits).
#include "common.h"
#include "handover.h"
#include "lds_dma.h"
#include "limb_split.h"

#include <stdlib.h>
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
constexpr int CTL = 16;                 // control words: [1..6] rows filled per buffer, [8..13] matrix waves done with it,
                                        // [14] matrix waves through P1's k-loop, [15] r waves whose r * h is written
constexpr int U = 128;                  // units = input width

struct GruArgs {
  const float* X; int64_t ldx;           // [V, 128] inputs (the aggregated messages)
  const float* H; int64_t ldh;           // [V, 128] states
  const uint16_t* B1;                    // limb image of the [256, 256] right operand of P1: rows = z | r columns, k = [x | h]
  const uint16_t* B2;                    // limb image of the [128, 256] right operand of P2: rows = candidate columns, k = [x | r * h]
  const float* bias;                     // [384]: b_z | b_r | b_h
  float* Z; float* R; float* RH; float* HH;   // [V, 128] each, dense rows: what the backward reads (all four or none)
  float* OUT;                            // [V, 128]
  int32_t V, act;
  int32_t units_base, units_rem, groups;
  int32_t* status;
#ifdef RELGNN_GRU_TIMING
  unsigned long long* timing;
#endif
};

struct Frag { bf16x8 hi, mid, lo; };

#ifdef RELGNN_GRU_TIMING
// diagnostic build (scripts/build_timing_variant.sh gru_cell RELGNN_GRU_TIMING): s_memtime totals per wave, [workgroup][wave][8]:
// 0 total, 1 in polls, 2 polls that waited, 3 k-loop halves, 4 gate epilogue (+ r * h write), 5 blend epilogue / producers: row wait,
// 6 producers: split + write
unsigned long long* g_gru_timing = nullptr;
#define TSTAMP(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define TACC(slot, t1, t0) tacc[slot] += (t1) - (t0)
#else
#define TSTAMP(v)
#define TACC(slot, t1, t0)
#endif

// element at `base + byte offset`: a uniform base and a 32-bit per-lane byte offset (the global_load / global_store form that
// needs no per-lane 64-bit address)
__device__ __forceinline__ f32x4 ld4(const float* base, uint32_t boff) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ void st4(float* base, uint32_t boff, f32x4 v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + boff) = v;
}
// the value as the compiler must take it here (an offset it cannot compute ahead of the k-loops: the addresses of every epilogue
// access, hoisted to the top of a panel, were seventy spilled registers reloaded one round trip at a time)
__device__ __forceinline__ uint32_t here(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }   // (gru.hip's expression)

template <bool TANH>
__global__ __launch_bounds__(1024) void gru_cell_fwd_kernel(const GruArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * SLAB + CTL * 4 + 3 * U * 4];      // sub-slabs, control words, bias
  int* ctl = reinterpret_cast<int*>(lds + NBUF * SLAB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  if (tid < CTL) ctl[tid] = 0;
  if (tid >= 64 && tid < 64 + 3 * U / 4)
    reinterpret_cast<f32x4*>(lds + NBUF * SLAB + CTL * 4)[tid - 64] = reinterpret_cast<const f32x4*>(a.bias)[tid - 64];
  __syncthreads();
  const int nfull = nu >> 1;                                 // panels of two units; an odd unit left over is a panel of one row tile
  const int npan = (nu + 1) >> 1;
  const int nseq = nfull * 4 + (nu & 1) * 2;                 // sub-slabs: (x, tile 0), (x, tile 1), (h, tile 0), (h, tile 1) per panel
  const int rend = min((u0 + nu) * 32, a.V);
  bool dead = false;
  const int spin_limit = handover_limit(a.status);
#ifdef RELGNN_GRU_TIMING
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
    while (__builtin_amdgcn_readfirstlane(handover_counter(p)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) { dead = true; if (lane == 0 && a.status) atomicOr(a.status, 4 + (wave < 8 ? 0 : 4)); break; }
    }
    handover_fence();
#ifdef RELGNN_GRU_TIMING
    TSTAMP(tp1);
    tacc[1] += tp1 - tp0;
    if (spins) tacc[2] += 1;
#endif
  };

  if (wave < 8) {
    // =================================================== matrix waves ===================================================
    const bool zrole = wave < 4;
    const int i32 = lane & 31, h32 = lane >> 5;
    Frag wr[4];
    // the W fragments a wave will need, in order: the 16 k-tiles of its column group of B1 (every panel), and for a z wave the 16
    // k-tiles of its column group of B2 behind them
    const uint16_t* w1 = a.B1 + (int64_t)wave * 16 * 1536 + 8 * lane;
    const uint16_t* w2 = a.B2 + (int64_t)(wave & 3) * 16 * 1536 + 8 * lane;
    const int wlen = zrole ? 32 : 16;
    int wt = 0;
    auto wload = [&](Frag& f) {
      const uint16_t* p = wt < 16 ? w1 + wt * 1536 : w2 + (wt - 16) * 1536;
      f.hi = *reinterpret_cast<const bf16x8*>(p);
      f.mid = *reinterpret_cast<const bf16x8*>(p + 512);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
      if (++wt == wlen) wt = 0;
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
    int b0 = 0, gen0 = 0;                                     // buffer / generation of the panel's first sub-slab
    const int xlane = h32 * PIECE + i32 * 16;
    auto buf_of = [&](int i) { const int b = b0 + i; return b >= NBUF ? b - NBUF : b; };      // i <= 3 < NBUF
    auto poll_buf = [&](int i) {
      const int b = b0 + i;
      if (b >= NBUF) poll(ctl + 1 + b - NBUF, 32 * (gen0 + 2)); else poll(ctl + 1 + b, 32 * (gen0 + 1));
    };
    auto release = [&](int n) {
      wait_lgkm0();                                            // my reads of (and writes to) these buffers are done
      handover_fence();
      if (lane == 0)
        for (int i = 0; i < n; ++i) __hip_atomic_fetch_add(ctl + 8 + buf_of(i), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      b0 += n;
      if (b0 >= NBUF) { b0 -= NBUF; ++gen0; }
    };
    const int cw = (wave & 3) * 32;                            // my 32 columns of the 128
    float* const zpark = a.Z ? a.Z : a.OUT;                    // an inference pass keeps z where the output will go
    const float* const lbias = reinterpret_cast<const float*>(lds + NBUF * SLAB + CTL * 4) + cw + 4 * h32;
    // the accumulators start from the bias of their columns (a column's bias is the same in every row; the kernel's first
    // instructions put the 384 values into LDS): nothing to add in the epilogues
    auto from_bias = [&](f32x16& acc0, f32x16& acc1, int first) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + first + 8 * c);
        acc0[4 * c] = bv[0]; acc0[4 * c + 1] = bv[1]; acc0[4 * c + 2] = bv[2]; acc0[4 * c + 3] = bv[3];
      }
      acc1 = acc0;
    };
    for (int pi = 0; pi < npan; ++pi) {
      const int m0 = (u0 + 2 * pi) * 32;
      const int rows_here = min(64, rend - m0);
      const bool two = pi < nfull;
      const int per = two ? 2 : 1;
      // 32 x 32 tile: lane holds row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register 4 c + {0..3}).
      // Epilogues, per row tile: every load first, then the arithmetic, then every store.  Written load / compute / store per column
      // group, the stores of one group stood between the loads of the next — z may be parked where the output goes, so they may
      // alias — and every group paid a store's and a load's round trip: 60 % of a z wave's time (scripts/bench_gru_cell_timing.py).
      // Addresses: a uniform base + a 32-bit byte offset made HERE (here()).  (Both row tiles' loads in front of everything, with
      // the W ring's three requests dropped and repeated behind the blend to make room: 96 spilled registers, some inside the
      // k-loops.)
      auto row_off = [&](int r, uint32_t row_bytes) {
        return ((uint32_t)m0 + here((uint32_t)r)) * row_bytes + (cw + 4 * h32) * 4;
      };
      auto gate_z = [&](const f32x16& acc, int r) {
        if (r >= rows_here) return;
        const uint32_t off = row_off(r, U * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          st4(zpark, off + 32 * c, f32x4{hard_sigmoid(acc[4 * c]), hard_sigmoid(acc[4 * c + 1]), hard_sigmoid(acc[4 * c + 2]),
                                         hard_sigmoid(acc[4 * c + 3])});
      };
      // r = hs(.); r * h replaces the accumulator
      auto gate_r = [&](f32x16& acc, int r) {
        const bool live = r < rows_here;
        const int rr = live ? r : 0;                          // (a row past the end: a valid address, the product is never stored)
        const uint32_t off = row_off(rr, U * 4), offh = row_off(rr, (uint32_t)a.ldh * 4);
        f32x4 hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) hv[c] = ld4(a.H, offh + 32 * c);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = f32x4{hard_sigmoid(acc[4 * c]), hard_sigmoid(acc[4 * c + 1]), hard_sigmoid(acc[4 * c + 2]),
                                hard_sigmoid(acc[4 * c + 3])};
          const f32x4 pr = f32x4{live ? v[0] * hv[c][0] : 0.f, live ? v[1] * hv[c][1] : 0.f, live ? v[2] * hv[c][2] : 0.f,
                                 live ? v[3] * hv[c][3] : 0.f};
          if (a.R && live) {
            st4(a.R, off + 32 * c, v);
            st4(a.RH, off + 32 * c, pr);
          }
          acc[4 * c] = pr[0]; acc[4 * c + 1] = pr[1]; acc[4 * c + 2] = pr[2]; acc[4 * c + 3] = pr[3];
        }
      };
      // a 32 x 32 patch of r * h as limbs, in the producers' layout: 4 columns = 8 bytes per limb at
      // row * 16 + (col4 >> 1) * PIECE + (col4 & 1) * 8, col4 = column / 4
      auto put = [&](const f32x16& acc, int buf) {
        unsigned char* base = lds + buf * SLAB + i32 * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col4 = (cw >> 2) + 2 * c + h32;
          uint32_t h0, m0_, l0, h1, m1, l1;
          split_pair(acc[4 * c], acc[4 * c + 1], h0, m0_, l0);
          split_pair(acc[4 * c + 2], acc[4 * c + 3], h1, m1, l1);
          if (__builtin_expect(max3_abs(max3_abs(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2]), acc[4 * c + 3], acc[4 * c + 3]) >=
                               __uint_as_float(0x7F7F8000u), 0)) {
            split_pair_sat(acc[4 * c], acc[4 * c + 1], h0, m0_, l0);
            split_pair_sat(acc[4 * c + 2], acc[4 * c + 3], h1, m1, l1);
          }
          unsigned char* o = base + (col4 >> 1) * PIECE + (col4 & 1) * 8;
          *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(o + PLANE) = make_uint2(m0_, m1);
          *reinterpret_cast<uint2*>(o + 2 * PLANE) = make_uint2(l0, l1);
        }
      };
      auto blend = [&](const f32x16& acc, int r) {
        if (r >= rows_here) return;
        const uint32_t off = row_off(r, U * 4), offh = row_off(r, (uint32_t)a.ldh * 4);
        f32x4 zz[4], hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          zz[c] = ld4(zpark, off + 32 * c);                    // (my own stores of this panel)
          hv[c] = ld4(a.H, offh + 32 * c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 t = f32x4{acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
          if constexpr (TANH) {
            // (v_exp_f32 / v_rcp_f32, 2e-7 absolute: common.h; tanhf inlines sixteen times per tile into branchy code)
            t = f32x4{tanh_fast(t[0]), tanh_fast(t[1]), tanh_fast(t[2]), tanh_fast(t[3])};
          } else if (a.act == RELGNN_ACT_RELU) {
            t = f32x4{act_fwd<RELGNN_ACT_RELU>(t[0]), act_fwd<RELGNN_ACT_RELU>(t[1]), act_fwd<RELGNN_ACT_RELU>(t[2]),
                      act_fwd<RELGNN_ACT_RELU>(t[3])};
          } else if (a.act == RELGNN_ACT_LEAKY_RELU) {
            t = f32x4{act_fwd<RELGNN_ACT_LEAKY_RELU>(t[0]), act_fwd<RELGNN_ACT_LEAKY_RELU>(t[1]),
                      act_fwd<RELGNN_ACT_LEAKY_RELU>(t[2]), act_fwd<RELGNN_ACT_LEAKY_RELU>(t[3])};
          }
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = zz[c][j] * hv[c][j] + (1.0f - zz[c][j]) * t[j];       // (gru.hip's expression)
          if (a.HH) st4(a.HH, off + 32 * c, t);
          st4(a.OUT, off + 32 * c, o);
        }
      };
      // One call site for the k-tiles (two copies of them spill registers into the k-loops: limb_gemm_pc.hip, and again here).  Pass 0
      // is P1 (all eight waves), pass 1 P2 (the z waves): 16 k-tiles each, 0-7 from the x sub-slabs (buffers 0 .. per - 1), 8-15 from
      // the h / r * h sub-slabs behind them.
      const int passes = zrole ? 2 : 1;
      for (int ps = 0; ps < passes; ++ps) {
        f32x16 acc0, acc1;
        from_bias(acc0, acc1, ps ? 2 * U : (zrole ? 0 : U));
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
          // what this half reads: P1 the panel's x (hs 0) / h (hs 1) sub-slabs as the producers fill them — an r wave is through
          // its panel while the z waves are in P2 and takes the NEXT panel's x half meanwhile (its h sub-slabs are the buffers
          // the z waves still read); P2 the same x sub-slabs, then r * h where h was: the r waves' epilogue hides under the x half
          if (ps == 0) {
            for (int i = 0; i < per; ++i) poll_buf(hs * per + i);
          } else if (hs == 1) {
            poll(ctl + 15, 4 * (pi + 1));                      // the four r waves have written r * h over the h sub-slabs
          }
          TSTAMP(tk0);
          const unsigned char* x0b = lds + buf_of(hs * per) * SLAB + xlane;
          const unsigned char* x1b = lds + buf_of(hs * per + per - 1) * SLAB + xlane;
          Frag x0 = xread(x0b), x1 = x0;
          if (two) x1 = xread(x1b);
#pragma unroll
          for (int kt = 0; kt < 8; ++kt) {
            wload(wr[(kt + 3) & 3]);
            __builtin_amdgcn_s_waitcnt(0x0F79);               // vmcnt(9): the W fragments of this k-tile have landed, three k-tiles stay in flight
            acc0 = products(acc0, wr[kt & 3], x0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < 8) x0 = xread(x0b + (kt + 1) * 2 * PIECE);
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
              acc1 = products(acc1, wr[kt & 3], x1);
              __builtin_amdgcn_sched_barrier(0);
              if (kt + 1 < 8) x1 = xread(x1b + (kt + 1) * 2 * PIECE);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          TSTAMP(tk1);
          TACC(3, tk1, tk0);
        }
        TSTAMP(te0);
        if (ps == 0) {
          wait_lgkm0();                                        // my P1 reads of the h sub-slabs have returned
          handover_fence();
          if (lane == 0) __hip_atomic_fetch_add(ctl + 14, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (zrole) {
            gate_z(acc0, i32);
            if (two) gate_z(acc1, 32 + i32);
          } else {
            gate_r(acc0, i32);
            if (two) gate_r(acc1, 32 + i32);
            poll(ctl + 14, 8 * (pi + 1));                      // nobody reads the h sub-slabs any more
            put(acc0, buf_of(per));
            if (two) put(acc1, buf_of(per + 1));
            wait_lgkm0();
            handover_fence();
            if (lane == 0) __hip_atomic_fetch_add(ctl + 15, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            release(2 * per);
          }
        } else {
          release(2 * per);
          blend(acc0, i32);
          if (two) blend(acc1, 32 + i32);
        }
#ifdef RELGNN_GRU_TIMING
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // (vmcnt(0): the epilogue's loads and stores are inside its stamp)
        TSTAMP(te1);
        TACC(ps == 0 ? 4 : 5, te1, te0);
#endif
      }
    }
#ifdef RELGNN_GRU_TIMING
    tflush();
#endif
    return;
  }

  // ===================================================== producer waves =====================================================
  // limb_gemm_pc.hip's producers with two sources: sub-slabs 0, 1 of a panel are rows of x, 2, 3 rows of h.  Wave p streams rows
  // 4 p .. 4 p + 3 of every sub-slab: two 1 KiB loads (two rows x 128 k each: lane = row (lane >> 5), 4 k), the split, three 8-byte
  // LDS writes per load; the loads of the three sub-slabs behind the current one are in flight.
  const int pw = wave - 8;
  const int col4 = lane & 31, rsub = lane >> 5;
  const int wr_lane = (col4 >> 1) * PIECE + (col4 & 1) * 8;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(a.X);
  const f32x4* H4 = reinterpret_cast<const f32x4*>(a.H);
  const int64_t ldx4 = a.ldx >> 2, ldh4 = a.ldh >> 2;
  struct Pos { int g, pi, hs, tm; };                          // sequence position -> (panel, source, row tile)
  auto advance = [&](Pos& p) {
    ++p.g;
    if (p.pi < nfull && p.tm == 0) { p.tm = 1; return; }
    p.tm = 0;
    if (++p.hs == 2) { p.hs = 0; ++p.pi; }
  };
  auto row0 = [&](const Pos& p) { return (u0 + 2 * p.pi) * 32 + p.tm * 32 + 4 * pw + rsub; };
  auto issue = [&](const Pos& p, f32x4 (&v)[2]) {
    const int r0 = row0(p);
    const f32x4* S4 = p.hs ? H4 : X4;
    const int64_t ld4 = p.hs ? ldh4 : ldx4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = p.g < nseq && r0 + 2 * j < rend;
      v[j] = S4[(ok ? (int64_t)(r0 + 2 * j) * ld4 : 0) + col4];                       // (row 0 exists: V > 0)
    }
  };
  auto process = [&](const Pos& p, f32x4 (&v)[2]) {
    TSTAMP(tw0);
    __builtin_amdgcn_s_waitcnt(0x0F76);                        // vmcnt(6): everything but the six loads issued last has landed
    TSTAMP(tw1);
    TACC(5, tw1, tw0);
    const int fill = p.g % NBUF, gen = p.g / NBUF;
    poll(ctl + 8 + fill, 8 * gen);                            // the buffer's previous user has been consumed by the eight matrix waves
    const int r0 = row0(p);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 x = v[j];
      if (r0 + 2 * j >= rend) x = f32x4{0.f, 0.f, 0.f, 0.f};
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
    TSTAMP(tw2);
    TACC(6, tw2, tw1);
  };
  Pos p0{0, 0, 0, 0}, p1, p2, p3;
  f32x4 v0[2], v1[2], v2[2], v3[2];
  p1 = p0; advance(p1); p2 = p1; advance(p2); p3 = p2; advance(p3);
  issue(p0, v0); issue(p1, v1); issue(p2, v2);
  for (;;) {                                                   // step: the loads of sub-slab g + 3, then the split of sub-slab g
    if (p0.g >= nseq) break;
    issue(p3, v3); process(p0, v0); p0 = p3; advance(p0);
    if (p1.g >= nseq) break;
    issue(p0, v0); process(p1, v1); p1 = p0; advance(p1);
    if (p2.g >= nseq) break;
    issue(p1, v1); process(p2, v2); p2 = p1; advance(p2);
    if (p3.g >= nseq) break;
    issue(p2, v2); process(p3, v3); p3 = p2; advance(p3);
  }
#ifdef RELGNN_GRU_TIMING
  tflush();
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The backward of the same cell (what gru.hip's gru_out_bwd / gru_gates_bwd and three limb products did in seven launches, the
// gradients of the z / r pre-activations and of r * h going through HBM in between):
//     gpre = g (1 - z) act'(hh);   gzp = g (h - hh) hs'(z)                                            (producer waves, elementwise)
//     [gx_h | grh] = gpre @ [K_h; U_h]^T                                                              B2: K = 128, N = 256
//     grp = grh h hs'(r);   gh0 = g z + grh r                                                         (epilogue of the grh columns)
//     [gx | gh] = [gx_h | gh0] + [gzp | grp] @ [[K_z K_r]; [U_z U_r]]^T                               B4: K = 256, N = 256
//     gxk = [gzp | grp | gpre]   (the right operand of the cell's weight gradients: relgnn_gemm_tn_stream_group_f32)
// Eight matrix waves, wave w = output columns 32 w .. 32 w + 31 of the 256 (0-3: the x half, 4-7: the h half), both row tiles of a
// 64-row panel; the accumulators of B2 ARE the start of B4's (the x half untouched, the h half after its epilogue).  Six LDS
// sub-slabs with fixed jobs: 0, 1 gpre (tiles 0, 1), 2, 3 gzp — both from the producer waves, which read g, z, h, hh once and
// also store the two thirds of gxk they make —, 4, 5 grp, written by waves 4-7 behind B2.  Counters as in the forward kernel.
struct GruBwdArgs {
  const float* G; int64_t ldg;           // [V, 128] gradient of the new states
  const float* Z; const float* R; const float* HH;   // [V, 128] dense: what the forward kept
  const float* H; int64_t ldh;           // [V, 128] states
  const uint16_t* B2;                    // limb image of [K[:, 2u:]; U[:, 2u:]] as [256, 128] (n = x | h columns, k = candidate columns)
  const uint16_t* B4;                    // limb image of [K[:, :2u]; U[:, :2u]] as [256, 256] (n = x | h columns, k = z | r columns)
  float* GXK;                            // [V, 384] dense
  float* GX; float* GH;                  // [V, 128] dense
  int32_t V, act;
  int32_t units_base, units_rem, groups;
  int32_t* status;
};

__device__ __forceinline__ float hs_grad(float y) { return (y > 0.f && y < 1.f) ? 0.2f : 0.f; }      // (gru.hip's expressions)
__device__ __forceinline__ float act_grad_out(int act, float y) {
  switch (act) {
    case RELGNN_ACT_TANH: return 1.f - y * y;
    case RELGNN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RELGNN_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.2f;
    default: return 1.f;
  }
}

constexpr int BW_BUFS = 4;              // sub-slabs of the backward kernel: 0, 1 gpre then grp (row tiles 0, 1), 2, 3 gzp
constexpr int EROW = 528;               // bytes per row of the fp32 exchange tiles (128 floats + 16 bytes: rows start in different banks)

__global__ __launch_bounds__(1024) void gru_cell_bwd_kernel(const GruBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[BW_BUFS * SLAB + 64 * EROW + 32 * 4];
  unsigned char* const ex = lds + BW_BUFS * SLAB;             // [2 row tiles][32 rows][EROW]: grh from the h waves, gh0 back to them
  // counters (monotonic): [0, 1] rows of gpre filled (tile 0, 1), [2, 3] rows of gzp, [4, 5] rows of grp (and gh0 in the exchange
  // tile), [6] matrix waves through B2's k-loop, [7] h waves whose grh is in the exchange tiles, [8] matrix waves through the gzp
  // part, [9] through the grp part
  int* ctl = reinterpret_cast<int*>(lds + BW_BUFS * SLAB + 64 * EROW);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  if (tid < 32) ctl[tid] = 0;
  __syncthreads();
  const int nfull = nu >> 1;
  const int npan = (nu + 1) >> 1;
  const int rend = min((u0 + nu) * 32, a.V);
  bool dead = false;
  const int spin_limit = handover_limit(a.status);
  auto poll = [&](int* p, int target) {
    if (dead) return;
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(handover_counter(p)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) { dead = true; if (lane == 0 && a.status) atomicOr(a.status, 4 + (wave < 8 ? 0 : 4)); break; }
    }
    handover_fence();
  };
  auto bump = [&](int* p, int by) {                           // (behind the LDS accesses it reports)
    wait_lgkm0();
    handover_fence();
    if (lane == 0) __hip_atomic_fetch_add(p, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // four columns of a row tile as limbs into a sub-slab: 8 bytes per limb at row * 16 + (col4 >> 1) * PIECE + (col4 & 1) * 8
  auto put4 = [&](unsigned char* slab, int row, int col4, f32x4 x) {
    uint32_t h0, m0_, l0, h1, m1, l1;
    split_pair(x[0], x[1], h0, m0_, l0);
    split_pair(x[2], x[3], h1, m1, l1);
    if (__builtin_expect(max3_abs(max3_abs(x[0], x[1], x[2]), x[3], x[3]) >= __uint_as_float(0x7F7F8000u), 0)) {
      split_pair_sat(x[0], x[1], h0, m0_, l0);
      split_pair_sat(x[2], x[3], h1, m1, l1);
    }
    unsigned char* o = slab + row * 16 + (col4 >> 1) * PIECE + (col4 & 1) * 8;
    *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(o + PLANE) = make_uint2(m0_, m1);
    *reinterpret_cast<uint2*>(o + 2 * PLANE) = make_uint2(l0, l1);
  };

  if (wave < 8) {
    // =================================================== matrix waves ===================================================
    // MFMAs, LDS and the final stores only: every elementwise step is the producers' (they hold g, z, h, r of the rows anyway; an
    // epilogue that loaded them here — 32 more live registers next to two accumulators and the W ring — spilled 210).
    const bool hrole = wave >= 4;
    const int i32 = lane & 31, h32 = lane >> 5;
    const int cw = (wave & 3) * 32;
    Frag wr[4];
    // the W fragments of a panel, in order: the 8 k-tiles of my column group of B2, the 16 of my column group of B4.
    // (a uniform base + one per-lane byte offset: with per-lane pointers and every wave on the same 24-tile cycle the compiler
    //  computed the 48 fragment addresses of a panel ahead of the loop — 96 registers, spilled)
    const char* const w2 = reinterpret_cast<const char*>(a.B2) + (int64_t)wave * 8 * 3072;
    const char* const w4 = reinterpret_cast<const char*>(a.B4) + (int64_t)wave * 16 * 3072;
    const uint32_t wlane = 16 * lane;
    int wt = 0;
    auto wload = [&](Frag& f) {
      const char* p = wt < 8 ? w2 + wt * 3072 : w4 + (wt - 8) * 3072;
      f.hi = *reinterpret_cast<const bf16x8*>(p + wlane);
      f.mid = *reinterpret_cast<const bf16x8*>(p + wlane + 1024);
      f.lo = *reinterpret_cast<const bf16x8*>(p + wlane + 2048);
      if (++wt == 24) wt = 0;
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
    const int xlane = h32 * PIECE + i32 * 16;
    unsigned char* const epatch = ex + i32 * EROW + (cw + 4 * h32) * 4;        // my patch of exchange tile 0 (tile 1: + 32 rows)
    for (int pi = 0; pi < npan; ++pi) {
      const int m0 = (u0 + 2 * pi) * 32;
      const int rows_here = min(64, rend - m0);
      const bool two = pi < nfull;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      // three parts of eight k-tiles through ONE call site: B2 over gpre (sub-slabs 0, 1), B4 over gzp (2, 3), B4 over grp (0, 1 again)
#pragma clang loop unroll(disable)
      for (int part = 0; part < 3; ++part) {
        const int bt = part == 1 ? 2 : 0;
        int* const filled = ctl + (part == 0 ? 0 : part == 1 ? 2 : 4);
        poll(filled, 32 * (pi + 1));
        if (two) poll(filled + 1, 32 * (pi + 1));
        const unsigned char* x0b = lds + bt * SLAB + xlane;
        const unsigned char* x1b = lds + (two ? bt + 1 : bt) * SLAB + xlane;
        Frag x0 = xread(x0b), x1 = x0;
        if (two) x1 = xread(x1b);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          wload(wr[(kt + 3) & 3]);
          __builtin_amdgcn_s_waitcnt(0x0F79);                 // vmcnt(9): the W fragments of this k-tile have landed, three k-tiles stay in flight
          acc0 = products(acc0, wr[kt & 3], x0);
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < 8) x0 = xread(x0b + (kt + 1) * 2 * PIECE);
          __builtin_amdgcn_sched_barrier(0);
          if (two) {
            acc1 = products(acc1, wr[kt & 3], x1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < 8) x1 = xread(x1b + (kt + 1) * 2 * PIECE);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (part == 0) {
          if (hrole) {
            // grh to the producers through the exchange tiles; B4 starts from zero here, gh0 = g z + grh r comes back at the end
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              *reinterpret_cast<f32x4*>(epatch + 32 * c) = f32x4{acc0[4 * c], acc0[4 * c + 1], acc0[4 * c + 2], acc0[4 * c + 3]};
              if (two)
                *reinterpret_cast<f32x4*>(epatch + 32 * EROW + 32 * c) = f32x4{acc1[4 * c], acc1[4 * c + 1], acc1[4 * c + 2], acc1[4 * c + 3]};
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            bump(ctl + 7, 1);
          }
          bump(ctl + 6, 1);
        } else {
          bump(ctl + 7 + part, 1);
        }
      }
      // [gx | gh]: my 32 columns of both row tiles
      if (hrole) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(epatch + 32 * c);
          acc0[4 * c] += e0[0]; acc0[4 * c + 1] += e0[1]; acc0[4 * c + 2] += e0[2]; acc0[4 * c + 3] += e0[3];
          if (two) {
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(epatch + 32 * EROW + 32 * c);
            acc1[4 * c] += e1[0]; acc1[4 * c + 1] += e1[1]; acc1[4 * c + 2] += e1[2]; acc1[4 * c + 3] += e1[3];
          }
        }
      }
      float* const dst = hrole ? a.GH : a.GX;
      if (i32 < rows_here) {
        const uint32_t off = ((uint32_t)m0 + here((uint32_t)i32)) * (U * 4) + (cw + 4 * h32) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) st4(dst, off + 32 * c, f32x4{acc0[4 * c], acc0[4 * c + 1], acc0[4 * c + 2], acc0[4 * c + 3]});
      }
      if (two && 32 + i32 < rows_here) {
        const uint32_t off = ((uint32_t)m0 + here((uint32_t)(32 + i32))) * (U * 4) + (cw + 4 * h32) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) st4(dst, off + 32 * c, f32x4{acc1[4 * c], acc1[4 * c + 1], acc1[4 * c + 2], acc1[4 * c + 3]});
      }
    }
    return;
  }

  // ===================================================== producer waves =====================================================
  // wave p owns rows 4 p .. 4 p + 3 of a row tile: lane = row (lane >> 5) of a pair, 4 columns.  Per panel, first
  //   g, z, h, hh of its rows -> gpre into sub-slab 0 / 1 and gxk[:, 2u:], gzp into sub-slab 2 / 3 and gxk[:, :u],
  // then, behind B2 (whose grh the h waves leave in the exchange tiles),
  //   grp = grh h hs'(r) over the gpre sub-slabs (nobody reads those any more) and into gxk[:, u:2u]; gh0 = g z + grh r back into
  //   the exchange tiles.
  const int pw = wave - 8;
  const int col4 = lane & 31, rsub = lane >> 5;
  for (int pi = 0; pi < npan; ++pi) {
    const int tiles = pi < nfull ? 2 : 1;
    f32x4 gv[2][2], zv[2][2], hv[2][2], rv[2][2];
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
      const int r0 = (u0 + 2 * pi) * 32 + 32 * t + 4 * pw + rsub;
      f32x4 cv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t row = r0 + 2 * j < rend ? r0 + 2 * j : 0;                          // (row 0 exists)
        gv[t][j] = *reinterpret_cast<const f32x4*>(a.G + row * a.ldg + 4 * col4);
        zv[t][j] = *reinterpret_cast<const f32x4*>(a.Z + row * U + 4 * col4);
        hv[t][j] = *reinterpret_cast<const f32x4*>(a.H + row * a.ldh + 4 * col4);
        rv[t][j] = *reinterpret_cast<const f32x4*>(a.R + row * U + 4 * col4);
        cv[j] = *reinterpret_cast<const f32x4*>(a.HH + row * U + 4 * col4);
      }
      f32x4 gpre[2], gzp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool live = r0 + 2 * j < rend;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = gv[t][j][e], zz = zv[t][j][e], cand = cv[j][e];
          gpre[j][e] = live ? g * (1.0f - zz) * act_grad_out(a.act, cand) : 0.f;          // (gru.hip's expressions)
          gzp[j][e] = live ? (g * (hv[t][j][e] - cand)) * hs_grad(zz) : 0.f;
        }
        if (live) {
          float* krow = a.GXK + (int64_t)(r0 + 2 * j) * (3 * U) + 4 * col4;
          *reinterpret_cast<f32x4*>(krow) = gzp[j];
          *reinterpret_cast<f32x4*>(krow + 2 * U) = gpre[j];
        }
      }
      poll(ctl + 9, 8 * pi);                                   // the previous panel's grp (same sub-slabs) has been read
      put4(lds + t * SLAB, 4 * pw + rsub, col4, gpre[0]);
      put4(lds + t * SLAB, 4 * pw + 2 + rsub, col4, gpre[1]);
      bump(ctl + t, 4);
      poll(ctl + 8, 8 * pi);                                   // the previous panel's gzp likewise
      put4(lds + (2 + t) * SLAB, 4 * pw + rsub, col4, gzp[0]);
      put4(lds + (2 + t) * SLAB, 4 * pw + 2 + rsub, col4, gzp[1]);
      bump(ctl + 2 + t, 4);
    }
    poll(ctl + 6, 8 * (pi + 1));                               // B2 is through: the gpre sub-slabs are free, and
    poll(ctl + 7, 4 * (pi + 1));                               // grh is in the exchange tiles
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
      const int r0 = (u0 + 2 * pi) * 32 + 32 * t + 4 * pw + rsub;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 4 * pw + 2 * j + rsub;                  // row inside the tile
        const bool live = r0 + 2 * j < rend;
        unsigned char* ep = ex + (32 * t + lr) * EROW + 16 * col4;
        const f32x4 grh = *reinterpret_cast<const f32x4*>(ep);
        f32x4 grp, gh0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          grp[e] = live ? (grh[e] * hv[t][j][e]) * hs_grad(rv[t][j][e]) : 0.f;           // (gru.hip's expressions)
          gh0[e] = gv[t][j][e] * zv[t][j][e] + grh[e] * rv[t][j][e];
        }
        if (live) *reinterpret_cast<f32x4*>(a.GXK + (int64_t)(r0 + 2 * j) * (3 * U) + U + 4 * col4) = grp;
        *reinterpret_cast<f32x4*>(ep) = gh0;
        put4(lds + t * SLAB, lr, col4, grp);
      }
      bump(ctl + 4 + t, 4);
    }
  }
}

}  // namespace

extern "C" {

int relgnn_gru_cell_fwd_supported(int32_t act, int32_t units, int32_t in_dim) {
  return units == U && in_dim == U &&
         (act == RELGNN_ACT_LINEAR || act == RELGNN_ACT_TANH || act == RELGNN_ACT_RELU || act == RELGNN_ACT_LEAKY_RELU);
}

int relgnn_gru_cell_fwd_xf32(const float* x, int64_t ldx, const float* h, int64_t ldh, const uint16_t* w_zr_limbs,
                             const uint16_t* w_h_limbs, const float* bias, int32_t act, float* z, float* r, float* rh, float* hh,
                             float* out, int64_t num_nodes, int32_t units, int32_t in_dim, int32_t* status, void* stream) {
  if (num_nodes < 0 || units <= 0 || in_dim <= 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!x || !h || !w_zr_limbs || !w_h_limbs || !bias || !out) return RELGNN_EINVAL;
  const int saved = (z != nullptr) + (r != nullptr) + (rh != nullptr) + (hh != nullptr);
  if (saved != 0 && saved != 4) return RELGNN_EINVAL;
  if (!relgnn_gru_cell_fwd_supported(act, units, in_dim) || num_nodes > ((int64_t)1 << 22) || num_nodes * ldh >= ((int64_t)1 << 30))
    return RELGNN_EUNSUPPORTED;                              // (32-bit byte offsets in the epilogues)
  if (ldx < in_dim || ldh < units) return RELGNN_EINVAL;
  if (!aligned16(x) || !aligned16(h) || !aligned16(w_zr_limbs) || !aligned16(w_h_limbs) || !aligned16(bias) || !aligned16(out) ||
      (saved && (!aligned16(z) || !aligned16(r) || !aligned16(rh) || !aligned16(hh))) || ldx % 4 || ldh % 4)
    return RELGNN_EUNSUPPORTED;
  GruArgs a{};
  a.X = x; a.ldx = ldx; a.H = h; a.ldh = ldh; a.B1 = w_zr_limbs; a.B2 = w_h_limbs; a.bias = bias;
  a.Z = z; a.R = r; a.RH = rh; a.HH = hh; a.OUT = out; a.V = (int32_t)num_nodes; a.act = act; a.status = status;
#ifdef RELGNN_GRU_TIMING
  a.timing = g_gru_timing;
#endif
  const int units32 = (int)((num_nodes + 31) / 32);
  int groups = (units32 + 1) / 2;                             // at least one full panel per workgroup
  if (groups > 256) groups = 256;
  const int longest = (units32 + groups - 1) / groups;
  groups = (units32 + longest - 1) / longest;
  a.groups = groups; a.units_base = units32 / groups; a.units_rem = units32 % groups;
  const unsigned grid = (unsigned)(8 * ((groups + 7) / 8));
  hipStream_t st = as_stream(stream);
  if (act == RELGNN_ACT_TANH) gru_cell_fwd_kernel<true><<<grid, 1024, 0, st>>>(a);
  else gru_cell_fwd_kernel<false><<<grid, 1024, 0, st>>>(a);
  return launch_status();
}

int relgnn_gru_cell_bwd_xf32(const float* gout, int64_t ldg, const float* z, const float* r, const float* h, int64_t ldh,
                             const float* hh, const uint16_t* w_h_nt_limbs, const uint16_t* w_zr_nt_limbs, int32_t act, float* gxk,
                             float* gx, float* gh, int64_t num_nodes, int32_t units, int32_t in_dim, int32_t* status, void* stream) {
  if (num_nodes < 0 || units <= 0 || in_dim <= 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!gout || !z || !r || !h || !hh || !w_h_nt_limbs || !w_zr_nt_limbs || !gxk || !gx || !gh) return RELGNN_EINVAL;
  if (!relgnn_gru_cell_fwd_supported(act, units, in_dim) || num_nodes > ((int64_t)1 << 21) || num_nodes * ldh >= ((int64_t)1 << 30) ||
      num_nodes * ldg >= ((int64_t)1 << 30))
    return RELGNN_EUNSUPPORTED;                              // (32-bit byte offsets in the epilogues; gxk rows are 1536 bytes)
  if (ldg < units || ldh < units) return RELGNN_EINVAL;
  if (!aligned16(gout) || !aligned16(z) || !aligned16(r) || !aligned16(h) || !aligned16(hh) || !aligned16(w_h_nt_limbs) ||
      !aligned16(w_zr_nt_limbs) || !aligned16(gxk) || !aligned16(gx) || !aligned16(gh) || ldg % 4 || ldh % 4)
    return RELGNN_EUNSUPPORTED;
  GruBwdArgs a{};
  a.G = gout; a.ldg = ldg; a.Z = z; a.R = r; a.HH = hh; a.H = h; a.ldh = ldh; a.B2 = w_h_nt_limbs; a.B4 = w_zr_nt_limbs;
  a.GXK = gxk; a.GX = gx; a.GH = gh; a.V = (int32_t)num_nodes; a.act = act; a.status = status;
  const int units32 = (int)((num_nodes + 31) / 32);
  int groups = (units32 + 1) / 2;
  if (groups > 256) groups = 256;
  const int longest = (units32 + groups - 1) / groups;
  groups = (units32 + longest - 1) / longest;
  a.groups = groups; a.units_base = units32 / groups; a.units_rem = units32 % groups;
  gru_cell_bwd_kernel<<<(unsigned)(8 * ((groups + 7) / 8)), 1024, 0, as_stream(stream)>>>(a);
  return launch_status();
}

#ifdef RELGNN_GRU_TIMING
void relgnn_gru_timing_buffer(unsigned long long* p) { g_gru_timing = p; }
#endif

}  // extern "C"
