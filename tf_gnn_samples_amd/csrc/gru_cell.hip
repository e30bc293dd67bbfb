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
// Structure: limb_gemm_pc.hip's wave roles.  A persistent 16-wave workgroup per CU owns a contiguous range of 32-row units, taken
// as 64-row panels.  Eight MATRIX waves (W fragments straight from L2 into registers three k-tiles ahead, X fragments from LDS
// sub-slabs of 32 rows x 128 k as three bf16 limbs) issue MFMAs, LDS accesses and W loads and nothing else: waves 0-3 the z columns
// (P1) and then the candidate columns (P2) of both row tiles, waves 4-7 the r columns.  Eight PRODUCER waves do everything
// elementwise: they split the panel's rows of x and h into the sub-slabs, take the z, r and candidate pre-activations from the
// matrix waves through a pair of fp32 exchange tiles in LDS, write r * h as limbs over the h sub-slabs for P2's second half, blend,
// and issue every global store.  Hand-over by monotonic counters in LDS, bounded polls (handover.h).  P2 keeps only the z waves
// busy (a third more matrix time per panel than a balanced split): at three panels per CU the kernel is bound by its pipeline's
// fill, not by the matrix pipe.
// (The first form — commit fe70cf4, `profiles/r06_gru_cell_cycles.txt` — had the gates and the blend in the matrix waves' epilogues:
// their global loads of h and of the parked z cost a round trip per row tile, 85 us against 78 at 50 k nodes, 345 against 285 at 200 k.)
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
};

struct Frag { bf16x8 hi, mid, lo; };

// store at `base + byte offset`: a uniform base and a 32-bit per-lane byte offset (the global_store form that needs no per-lane
// 64-bit address)
__device__ __forceinline__ void st4(float* base, uint32_t boff, f32x4 v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + boff) = v;
}
// the value as the compiler must take it here (an offset it cannot compute ahead of the k-loops: the addresses of every epilogue
// access, hoisted to the top of a panel, were seventy spilled registers reloaded one round trip at a time)
__device__ __forceinline__ uint32_t here(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }   // (gru.hip's expression)

// ---------------------------------------------------------------------------------------------------------------------------------
// The forward.  Four sub-slabs with fixed jobs — 0, 1 the x tiles, 2, 3 the h
// tiles and then r * h — and ONE pair of fp32 exchange tiles (64 rows x 128 columns, rows padded to 528 B) that carries, one after
// the other, the z pre-activations (z waves -> producers, which keep z = hs(.) in registers), the r pre-activations (r waves ->
// producers: r, r * h as limbs over the h sub-slabs), the candidate pre-activations (z waves -> producers: act, blend, stores).
// The producers hold h of their rows from the load that fed the h sub-slabs; every global store is theirs, whole 512-byte row
// pieces per half wave instead of the matrix waves' 32-byte pieces of 32 rows.
template <bool TANH>
__global__ __launch_bounds__(1024) void gru_cell_fwd_kernel(const GruArgs a) {
  constexpr int NB = 4, EROWB = 528;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NB * SLAB + 64 * EROWB + 3 * U * 4 + 32 * 4];
  unsigned char* const ex = lds + NB * SLAB;
  float* const lbias_all = reinterpret_cast<float*>(lds + NB * SLAB + 64 * EROWB);
  // counters (monotonic): [0, 1] rows of x filled, [2, 3] rows of h, [4, 5] rows of r * h, [6] matrix waves through P1's k-loop,
  // [7] z waves through P2's k-loop, [8] z waves whose z pre-activations are in the exchange tiles, [9] producer waves that have
  // read them, [10] r waves whose pre-activations are there, [11] z waves whose candidate pre-activations are there, [12] producer
  // waves that have consumed those
  int* ctl = reinterpret_cast<int*>(lds + NB * SLAB + 64 * EROWB + 3 * U * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  if (tid < 32) ctl[tid] = 0;
  if (tid >= 64 && tid < 64 + 3 * U / 4) reinterpret_cast<f32x4*>(lbias_all)[tid - 64] = reinterpret_cast<const f32x4*>(a.bias)[tid - 64];
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

  if (wave < 8) {
    // =================================================== matrix waves ===================================================
    const bool zrole = wave < 4;
    const int i32 = lane & 31, h32 = lane >> 5;
    const int cw = (wave & 3) * 32;
    Frag wr[4];
    const char* const w1 = reinterpret_cast<const char*>(a.B1) + (int64_t)wave * 16 * 3072;
    const char* const w2 = reinterpret_cast<const char*>(a.B2) + (int64_t)(wave & 3) * 16 * 3072;
    const uint32_t wlane = 16 * lane;
    const int wlen = zrole ? 32 : 16;
    int wt = 0;
    auto wload = [&](Frag& f) {
      const char* p = wt < 16 ? w1 + wt * 3072 : w2 + (wt - 16) * 3072;
      f.hi = *reinterpret_cast<const bf16x8*>(p + wlane);
      f.mid = *reinterpret_cast<const bf16x8*>(p + wlane + 1024);
      f.lo = *reinterpret_cast<const bf16x8*>(p + wlane + 2048);
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
    const int xlane = h32 * PIECE + i32 * 16;
    unsigned char* const epatch = ex + i32 * EROWB + (cw + 4 * h32) * 4;       // my patch of exchange tile 0 (tile 1: + 32 rows)
    const float* const lbias = lbias_all + cw + 4 * h32;
    const int passes = zrole ? 2 : 1;
    for (int pi = 0; pi < npan; ++pi) {
      const bool two = pi < nfull;
#pragma clang loop unroll(disable)
      for (int ps = 0; ps < passes; ++ps) {
        f32x16 acc0, acc1;
        {
          const int first = ps ? 2 * U : (zrole ? 0 : U);     // the accumulators start from the bias of their columns
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + first + 8 * c);
            acc0[4 * c] = bv[0]; acc0[4 * c + 1] = bv[1]; acc0[4 * c + 2] = bv[2]; acc0[4 * c + 3] = bv[3];
          }
          acc1 = acc0;
        }
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
          // k-tiles 0-7 from the x sub-slabs, 8-15 from the h sub-slabs (P1) / r * h in their place (P2)
          int* const filled = ctl + (hs == 0 ? 0 : ps == 0 ? 2 : 4);
          if (ps == 0 || hs == 1) {
            poll(filled, 32 * (pi + 1));
            if (two) poll(filled + 1, 32 * (pi + 1));
          }
          const unsigned char* x0b = lds + (2 * hs) * SLAB + xlane;
          const unsigned char* x1b = lds + (2 * hs + (two ? 1 : 0)) * SLAB + xlane;
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
        }
        bump(ctl + (ps == 0 ? 6 : 7), 1);                      // my reads of this pass's sub-slabs have returned
        // the pre-activations to the producers: the exchange tiles carry z's, then r's, then the candidate's
        if (ps == 0 && zrole) poll(ctl + 12, 8 * pi);          // the previous panel's candidates have been consumed
        if (ps == 0 && !zrole) poll(ctl + 9, 8 * (pi + 1));    // the producers have z in their registers
        if (ps == 1) poll(ctl + 4 + (two ? 1 : 0), 32 * (pi + 1));   // (r's have been read: r * h is written — polled for the k-loop already)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<f32x4*>(epatch + 32 * c) = f32x4{acc0[4 * c], acc0[4 * c + 1], acc0[4 * c + 2], acc0[4 * c + 3]};
          if (two)
            *reinterpret_cast<f32x4*>(epatch + 32 * EROWB + 32 * c) = f32x4{acc1[4 * c], acc1[4 * c + 1], acc1[4 * c + 2], acc1[4 * c + 3]};
        }
        bump(ctl + (ps == 1 ? 11 : zrole ? 8 : 10), 1);
      }
    }
    return;
  }

  // ===================================================== producer waves =====================================================
  // wave p owns rows 4 p .. 4 p + 3 of both row tiles: lane = row (lane >> 5) of a pair, 4 columns.
  const int pw = wave - 8;
  const int col4 = lane & 31, rsub = lane >> 5;
  auto put4 = [&](unsigned char* slab, int row, f32x4 x) {
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
  f32x4 xv[2][2], hv[2][2];
  auto request = [&](int pi) {                                 // x and h of my rows of panel pi (past the end: row 0, zeroed later)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (u0 + 2 * pi) * 32 + 32 * t + 4 * pw + 2 * j + rsub;
        const int64_t row = (pi < npan && r < rend) ? r : 0;
        xv[t][j] = *reinterpret_cast<const f32x4*>(a.X + row * a.ldx + 4 * col4);
        hv[t][j] = *reinterpret_cast<const f32x4*>(a.H + row * a.ldh + 4 * col4);
      }
  };
  request(0);
  for (int pi = 0; pi < npan; ++pi) {
    const int tiles = pi < nfull ? 2 : 1;
    const int rbase = (u0 + 2 * pi) * 32 + 4 * pw + rsub;
    // ---- the panel's rows as limbs (the sub-slabs are free once the z waves are through the previous panel's P2)
    poll(ctl + 6, 8 * pi);
    poll(ctl + 7, 4 * pi);
    f32x4 hk[2][2];                                            // h of my rows, kept for r * h and the blend
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool live = rbase + 32 * t + 2 * j < rend;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        hk[t][j] = live ? hv[t][j] : zero;
        put4(lds + t * SLAB, 4 * pw + 2 * j + rsub, live ? xv[t][j] : zero);
        put4(lds + (2 + t) * SLAB, 4 * pw + 2 * j + rsub, hk[t][j]);
      }
      bump(ctl + t, 4);
      bump(ctl + 2 + t, 4);
    }
    request(pi + 1);                                           // the next panel's rows are on their way meanwhile
    // ---- z = hs(.) of my rows, into registers (and to the backward's tensor)
    f32x4 zk[2][2];
    poll(ctl + 8, 4 * (pi + 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 4 * pw + 2 * j + rsub;
        const f32x4 p = *reinterpret_cast<const f32x4*>(ex + (32 * t + lr) * EROWB + 16 * col4);
        zk[t][j] = f32x4{hard_sigmoid(p[0]), hard_sigmoid(p[1]), hard_sigmoid(p[2]), hard_sigmoid(p[3])};
        const int r = rbase + 32 * t + 2 * j;
        if (a.Z && r < rend) *reinterpret_cast<f32x4*>(a.Z + (int64_t)r * U + 4 * col4) = zk[t][j];
      }
    }
    bump(ctl + 9, 1);
    // ---- r = hs(.), r * h: to the backward's tensors and, as limbs, over the h sub-slabs (all eight matrix waves are through P1)
    poll(ctl + 10, 4 * (pi + 1));
    poll(ctl + 6, 8 * (pi + 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 4 * pw + 2 * j + rsub;
        const f32x4 p = *reinterpret_cast<const f32x4*>(ex + (32 * t + lr) * EROWB + 16 * col4);
        const f32x4 rr = f32x4{hard_sigmoid(p[0]), hard_sigmoid(p[1]), hard_sigmoid(p[2]), hard_sigmoid(p[3])};
        const f32x4 rh = f32x4{rr[0] * hk[t][j][0], rr[1] * hk[t][j][1], rr[2] * hk[t][j][2], rr[3] * hk[t][j][3]};
        const int r = rbase + 32 * t + 2 * j;
        if (a.R && r < rend) {
          *reinterpret_cast<f32x4*>(a.R + (int64_t)r * U + 4 * col4) = rr;
          *reinterpret_cast<f32x4*>(a.RH + (int64_t)r * U + 4 * col4) = rh;
        }
        put4(lds + (2 + t) * SLAB, lr, rh);
      }
      bump(ctl + 4 + t, 4);
    }
    // ---- the candidate and the blend
    poll(ctl + 11, 4 * (pi + 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 4 * pw + 2 * j + rsub;
        f32x4 v = *reinterpret_cast<const f32x4*>(ex + (32 * t + lr) * EROWB + 16 * col4);
        if constexpr (TANH) {
          v = f32x4{tanh_fast(v[0]), tanh_fast(v[1]), tanh_fast(v[2]), tanh_fast(v[3])};     // (v_exp_f32 / v_rcp_f32, 2e-7 absolute: common.h)
        } else if (a.act == RELGNN_ACT_RELU) {
          v = f32x4{act_fwd<RELGNN_ACT_RELU>(v[0]), act_fwd<RELGNN_ACT_RELU>(v[1]), act_fwd<RELGNN_ACT_RELU>(v[2]),
                    act_fwd<RELGNN_ACT_RELU>(v[3])};
        } else if (a.act == RELGNN_ACT_LEAKY_RELU) {
          v = f32x4{act_fwd<RELGNN_ACT_LEAKY_RELU>(v[0]), act_fwd<RELGNN_ACT_LEAKY_RELU>(v[1]), act_fwd<RELGNN_ACT_LEAKY_RELU>(v[2]),
                    act_fwd<RELGNN_ACT_LEAKY_RELU>(v[3])};
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = zk[t][j][e] * hk[t][j][e] + (1.0f - zk[t][j][e]) * v[e];       // (gru.hip's expression)
        const int r = rbase + 32 * t + 2 * j;
        if (r < rend) {
          if (a.HH) *reinterpret_cast<f32x4*>(a.HH + (int64_t)r * U + 4 * col4) = v;
          *reinterpret_cast<f32x4*>(a.OUT + (int64_t)r * U + 4 * col4) = o;
        }
      }
    }
    bump(ctl + 12, 1);
  }
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
  f32x4 gv[2][2], zv[2][2], hv[2][2], rv[2][2], cv[2][2];
  auto request = [&](int pi) {                                 // g, z, h, r, hh of my rows of panel pi (past the end: row 0, masked later)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (u0 + 2 * pi) * 32 + 32 * t + 4 * pw + 2 * j + rsub;
        const int64_t row = (pi < npan && r < rend) ? r : 0;
        gv[t][j] = *reinterpret_cast<const f32x4*>(a.G + row * a.ldg + 4 * col4);
        zv[t][j] = *reinterpret_cast<const f32x4*>(a.Z + row * U + 4 * col4);
        hv[t][j] = *reinterpret_cast<const f32x4*>(a.H + row * a.ldh + 4 * col4);
        rv[t][j] = *reinterpret_cast<const f32x4*>(a.R + row * U + 4 * col4);
        cv[t][j] = *reinterpret_cast<const f32x4*>(a.HH + row * U + 4 * col4);
      }
  };
  request(0);
  for (int pi = 0; pi < npan; ++pi) {
    const int tiles = pi < nfull ? 2 : 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= tiles) break;
      const int r0 = (u0 + 2 * pi) * 32 + 32 * t + 4 * pw + rsub;
      f32x4 gpre[2], gzp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool live = r0 + 2 * j < rend;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = gv[t][j][e], zz = zv[t][j][e], cand = cv[t][j][e];
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
#pragma unroll
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
    request(pi + 1);                                           // the next panel's rows come in under this panel's B4 (requested ahead
  }                                                            //  of the elementwise step above they are 80 registers too many: 71 spills)
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

}  // extern "C"
