// relgnn_limb_gemm_xf32 with the work split between wave ROLES instead of between k-loop phases: C = act(bias + A @ B^T) (* dact'(Y)),
// A fp32 [M, K] split into three bf16 limbs on its way into LDS, B the limb image of a weight operand — the tall Dense products of
// gnns/rgcn.py:96-98 in the aggregate-first order (forward [V, L*256] @ [L*256, 256]; input gradient [V, 256] @ [256, L*256]).
//
// limb_gemm_kernel (limb_gemm.hip) runs eight waves that ALL feed the matrix pipe and of which five also split the streamed
// operand at the top of every k-tile, behind a workgroup barrier per k-tile, with both operands staged in a 156 KB LDS ring: 75-86 us
// for [36 096, 768] x [768, 256] against 59.5 us for its own MFMA-only loop (LABNOTES 7.4).  The matrix waves of rgcn_fused.hip —
// X limbs from LDS sub-slabs that other waves fill, W fragments straight from L2 into registers three k-tiles ahead, no barrier,
// hand-over by counters — ran at the MFMA-only time there.  This kernel keeps those matrix waves and replaces the gather waves by
// eight STREAMING producer waves: each owns four rows of every 32-row x 256-k sub-slab (one coalesced 1 KiB load per row), splits
// them (limb_split.h: the same limbs) and writes them in the sub-slab layout; the next sub-slab's rows are in flight meanwhile.
// Same k-tile order and limb-product order per accumulator as limb_gemm_kernel: bit-identical results.
//
// A persistent 16-wave workgroup per CU owns a contiguous range of 32-row units, taken as 64-row panels.  A sub-slab is one 32-row
// tile x 128 k (eight k-tiles) as limbs, 24.75 KB; SIX of them rotate through LDS: the matrix waves hold a tile pair while the
// producers are up to four sub-slabs ahead (with three sub-slabs of 256 k — the geometry rgcn_fused.hip needs for its whole-row
// gathers — the matrix waves waited for the second tile of every pair: measured 74 us per product on average against 94 us for
// limb_gemm_kernel; this geometry: see profiles/).  The panel's sub-slabs come in k order, (hs, tile 0), (hs, tile 1).  Either
// N = 256 (one column chunk; any K % 128 == 0) or K <= 256 (any number of 256-column chunks: the panel's <= 4 sub-slabs stay
// while the matrix waves pass over them once per chunk).  Everything else: limb_gemm_kernel.
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
constexpr int CTL = 16;                 // control words: [1..6] rows filled per buffer, [8..13] matrix waves done with it

struct PcArgs {
  const float* A; int64_t lda;
  const uint16_t* B;                     // limb tiles of the [N, K] right operand
  const float* bias;
  const float* dy; int64_t ldy; int32_t dact;    // epilogue of an input-gradient product: C *= dact'(Y) (limb_gemm.hip)
  float* C; int64_t ldc;
  int32_t M, N, K, act;
  int32_t units_base, units_rem, groups;
  int32_t* status;
};

struct Frag { bf16x8 hi, mid, lo; };

// act'(x) as a function of y = act(x) — the expressions of act_bwd_from_output_kernel (seg_reduce.hip) and limb_gemm.hip
__device__ __forceinline__ float dact_from_output(int act, float yy) {
  switch (act) {
    case RELGNN_ACT_TANH: return 1.f - yy * yy;
    case RELGNN_ACT_RELU: return yy > 0.f ? 1.f : 0.f;
    case RELGNN_ACT_LEAKY_RELU: return yy > 0.f ? 1.f : 0.2f;
    case RELGNN_ACT_ELU: return yy > 0.f ? 1.f : yy + 1.f;
    case RELGNN_ACT_SELU: return yy > 0.f ? 1.0507009873554804934193349852946f : yy + 1.7580993408473768599402175208123f;
    default: return 1.f;
  }
}

// S2 = K / 128 is a template parameter: the k-tiles of a pass are straight-line code.  (With a run-time loop over the half slabs
// hipcc's wait insertion loses track of the W fragments that are in flight across the loop's back edge and drains them at every
// loop header — vmcnt(0) in front of the first MFMA of every four k-tiles: a full L2 round trip per 1.7 us of matrix work.)
// TANH: the epilogue's activation is tanh (the Dense layers between GNN layers, models/sparse_graph_model.py:194-200) instead of
// ReLU / none — a variant of its own: tanhf inlines thirty-two times into the stores.
template <int S2, bool TANH = false>
__global__ __launch_bounds__(1024) void limb_gemm_pc_kernel(const PcArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * SLAB + CTL * 4];
  int* ctl = reinterpret_cast<int*>(lds + NBUF * SLAB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = (int)xcd_logical_block(a.groups);
  if (q < 0) return;
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  if (tid < CTL) ctl[tid] = 0;
  __syncthreads();
  const int chunks = a.N >> 8;
  const int ntiles = a.K >> 4;
  const int nfull = nu >> 1;                                 // panels of two units; an odd unit left over is a panel of one row tile
  const int npan = (nu + 1) >> 1;
  const int nseq = nfull * 2 * S2 + (nu & 1) * S2;
  const int rend = min((u0 + nu) * 32, a.M);
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

  if (wave < 8) {
    // =================================================== matrix waves ===================================================
    const int i32 = lane & 31, h32 = lane >> 5;
    Frag wr[4];
    int wc = 0, wt = 0;                                       // (column chunk, k-tile) whose W fragments are requested next
    auto wload = [&](Frag& f) {
      const uint16_t* p = a.B + ((int64_t)(wc * 8 + wave) * ntiles + wt) * 1536 + 8 * lane;
      f.hi = *reinterpret_cast<const bf16x8*>(p);
      f.mid = *reinterpret_cast<const bf16x8*>(p + 512);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
      if (++wt == ntiles) { wt = 0; if (++wc == chunks) wc = 0; }
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
    auto buf_of = [&](int i) { const int b = b0 + i; return b >= NBUF ? b - NBUF : b; };      // i <= 4 < NBUF
    auto poll_buf = [&](int i) {
      const int b = b0 + i;
      if (b >= NBUF) poll(ctl + 1 + b - NBUF, 32 * (gen0 + 2)); else poll(ctl + 1 + b, 32 * (gen0 + 1));
    };
    auto release = [&](int n) {
      wait_lgkm0();                                            // my reads of these buffers have returned
      handover_fence();
      if (lane == 0)
        for (int i = 0; i < n; ++i) __hip_atomic_fetch_add(ctl + 8 + buf_of(i), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      b0 += n;
      if (b0 >= NBUF) { b0 -= NBUF; ++gen0; }
    };
    for (int pi = 0; pi < npan; ++pi) {
      const int m0 = (u0 + 2 * pi) * 32;
      const int rows_here = min(64, rend - m0);
      const bool two = pi < nfull;
      const int per = two ? 2 : 1;
      f32x16 acc0, acc1;
      // 32 x 32 tile: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register 4 c + {0..3})
      auto store_tile = [&](const f32x16& acc, int r, int colw) {
        if (r >= rows_here) return;
        float* crow = a.C + (int64_t)(m0 + r) * a.ldc;
        const float* yrow = a.dy ? a.dy + (int64_t)(m0 + r) * a.ldy : nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cc = colw + 8 * c + 4 * h32;
          f32x4 v = f32x4{acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
          if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + cc);
          if constexpr (TANH) {
            v[0] = act_fwd<RELGNN_ACT_TANH>(v[0]); v[1] = act_fwd<RELGNN_ACT_TANH>(v[1]);
            v[2] = act_fwd<RELGNN_ACT_TANH>(v[2]); v[3] = act_fwd<RELGNN_ACT_TANH>(v[3]);
          } else if (a.act == RELGNN_ACT_RELU) {
            v[0] = act_fwd<RELGNN_ACT_RELU>(v[0]); v[1] = act_fwd<RELGNN_ACT_RELU>(v[1]);
            v[2] = act_fwd<RELGNN_ACT_RELU>(v[2]); v[3] = act_fwd<RELGNN_ACT_RELU>(v[3]);
          }
          if (yrow) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(yrow + cc);
            v = f32x4{v[0] * dact_from_output(a.dact, y[0]), v[1] * dact_from_output(a.dact, y[1]),
                      v[2] * dact_from_output(a.dact, y[2]), v[3] * dact_from_output(a.dact, y[3])};
          }
          *reinterpret_cast<f32x4*>(crow + cc) = v;
        }
      };
      // One call site for the k-tiles (two copies of it, and of the stores, spilled registers).  A panel is `chunks` passes over
      // its S2 half slabs.  One chunk: a tile pair is acquired and released per half slab.  Several chunks (K <= 256): all the
      // panel's sub-slabs are acquired before the first pass and released behind the last.
      if (chunks > 1)
        for (int i = 0; i < S2 * per; ++i) poll_buf(i);
      for (int ps = 0; ps < chunks; ++ps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int hs = 0; hs < S2; ++hs) {
          const int i0 = chunks > 1 ? hs * per : 0;
          if (chunks == 1) { poll_buf(0); if (two) poll_buf(1); }
          const unsigned char* x0b = lds + buf_of(i0) * SLAB + xlane;
          const unsigned char* x1b = lds + buf_of(i0 + per - 1) * SLAB + xlane;
          // the X fragments of k-tile t + 1 are requested right behind the MFMAs that read those of k-tile t (the matrix pipe has read
          // its operands by then): an LDS round trip hides under six MFMAs instead of standing in front of them
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
          if (chunks == 1) release(per);
        }
        const int colw = ps * 256 + wave * 32;
        store_tile(acc0, i32, colw);
        if (two) store_tile(acc1, 32 + i32, colw);
      }
      if (chunks > 1) release(S2 * per);
    }
    return;
  }

  // ===================================================== producer waves =====================================================
  // wave p streams rows 4 p .. 4 p + 3 of every sub-slab: two 1 KiB loads (two rows x 128 k each: lane = row (lane >> 5), 4 k), the
  // split, three 8-byte LDS writes per load.  The loads of the three sub-slabs behind the current one are in flight; the four
  // register sets alternate by name and the loads are unconditional (a row past the end reads a valid address and is zeroed), so
  // that the wait in front of the split is exactly "all but the six loads issued last".
  const int pw = wave - 8;
  const int col4 = lane & 31, rsub = lane >> 5;
  const int wr_lane = (col4 >> 1) * PIECE + (col4 & 1) * 8;   // k-tile col4 >> 2, k half (col4 >> 1) & 1, k 4 (col4 & 1) .. + 3
  const f32x4* A4 = reinterpret_cast<const f32x4*>(a.A);
  const int64_t lda4 = a.lda >> 2;
  struct Pos { int g, pi, hs, tm; };                          // sequence position -> (panel, half slab, row tile)
  auto advance = [&](Pos& p) {
    ++p.g;
    if (p.pi < nfull && p.tm == 0) { p.tm = 1; return; }
    p.tm = 0;
    if (++p.hs == S2) { p.hs = 0; ++p.pi; }
  };
  auto row0 = [&](const Pos& p) { return (u0 + 2 * p.pi) * 32 + p.tm * 32 + 4 * pw + rsub; };
  auto issue = [&](const Pos& p, f32x4 (&v)[2]) {
    const int r0 = row0(p);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = p.g < nseq && r0 + 2 * j < rend;
      v[j] = A4[(ok ? (int64_t)(r0 + 2 * j) * lda4 + p.hs * 32 : 0) + col4];          // (row 0 exists: M > 0, K >= 128)
    }
  };
  auto process = [&](const Pos& p, f32x4 (&v)[2]) {
    __builtin_amdgcn_s_waitcnt(0x0F76);                        // vmcnt(6): everything but the six loads issued last has landed
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
}

}  // namespace

extern "C" {

// the shapes the kernel is instantiated for: K / 128 in {1, 2, 3, 4, 6, 8} (tanh epilogue: {1, 2, 4}), whole 256-column chunks,
// several chunks only while a pass's W fragments stay short (K <= 256)
int relgnn_limb_gemm_xf32_pc_supported(int32_t act, int32_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  if (K % 128 != 0 || K > 1024 || K == 640 || K == 896 || N % 256 != 0 || (N != 256 && K > 256)) return 0;
  if (act == RELGNN_ACT_TANH) return K == 128 || K == 256 || K == 512;
  return act == RELGNN_ACT_LINEAR || act == RELGNN_ACT_RELU;
}

int relgnn_limb_gemm_xf32_pc(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, int32_t dact,
                             const float* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t* status,
                             void* stream) {
  if (M < 0 || N < 0 || K < 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C) return RELGNN_EINVAL;
  if (!relgnn_limb_gemm_xf32_pc_supported(act, M, N, K)) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias && !aligned16(bias)) || ldc % 4 || ldc < N || lda % 4 || lda < K)
    return RELGNN_EUNSUPPORTED;
  if (Y) {
    if (dact < RELGNN_ACT_LINEAR || dact > RELGNN_ACT_SELU) return dact == RELGNN_ACT_GELU ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
    if (!aligned16(Y) || ldy % 4 || ldy < N) return RELGNN_EUNSUPPORTED;
  }
  PcArgs a{};
  a.A = A; a.lda = lda; a.B = B; a.bias = bias; a.dy = Y; a.ldy = ldy; a.dact = dact; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.act = act;
  a.status = status;
  // the fewest workgroups that keep the longest range: 1128 units over 256 CUs are ranges of 4 and 5 units — 226 ranges of 5 finish
  // at the same time and leave 30 CUs to whatever runs next to this kernel (the weight gradient on the side stream)
  const int units = (M + 31) / 32;
  int groups = (units + 1) / 2;                               // at least one full panel per workgroup
  if (groups > 256) groups = 256;
  const int longest = (units + groups - 1) / groups;
  groups = (units + longest - 1) / longest;
  a.groups = groups; a.units_base = units / groups; a.units_rem = units % groups;
  const unsigned grid = (unsigned)(8 * ((groups + 7) / 8));
  hipStream_t st = as_stream(stream);
  if (act == RELGNN_ACT_TANH) {
    switch (K >> 7) {
      case 1: limb_gemm_pc_kernel<1, true><<<grid, 1024, 0, st>>>(a); break;
      case 2: limb_gemm_pc_kernel<2, true><<<grid, 1024, 0, st>>>(a); break;
      case 4: limb_gemm_pc_kernel<4, true><<<grid, 1024, 0, st>>>(a); break;
      default: return RELGNN_EUNSUPPORTED;
    }
    return launch_status();
  }
  switch (K >> 7) {
    case 1: limb_gemm_pc_kernel<1><<<grid, 1024, 0, st>>>(a); break;
    case 2: limb_gemm_pc_kernel<2><<<grid, 1024, 0, st>>>(a); break;
    case 3: limb_gemm_pc_kernel<3><<<grid, 1024, 0, st>>>(a); break;
    case 4: limb_gemm_pc_kernel<4><<<grid, 1024, 0, st>>>(a); break;
    case 6: limb_gemm_pc_kernel<6><<<grid, 1024, 0, st>>>(a); break;
    case 8: limb_gemm_pc_kernel<8><<<grid, 1024, 0, st>>>(a); break;
    default: return RELGNN_EUNSUPPORTED;
  }
  return launch_status();
}

}  // extern "C"
