// fp32 products on the bf16 matrix pipe: every fp32 operand is carried as THREE bf16 limbs, x = x_hi + x_mid + x_lo EXACTLY
// (3 x 8 significant bits = the 24 of an fp32 value; bf16 has fp32's exponent range), and a product is the six limb products
// whose weight is >= 2^-16 of the leading one, each exact in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//
//     x * w  ~=  x_hi w_hi + (x_hi w_mid + x_mid w_hi) + (x_hi w_lo + x_mid w_mid + x_lo w_hi)
//
// The three dropped products (mid*lo, lo*mid, lo*lo) are below 2^-23 of |x w| — less than the rounding of ONE fp32 accumulation
// step — so the result is in the same error class as the exact-fp32 pipe (v_mfma_f32_32x32x2_f32): measured against float64 on
// the C2 layer shapes in tests/test_gpu_limb_gemm.py and, end to end, in tests/test_gpu_parity_margin.py.  Why bother: gfx950
// has no xf32 / TF32 form and its fp32-input MFMA runs at the vector rate, 1/16 of the bf16 rate (MI355X_MICROARCH.md); six
// bf16 products per fp32 product leave a factor 16 / 6 = 2.7 on the matrix pipe.
//
// What it replaces: the node-side Dense products of gnns/rgcn.py:96-98 in the aggregate-first order, out = A [V, L*D] @ W, and
// their input gradient dA = dOut @ W^T — C[m][n] = act(bias[n] + sum_k A[m][k] * B[n][k]), both operands k-contiguous.
//
// Limb format ("limb tiles"): an [R, C] matrix (C % 16 == 0) is stored as ceil(R / 32) x C / 16 tiles of 32 rows x 16 columns;
// a tile is three 1 KiB blocks (hi, mid, lo), a block holds element (i, k) at bf16 index (k / 8) * 256 + i * 8 + k % 8 — which is
// the LDS image AND the MFMA operand layout, so one DMA instruction moves one block as 1 KiB of consecutive bytes (eight full
// cache lines; with plain row-major planes a block is 32 row segments of 32 bytes, every cache line is fetched into L1 four times
// and the kernel runs at the vector-memory rate: 113 us instead of 60 for [36 k, 768] x [768, 256], profiles/r03_limb_gemm.jsonl).
// Rows past R in the last tile row are zeros.  relgnn_limb_split_f32 writes the format from an fp32 matrix (optionally
// transposed: the forward product wants W^T); producers that hold the values in registers can write it directly.
//
// Geometry and staging follow panel_gemm.hip: a workgroup (8 waves = 1 x 8) owns a panel of 32*T32 <= 160 rows x 256 columns
// x full K; k-tile = 16; operands go global -> LDS by global_load_lds_dwordx4 into a 4-stage ring (counted vmcnt, raw s_barrier),
// one loader wave per SIMD.  An LDS block is one limb block = one DMA instruction = one ds_read_b128 per lane: lane
// (i = lane & 31, h = lane >> 5) owns row i, k = 8 h .. 8 h + 7, stored at 16 B * lane — conflict-free for the 16-lane groups of gfx950's ds_read_b128 (rows 0-3, 12-15, 20-27 of one k half hit 16
// different bank quads).  The W limbs are the MFMA's A operand and the X limbs its B operand, so a lane ends up with four
// consecutive output columns of one output row (one 16-byte store), as in panel_gemm.hip.
#include "common.h"
#include "lds_dma.h"
#include "limb_split.h"

#include <stdlib.h>
#include <type_traits>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK = 16;
constexpr int STAGES = 4;
#ifdef RELGNN_LIMB_TIMING
__device__ unsigned long long* g_limb_timing_dev = nullptr;
unsigned long long* g_limb_timing = nullptr;   // diagnostic build: per-wave cycle totals of the k-loop's segments
#define TSTAMP(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define TSTAMP(v)
#endif

struct LimbArgs {
  const uint16_t* A;                   // limb tiles of the [M, K] left operand (XF32: unused)
  const float* Ax; int64_t lda;        // XF32: the left operand as fp32 [M, K], row-major
  const uint16_t* B;                   // limb tiles of the [N, K] right operand
  const float* bias; const uint16_t* zeros;
  float* C; int64_t ldc;
  int32_t M, N, K, act;
  int32_t units_base, units_rem;       // panel q covers 32-row units [q*base + min(q, rem), +base + (q < rem))
  int32_t panels, chunks;              // row panels x 256-column chunks = logical workgroups
  // two-fp16-limb arithmetic (NL = 2): per-row magnitudes of the left operand (xmax[m * xgroups + g], the row's scale comes from
  // their maximum) and the magnitude the right operand's limbs were scaled by (wmax[0]); both in device memory
  const float* xmax; int32_t xgroups; const float* wmax;
  // epilogue of an input-gradient product: C *= act'(Y) with the derivative taken from the OUTPUT Y [M, N] of activation `dact`
  // (what relgnn_act_bwd_from_output computes in a pass of its own); dy = nullptr: none
  const float* dy; int64_t ldy; int32_t dact;
#ifdef RELGNN_LIMB_TIMING
  unsigned long long* timing;          // [workgroup][wave][8] cycle totals per loop segment (diagnostic build only)
#endif
};

// act'(x) as a function of y = act(x) — the same expressions, in the same order, as act_bwd_from_output_kernel (seg_reduce.hip)
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

// ---- fp32 -> two fp16 limbs behind an exact power-of-two scale ----------------------------------------------------------
// x * s = hi + lo + r with hi = fp16(x s), lo = fp16(x s - hi), |r| <= 2^-22 |x s|; s = 2^j puts the largest magnitude of the
// row (of the matrix, for weights) into [2^14, 2^15): fp16's mantissa is enough for two limbs, its exponent range is what the
// scale is for.  Three products per fp32 product (hi hi, hi lo, lo hi: each exact in fp32) instead of the six of the bf16 triple.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t limb16_scale_bits(float mx) {          // exponent field of s (s = 1 for 0 / denormal / inf / nan)
  const uint32_t e = (__float_as_uint(mx) >> 23) & 0xFFu;
  if (e == 0u || e == 255u) return 127u;
  const uint32_t f = 268u - e;                                              // 2^(141 - e): mx * s in [2^14, 2^15)
  return f > 253u ? 253u : f;
}
__device__ __forceinline__ float limb16_scale(float mx) { return __uint_as_float(limb16_scale_bits(mx) << 23); }
__device__ __forceinline__ float limb16_unscale(float mx) { return __uint_as_float((254u - limb16_scale_bits(mx)) << 23); }
// log2 of 1 / scale: scales are removed with ldexpf(x, k_left + k_right) — ONE exact scaling by the sum of the exponents.  Multiplying
// by the two reciprocal scales one after the other overflows on the way when a huge left operand (scale 2^-113 for float32 lowest:
// acc * 2^113 = inf) meets a tiny right one (2^-29 would have brought it back): found by tests/test_gpu_extreme_values.py.
__device__ __forceinline__ int limb16_unscale_exp(float mx) { return 127 - (int)limb16_scale_bits(mx); }
// (vector-typed on purpose: hipcc then issues the scale and the subtraction as v_pk_mul_f32 / v_pk_add_f32 — two values per VALU
//  instruction, 24 instead of 32 per eight values; the split's issue slots are what gates these kernels, LABNOTES 7.4)
__device__ __forceinline__ void split_pair16(f32x2 xs, uint32_t& h, uint32_t& l) {
  const f16x2 hh = __builtin_convertvector(xs, f16x2);                     // round to nearest even
  const f32x2 hf = __builtin_convertvector(hh, f32x2);
  const f16x2 ll = __builtin_convertvector(xs - hf, f16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, ll);
}
__device__ __forceinline__ void split8_16(const float* v, float s, uint4& h, uint4& l) {
  split_pair16(f32x2{v[0], v[1]} * s, h.x, l.x);
  split_pair16(f32x2{v[2], v[3]} * s, h.y, l.y);
  split_pair16(f32x2{v[4], v[5]} * s, h.z, l.z);
  split_pair16(f32x2{v[6], v[7]} * s, h.w, l.w);
}

// ---- fp32 -> three bf16 limbs: limb_split.h (split_pair, split_pair_sat, split8) ----------------------------------------

// XF32 = false: both operands arrive as limb tiles (DMA for everything).
// XF32 = true : the LEFT operand is plain fp32 [M, K] — what the producers of the path write (seg_reduce buckets, gradients) and
//               what the weight-gradient product reads — and is split on its way into LDS: waves 0 .. T32-1 load it (thread = one
//               row x 16 k = 64 B, two neighbouring threads one 128-byte line of a 32-k super-tile), split it in registers
//               (v_cvt_pk_bf16_f32 + exact subtractions, ~6 VALU per element, every element once per workgroup) and write the
//               limb blocks with ds_write_b128; the other waves issue the DMA of the W limb tiles.  No limb copy of the left
//               operand ever exists in HBM (it would be 6 B per element written and read again; the split pass alone costs
//               48 us for [36 k, 768], profiles/r03_limb_gemm.jsonl).
// ABL (experiments only, results wrong when != 0): bit 0 no DMA after the prologue, bit 1 no fragment reads in the loop,
// bit 2 no waits / barrier in the loop.
// NL = 3: three bf16 limbs per operand, six products; NL = 2 (XF32 only): two fp16 limbs behind power-of-two scales, three products.
template <int T32, bool XF32, int ABL = 0, int NL = 3>
__global__ __launch_bounds__(512) void limb_gemm_kernel(const LimbArgs a) {
  static_assert(NL == 3 || (NL == 2 && XF32), "two limbs: the in-flight form only");
  constexpr int NC = 256;
  constexpr int PA = NL * T32, PB = NL * (NC / 32), P = PA + PB;   // 1 KiB blocks per k-tile
  constexpr int STAGE_BYTES = P * 1024;
  // DMA: XF32: the W blocks only, by the last 4 (T32 <= 4) or 3 waves; else everything, by waves 0-3 (one per SIMD)
  constexpr int LOADERS = XF32 ? (T32 <= 4 ? 4 : 3) : 4;
  constexpr int FIRST_LOADER = XF32 ? 8 - LOADERS : 0;
  constexpr int PD = XF32 ? PB : P;                                // blocks that arrive by DMA
  constexpr int G = (PD + LOADERS - 1) / LOADERS;
  static_assert(STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // chunk index fastest and one contiguous range of logical workgroups per XCD: the column chunks of a row panel (dA: three)
  // run next to each other under one L2 and fetch the panel's left operand from HBM once
  const int64_t lb = xcd_logical_block((int64_t)a.panels * a.chunks);
  if (lb < 0) return;
  const int q = (int)(lb / a.chunks), chunk = (int)(lb % a.chunks);
  const int u0 = q * a.units_base + min(q, a.units_rem);
  const int nu = a.units_base + (q < a.units_rem ? 1 : 0);
  const int m0 = u0 * 32;
  const int rows_here = min(nu * 32, a.M - m0);
  const int n0 = chunk * NC;
  const int ntiles = a.K / BK;

  // ---- DMA sources ---------------------------------------------------------------------------------------------------
  const bool loader = wave >= FIRST_LOADER && wave < FIRST_LOADER + LOADERS;
  const uint16_t* src[G];
  int step[G];
  int piece[G];
  constexpr int TILE = NL * 512;                      // 16-bit elements of one limb tile
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // wave-uniform; past the end: a duplicate of the last block
    const int lw = max(wave - FIRST_LOADER, 0);
    const int c = (XF32 ? PA : 0) + min(lw + LOADERS * g, PD - 1);
    piece[g] = c;
    if (c < PA) {
      const int tm = c / NL, pl = c % NL;
      const bool ok = tm < nu;                        // a tile row of this panel (rows past M inside it are stored zeros)
      src[g] = ok ? a.A + ((int64_t)(u0 + tm) * ntiles) * TILE + pl * 512 + 8 * lane : a.zeros + 8 * lane;
      step[g] = ok ? TILE : 0;
    } else {
      const int cb = c - PA;
      const int wn_ = cb / NL, pl = cb % NL;
      src[g] = a.B + ((int64_t)(n0 / 32 + wn_) * ntiles) * TILE + pl * 512 + 8 * lane;
      step[g] = TILE;
    }
  }
  // (the next k-tile of my blocks [G0, G1): a k-tile's DMA may be issued in parts, in block order)
  auto issue_part = [&](int stage, auto g0_c, auto g1_c) {
    constexpr int G0 = decltype(g0_c)::value, G1 = decltype(g1_c)::value;
    if (!loader) return;
    unsigned char* dst = lds + stage * STAGE_BYTES;
#pragma unroll
    for (int g = G0; g < G1; ++g) {
      dma16(src[g], dst + piece[g] * 1024);
      src[g] += step[g];
    }
  };
  auto issue = [&](int stage) { issue_part(stage, std::integral_constant<int, 0>{}, std::integral_constant<int, G>{}); };
  auto wait_dma = [&](int tiles) {                   // leave `tiles` k-tiles of my own DMA in flight
    if (!loader) return;
    if (tiles >= 3) wait_vm<3 * G>();
    else if (tiles == 2) wait_vm<2 * G>();
    else if (tiles == 1) wait_vm<G>();
    else wait_vm<0>();
  };

  // ---- XF32: the left operand's way into LDS -------------------------------------------------------------------------
  // Thread (r = tid >> 1, hf = tid & 1) of waves 0 .. T32-1 owns row r of the panel, k-tile 2 S + hf of every 32-k super-tile S:
  // 16 fp32 = 64 B per super-tile (4 x dwordx4), two limb chunks (k 0-7, k 8-15) per plane.  Super-tile S is loaded during
  // k-tile 2S-4, its first chunks are split and stored at the top of k-tile 2S-2, the second ones at the top of k-tile 2S-1:
  // after the barrier inside k-tile 2S-3 released the two stages, before the barrier inside k-tile 2S-1 publishes k-tile 2S.
  const bool xwave = XF32 && wave < T32;
  const int xr = tid >> 1, xhf = tid & 1;
  const bool xrow_ok = xwave && xr < rows_here;
  const float* xsrc = XF32 ? a.Ax + (int64_t)(m0 + (xrow_ok ? xr : 0)) * a.lda + 16 * xhf : nullptr;
  const int xblock = (NL * (xr >> 5)) * 1024 + (xr & 31) * 16;     // byte offset of (tile row, row) inside a stage, chunk 0, plane hi
  float xscale = 1.f;                                              // NL = 2: the power of two that lifts my row into fp16's range
  if constexpr (NL == 2) {
    float mx = 0.f;
    if (xrow_ok)
      for (int g = 0; g < a.xgroups; ++g) mx = fmaxf(mx, a.xmax[(int64_t)(m0 + xr) * a.xgroups + g]);
    xscale = limb16_scale(mx);
  }
  f32x4 xv[4];                                                     // the super-tile in flight
  float xh[8];                                                     // second chunk of the super-tile being stored
  // The loads are issued by EVERY wave, outside any branch, from an address that is always valid (a wave or row that has nothing to
  // split reads the zero block, a super-tile past the end re-reads the last one; what must not count is zeroed in x_store).  Inside
  // the `if (xwave)` region hipcc loaded into temporaries and copied them into the loop-carried registers at the region's end,
  // behind an s_waitcnt vmcnt(0): every load was synchronous — the even k-tiles' split step took 1550-1750 ticks against 335 without
  // its loads (s_memtime stamps, scripts/bench_limb_timing.py), and all eight waves waited for it at the barrier.
  const float* xbase = xrow_ok ? xsrc : reinterpret_cast<const float*>(a.zeros);
  const int xkmax = xrow_ok ? a.K - 16 - 16 * xhf : 0;
  auto x_load = [&](int S) {
    const float* p = xbase + min(32 * S, xkmax);
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(p + 4 * j);
  };
  auto x_store = [&](int S, int half, const float* v) {            // 8 values -> chunk `half` of k-tile 2 S + xhf
    if (2 * S + xhf >= ntiles || xr >= 32 * nu) return;           // (a tile row the panel does not have is never multiplied)
    // (rows past M inside the panel were LOADED from the zero block, xbase: nothing to mask here — eight v_cndmask per eight values
    //  went with that, a fifth of the split's instructions)
    const float* z = v;
    unsigned char* p = lds + ((2 * S + xhf) % STAGES) * STAGE_BYTES + xblock + half * 512;
    if constexpr (NL == 2) {
      uint4 h, l;
      split8_16(z, xscale, h, l);
      *reinterpret_cast<uint4*>(p) = h;
      *reinterpret_cast<uint4*>(p + 1024) = l;
    } else {
      uint4 h, m, l;
      split8(z, h, m, l);
      *reinterpret_cast<uint4*>(p) = h;
      *reinterpret_cast<uint4*>(p + 1024) = m;
      *reinterpret_cast<uint4*>(p + 2048) = l;
    }
  };
  auto x_first = [&](int S) {                                      // chunk 0 now, chunk 1 kept for the next k-tile
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = xv[0][j]; v[4 + j] = xv[1][j]; xh[j] = xv[2][j]; xh[4 + j] = xv[3][j]; }
    x_store(S, 0, v);
  };

  // ---- fragments -----------------------------------------------------------------------------------------------------
  struct Limbs { bf16x8 hi, mid, lo; };
  auto read_planes = [&](const unsigned char* p) {          // (NL = 2: hi, lo; `mid` stays unused)
    Limbs f;
    f.hi = *reinterpret_cast<const bf16x8*>(p);
    if constexpr (NL == 3) {
      f.mid = *reinterpret_cast<const bf16x8*>(p + 1024);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 2048);
    } else {
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
      f.mid = f.lo;
    }
    return f;
  };
  auto read_x = [&](int stage, int tm) { return read_planes(lds + stage * STAGE_BYTES + (NL * tm) * 1024 + 16 * lane); };
  auto read_w = [&](int stage) { return read_planes(lds + stage * STAGE_BYTES + (PA + NL * wave) * 1024 + 16 * lane); };
  f32x16 acc[T32];
#pragma unroll
  for (int tm = 0; tm < T32; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
  // small terms first: the three 2^-16 products, then the two 2^-8 ones, then the leading one
  auto products = [&](f32x16 c, const Limbs& w, const Limbs& x) {
    if constexpr (NL == 2) {
      const f16x8 wh = __builtin_bit_cast(f16x8, w.hi), wl = __builtin_bit_cast(f16x8, w.lo);
      const f16x8 xh_ = __builtin_bit_cast(f16x8, x.hi), xl = __builtin_bit_cast(f16x8, x.lo);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh_, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh_, c, 0, 0, 0);
      return c;
    } else {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.hi, c, 0, 0, 0);
      return c;
    }
  };

  // ---- pipeline ------------------------------------------------------------------------------------------------------
  // The W limbs of a k-tile stay in registers for the whole tile; the X limbs rotate through two register sets, row tile
  // tm+1 read while tile tm is in the matrix pipe.  (The six MFMAs of a row tile form a dependent chain on one accumulator;
  // taking the row tiles in pairs with the 2 x 6 MFMAs alternating between two accumulators was measured: no change, 75.3 us and
  // 59.5 us for the MFMA-only loop either way — the chain is not what keeps the bf16 pipe at ~60 % busy.  A copy of the loop per
  // wave role with a straight-line steady-state form — no role tests, constant wait counts, W register sets alternating by the
  // parity of t — was measured too: -12 % / -2 % / -1 % at 64 / 96 / 128-row panels, but 256 VGPRs and spills at 160 rows, the
  // panel height of the C2 batches: +5 %.  Not kept.)  The synchronisation point sits in front of the LAST row tile of k-tile t:
  // by then every read of stage t % 4 has been issued, so the barrier releases that stage, and the reads of k-tile t+1 (its W
  // limbs, its first X tile) go out under the last row tile's six MFMAs.  The DMA of k-tile t+3 is issued during k-tile t, a few
  // instructions in front of every row tile but the last (its stage, (t-1) % 4, was released inside k-tile t-1).
  Limbs w_cur, w_nxt, xs[2];
  constexpr int SLOTS = T32 > 1 ? T32 - 1 : 1;
  constexpr int PER = (G + SLOTS - 1) / SLOTS;
  constexpr int C1 = PER < G ? PER : G, C2 = 2 * PER < G ? 2 * PER : G, C3 = 3 * PER < G ? 3 * PER : G;
  if (ntiles > 0) {
#pragma unroll
    for (int i = 0; i < STAGES; ++i)
      if (i < ntiles) issue(i);
    if constexpr (XF32) {
      x_load(0);
      if (xwave) {
        x_first(0);
        x_store(0, 1, xh);
      }
      x_load(1);
    }
    wait_dma(min(STAGES - 1, ntiles - 1));
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    w_cur = read_w(0);
    xs[0] = read_x(0, 0);
  }
  // k-tile t; row tile tm lives in register set (tm + PAR) & 1 (PAR alternates from k-tile to k-tile when T32 is odd)
#ifdef RELGNN_LIMB_TIMING
  unsigned long long seg[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  auto ktile = [&](int t, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const int stage = t % STAGES;
    const bool more = t + 1 < ntiles;
    const bool feed = !(ABL & 1) && t >= 1 && t + STAGES - 1 < ntiles;
    TSTAMP(ts0);
    const int fstage = (t + STAGES - 1) % STAGES;
    // XF32: this k-tile's share of the split, at the top of the k-tile (the SIMD partner feeds the matrix pipe meanwhile; moving
    // wave 4's share — it shares SIMD 0 with wave 0 — two row tiles down was measured: 83.3 vs 81.3 us, no gain).
    // XF32: this k-tile's step of the split, at the top of the k-tile (the SIMD partner feeds the matrix pipe meanwhile; wave 4
    // shares SIMD 0 with wave 0 — moving its step two row tiles down only moves the imbalance: 83.0 / 75.1 us against 81.4 / 68.1 at
    // 160 / 128-row panels).  The loads of the next super-tile follow for ALL waves, outside any branch.
    if constexpr (XF32) {
      const int S = (t >> 1) + 1;
      if ((t & 1) == 0) {
        if (xwave && 2 * S < ntiles) x_first(S);
        x_load(S + 1);
      } else {
        if (xwave && 2 * S < ntiles) x_store(S, 1, xh);
      }
    }
#ifdef RELGNN_LIMB_TIMING
    TSTAMP(ts1);
    if (t & 1) seg[6] += ts1 - ts0; else seg[0] += ts1 - ts0;
    unsigned long long tsb = ts1;
#endif
#pragma unroll
    for (int tm = 0; tm < T32; ++tm) {
      Limbs& xc = xs[(tm + PAR) & 1];
      Limbs& xn = xs[(tm + PAR + 1) & 1];
      if (feed) {                                              // blocks [tm * PER, (tm + 1) * PER) of k-tile t + 3
        if (tm == 0) issue_part(fstage, std::integral_constant<int, 0>{}, std::integral_constant<int, C1>{});
        if (tm == 1 && SLOTS > 1) issue_part(fstage, std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{});
        if (tm == 2 && SLOTS > 2) issue_part(fstage, std::integral_constant<int, C2>{}, std::integral_constant<int, C3>{});
        if (tm == 3 && SLOTS > 3) issue_part(fstage, std::integral_constant<int, C3>{}, std::integral_constant<int, G>{});
      }
      if (tm == T32 - 1) {
        if (more) {
          TSTAMP(ts2);
          if constexpr (!(ABL & 4)) {
            // my blocks of k-tile t+1 have landed (issued so far: up to k-tile t+3)
            if constexpr (!(ABL & 1)) wait_dma(min(STAGES - 2, ntiles - 2 - t));
            TSTAMP(ts3);
            wait_lgkm0();                                      // my reads of stage t % 4 and my limb stores are done
            TSTAMP(ts4);
            __builtin_amdgcn_s_barrier();                      // -> k-tile t+1 complete for everybody, stage t % 4 free
            TSTAMP(ts5);
#ifdef RELGNN_LIMB_TIMING
            seg[1] += ts2 - tsb; seg[2] += ts3 - ts2; seg[3] += ts4 - ts3; seg[4] += ts5 - ts4; tsb = ts5;
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(ABL & 2)) {
            w_nxt = read_w((t + 1) % STAGES);
            xn = read_x((t + 1) % STAGES, 0);
          } else { w_nxt = w_cur; xn = xc; }
        }
        acc[tm] = products(acc[tm], w_cur, xc);
#ifdef RELGNN_LIMB_TIMING
        { TSTAMP(ts6); seg[5] += ts6 - tsb; }
#endif
      } else {
        if constexpr (!(ABL & 2)) xn = read_x(stage, tm + 1); else xn = xc;
        acc[tm] = products(acc[tm], w_cur, xc);
      }
    }
    w_cur = w_nxt;
  };
  if constexpr (T32 & 1) {
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
      ktile(t, std::integral_constant<int, 0>{});
      ktile(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < ntiles) ktile(t, std::integral_constant<int, 0>{});
  } else {
    for (int t = 0; t < ntiles; ++t) ktile(t, std::integral_constant<int, 0>{});
  }

#ifdef RELGNN_LIMB_TIMING
  if (a.timing && lane == 0 && lb < 64) {
#pragma unroll
    for (int i = 0; i < 7; ++i) a.timing[(lb * 8 + wave) * 8 + i] = seg[i];
    a.timing[(lb * 8 + wave) * 8 + 7] = ntiles;
  }
#endif
  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // 32x32 tile: lane holds output row (lane & 31) x columns 8 c + 4 h + {0..3}, c = 0..3 (register r = 4 c + {0..3})
  const int i32 = lane & 31, h32 = lane >> 5;
  const int colw = n0 + wave * 32;
  auto finish = [&](f32x4 v, int col) {
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
    if (a.act != RELGNN_ACT_LINEAR) {
      v[0] = act_rt(a.act, v[0]); v[1] = act_rt(a.act, v[1]); v[2] = act_rt(a.act, v[2]); v[3] = act_rt(a.act, v[3]);
    }
    return v;
  };
#pragma unroll
  for (int tm = 0; tm < T32; ++tm) {
    const int r = tm * 32 + i32;
    if (r < rows_here) {
      float* crow = a.C + (int64_t)(m0 + r) * a.ldc;
      int unscale = 0;
      if constexpr (NL == 2) {                          // the row's and the weights' powers of two leave: exact (one ldexp)
        float mx = 0.f;
        for (int g = 0; g < a.xgroups; ++g) mx = fmaxf(mx, a.xmax[(int64_t)(m0 + r) * a.xgroups + g]);
        unscale = limb16_unscale_exp(mx) + limb16_unscale_exp(a.wmax[0]);
      }
      const float* yrow = a.dy ? a.dy + (int64_t)(m0 + r) * a.ldy : nullptr;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = colw + 8 * c + 4 * h32;
        f32x4 v = f32x4{acc[tm][4 * c], acc[tm][4 * c + 1], acc[tm][4 * c + 2], acc[tm][4 * c + 3]};
        if constexpr (NL == 2) v = f32x4{ldexpf(v[0], unscale), ldexpf(v[1], unscale), ldexpf(v[2], unscale), ldexpf(v[3], unscale)};
        v = finish(v, col);
        if (yrow) {
          const f32x4 y = *reinterpret_cast<const f32x4*>(yrow + col);
          v = f32x4{v[0] * dact_from_output(a.dact, y[0]), v[1] * dact_from_output(a.dact, y[1]),
                    v[2] * dact_from_output(a.dact, y[2]), v[3] * dact_from_output(a.dact, y[3])};
        }
        *reinterpret_cast<f32x4*>(crow + col) = v;
      }
    }
  }
}

// ---- 128 x 128 panels, gathered rows, per-panel weights: the per-(node, type) transforms of many-type graphs --------------------------
//     C[r][n] = act(bias[n] + sum_k A[rows[r]][k] * B_sel(r)[n][k]),   n in a 128-column chunk     (gnns/gnn_film.py:92-106; rgcn.py,
//     ggnn.py on VarMisuse-shaped batches; K = 128 there: eight k-tiles per panel)
// With so short a reduction a panel is mostly latency (gather the rows, split, 8 k-tiles, 64 KB of stores), so the geometry is chosen
// for TWO workgroups per CU: 128 rows x 128 columns, 8 waves = 2 row groups x 4 column tiles (a wave: 2 row tiles x 32 columns),
// 3 stages of 24 KiB, <= 128 VGPRs.  Waves 0-3 issue the DMA of the W limb blocks (3 per k-tile each); waves 4-7 split the fp32 rows,
// waves 4-5 the even k-tiles and waves 6-7 the odd ones: a thread owns one row x one whole k-tile (64 B) every second k-tile, loaded
// four k-tiles before it is split (by every wave, outside any branch: see limb_gemm_kernel) into the stage the barrier of the
// previous k-tile released.  rows (nullable): output row r reads A[rows[r]] (< 0: zeros) — the tf.nn.embedding_lookup of the
// (node, type) tables in the load addresses; b_select (nullable): rows [p * rows_per_select, ..) use the limb tiles at
// B + b_select[p] * b_stride (one Edge_%i kernel per 512-row tile; rows_per_select % 128 == 0).
constexpr int SEL_STAGES = 3;

struct LimbSelArgs {
  const float* Ax; int64_t lda; const int32_t* rows;
  const uint16_t* B; const int32_t* b_select; int32_t rows_per_select; int64_t b_stride;
  const float* bias; const uint16_t* zeros; float* C; int64_t ldc;
  int32_t M, N, K, act;
  int32_t panels, chunks;
  int32_t n_valid;                     // columns that exist in C (<= N = chunks * 128: the last chunk may be cut, e.g. 121 labels)
};

__global__ __launch_bounds__(512, 2) void limb_gemm_sel_kernel(const LimbSelArgs a) {
  constexpr int TW = 2, T32 = 4, PR = 128, NC = 128;
  constexpr int PA = 3 * T32, PB = 3 * (NC / 32), P = PA + PB;     // 12 + 12 blocks
  constexpr int STAGE_BYTES = P * 1024;
  constexpr int G = PB / 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[SEL_STAGES * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t lb = xcd_logical_block((int64_t)a.panels * a.chunks);
  if (lb < 0) return;
  const int q = (int)(lb / a.chunks), chunk = (int)(lb % a.chunks);
  const int m0 = q * PR;
  const int rows_here = min(PR, a.M - m0);
  const int n0 = chunk * NC;
  const int ntiles = a.K / BK;
  const uint16_t* Bp = a.B + (a.b_select ? (int64_t)a.b_select[m0 / a.rows_per_select] * a.b_stride : 0);

  // ---- W by DMA (waves 0-3) -------------------------------------------------------------------------------------------------
  const bool loader = wave < 4;
  constexpr int TILE = 3 * 512;
  const uint16_t* src[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int cb = (wave & 3) + 4 * g;
    src[g] = Bp + ((int64_t)(n0 / 32 + cb / 3) * ntiles) * TILE + (cb % 3) * 512 + 8 * lane;
  }
  auto issue_w = [&](int stage) {
    if (!loader) return;
    unsigned char* dst = lds + stage * STAGE_BYTES + PA * 1024;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      dma16(src[g], dst + ((wave & 3) + 4 * g) * 1024);
      src[g] += TILE;
    }
  };
  // (the loads of the split below are issued by the loader waves too, from the zero block: a count that leaves them out waits for
  //  MORE of the older DMA instructions than necessary, never for fewer)
  auto wait_w = [&](int tiles) {
    if (!loader) return;
    if (tiles >= 1) wait_vm<G>(); else wait_vm<0>();
  };

  // ---- the split (waves 4-7): thread x = tid & 255 owns row x & 127, chunk x >> 7 (k 8h .. 8h+7) of EVERY k-tile: 32 B per k-tile,
  // two register sets (even / odd k-tiles), each loaded two k-tiles before it is split ------------------------------------------
  const bool xwave = wave >= 4;
  const int x = tid & 255;
  const int xr = x & 127, xh_ = x >> 7;
  int64_t row = -1;
  if (xwave && xr < rows_here) row = a.rows ? (int64_t)a.rows[m0 + xr] : (int64_t)(m0 + xr);
  const bool xok = row >= 0;
  const float* xbase = xok ? a.Ax + row * a.lda + 8 * xh_ : reinterpret_cast<const float*>(a.zeros);
  const int xkmax = xok ? a.K - 16 : 0;
  const int xblock = (3 * (xr >> 5)) * 1024 + xh_ * 512 + (xr & 31) * 16;
  f32x4 xva[2], xvb[2];                               // my chunk of an even / an odd k-tile in flight
  auto x_load = [&](f32x4 (&v)[2], int kt) {          // (every wave, no branch; past the end: the last k-tile again)
    const float* p = xbase + min(16 * kt, xkmax);
    v[0] = *reinterpret_cast<const f32x4*>(p);
    v[1] = *reinterpret_cast<const f32x4*>(p + 4);
  };
  auto x_split = [&](const f32x4 (&v)[2], int kt) {   // my chunk of k-tile kt -> stage kt % 3
    float z[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i] = xok ? v[0][i] : 0.f; z[4 + i] = xok ? v[1][i] : 0.f; }
    uint4 h, m, l;
    split8(z, h, m, l);
    unsigned char* p = lds + (kt % SEL_STAGES) * STAGE_BYTES + xblock;
    *reinterpret_cast<uint4*>(p) = h;
    *reinterpret_cast<uint4*>(p + 1024) = m;
    *reinterpret_cast<uint4*>(p + 2048) = l;
  };

  // ---- fragments / products -----------------------------------------------------------------------------------------------
  struct Limbs { bf16x8 hi, mid, lo; };
  auto read_blk = [&](int stage, int blk) {
    const unsigned char* p = lds + stage * STAGE_BYTES + blk * 1024 + 16 * lane;
    Limbs f;
    f.hi = *reinterpret_cast<const bf16x8*>(p);
    f.mid = *reinterpret_cast<const bf16x8*>(p + 1024);
    f.lo = *reinterpret_cast<const bf16x8*>(p + 2048);
    return f;
  };
  f32x16 acc[TW];
#pragma unroll
  for (int tm = 0; tm < TW; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
  auto products = [&](f32x16 c, const Limbs& w, const Limbs& xx) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xx.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, xx.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xx.mid, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xx.mid, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xx.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xx.hi, c, 0, 0, 0);
    return c;
  };

  // ---- pipeline: k-tile t is multiplied while k-tile t+1 is complete in LDS and k-tile t+2 arrives (W by DMA, X split from the
  // register set of its parity, whose next load — k-tile t+4 — follows) ----------------------------------------------------------
  if (ntiles > 0) {
    issue_w(0);
    if (1 < ntiles) issue_w(1);
    x_load(xva, 0);
    x_load(xvb, 1);
    if (xwave) {
      x_split(xva, 0);
      if (1 < ntiles) x_split(xvb, 1);
    }
    x_load(xva, 2);
    x_load(xvb, 3);
    wait_w(min(1, ntiles - 1));
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
  }
  Limbs w_cur, w_nxt, x0, x1;
  if (ntiles > 0) {
    w_cur = read_blk(0, PA + 3 * wn);
    x0 = read_blk(0, 3 * (wm * TW));
  }
  // (four register sets — a lead of four k-tiles instead of two — need 138 VGPRs: one workgroup per CU, 278 instead of 247 us)
  auto ktile = [&](int t, auto odd_c) {
    constexpr int ODD = decltype(odd_c)::value;
    f32x4 (&xv)[2] = ODD ? xvb : xva;
    const int stage = t % SEL_STAGES;
    const bool more = t + 1 < ntiles;
    if (t + 2 < ntiles) issue_w((t + 2) % SEL_STAGES);
    if (xwave && t + 2 < ntiles) x_split(xv, t + 2);
    x_load(xv, t + 4);
    x1 = read_blk(stage, 3 * (wm * TW + 1));
    acc[0] = products(acc[0], w_cur, x0);
    if (more) {
      // W tile t+1 was issued at k-tile t-1 (or in the prologue); after it: W tile t+2 (this k-tile) and loads of the split
      wait_w(t + 2 < ntiles ? 1 : 0);
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      w_nxt = read_blk((t + 1) % SEL_STAGES, PA + 3 * wn);
      x0 = read_blk((t + 1) % SEL_STAGES, 3 * (wm * TW));
    }
    acc[1] = products(acc[1], w_cur, x1);
    w_cur = w_nxt;
  };
  {
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
      ktile(t, std::integral_constant<int, 0>{});
      ktile(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < ntiles) ktile(t, std::integral_constant<int, 0>{});
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  const int i32 = lane & 31, h32 = lane >> 5;
  const int colw = n0 + wn * 32;
  auto finish = [&](f32x4 v, int col) {
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
    if (a.act != RELGNN_ACT_LINEAR) {
      v[0] = act_rt(a.act, v[0]); v[1] = act_rt(a.act, v[1]); v[2] = act_rt(a.act, v[2]); v[3] = act_rt(a.act, v[3]);
    }
    return v;
  };
  if (a.n_valid == a.N && a.ldc % 4 == 0) {
#pragma unroll
    for (int tm = 0; tm < TW; ++tm) {
      const int r = (wm * TW + tm) * 32 + i32;
      if (r < rows_here) {
        float* crow = a.C + (int64_t)(m0 + r) * a.ldc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col = colw + 8 * c + 4 * h32;
          const f32x4 v = f32x4{acc[tm][4 * c], acc[tm][4 * c + 1], acc[tm][4 * c + 2], acc[tm][4 * c + 3]};
          *reinterpret_cast<f32x4*>(crow + col) = finish(v, col);
        }
      }
    }
  } else {
    // a cut last chunk and / or rows that are not 16-byte aligned (the [V, 121] logits of the PPI head): element by element
#pragma unroll
    for (int tm = 0; tm < TW; ++tm) {
      const int r = (wm * TW + tm) * 32 + i32;
      if (r < rows_here) {
        float* crow = a.C + (int64_t)(m0 + r) * a.ldc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col0 = colw + 8 * c + 4 * h32;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[tm][4 * c + e];
            if (a.bias && col0 + e < a.n_valid) v[e] += a.bias[col0 + e];
            if (a.act != RELGNN_ACT_LINEAR) v[e] = act_rt(a.act, v[e]);
          }
          if (col0 + 3 < a.n_valid) {                 // (4-byte aligned 16-byte store: global memory takes it)
            typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
            *reinterpret_cast<f32x4_u*>(crow + col0) = f32x4_u{v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col0 + e < a.n_valid) crow[col0 + e] = v[e];
          }
        }
      }
    }
  }
}

// ---- K = 128 with the weights RESIDENT in LDS: persistent workgroups over runs of 128-row panels --------------------------------
// The per-(node, type) transforms of the many-type models (gnns/gnn_film.py:92-106) are K = 128 products over ~1e6 gathered rows: in
// limb_gemm_sel_kernel every 128-row panel fetched its 96 KB of weight limbs again (more than its 64 KB of rows or its 64 KB of
// results) behind a cold pipeline of 8 k-tiles.  Here one workgroup per CU owns a contiguous run of panels of one 128-column chunk:
// the 8 x 12 KB of weight limbs stay in LDS until the edge type of the select tile changes (the tiles are sorted by type); the rows
// stream through a 4-stage ring that never drains — all 512 threads gather (one 32-byte chunk per 32 k, a panel ahead, in four
// register sets; the row ids two panels ahead) and split; the stores of panel p leave while panel p+1 is multiplied.
// 160 KB of LDS, one workgroup per CU.
constexpr int TILE_RING = 4;

__global__ __launch_bounds__(512) void limb_gemm_tile_kernel(const LimbSelArgs a) {
  constexpr int TW = 2, PR = 128, NC = 128, KT = 8;
  constexpr int PA = 12, WB = 12;                      // 1 KiB blocks per k-tile: rows 4 x 3 limbs, weights 4 x 3 limbs
  constexpr int W_BYTES = KT * WB * 1024, STAGE_BYTES = PA * 1024;
  constexpr int OUT_BYTES = 8 * 2048;                  // per wave: 16 result rows x 128 B on their way out
  __shared__ __attribute__((aligned(16))) unsigned char lds[W_BYTES + TILE_RING * STAGE_BYTES + OUT_BYTES];
  unsigned char* const ring = lds + W_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  TSTAMP(t_begin);
  const int64_t lb = xcd_logical_block((int64_t)a.panels * a.chunks);      // panels: workgroups per column chunk here
  if (lb < 0) return;
  const int g = (int)(lb / a.chunks), chunk = (int)(lb % a.chunks);
  const int all = (a.M + PR - 1) / PR;                // 128-row panels of the product
  const int base = all / a.panels, rem = all % a.panels;
  const int p0 = g * base + min(g, rem), np = base + (g < rem ? 1 : 0);     // my run of panels
  if (np == 0) return;
  const int n0 = chunk * NC;
  auto type_of = [&](int p) { return a.b_select ? a.b_select[((p0 + p) * PR) / a.rows_per_select] : 0; };

  // ---- the weights: 96 blocks, 12 per wave --------------------------------------------------------------------------------
  constexpr int TILE = 3 * 512;
  auto load_w = [&](int type) {
    const uint16_t* Bp = a.B + (int64_t)type * a.b_stride;
#pragma unroll
    for (int gq = 0; gq < KT * WB / 8; ++gq) {
      const int i = wave + 8 * gq, kt = i / WB, cb = i % WB;
      dma16(Bp + ((int64_t)(n0 / 32 + cb / 3) * KT + kt) * TILE + (cb % 3) * 512 + 8 * lane, lds + i * 1024);
    }
  };
  int type_cur = type_of(0);
  load_w(type_cur);

  // ---- the rows: thread = (row xr of the panel, chunk c4 of a 32-k super-tile) ------------------------------------------------
  // (consecutive lanes = consecutive rows: the limb stores of 8 lanes are 128 contiguous bytes, conflict-free; four lanes per row —
  //  one 128-byte line per quad — loaded no faster and cost 4-way conflicts on those stores)
  const int xr = tid & 127, c4 = tid >> 7;
  auto row_id = [&](int p) -> int64_t {               // the table row behind my row of panel p; < 0: a row of zeros
    const int r = (p0 + min(p, np - 1)) * PR + xr;
    int64_t row = a.rows ? (int64_t)a.rows[min(r, a.M - 1)] : (int64_t)r;
    return (p < np && r < a.M) ? row : -1;
  };
  auto row_ptr = [&](int64_t id) -> const float* { return id >= 0 ? a.Ax + id * a.lda + 8 * c4 : nullptr; };
  // k-tile 2j + (c4 >> 1) of a panel lives in ring stage (2j + (c4 >> 1)) % 4 (8 k-tiles per panel: the same stage in every panel)
  const int xblock = (3 * (xr >> 5)) * 1024 + (c4 & 1) * 512 + (xr & 31) * 16;
  f32x4 xv[4][2];                                     // super-tile j of a panel in flight
  auto x_load = [&](f32x4 (&v)[2], const float* rp, int j) {      // (every thread, no branch)
    const float* p = rp ? rp + 32 * j : reinterpret_cast<const float*>(a.zeros);
    v[0] = *reinterpret_cast<const f32x4*>(p);
    v[1] = *reinterpret_cast<const f32x4*>(p + 4);
  };
  auto x_split = [&](const f32x4 (&v)[2], bool ok, int j) {
    float z[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i] = ok ? v[0][i] : 0.f; z[4 + i] = ok ? v[1][i] : 0.f; }
    uint4 h, m, l;
    split8(z, h, m, l);
    unsigned char* p = ring + ((2 * j + (c4 >> 1)) % TILE_RING) * STAGE_BYTES + xblock;
    *reinterpret_cast<uint4*>(p) = h;
    *reinterpret_cast<uint4*>(p + 1024) = m;
    *reinterpret_cast<uint4*>(p + 2048) = l;
  };

  struct Limbs { bf16x8 hi, mid, lo; };
  auto read3 = [&](const unsigned char* p) {
    Limbs f;
    f.hi = *reinterpret_cast<const bf16x8*>(p);
    f.mid = *reinterpret_cast<const bf16x8*>(p + 1024);
    f.lo = *reinterpret_cast<const bf16x8*>(p + 2048);
    return f;
  };
  auto read_w = [&](int kt) { return read3(lds + (kt * WB + 3 * wn) * 1024 + 16 * lane); };
  auto read_x = [&](int kt, int tm) { return read3(ring + (kt % TILE_RING) * STAGE_BYTES + (3 * (wm * TW + tm)) * 1024 + 16 * lane); };
  f32x16 acc[TW];
#pragma unroll
  for (int tm = 0; tm < TW; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
  const int i32 = lane & 31, h32 = lane >> 5;
  const int colw = n0 + wn * 32;
  const bool plain_out = !a.bias && a.act == RELGNN_ACT_LINEAR;
  auto finish = [&](f32x4 v, int col) {
    if (plain_out) return v;
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
    if (a.act != RELGNN_ACT_LINEAR) {
      v[0] = act_rt(a.act, v[0]); v[1] = act_rt(a.act, v[1]); v[2] = act_rt(a.act, v[2]); v[3] = act_rt(a.act, v[3]);
    }
    return v;
  };
  // A lane of the 32 x 32 result holds 4 x 16 bytes of ONE row (columns 8c + 4h ..): stored directly, every 128-byte line of C is
  // written as four 32-byte pieces by four instructions.  Each wave turns its tile through 2 KB of LDS (16 rows at a time, 16-byte
  // slots XOR-swizzled by the row) so that 8 consecutive lanes write one whole line.
  unsigned char* const obuf = lds + W_BYTES + TILE_RING * STAGE_BYTES + wave * 2048;
  auto store_panel = [&](int p) __attribute__((always_inline)) {
    const int m0 = (p0 + p) * PR;
#pragma unroll
    for (int tm = 0; tm < TW; ++tm) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if ((i32 >> 4) == half) {
          const int rr = i32 & 15;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = colw + 8 * c + 4 * h32;
            const f32x4 v = f32x4{acc[tm][4 * c], acc[tm][4 * c + 1], acc[tm][4 * c + 2], acc[tm][4 * c + 3]};
            *reinterpret_cast<f32x4*>(obuf + rr * 128 + (((2 * c + h32) ^ (rr & 7)) << 4)) = finish(v, col);
          }
        }
        wait_lgkm0();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int rr = (lane >> 3) + 8 * k, p8 = lane & 7;
          const f32x4 v = *reinterpret_cast<const f32x4*>(obuf + rr * 128 + ((p8 ^ (rr & 7)) << 4));
          const int r = m0 + (wm * TW + tm) * 32 + 16 * half + rr;
          if (r < a.M) *reinterpret_cast<f32x4*>(a.C + (int64_t)r * a.ldc + colw + 4 * p8) = v;
        }
        wait_lgkm0();                                  // (the reads are done before the next half overwrites the slots)
      }
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) acc[tm][r16] = 0.f;
    }
  };

  // ---- prologue: panel 0's four super-tiles in flight, super-tile 0 split; row pointers one panel, row ids two panels ahead ------
  const int64_t id0 = row_id(0);
  int64_t id_nxt = row_id(1), id_nn = row_id(2);
  const float* const ptr0 = row_ptr(id0);
  bool ok_cur = id0 >= 0;
  x_load(xv[0], ptr0, 0);
  x_load(xv[1], ptr0, 1);
  x_load(xv[2], ptr0, 2);
  x_load(xv[3], ptr0, 3);
  x_split(xv[0], ok_cur, 0);
  wait_vm<0>();                                       // the weights have landed (and the rest of panel 0, needed next anyway)
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
#ifdef RELGNN_LIMB_TIMING
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TSTAMP(t_pro);
  seg[0] = t_pro - t_begin;
#endif

  // Super-tile j of panel p (k-tiles 2j, 2j+1 in ring stages 2j % 4, (2j+1) % 4): ONE barrier per 32 k.  While its 24 products run
  // (the two row tiles alternate, so that no MFMA waits for the one before it), super-tile j+1 (of the next panel when j = 3) is
  // split from its register set into the other two stages — they were read one super-tile ago, before the last barrier — and
  // set j, split one super-tile ago, is reloaded for the next panel (three super-tiles, 48 KB per CU, ahead of its split; a lead of
  // seven super-tiles in eight register sets only lengthened the queues in front of the stores: 259 vs 227 us).
  auto two = [&](const Limbs& w, const Limbs& xa, const Limbs& xb) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xa.lo, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xb.lo, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, xa.hi, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, xb.hi, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xa.mid, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xb.mid, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xa.mid, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xb.mid, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xa.hi, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, xb.hi, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xa.hi, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, xb.hi, acc[1], 0, 0, 0);
  };
  for (int p = 0; p < np; ++p) {
    const float* const ptr_nxt = row_ptr(id_nxt);
    const bool ok_nxt = id_nxt >= 0;
    const int64_t id_n3 = row_id(p + 3);              // (needed two panels from now)
    auto supertile = [&](auto j_c) __attribute__((always_inline)) {
      constexpr int j = decltype(j_c)::value;
      constexpr int jn = (j + 1) % 4;
      TSTAMP(q0);
      const Limbs wa = read_w(2 * j), xa0 = read_x(2 * j, 0), xa1 = read_x(2 * j, 1);
      x_split(xv[jn], j == 3 ? ok_nxt : ok_cur, jn);  // (past the last panel: zeros that nobody reads)
      TSTAMP(q1);
      x_load(xv[j], ptr_nxt, j);
      const Limbs wb = read_w(2 * j + 1), xb0 = read_x(2 * j + 1, 0), xb1 = read_x(2 * j + 1, 1);
      two(wa, xa0, xa1);
      two(wb, xb0, xb1);
      TSTAMP(q2);
      if constexpr (j == 3) store_panel(p);
      TSTAMP(q3);
      wait_lgkm0();
      TSTAMP(q4);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifdef RELGNN_LIMB_TIMING
      { TSTAMP(q5); seg[1] += q1 - q0; seg[2] += q2 - q1; seg[3] += q3 - q2; seg[4] += q4 - q3; seg[5] += q5 - q4; }
#endif
    };
    supertile(std::integral_constant<int, 0>{});
    supertile(std::integral_constant<int, 1>{});
    supertile(std::integral_constant<int, 2>{});
    supertile(std::integral_constant<int, 3>{});
    ok_cur = ok_nxt;
    id_nxt = id_nn;
    id_nn = id_n3;
    if (p + 1 < np) {                                  // the next panel belongs to another edge type: its weights replace these
      const int t = type_of(p + 1);
      if (t != type_cur) {                             // (every wave is past the barrier behind the last product of panel p)
        type_cur = t;
        load_w(t);
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#ifdef RELGNN_LIMB_TIMING
  if (g_limb_timing_dev && lane == 0 && lb < 64) {
    TSTAMP(t_end);
    seg[6] = t_end - t_begin; seg[7] = np;
#pragma unroll
    for (int i = 0; i < 8; ++i) g_limb_timing_dev[(lb * 8 + wave) * 8 + i] = seg[i];
  }
#endif
}

// ---- weight gradients: P[z] = A[rows of chunk z]^T @ G[rows of chunk z] ------------------------------------------------------
// dW = A^T G for A [V, J], G [V, 256 c] (both fp32 row-major: the reduction index is the ROW of both).  Same matrix-pipe core,
// same LDS blocks, but both operands are split in flight and TRANSPOSED on the way: a thread loads a 4-column x 8-row patch
// (8 x dwordx4, a wave covers whole 1 KiB rows of G / 512-byte row pieces of A), and the 8 values of one column — 8 consecutive
// reduction indices — are exactly one 16-byte limb chunk.  No DMA, no limb copy of anything in HBM.  A workgroup owns 32*T32
// output rows (j) x 256 columns (c) x one chunk of the reduction; the chunks' partial products are summed afterwards in chunk
// order (relgnn_sum_slabs_tail_f32: deterministic, like the fp32 split-K route it replaces).
struct LimbTnArgs {
  const float* A; int64_t lda;         // [V, J]
  const float* G; int64_t ldg;         // [V, C]
  float* P;                            // [Z][J][C] partial products
  int32_t V, J, C;
  int32_t rows_per_chunk;              // % 32 == 0
  int32_t panels, chunks, Z;
  // NL = 2: magnitudes the power-of-two scales come from (device floats): column j of A takes amax[j / a_cols] (a_cols = 1: one
  // scale per column, J: one for the operand, anything between: per group of a_cols consecutive columns — e.g. per edge type of
  // the aggregate-first layer's [V, L*D] operand), column c of G gmax[c / g_cols].  A column's scale factors out of its row /
  // column of the product.
  const float* amax; const float* gmax; int32_t a_cols, g_cols;
  // GATHER: reduction row r of A is row a_rows[r] of the [*, J] table A (< 0: a row of zeros, read from `zeros`); G is read in place
  const int32_t* a_rows; const float* zeros;
};

// NL = 3: bf16 triples, six products; NL = 2: fp16 pairs behind one power-of-two scale per COLUMN of each operand (the reduction
// runs over the rows of both operands, so a row's scale would not factor out, a column's does: out[j][c] carries sa[j] * sg[c]).
// An element 2^-18 below its column's largest loses low bits of a term that is 2^-18 of the largest terms of ITS sum; with one
// scale per operand (astride = gstride = 0, the first form) that was 2^-18 of the operand's largest, and a column of small
// gradients lost relative precision (measured: tests/test_gpu_limb_gemm.py, column magnitudes 1 .. 1e-10).  Three products.
// NC = 256: a wave owns 32 of the 256 output columns and all T32 row tiles.  NC = 128 (the typed [128, 128] partials of many-type
// graphs): four column blocks x two row groups — wave w owns column block w % 4 and row tiles (w / 4) * T32 / 2 .. + T32 / 2.
// GATHER: the reduction rows of A are gathered through a_rows (the compact pair tables' row -> node map, ops.typed_linear): a chunk
// is then one 512-row tile of ONE edge type and P[z] its partial weight gradient.
template <int T32, int NL = 3, int NC = 256, bool GATHER = false>
__global__ __launch_bounds__(512) void limb_gemm_tn_kernel(const LimbTnArgs a) {
  constexpr int PR = 32 * T32;
  constexpr int CW = NC / 32, RW = 8 / CW, TPW = T32 / RW;          // column waves, row groups, row tiles per wave
  static_assert((NC == 256 || NC == 128) && T32 % RW == 0 && TPW >= 1, "wave roles");
  constexpr int PA = NL * T32, PB = NL * (NC / 32), P = PA + PB;
  constexpr int STAGE_BYTES = P * 1024;
  constexpr int XG = PR / 4;                          // 4-column groups of the panel's rows
  constexpr int GG = NC / 4;                          // 4-column groups of G's column chunk
  constexpr int ITEMS = NC + 4 * XG;                  // per 32-row super-tile: G: GG groups x 4 row octets, A: XG x 4
  static_assert(ITEMS <= 512 && STAGES * STAGE_BYTES <= 160 * 1024, "geometry");
  // (+ 3 KiB that nobody reads: where the limb stores of idle threads and of k-tiles past the end go — the split code has no
  //  branches, so that it sits in one basic block with the MFMAs of its k-tile and the scheduler can interleave the two)
  constexpr int DUMP = STAGES * STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE_BYTES + 3072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_z = a.panels * a.chunks;
  const int64_t lb = xcd_logical_block((int64_t)per_z * a.Z);
  if (lb < 0) return;
  const int z = (int)(lb / per_z), pc = (int)(lb % per_z);
  const int q = pc / a.chunks, chunk = pc % a.chunks;
  const int j0 = q * PR, n0 = chunk * NC;
  const int r_beg = z * a.rows_per_chunk;
  const int r_end = min(a.V, r_beg + a.rows_per_chunk);
  const int ntiles = (r_end - r_beg) / BK;                         // even and > 0: whole 32-row super-tiles only

  // ---- my patch: 4 columns x 8 reduction rows of every super-tile ---------------------------------------------------
  const bool is_g = tid < NC;
  const bool active = tid < ITEMS;
  const int idx = is_g ? tid : (active ? tid - NC : 0);            // (idle threads shadow item 0 of A: loads in bounds, stores dumped)
  const int grp = is_g ? (idx % GG) : (idx % XG);                  // column group
  const int oct = is_g ? (idx / GG) : (idx / XG);                  // row octet inside the super-tile: k-tile oct / 2, chunk oct % 2
  const int col = 4 * grp;                                         // first of my 4 columns inside the tile rows of the operand
  const float* base = is_g ? a.G + n0 + col : a.A + j0 + col;
  const int64_t ld = is_g ? a.ldg : a.lda;
  // LDS blocks of THIS kernel keep row i of a tile at slot i ^ ((i >> 3) & 3) (16 B each): the fragment reads stay conflict-free
  // (the XOR permutes inside aligned groups of four slots) and the limb stores — 8 lanes = 8 column groups = rows 4 a + c — hit
  // 8 different bank quads instead of two (4-way conflicts: 114 -> 9x us at [36 k, 768]^T x [36 k, 256])
  const int blk = ((is_g ? PA : 0) + NL * (col >> 5)) * 1024 + (oct & 1) * 512 + (col & 31) * 16;
  float pscale[4] = {1.f, 1.f, 1.f, 1.f};                         // NL = 2: the scales of my four columns
  if constexpr (NL == 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      pscale[c] = limb16_scale(is_g ? a.gmax[(n0 + col + c) / a.g_cols] : a.amax[(j0 + col + c) / a.a_cols]);
  }
  const int cswz = ((col & 31) >> 3) & 3;                          // column c of my patch goes to slot (col & 31) + (c ^ cswz)
  const int dump = DUMP + lane * 16;
  f32x4 pv[8];                                                      // the patch in flight: pv[m][c]
  float ph[16];                                                     // columns 2, 3 of the patch being stored
  const int nsuper = ntiles / 2;                                    // whole super-tiles only (the host hands over V - V % 32 rows)
  // GATHER: the row ids of the NEXT super-tile are fetched one load step ahead (two dependent loads would otherwise sit inside the
  // two k-tiles between a patch's load and its split)
  int nix[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto ix_load = [&](int S) {
    if constexpr (GATHER) {
      const int r0 = min(r_beg + 32 * S + 8 * oct, a.V - 8);       // (r0 % 8 == 0, V % 32 == 0: whole, aligned octets)
      const int4 i0 = *reinterpret_cast<const int4*>(a.a_rows + r0), i1 = *reinterpret_cast<const int4*>(a.a_rows + r0 + 4);
      nix[0] = i0.x; nix[1] = i0.y; nix[2] = i0.z; nix[3] = i0.w; nix[4] = i1.x; nix[5] = i1.y; nix[6] = i1.z; nix[7] = i1.w;
    }
  };
  auto p_load = [&](int S) {                                        // (past the end: some valid row; such a patch is never stored)
    const int r0 = r_beg + 32 * S + 8 * oct;
    if constexpr (GATHER) {
      const float* zsrc = a.zeros;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int64_t row = is_g ? (int64_t)min(r0 + m, a.V - 1) : (int64_t)nix[m];
        const float* p = row >= 0 ? base + row * ld : zsrc;
        pv[m] = *reinterpret_cast<const f32x4*>(p);
      }
      ix_load(S + 1);
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) pv[m] = *reinterpret_cast<const f32x4*>(base + (int64_t)min(r0 + m, a.V - 1) * ld);
    }
  };
  auto p_store = [&](int S, int c, const float* v) {                // 8 reduction values of column c -> one chunk per plane
    const int off = (active && S < nsuper) ? ((2 * S + (oct >> 1)) % STAGES) * STAGE_BYTES + blk + (c ^ cswz) * 16 : dump;
    if constexpr (NL == 2) {
      uint4 h, l;
      split8_16(v, pscale[c], h, l);
      *reinterpret_cast<uint4*>(lds + off) = h;
      *reinterpret_cast<uint4*>(lds + off + 1024) = l;
    } else {
      uint4 h, m, l;
      split8(v, h, m, l);
      *reinterpret_cast<uint4*>(lds + off) = h;
      *reinterpret_cast<uint4*>(lds + off + 1024) = m;
      *reinterpret_cast<uint4*>(lds + off + 2048) = l;
    }
  };
  // step 0 .. 3 of a super-tile's split: columns 0 (+ keep columns 2, 3 aside: the next load reuses pv), 1 | 2, 3
  auto p_step = [&](int S, auto step_c) {
    constexpr int STEP = decltype(step_c)::value;
    float v[8];
    if constexpr (STEP == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) { v[m] = pv[m][0]; ph[m] = pv[m][2]; ph[8 + m] = pv[m][3]; }
      p_store(S, 0, v);
    } else if constexpr (STEP == 1) {
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = pv[m][1];
      p_store(S, 1, v);
    } else if constexpr (STEP == 2) {
      p_store(S, 2, ph);
    } else {
      p_store(S, 3, ph + 8);
    }
  };

  // ---- fragments / products (as in limb_gemm_kernel) ------------------------------------------------------------------
  struct Limbs { bf16x8 hi, mid, lo; };
  auto read_planes = [&](const unsigned char* p) {
    Limbs f;
    f.hi = *reinterpret_cast<const bf16x8*>(p);
    if constexpr (NL == 3) {
      f.mid = *reinterpret_cast<const bf16x8*>(p + 1024);
      f.lo = *reinterpret_cast<const bf16x8*>(p + 2048);
    } else {
      f.lo = *reinterpret_cast<const bf16x8*>(p + 1024);
      f.mid = f.lo;
    }
    return f;
  };
  const int cw = wave % CW, tm0 = (wave / CW) * TPW;               // my column block, my first row tile
  auto read_x = [&](int stage, int tm) {
    return read_planes(lds + stage * STAGE_BYTES + (NL * (tm0 + tm)) * 1024 + 16 * (lane ^ ((lane >> 3) & 3)));
  };
  auto read_w = [&](int stage) {
    return read_planes(lds + stage * STAGE_BYTES + (PA + NL * cw) * 1024 + 16 * (lane ^ ((lane >> 3) & 3)));
  };
  f32x16 acc[TPW];
#pragma unroll
  for (int tm = 0; tm < TPW; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
  auto products = [&](f32x16 c, const Limbs& w, const Limbs& x) {
    if constexpr (NL == 2) {
      const f16x8 wh = __builtin_bit_cast(f16x8, w.hi), wl = __builtin_bit_cast(f16x8, w.lo);
      const f16x8 xh_ = __builtin_bit_cast(f16x8, x.hi), xl = __builtin_bit_cast(f16x8, x.lo);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh_, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh_, c, 0, 0, 0);
      return c;
    } else {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.mid, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, x.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.hi, c, 0, 0, 0);
      return c;
    }
  };

  // ---- pipeline: super-tile S is loaded during k-tile 2S-4, split and stored during k-tiles 2S-2 (columns 0, 1) and 2S-1
  // (columns 2, 3) — after the barrier inside k-tile 2S-3 released its two stages, before the barrier inside k-tile 2S-1
  // publishes k-tile 2S ---------------------------------------------------------------------------------------------------
  // Super-tile S is loaded during k-tile 2S-4, split and stored during k-tiles 2S-2 (columns 0, 1) and 2S-1 (columns 2, 3): after
  // the barrier inside k-tile 2S-3 released its two stages, before the barrier inside k-tile 2S-1 publishes k-tile 2S.
  // Program order inside a k-tile: [reads of the next row tile, six MFMAs, one step of the split] per row tile in front of the
  // barrier, with a scheduling fence behind each — the MFMAs go out first and the ~55 VALU instructions of the step run while
  // they occupy the pipe.  (Split first, MFMAs after — what hipcc makes of it when left alone — puts both waves of a SIMD in
  // their VALU phase at the same time, the barrier keeps them in step, and the pipe idles: 124 us instead of 100 at
  // [36 k, 768]^T x [36 k, 256].)
  Limbs w_cur, w_nxt, xs[2];
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  ix_load(0);
  p_load(0);
  p_step(0, I0{}); p_step(0, I1{}); p_step(0, I2{}); p_step(0, I3{});
  p_load(1);
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  w_cur = read_w(0);
  xs[0] = read_x(0, 0);
  auto ktile = [&](int t, auto odd_c) {
    constexpr int ODD = decltype(odd_c)::value;
    constexpr int PAR = (TPW & 1) ? ODD : 0;                   // row tile tm lives in register set (tm + PAR) & 1
    constexpr int PRE = TPW - 1;                               // row tiles in front of the barrier
    const int stage = t % STAGES;
    const bool more = t + 1 < ntiles;
    const int S = (t >> 1) + 1;
    if constexpr (PRE == 0) {                                  // one row tile: nothing to hide the split under
      if constexpr (ODD == 0) { p_step(S, I0{}); p_step(S, I1{}); p_load(S + 1); } else { p_step(S, I2{}); p_step(S, I3{}); }
    }
#pragma unroll
    for (int tm = 0; tm < TPW; ++tm) {
      Limbs& xc = xs[(tm + PAR) & 1];
      Limbs& xn = xs[(tm + PAR + 1) & 1];
      if (tm == TPW - 1) {
        if (more) {
          wait_lgkm0();                                        // my reads of stage t % 4 and my limb stores are done
          __builtin_amdgcn_s_barrier();                        // -> k-tile t+1 complete for everybody, stage t % 4 free
          __builtin_amdgcn_sched_barrier(0);
          w_nxt = read_w((t + 1) % STAGES);
          xn = read_x((t + 1) % STAGES, 0);
        }
      } else {
        xn = read_x(stage, tm + 1);
      }
      acc[tm] = products(acc[tm], w_cur, xc);
      if constexpr (PRE >= 1) {
        if (tm < PRE) {
          constexpr int A_AT = 0, B_AT = PRE >= 2 ? 1 : 0, L_AT = PRE >= 3 ? 2 : PRE - 1;
          if (tm == A_AT) { if constexpr (ODD == 0) p_step(S, I0{}); else p_step(S, I2{}); }
          if (tm == B_AT) { if constexpr (ODD == 0) p_step(S, I1{}); else p_step(S, I3{}); }
          if (tm == L_AT) { if constexpr (ODD == 0) p_load(S + 1); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    w_cur = w_nxt;
  };
  {
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
      ktile(t, I0{});
      ktile(t + 1, I1{});
    }
  }

  // ---- partial product of this chunk ------------------------------------------------------------------------------------
  const int i32 = lane & 31, h32 = lane >> 5;
  const int colw = n0 + cw * 32;
  float* slab = a.P + (int64_t)z * a.J * a.C;
  int ug[4][4];                                                     // NL = 2: log2(1 / scale) of my output columns
  if constexpr (NL == 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) ug[c][e] = limb16_unscale_exp(a.gmax[(colw + 8 * c + 4 * h32 + e) / a.g_cols]);
  }
#pragma unroll
  for (int tm = 0; tm < TPW; ++tm) {
    const int j = j0 + (tm0 + tm) * 32 + i32;
    float* crow = slab + (int64_t)j * a.C;
    int ua = 0;
    if constexpr (NL == 2) ua = limb16_unscale_exp(a.amax[j / a.a_cols]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v = f32x4{acc[tm][4 * c], acc[tm][4 * c + 1], acc[tm][4 * c + 2], acc[tm][4 * c + 3]};
      if constexpr (NL == 2)          // (one exact scaling by the sum of the two exponents: no overflow on the way, see limb16_unscale_exp)
        v = f32x4{ldexpf(v[0], ua + ug[c][0]), ldexpf(v[1], ua + ug[c][1]), ldexpf(v[2], ua + ug[c][2]), ldexpf(v[3], ua + ug[c][3])};
      *reinterpret_cast<f32x4*>(crow + colw + 8 * c + 4 * h32) = v;
    }
  }
}

// |x| as its bit pattern if x is finite, 0 for inf / NaN: magnitudes order like unsigned integers, and a scale derived from the
// largest FINITE magnitude keeps every finite element of the operand representable — a non-finite element then spoils the sums it
// takes part in (as it does in fp32) and nothing else
__device__ __forceinline__ uint32_t finite_mag_bits(float x) {
  const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu;
  return u < 0x7F800000u ? u : 0u;
}
__device__ __forceinline__ float finite_mag(float x) { return __uint_as_float(finite_mag_bits(x)); }

// the largest finite magnitude of n floats (atomicMax on the bit pattern; the caller zeroes *out first)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ out) {
  float m = 0.f;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(finite_mag(v[0]), finite_mag(v[1]))), fmaxf(finite_mag(v[2]), finite_mag(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - 4 * n4)) m = fmaxf(m, finite_mag(x[4 * n4 + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}

// The largest finite magnitude of every COLUMN of X [rows, cols]: the per-column / per-group scales of the two-fp16-limb weight
// gradient.  Two stages, no atomics (564 workgroups x 256 atomicMax on 256 addresses measured 44 us for the [36 k, 256] gradient
// of a C2 layer, inside a step: the contention, not the 37 MB read): stage 1, a workgroup = 64 column quads x 4 row lanes over a
// range of rows with four independent 16-byte loads in flight per thread, writes its 256 column maxima to part[block]; stage 2
// takes the maximum over the blocks.  Magnitudes as bit patterns (they order like unsigned integers): deterministic.
__global__ __launch_bounds__(256) void col_absmax_part_kernel(const float* __restrict__ X, int64_t ldx, int32_t rows, int32_t cols,
                                                              int32_t rows_per_block, uint32_t* __restrict__ part) {
  const int q = blockIdx.x * 64 + (threadIdx.x & 63);             // column quad
  const int y = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  uint32_t m[4] = {0u, 0u, 0u, 0u};
  if (4 * q < cols) {
    const float* p = X + 4 * q;
    int r = r0 + y;
    for (; r + 12 < r1; r += 16) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (int64_t)(r + 4 * u) * ldx));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = max(m[e], finite_mag_bits(v[u][e]));
    }
    for (; r < r1; r += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + (int64_t)r * ldx);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = max(m[e], finite_mag_bits(v[e]));
    }
  }
  __shared__ uint32_t sh[4][64][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) sh[y][threadIdx.x & 63][e] = m[e];
  __syncthreads();
  if (y == 0 && 4 * q < cols) {
    uint4 v;
    v.x = max(max(m[0], sh[1][threadIdx.x][0]), max(sh[2][threadIdx.x][0], sh[3][threadIdx.x][0]));
    v.y = max(max(m[1], sh[1][threadIdx.x][1]), max(sh[2][threadIdx.x][1], sh[3][threadIdx.x][1]));
    v.z = max(max(m[2], sh[1][threadIdx.x][2]), max(sh[2][threadIdx.x][2], sh[3][threadIdx.x][2]));
    v.w = max(max(m[3], sh[1][threadIdx.x][3]), max(sh[2][threadIdx.x][3], sh[3][threadIdx.x][3]));
    *reinterpret_cast<uint4*>(part + (int64_t)blockIdx.y * cols + 4 * q) = v;
  }
}

// stage 2: 4 columns x 64 block lanes per workgroup of 256 threads — every thread issues its <= 16 loads at once (one memory
// latency), the 64 lanes of a column meet in LDS.  Small workgroups on purpose: this kernel runs on the side stream next to the
// gather, whose 27 k four-wave workgroups keep every wave slot and register of the chip taken; a 1024-thread workgroup needs 16
// free slots on ONE CU at the same moment and starved for 38-50 us (measured), a four-wave one slips in as soon as one retires.
// (One thread per column walking all ~500 partial rows: 500 dependent loads, measured 177 us.)
__global__ __launch_bounds__(256) void col_absmax_final_kernel(const uint32_t* __restrict__ part, int32_t blocks, int32_t cols,
                                                               float* __restrict__ out) {
  const int cl = threadIdx.x & 3, y = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  uint32_t m = 0u;
  if (c < cols) {
    for (int b0 = y; b0 < blocks; b0 += 64 * 16) {
      uint32_t v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = b0 + 64 * u;
        v[u] = b < blocks ? part[(int64_t)b * cols + c] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) m = max(m, v[u]);
    }
  }
  __shared__ uint32_t sh[64][4];
  sh[y][cl] = m;
  __syncthreads();
  if (y == 0 && c < cols) {
#pragma unroll 8
    for (int k = 1; k < 64; ++k) m = max(m, sh[k][cl]);
    out[c] = __uint_as_float(m);
  }
}

// the same for a NARROW matrix (cols <= 16, any cols: the [V, L] bucket magnitudes of the gather -> one magnitude per edge type): a
// thread walks whole rows, one workgroup reduces in LDS, out[] (zeroed by the caller) takes one atomicMax per column and block
template <int MAXC>
__global__ __launch_bounds__(256) void col_absmax_narrow_kernel(const float* __restrict__ X, int64_t ldx, int32_t rows, int32_t cols,
                                                                float* __restrict__ out) {
  uint32_t m[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) m[c] = 0u;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256)
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < cols) m[c] = max(m[c], finite_mag_bits(X[r * ldx + c]));
  __shared__ uint32_t sh[256];
  for (int c = 0; c < cols; ++c) {
    uint32_t v = 0u;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) if (k == c) v = m[k];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0]) atomicMax(reinterpret_cast<unsigned int*>(out) + c, sh[0]);
    __syncthreads();
  }
}

// One workgroup = 32 rows x 64 columns of the OUTPUT matrix (4 waves; a wave: 8 rows x 8 groups of 8 k): every wave writes whole
// 128-byte lines of the limb blocks (8 consecutive rows x 16 bytes) and, untransposed, reads 256 consecutive bytes per row.
template <bool TRANSPOSE>
__device__ __forceinline__ void limb_split_tile(const float* __restrict__ X, int64_t ldx, int rows, int cols,
                                                uint16_t* __restrict__ out, int KT, int kt_off, int bx, int rb) {
  // output matrix: [R, C] = X (or X^T), written as k-tiles kt_off .. kt_off + C / 16 - 1 of a limb matrix with KT k-tiles per row block
  const int R = TRANSPOSE ? cols : rows, C = TRANSPOSE ? rows : cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = 8 * wave + (lane & 7);                // row inside the 32-row tile
  const int g8 = bx * 8 + (lane >> 3);                // group of 8 k
  // C need not be a multiple of 16: the last k-tile is filled up with zeros (a reduction length of 121 — the PPI head's input
  // gradient — or 50 becomes 128 / 64 against a left operand whose rows are zero-padded the same way)
  if (g8 * 8 >= ((C + 15) & ~15)) return;
  const int r = rb * 32 + i;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (r < R) {
    if constexpr (TRANSPOSE) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (8 * g8 + j < C) v[j] = X[(int64_t)(8 * g8 + j) * ldx + r];
    } else if (8 * g8 + 8 <= C && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + (int64_t)r * ldx + 8 * g8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(X + (int64_t)r * ldx + 8 * g8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (8 * g8 + j < C) v[j] = X[(int64_t)r * ldx + 8 * g8 + j];
    }
  }
  uint4 h, m, l;
  split8(v, h, m, l);
  const int kt = kt_off + (g8 >> 1), hh = g8 & 1;
  uint16_t* o = out + ((int64_t)rb * KT + kt) * 1536 + hh * 256 + i * 8;
  *reinterpret_cast<uint4*>(o) = h;
  *reinterpret_cast<uint4*>(o + 512) = m;
  *reinterpret_cast<uint4*>(o + 1024) = l;
}

template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void limb_split_kernel(const float* __restrict__ X, int64_t ldx, int rows, int cols,
                                                         uint16_t* __restrict__ out, int64_t x_stride, int64_t out_stride) {
  X += (int64_t)blockIdx.z * x_stride;                // batch: one matrix per blockIdx.z
  out += (int64_t)blockIdx.z * out_stride;
  limb_split_tile<TRANSPOSE>(X, ldx, rows, cols, out, ((TRANSPOSE ? rows : cols) + 15) / 16, 0, blockIdx.x, blockIdx.y);
}

// Several matrices in one launch (the weight operands of a training step, split once after the optimizer's update): item d owns
// the blocks [block_end[d-1], block_end[d]); an item may be a column range (k-tiles kt_off ..) of a wider limb matrix.
constexpr int SPLIT_MULTI_MAX = 24;
struct SplitItem {
  const float* X; int64_t ldx; uint16_t* out;
  int32_t rows, cols, transpose, kt_off, kt_total, gx, block_end;
  int32_t image;                       // (two-limb images: which scale / magnitude slot the matrix belongs to)
};
struct SplitMultiArgs { SplitItem it[SPLIT_MULTI_MAX]; int32_t n; };

__global__ __launch_bounds__(256) void limb_split_multi_kernel(const SplitMultiArgs a) {
  int d = 0;
  while (d + 1 < a.n && (int)blockIdx.x >= a.it[d].block_end) ++d;
  const SplitItem& it = a.it[d];
  const int local = (int)blockIdx.x - (d ? a.it[d - 1].block_end : 0);
  const int bx = local % it.gx, rb = local / it.gx;
  if (it.transpose) limb_split_tile<true>(it.X, it.ldx, it.rows, it.cols, it.out, it.kt_total, it.kt_off, bx, rb);
  else limb_split_tile<false>(it.X, it.ldx, it.rows, it.cols, it.out, it.kt_total, it.kt_off, bx, rb);
}

// ---- two fp16 limbs: the weight images (all matrices of an image share ONE scale: they are summed over in one product) -------------
// block = the 32 x 64 tile of the OUTPUT matrix limb_split_tile takes; first pass: the largest magnitude of the image (atomicMax on
// the bit pattern: magnitudes order like unsigned integers), second pass: the limbs of x * 2^j, j from that magnitude.
__device__ __forceinline__ void limb16_item_of_block(const SplitMultiArgs& a, int& d, int& bx, int& rb) {
  d = 0;
  while (d + 1 < a.n && (int)blockIdx.x >= a.it[d].block_end) ++d;
  const int local = (int)blockIdx.x - (d ? a.it[d - 1].block_end : 0);
  bx = local % a.it[d].gx; rb = local / a.it[d].gx;
}

template <bool TRANSPOSE>
__device__ __forceinline__ bool limb16_load8(const SplitItem& it, int bx, int rb, float (&v)[8], int& i, int& g8) {
  const int R = TRANSPOSE ? it.cols : it.rows, C = TRANSPOSE ? it.rows : it.cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i = 8 * wave + (lane & 7);
  g8 = bx * 8 + (lane >> 3);
  if (g8 * 8 >= C) return false;
  const int r = rb * 32 + i;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (r < R) {
    if constexpr (TRANSPOSE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = it.X[(int64_t)(8 * g8 + j) * it.ldx + r];
    } else {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(it.X + (int64_t)r * it.ldx + 8 * g8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(it.X + (int64_t)r * it.ldx + 8 * g8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    }
  }
  return true;
}

__global__ __launch_bounds__(256) void limb16_absmax_multi_kernel(const SplitMultiArgs a, float* __restrict__ wmax) {
  int d, bx, rb, i, g8;
  limb16_item_of_block(a, d, bx, rb);
  float v[8];
  const bool ok = a.it[d].transpose ? limb16_load8<true>(a.it[d], bx, rb, v, i, g8) : limb16_load8<false>(a.it[d], bx, rb, v, i, g8);
  float m = 0.f;
  if (ok)
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, finite_mag(v[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wave_max[4];                      // one atomic per block (every wave on one address measured 26 us for 18 matrices)
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(wmax + a.it[d].image), __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void limb16_split_multi_kernel(const SplitMultiArgs a, const float* __restrict__ wmax) {
  int d, bx, rb, i, g8;
  limb16_item_of_block(a, d, bx, rb);
  const SplitItem& it = a.it[d];
  float v[8];
  const bool ok = it.transpose ? limb16_load8<true>(it, bx, rb, v, i, g8) : limb16_load8<false>(it, bx, rb, v, i, g8);
  if (!ok) return;
  uint4 h, l;
  split8_16(v, limb16_scale(wmax[it.image]), h, l);
  const int kt = it.kt_off + (g8 >> 1), hh = g8 & 1;
  uint16_t* o = it.out + ((int64_t)rb * it.kt_total + kt) * 1024 + hh * 256 + i * 8;
  *reinterpret_cast<uint4*>(o) = h;
  *reinterpret_cast<uint4*>(o + 512) = l;
}

template <int T32, bool XF32, int ABL = 0, int NL = 3>
int launch_limb(const LimbArgs& a, hipStream_t st) {
  const int64_t logical = (int64_t)a.panels * a.chunks;
  limb_gemm_kernel<T32, XF32, ABL, NL><<<(unsigned)(8 * ((logical + 7) / 8)), 512, 0, st>>>(a);
  return launch_status();
}

template <bool XF32, int NL = 3>
int dispatch_limb(LimbArgs a, hipStream_t st) {
  a.chunks = a.N / 256;
#ifdef RELGNN_LIMB_TIMING
  a.timing = g_limb_timing;
#endif
  // panels: the fewest 32-row units per panel such that panels x chunks fills a whole number of rounds of the 256 CUs
  const int units = (a.M + 31) / 32;
  int want = 256 / a.chunks;
  if (want < 1) want = 1;
  int best_cap = 0, best_panels = 0;
  double best_cost = 1e30;
  for (int cap = 1; cap <= 5; ++cap) {
    int rounds = (units + want * cap - 1) / (want * cap);
    if (rounds < 1) rounds = 1;
    int panels = rounds * want;
    if (panels > units) panels = units;
    if ((units + panels - 1) / panels > cap) continue;
    const double cost = (double)((panels + want - 1) / want) * (cap + 0.35) + 1e-6 * panels;    // 0.35: prologue + epilogue
    if (cost < best_cost) { best_cost = cost; best_cap = cap; best_panels = panels; }
  }
  if (!best_cap) return RELGNN_EUNSUPPORTED;
  a.panels = best_panels; a.units_base = units / best_panels; a.units_rem = units % best_panels;
#ifdef RELGNN_LIMB_ABLATE
  if (best_cap == 5) {
    const char* e = getenv("RELGNN_LIMB_ABLATE");
    switch (e ? atoi(e) : 0) {
      case 1: return launch_limb<5, XF32, 1>(a, st);
      case 2: return launch_limb<5, XF32, 2>(a, st);
      case 3: return launch_limb<5, XF32, 3>(a, st);
      case 7: return launch_limb<5, XF32, 7>(a, st);
      default: break;
    }
  }
#endif
  switch (best_cap) {
    case 1: return launch_limb<1, XF32, 0, NL>(a, st);
    case 2: return launch_limb<2, XF32, 0, NL>(a, st);
    case 3: return launch_limb<3, XF32, 0, NL>(a, st);
    case 4: return launch_limb<4, XF32, 0, NL>(a, st);
    default: return launch_limb<5, XF32, 0, NL>(a, st);
  }
}

}  // namespace

extern "C" {

#ifdef RELGNN_LIMB_TIMING
void relgnn_limb_timing_buffer(unsigned long long* p) {
  g_limb_timing = p;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_limb_timing_dev), &p, sizeof(p));
}
#endif

int64_t relgnn_limb_elements(int64_t rows, int64_t cols) { return ((rows + 31) / 32) * (cols / 16) * 1536; }

int relgnn_limb_split_batch_f32(const float* X, int64_t ldx, int64_t x_batch_stride, int32_t rows, int32_t cols, int32_t transpose,
                                int32_t batch, uint16_t* out, void* stream) {
  if (rows < 0 || cols < 0 || ldx < cols || batch < 0) return RELGNN_EINVAL;
  if (rows == 0 || cols == 0 || batch == 0) return RELGNN_OK;
  if (!X || !out) return RELGNN_EINVAL;
  const int R = transpose ? cols : rows, C = transpose ? rows : cols;
  if (C % 16 != 0 || !aligned16(out) || (!transpose && (!aligned16(X) || ldx % 4 || x_batch_stride % 4)) || batch > 65535)
    return RELGNN_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 31) / 32), (unsigned)batch);
  const int64_t os = relgnn_limb_elements(R, C);
  if (transpose) limb_split_kernel<true><<<grid, 256, 0, st>>>(X, ldx, rows, cols, out, x_batch_stride, os);
  else limb_split_kernel<false><<<grid, 256, 0, st>>>(X, ldx, rows, cols, out, x_batch_stride, os);
  return launch_status();
}

int relgnn_limb_split_f32(const float* X, int64_t ldx, int32_t rows, int32_t cols, int32_t transpose, uint16_t* out, void* stream) {
  return relgnn_limb_split_batch_f32(X, ldx, 0, rows, cols, transpose, 1, out, stream);
}

int relgnn_limb_split_multi_f32(int32_t n, const float* const* X, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                                const int32_t* transpose, uint16_t* const* out, const int32_t* kt_offset, const int32_t* kt_total,
                                void* stream) {
  if (n < 0 || (n > 0 && (!X || !ldx || !rows || !cols || !transpose || !out || !kt_offset || !kt_total))) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  for (int32_t first = 0; first < n; first += SPLIT_MULTI_MAX) {
    SplitMultiArgs a{};
    int blocks = 0;
    a.n = 0;
    for (int32_t d = first; d < n && a.n < SPLIT_MULTI_MAX; ++d) {
      const int R = transpose[d] ? cols[d] : rows[d], C = transpose[d] ? rows[d] : cols[d];
      if (rows[d] < 0 || cols[d] < 0 || kt_offset[d] < 0 || kt_offset[d] + (C + 15) / 16 > kt_total[d]) return RELGNN_EINVAL;
      if (R == 0 || C == 0) continue;
      if (!X[d] || !out[d]) return RELGNN_EINVAL;
      if (!aligned16(out[d]) || ldx[d] < cols[d]) return RELGNN_EUNSUPPORTED;     // (rows of any alignment: read element-wise then)
      SplitItem& it = a.it[a.n++];
      it.X = X[d]; it.ldx = ldx[d]; it.out = out[d]; it.rows = rows[d]; it.cols = cols[d]; it.transpose = transpose[d];
      it.kt_off = kt_offset[d]; it.kt_total = kt_total[d]; it.gx = (C + 63) / 64;
      blocks += it.gx * ((R + 31) / 32);
      it.block_end = blocks;
    }
    if (!a.n) continue;
    limb_split_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(a);
    const int rc = launch_status();
    if (rc != RELGNN_OK) return rc;
  }
  return RELGNN_OK;
}

static int limb_common_checks(int32_t act, const void* A, const void* B, const float* bias, const void* zeros, float* C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K) {
  if (M < 0 || N < 0 || K < 0 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C || !zeros) return RELGNN_EINVAL;
  if (K == 0 || K % BK != 0 || N % 256 != 0) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || !aligned16(zeros) || (bias && !aligned16(bias)) || ldc % 4 || ldc < N)
    return RELGNN_EUNSUPPORTED;
  return -1;      // go on
}

int relgnn_limb_gemm_f32(int32_t act, const uint16_t* A, const uint16_t* B, const float* bias, const void* zeros, float* C,
                         int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  LimbArgs a{};
  a.A = A; a.B = B; a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.act = act;
  return dispatch_limb<false>(a, as_stream(stream));
}

int relgnn_limb_gemm_xf32(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, const void* zeros,
                          float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  if (lda % 4 || lda < K) return RELGNN_EUNSUPPORTED;
  LimbArgs a{};
  a.Ax = A; a.lda = lda; a.B = B; a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M;
  a.N = N; a.K = K; a.act = act;
  return dispatch_limb<true>(a, as_stream(stream));
}

static int dact_checks(int32_t dact, const float* Y, int64_t ldy, int32_t N) {
  if (!Y) return -1;
  if (dact < RELGNN_ACT_LINEAR || dact > RELGNN_ACT_SELU) return dact == RELGNN_ACT_GELU ? RELGNN_EUNSUPPORTED : RELGNN_EINVAL;
  if (!aligned16(Y) || ldy % 4 || ldy < N) return RELGNN_EUNSUPPORTED;
  return -1;
}

// relgnn_limb_gemm_xf32 with the activation gradient of the layer BELOW in its epilogue: C = act(bias + A @ B^T) * dact'(Y), the
// derivative of activation `dact` evaluated from its output Y [M, N] (row stride ldy): what TF's ReluGrad / TanhGrad compute in an op
// of their own behind the input-gradient MatMul (the backward of gnns/rgcn.py:114 / models/sparse_graph_model.py:198-200 meeting
// the backward of the next layer's Dense).  Same bits as relgnn_limb_gemm_xf32 followed by relgnn_act_bwd_from_output.
int relgnn_limb_gemm_xf32_dact(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, const void* zeros,
                               int32_t dact, const float* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                               void* stream) {
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  if (lda % 4 || lda < K) return RELGNN_EUNSUPPORTED;
  const int rd = dact_checks(dact, Y, ldy, N);
  if (rd >= 0) return rd;
  LimbArgs a{};
  a.Ax = A; a.lda = lda; a.B = B; a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M;
  a.N = N; a.K = K; a.act = act; a.dy = Y; a.ldy = ldy; a.dact = dact;
  return dispatch_limb<true>(a, as_stream(stream));
}

// ---- two fp16 limbs -------------------------------------------------------------------------------------------------------------
int64_t relgnn_limb16_elements(int64_t rows, int64_t cols) { return ((rows + 31) / 32) * (cols / 16) * 1024; }

int relgnn_limb16_split_multi_f32(int32_t n, const float* const* X, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                                  const int32_t* transpose, uint16_t* const* out, const int32_t* kt_offset, const int32_t* kt_total,
                                  const int32_t* image, int32_t n_images, float* wmax, void* stream) {
  if (n < 0 || n_images < 0 || (n_images > 0 && !wmax) ||
      (n > 0 && (!X || !ldx || !rows || !cols || !transpose || !out || !kt_offset || !kt_total || !image)))
    return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  for (int32_t d = 0; d < n; ++d) {
    const int C = transpose[d] ? rows[d] : cols[d];
    if (rows[d] < 0 || cols[d] < 0 || C % 16 != 0 || kt_offset[d] < 0 || kt_offset[d] + C / 16 > kt_total[d] || image[d] < 0 ||
        image[d] >= n_images)
      return RELGNN_EINVAL;
    if (rows[d] == 0 || cols[d] == 0) continue;
    if (!X[d] || !out[d]) return RELGNN_EINVAL;
    if (!aligned16(out[d]) || (!transpose[d] && (!aligned16(X[d]) || ldx[d] % 4)) || ldx[d] < cols[d]) return RELGNN_EUNSUPPORTED;
  }
  if (n_images > 0 && hipMemsetAsync(wmax, 0, sizeof(float) * n_images, st) != hipSuccess) return RELGNN_EHIP;
  // pass 0: every image's largest magnitude; pass 1: the limbs (both in launches of up to SPLIT_MULTI_MAX matrices)
  for (int pass = 0; pass < 2; ++pass)
    for (int32_t first = 0; first < n; first += SPLIT_MULTI_MAX) {
      SplitMultiArgs a{};
      int blocks = 0;
      for (int32_t d = first; d < n && d < first + SPLIT_MULTI_MAX; ++d) {
        const int R = transpose[d] ? cols[d] : rows[d], C = transpose[d] ? rows[d] : cols[d];
        if (R == 0 || C == 0) continue;
        SplitItem& it = a.it[a.n++];
        it.X = X[d]; it.ldx = ldx[d]; it.out = out[d]; it.rows = rows[d]; it.cols = cols[d]; it.transpose = transpose[d];
        it.kt_off = kt_offset[d]; it.kt_total = kt_total[d]; it.gx = (C + 63) / 64; it.image = image[d];
        blocks += it.gx * ((R + 31) / 32);
        it.block_end = blocks;
      }
      if (!a.n) continue;
      if (pass == 0) limb16_absmax_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(a, wmax);
      else limb16_split_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(a, wmax);
      const int rc = launch_status();
      if (rc != RELGNN_OK) return rc;
    }
  return RELGNN_OK;
}

int relgnn_limb16_gemm_xf32(int32_t act, const float* A, int64_t lda, const float* xmax, int32_t xgroups, const uint16_t* B,
                            const float* wmax, const float* bias, const void* zeros, float* C, int64_t ldc, int32_t M, int32_t N,
                            int32_t K, void* stream) {
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  if (lda % 4 || lda < K || !xmax || !wmax || xgroups < 1) return RELGNN_EUNSUPPORTED;
  LimbArgs a{};
  a.Ax = A; a.lda = lda; a.B = B; a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M;
  a.N = N; a.K = K; a.act = act; a.xmax = xmax; a.xgroups = xgroups; a.wmax = wmax;
  return dispatch_limb<true, 2>(a, as_stream(stream));
}

int relgnn_limb16_gemm_xf32_dact(int32_t act, const float* A, int64_t lda, const float* xmax, int32_t xgroups, const uint16_t* B,
                                 const float* wmax, const float* bias, const void* zeros, int32_t dact, const float* Y, int64_t ldy,
                                 float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  if (lda % 4 || lda < K || !xmax || !wmax || xgroups < 1) return RELGNN_EUNSUPPORTED;
  const int rd = dact_checks(dact, Y, ldy, N);
  if (rd >= 0) return rd;
  LimbArgs a{};
  a.Ax = A; a.lda = lda; a.B = B; a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M;
  a.N = N; a.K = K; a.act = act; a.xmax = xmax; a.xgroups = xgroups; a.wmax = wmax; a.dy = Y; a.ldy = ldy; a.dact = dact;
  return dispatch_limb<true, 2>(a, as_stream(stream));
}

// rows per chunk (% 32 == 0) and number of chunks for the V - V % 32 rows the kernel takes
static void limb_tn_geometry(int32_t V, int32_t J, int32_t C, int* t32, int* rows, int* Z) {
  const int units = J / 32;
  *t32 = units % 4 == 0 ? 4 : (units % 2 == 0 ? 2 : 1);
  const int per_z = (units / *t32) * (C / 256);
  const int vmain = V - V % 32;
  int z0 = 256 / per_z;
  if (z0 < 1) z0 = 1;
  const int max_z = (vmain + 255) / 256;               // a chunk is at least 256 rows
  if (z0 > max_z) z0 = max_z;
  if (z0 < 1) z0 = 1;
  *rows = ((vmain + z0 - 1) / z0 + 31) / 32 * 32;
  *Z = *rows > 0 ? (vmain + *rows - 1) / *rows : 0;
}

int64_t relgnn_limb_gemm_tn_chunks(int32_t V, int32_t J, int32_t C) {
  if (V < 32 || J <= 0 || C <= 0 || J % 32 != 0 || C % 256 != 0) return 0;
  int t32, rows, Z;
  limb_tn_geometry(V, J, C, &t32, &rows, &Z);
  return Z;
}

// P [chunks][J][C] = per-chunk partial products of A^T G over the first V - V % 32 rows (A [V, J], G [V, C], fp32 row-major;
// J % 32 == 0, C % 256 == 0, V >= 32); chunks = relgnn_limb_gemm_tn_chunks(V, J, C).  The caller sums the slabs in order and adds
// the last V % 32 rows' product: relgnn_sum_slabs_tail_f32 does both in one pass.
static int limb_tn_any(const float* A, int64_t lda, const float* G, int64_t ldg, const float* amax, const float* gmax,
                       int32_t a_cols, int32_t g_cols, float* P, int32_t V, int32_t J, int32_t C, void* stream);

int relgnn_limb_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, float* P, int32_t V, int32_t J, int32_t C,
                            void* stream) {
  return limb_tn_any(A, lda, G, ldg, nullptr, nullptr, 1, 1, P, V, J, C, stream);
}

int relgnn_limb16_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* amax, int32_t a_cols_per_scale,
                              const float* gmax, int32_t g_cols_per_scale, float* P, int32_t V, int32_t J, int32_t C, void* stream) {
  if (!amax || !gmax || a_cols_per_scale < 1 || g_cols_per_scale < 1) return RELGNN_EINVAL;
  return limb_tn_any(A, lda, G, ldg, amax, gmax, a_cols_per_scale, g_cols_per_scale, P, V, J, C, stream);
}

static void col_absmax_geometry(int32_t rows, int32_t cols, int* gx, int* per, int* nb) {
  *gx = (cols / 4 + 63) / 64;
  *per = 64;                                           // rows per workgroup: ~1024 workgroups at most (4 per CU)
  while ((int64_t)*gx * ((rows + *per - 1) / *per) > 1024) *per *= 2;
  *nb = (rows + *per - 1) / *per;
}

int64_t relgnn_col_absmax_workspace_bytes(int32_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 16 || cols % 4 != 0) return 0;
  int gx, per, nb;
  col_absmax_geometry(rows, cols, &gx, &per, &nb);
  return (int64_t)nb * cols * 4;
}

int relgnn_col_absmax_f32(const float* X, int64_t ldx, int32_t rows, int32_t cols, float* out, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  if (rows < 0 || cols < 0 || (cols > 0 && !out)) return RELGNN_EINVAL;
  if (cols == 0) return RELGNN_OK;
  hipStream_t st = as_stream(stream);
  if (rows == 0 || cols <= 16) {
    if (hipMemsetAsync(out, 0, sizeof(float) * cols, st) != hipSuccess) return RELGNN_EHIP;
    if (rows == 0) return RELGNN_OK;
    if (!X || ldx < cols) return RELGNN_EINVAL;
    int64_t blocks = ((int64_t)rows + 255) / 256;
    if (blocks > 64) blocks = 64;
    if (cols <= 4) col_absmax_narrow_kernel<4><<<(unsigned)blocks, 256, 0, st>>>(X, ldx, rows, cols, out);
    else col_absmax_narrow_kernel<16><<<(unsigned)blocks, 256, 0, st>>>(X, ldx, rows, cols, out);
    return launch_status();
  }
  if (!X) return RELGNN_EINVAL;
  if (cols % 4 != 0 || ldx % 4 != 0 || ldx < cols || !aligned16(X)) return RELGNN_EUNSUPPORTED;
  int gx, per, nb;
  col_absmax_geometry(rows, cols, &gx, &per, &nb);
  if (!workspace || workspace_bytes < (int64_t)nb * cols * 4 || !aligned16(workspace)) return RELGNN_EINVAL;
  dim3 grid((unsigned)gx, (unsigned)nb);
  col_absmax_part_kernel<<<grid, 256, 0, st>>>(X, ldx, rows, cols, per, static_cast<uint32_t*>(workspace));
  int rc = launch_status();
  if (rc != RELGNN_OK) return rc;
  col_absmax_final_kernel<<<(unsigned)((cols + 3) / 4), 256, 0, st>>>(static_cast<const uint32_t*>(workspace), nb, cols, out);
  return launch_status();
}

int relgnn_absmax_f32(const float* x, int64_t n, float* out, void* stream) {
  if (n < 0 || !out) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return RELGNN_EHIP;
  if (n == 0) return RELGNN_OK;
  if (!x) return RELGNN_EINVAL;
  if (!aligned16(x)) return RELGNN_EUNSUPPORTED;
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  absmax_kernel<<<(unsigned)blocks, 256, 0, st>>>(x, n4, n, out);
  return launch_status();
}

static int limb_tn_any(const float* A, int64_t lda, const float* G, int64_t ldg, const float* amax, const float* gmax,
                       int32_t a_cols, int32_t g_cols, float* P, int32_t V, int32_t J, int32_t C, void* stream) {
  if (V < 0 || J < 0 || C < 0) return RELGNN_EINVAL;
  if (J == 0 || C == 0) return RELGNN_OK;
  if (V < 32) return RELGNN_EUNSUPPORTED;
  if (!A || !G || !P) return RELGNN_EINVAL;
  if (J % 32 != 0 || C % 256 != 0 || lda % 4 || ldg % 4 || lda < J || ldg < C || !aligned16(A) || !aligned16(G) || !aligned16(P))
    return RELGNN_EUNSUPPORTED;
  LimbTnArgs a{};
  a.A = A; a.lda = lda; a.G = G; a.ldg = ldg; a.P = P; a.V = V - V % 32; a.J = J; a.C = C; a.amax = amax; a.gmax = gmax;
  a.a_cols = a_cols; a.g_cols = g_cols;
  int t32, rows, Z;
  limb_tn_geometry(V, J, C, &t32, &rows, &Z);
  a.panels = (J / 32) / t32; a.chunks = C / 256; a.rows_per_chunk = rows; a.Z = Z;
  const int64_t logical = (int64_t)a.panels * a.chunks * a.Z;
  const unsigned grid = (unsigned)(8 * ((logical + 7) / 8));
  hipStream_t st = as_stream(stream);
  if (amax) {
    switch (t32) {
      case 4: limb_gemm_tn_kernel<4, 2><<<grid, 512, 0, st>>>(a); break;
      case 2: limb_gemm_tn_kernel<2, 2><<<grid, 512, 0, st>>>(a); break;
      default: limb_gemm_tn_kernel<1, 2><<<grid, 512, 0, st>>>(a); break;
    }
    return launch_status();
  }
  switch (t32) {
    case 4: limb_gemm_tn_kernel<4><<<grid, 512, 0, st>>>(a); break;
    case 2: limb_gemm_tn_kernel<2><<<grid, 512, 0, st>>>(a); break;
    default: limb_gemm_tn_kernel<1><<<grid, 512, 0, st>>>(a); break;
  }
  return launch_status();
}

// Typed weight-gradient partials of many-type graphs (ops.typed_linear's backward; gnns/gnn_film.py:92-106, gnns/rgcn.py:96-98 per edge
// type): the P rows of a compact pair table come in tiles of `rows_per_tile` rows that belong to ONE edge type each; tile z's partial is
//   part[z] = A[a_rows[z * rows_per_tile ..]]^T @ G[z * rows_per_tile ..]          ([J, C] each; a_rows < 0: a row of zeros)
// from the exact three-bf16-limb split of both operands (six MFMA products, fp32 accumulation), the gather and both splits in flight.
// The caller sums the tiles of a type in tile order (a segment reduction over [tiles, J * C]).  J % 64 == 0, J <= 128 per panel
// geometry (J = 64: T32 = 2, J % 128 == 0: T32 = 4), C % 128 == 0, P % rows_per_tile == 0, rows_per_tile % 32 == 0.
int relgnn_limb_gemm_tn_tiles_f32(const float* A, int64_t lda, const int32_t* a_rows, const float* G, int64_t ldg,
                                  const void* zeros, float* part, int32_t P, int32_t rows_per_tile, int32_t J, int32_t C,
                                  void* stream) {
  if (P < 0 || J < 0 || C < 0 || rows_per_tile <= 0) return RELGNN_EINVAL;
  if (P == 0 || J == 0 || C == 0) return RELGNN_OK;
  if (!A || !a_rows || !G || !zeros || !part) return RELGNN_EINVAL;
  if (rows_per_tile % 32 != 0 || P % rows_per_tile != 0 || J % 64 != 0 || C % 128 != 0 || lda % 4 || ldg % 4 || lda < J || ldg < C ||
      !aligned16(A) || !aligned16(G) || !aligned16(part) || !aligned16(a_rows) || !aligned16(zeros))
    return RELGNN_EUNSUPPORTED;
  LimbTnArgs a{};
  a.A = A; a.lda = lda; a.G = G; a.ldg = ldg; a.P = part; a.V = P; a.J = J; a.C = C; a.a_cols = 1; a.g_cols = 1;
  a.a_rows = a_rows; a.zeros = static_cast<const float*>(zeros);
  const bool wide = C % 256 == 0;
  const int t32 = J % 128 == 0 ? 4 : 2;
  a.panels = (J / 32) / t32; a.chunks = C / (wide ? 256 : 128); a.rows_per_chunk = rows_per_tile; a.Z = P / rows_per_tile;
  const int64_t logical = (int64_t)a.panels * a.chunks * a.Z;
  const unsigned grid = (unsigned)(8 * ((logical + 7) / 8));
  hipStream_t st = as_stream(stream);
  if (wide) {
    if (t32 == 4) limb_gemm_tn_kernel<4, 3, 256, true><<<grid, 512, 0, st>>>(a);
    else limb_gemm_tn_kernel<2, 3, 256, true><<<grid, 512, 0, st>>>(a);
  } else {
    if (t32 == 4) limb_gemm_tn_kernel<4, 3, 128, true><<<grid, 512, 0, st>>>(a);
    else limb_gemm_tn_kernel<2, 3, 128, true><<<grid, 512, 0, st>>>(a);
  }
  return launch_status();
}

// launch of the 128-column panel kernels over limb images that exist (relgnn_limb_dense_sel_f32 splits first; relgnn_limb_gemm_sel_xf32
// takes them from the caller)
static int limb_sel_launch(int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const uint16_t* limbs, int64_t per,
                           const int32_t* b_select, int32_t rows_per_select, const float* bias, const void* zeros, float* C,
                           int64_t ldc, int32_t M, int32_t N, int32_t n_valid, int32_t K, bool cut, void* stream) {
  LimbSelArgs a{};
  a.n_valid = n_valid;
  a.Ax = A; a.lda = lda; a.rows = a_rows; a.B = limbs; a.b_select = b_select; a.rows_per_select = rows_per_select; a.b_stride = per;
  a.bias = bias; a.zeros = static_cast<const uint16_t*>(zeros); a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.act = act;
  a.chunks = N / 128;
  // K = 128, tall: persistent workgroups with the weights resident in LDS (limb_gemm_tile_kernel), one per CU
  static const bool tile_form = []() { const char* e = getenv("RELGNN_LIMB_TILE"); return !(e && e[0] == '0'); }();
  // (typed products — large by construction; a plain product needs ~4 panels per workgroup before this form pays: measured)
  if (tile_form && !cut && K == 128 && M >= 128 * 256 && (b_select || (int64_t)M * a.chunks >= (int64_t)128 * 256 * 4)) {
    if (!b_select) a.rows_per_select = 128;
    a.panels = 256 / a.chunks > 0 ? 256 / a.chunks : 1;          // workgroups per column chunk
    const int64_t logical = (int64_t)a.panels * a.chunks;
    limb_gemm_tile_kernel<<<(unsigned)(8 * ((logical + 7) / 8)), 512, 0, as_stream(stream)>>>(a);
    return launch_status();
  }
  a.panels = (M + 127) / 128;
  const int64_t logical = (int64_t)a.panels * a.chunks;
  limb_gemm_sel_kernel<<<(unsigned)(8 * ((logical + 7) / 8)), 512, 0, as_stream(stream)>>>(a);
  return launch_status();
}

// 128 x 128 panels with gathered rows and per-panel weights (limb_gemm_sel_kernel): the weights are `num_b` fp32 matrices at
// B + i * b_batch_stride (RELGNN_GEMM_NN: [K, N] each; RELGNN_GEMM_NT: [N, K] each), all split into limb_ws first (one launch).
int relgnn_limb_dense_sel_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const float* B,
                              int64_t ldb, int32_t num_b, int64_t b_batch_stride, const int32_t* b_select, int32_t rows_per_select,
                              const float* bias, const void* zeros, uint16_t* limb_ws, int64_t limb_ws_elements, float* C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  if (layout != RELGNN_GEMM_NN && layout != RELGNN_GEMM_NT) return RELGNN_EINVAL;
  if (M < 0 || N < 0 || K < 0 || num_b < 1 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B || !C || !limb_ws || !zeros) return RELGNN_EINVAL;
  if (K == 0 || K % BK != 0) return RELGNN_EUNSUPPORTED;
  // N % 128 != 0 (one weight matrix only): the last 128-column chunk is cut at N when it is stored; C rows need no alignment then
  const bool cut = N % 128 != 0 || ldc % 4 != 0;
  const int32_t n_valid = N;
  N = (N + 127) / 128 * 128;
  if (cut && (b_select || num_b != 1)) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B) || !aligned16(zeros) || ldc < n_valid || lda % 4 || lda < K) return RELGNN_EUNSUPPORTED;
  if (!cut && (!aligned16(C) || (bias && !aligned16(bias)))) return RELGNN_EUNSUPPORTED;
  if (b_select && (rows_per_select <= 0 || rows_per_select % 128 != 0)) return RELGNN_EINVAL;
  if (!b_select && num_b != 1) return RELGNN_EINVAL;
  const int64_t per = relgnn_limb_elements(N, K);      // (row blocks past ceil(n_valid / 32) are never written: they only feed columns that are not stored)
  if (limb_ws_elements < per * num_b) return RELGNN_EINVAL;
  const int sp = layout == RELGNN_GEMM_NN ? relgnn_limb_split_batch_f32(B, ldb, b_batch_stride, K, n_valid, 1, num_b, limb_ws, stream)
                                          : relgnn_limb_split_batch_f32(B, ldb, b_batch_stride, n_valid, K, 0, num_b, limb_ws, stream);
  if (sp != RELGNN_OK) return sp;
  return limb_sel_launch(act, A, lda, a_rows, limb_ws, per, b_select, rows_per_select, bias, zeros, C, ldc, M, N, n_valid, K, cut, stream);
}

// The same product with the weights ALREADY split: B_limbs holds num_b limb images of [N, K] operands one behind the other
// (relgnn_limb_elements(N, K) elements each; relgnn_limb_split_multi_f32 / _batch_f32 write them) — the form the package uses since
// round 6: the images of a step's weights are split once per optimizer step, not in front of every product.
int relgnn_limb_gemm_sel_xf32(int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const uint16_t* B_limbs, int32_t num_b,
                              const int32_t* b_select, int32_t rows_per_select, const float* bias, const void* zeros, float* C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  if (M < 0 || N < 0 || K < 0 || num_b < 1 || act < RELGNN_ACT_LINEAR || act > RELGNN_ACT_GELU) return RELGNN_EINVAL;
  if (M == 0 || N == 0) return RELGNN_OK;
  if (!A || !B_limbs || !C || !zeros) return RELGNN_EINVAL;
  if (K == 0 || K % BK != 0 || N % 128 != 0 || ldc % 4 != 0) return RELGNN_EUNSUPPORTED;
  if (!aligned16(A) || !aligned16(B_limbs) || !aligned16(zeros) || !aligned16(C) || (bias && !aligned16(bias)) || ldc < N || lda % 4 ||
      lda < K)
    return RELGNN_EUNSUPPORTED;
  if (b_select && (rows_per_select <= 0 || rows_per_select % 128 != 0)) return RELGNN_EINVAL;
  if (!b_select && num_b != 1) return RELGNN_EINVAL;
  return limb_sel_launch(act, A, lda, a_rows, B_limbs, relgnn_limb_elements(N, K), b_select, rows_per_select, bias, zeros, C, ldc, M, N,
                         N, K, false, stream);
}

// The Dense product as the path calls it: fp32 activations x fp32 weights.  The weights (N x K elements, a few hundred KB) are
// split into limb_ws first — one more ~3 us kernel on the same stream, no host round trip — then the product runs with the left
// operand split in flight.  layout NN: B is [K, N] (tf.layers.dense kernels as stored); NT: B is [N, K] (the same kernel for the
// input gradient).
int relgnn_limb_dense_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                          const void* zeros, uint16_t* limb_ws, int64_t limb_ws_elements, float* C, int64_t ldc, int32_t M,
                          int32_t N, int32_t K, void* stream) {
  if (layout != RELGNN_GEMM_NN && layout != RELGNN_GEMM_NT) return RELGNN_EINVAL;
  const int rc = limb_common_checks(act, A, B, bias, zeros, C, ldc, M, N, K);
  if (rc >= 0) return rc;
  if (!limb_ws || limb_ws_elements < relgnn_limb_elements(N, K)) return RELGNN_EINVAL;
  if (lda % 4 || lda < K) return RELGNN_EUNSUPPORTED;
  const int s = layout == RELGNN_GEMM_NN ? relgnn_limb_split_f32(B, ldb, K, N, 1, limb_ws, stream)
                                         : relgnn_limb_split_f32(B, ldb, N, K, 0, limb_ws, stream);
  if (s != RELGNN_OK) return s;
  return relgnn_limb_gemm_xf32(act, A, lda, limb_ws, bias, zeros, C, ldc, M, N, K, stream);
}

}  // extern "C"
