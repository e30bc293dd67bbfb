// fp32 -> three bf16 limbs, hi + mid + lo == x bit for bit (shared by limb_gemm.hip and rgcn_fused.hip): gfx950 only.
#pragma once
#include "common.h"

namespace relgnn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {      // two fp32 -> two bf16 (round to nearest even), packed
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));      // v_cvt_pk_bf16_f32
}
// (x0, x1) -> the packed limbs; each subtraction is exact, so hi + mid + lo == x bit for bit
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf16(x0, x1);
  const f32x2 r = f32x2{x0, x1} - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xFFFF0000u)};     // (v_pk_add_f32)
  m = cvt_pk_bf16(r[0], r[1]);
  const f32x2 t = r - f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xFFFF0000u)};
  l = cvt_pk_bf16(t[0], t[1]);
}
// |x| >= 0x7F7F8000 (3.3962e38 .. FLT_MAX, and inf) rounds to a bf16 INFINITY: hi = +-inf, mid = x - hi = -+inf, lo = NaN — and the
// float32 lowest that tf.unsorted_segment_max writes for an empty segment (utils/utils.py:23-33) is such a value: it came
// out of a Dense product as NaN where fp32 arithmetic gives a finite number.  For those values hi saturates at the largest finite
// bf16 (0x7F7F); the remainder x - hi is then at most 16 significant bits at 2^104 .. 2^120 and mid, lo hold it exactly as before,
// so hi + mid + lo == x still holds bit for bit for EVERY finite x.  inf / NaN inputs keep producing non-finite limbs (inf: hi = inf
// -> mid = NaN), i.e. a non-finite output row, as the fp32 product does.
// Detection costs four v_max3_f32 per eight values (written as asm: hipcc's fmaxf() first canonicalises every operand, one more
// VALU instruction per value) and one compare; the saturating split itself sits behind a branch that normal data never takes.
// (A signalling NaN among the eight makes the maximum a NaN and hides a huge finite neighbour from the compare — the eight values
// share one output row, which that NaN makes non-finite anyway.)
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ bool bf16_hi_overflows(const float* v) {       // any of v[0..7] rounds to a bf16 infinity
  float m = max3_abs(v[0], v[1], v[2]);
  m = max3_abs(m, v[3], v[4]);
  m = max3_abs(m, v[5], v[6]);
  m = max3_abs(m, v[7], v[7]);
  return m >= __uint_as_float(0x7F7F8000u);
}
__device__ __forceinline__ uint32_t bf16_sat_bits(float x) {              // bf16(x), round to nearest even; a finite x stays finite
  uint32_t h = cvt_pk_bf16(x, x) & 0xFFFFu;
  if ((h & 0x7FFFu) == 0x7F80u && (__float_as_uint(x) & 0x7FFFFFFFu) < 0x7F800000u) h -= 1u;      // 0x7F80 -> 0x7F7F, sign kept
  return h;
}
__device__ __forceinline__ void split_pair_sat(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t h0 = bf16_sat_bits(x0), h1 = bf16_sat_bits(x1);
  h = h0 | (h1 << 16);
  const float r0 = x0 - __uint_as_float(h0 << 16), r1 = x1 - __uint_as_float(h1 << 16);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ void split8(const float* v, uint4& h, uint4& m, uint4& l) {
  split_pair(v[0], v[1], h.x, m.x, l.x);
  split_pair(v[2], v[3], h.y, m.y, l.y);
  split_pair(v[4], v[5], h.z, m.z, l.z);
  split_pair(v[6], v[7], h.w, m.w, l.w);
  if (__builtin_expect(bf16_hi_overflows(v), 0)) {
    split_pair_sat(v[0], v[1], h.x, m.x, l.x);
    split_pair_sat(v[2], v[3], h.y, m.y, l.y);
    split_pair_sat(v[4], v[5], h.z, m.z, l.z);
    split_pair_sat(v[6], v[7], h.w, m.w, l.w);
  }
}

}  // namespace relgnn
