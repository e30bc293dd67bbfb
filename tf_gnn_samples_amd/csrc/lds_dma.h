// Direct-to-LDS staging helpers shared by the MFMA GEMM kernels (panel_gemm.hip, limb_gemm.hip): gfx950 only.
#pragma once
#include "common.h"

namespace relgnn {

// 16 bytes per lane, global -> LDS without passing through registers (global_load_lds_dwordx4).  The destination is
// wave-uniform base + 16 B * lane: any swizzle of the LDS image has to be expressed through the SOURCE address.
__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                       reinterpret_cast<uintptr_t>(src)),
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// lgkmcnt(0) through the builtin (gfx9 encoding: vmcnt 63 = bits [15:14|3:0], expcnt 7 = bits [6:4], lgkmcnt = bits [11:8]):
// hipcc's own wait insertion does not see an inline-asm wait and would add a conservative lgkmcnt(0) in front of the MFMA
// block — after the NEXT tile's fragment reads have been issued, which is exactly the overlap these loops exist for.
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

__device__ __forceinline__ float act_rt(int act, float x) {
  switch (act) {
    case RELGNN_ACT_TANH: return act_fwd<RELGNN_ACT_TANH>(x);
    case RELGNN_ACT_RELU: return act_fwd<RELGNN_ACT_RELU>(x);
    case RELGNN_ACT_LEAKY_RELU: return act_fwd<RELGNN_ACT_LEAKY_RELU>(x);
    case RELGNN_ACT_ELU: return act_fwd<RELGNN_ACT_ELU>(x);
    case RELGNN_ACT_SELU: return act_fwd<RELGNN_ACT_SELU>(x);
    case RELGNN_ACT_GELU: return act_fwd<RELGNN_ACT_GELU>(x);
    default: return x;
  }
}

}  // namespace relgnn
