// Hand-over between wave roles of one workgroup through monotonic counters in LDS (rgcn_fused.hip, limb_gemm_pc.hip): the pieces both
// kernels share.  gfx950 only.
#pragma once
#include "common.h"

namespace relgnn {

constexpr int HANDOVER_SPIN_LIMIT = 1 << 22;     // a poll that takes this long is a bug: give up, flag it, finish with wrong numbers

// a relaxed workgroup-scope atomic load: ds_read_b32 (a volatile load becomes a flat load behind vmcnt(0) lgkmcnt(0))
__device__ __forceinline__ int handover_counter(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void handover_fence() { asm volatile("" ::: "memory"); }

// The caller's status block (device memory, int32[2], may be null = nothing is reported; include/relgnn.h RELGNN_HANDOVER_*):
//   [0]  give-up bits, OR-ed in by a kernel whose poll ran out: bit 0 / 1 a matrix / gather wave of rgcn_fused_kernel,
//        bit 2 / 3 a matrix / producer wave of limb_gemm_pc_kernel.  The caller zeroes it and reads it where it syncs anyway.
//   [1]  the poll bound, 0 = HANDOVER_SPIN_LIMIT (tests write 1 to make every poll give up at once).
__device__ __forceinline__ int handover_limit(const int32_t* status) {
  const int v = status ? __builtin_amdgcn_readfirstlane(status[1]) : 0;
  return v > 0 ? v : HANDOVER_SPIN_LIMIT;
}

}  // namespace relgnn
