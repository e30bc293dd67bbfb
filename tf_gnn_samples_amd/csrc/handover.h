// Hand-over between wave roles of one workgroup through monotonic counters in LDS (rgcn_fused.hip, limb_gemm_pc.hip): the pieces both
// kernels share.  gfx950 only.
#pragma once
#include "common.h"

namespace relgnn {

constexpr int HANDOVER_SPIN_LIMIT = 1 << 22;     // a poll that takes this long is a bug: give up, flag it, finish with wrong numbers

// a relaxed workgroup-scope atomic load: ds_read_b32 (a volatile load becomes a flat load behind vmcnt(0) lgkmcnt(0))
__device__ __forceinline__ int handover_counter(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void handover_fence() { asm volatile("" ::: "memory"); }

// The device word the kernels OR their give-up bits into, allocated on first use.  nullptr (the kernels then report nothing): the
// allocation failed, or the first use falls into a stream capture (hipMalloc is not capturable; every later launch reports).
// bit 0 / 1: a matrix / gather wave of rgcn_fused_kernel, bit 2 / 3: a matrix / producer wave of limb_gemm_pc_kernel.
int32_t* handover_status_word(hipStream_t stream);

}  // namespace relgnn
