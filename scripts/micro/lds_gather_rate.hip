// Upper bound on an LDS-resident slab variant of the C2 gather (north_star "messages tiled through LDS").
//
// The design under test: a workgroup owns (graph, column slice); the graph's slice of the state table H sits in LDS
// (PPI-shaped graphs: 600 .. 3500 nodes; 3500 rows x 8 floats = 112 KB, 2245 rows x 16 floats = 144 KB), every bucket's messages
// are folded SEQUENTIALLY by its owner (the reference's summation order: bit-exactness is not negotiable), reading the source
// rows with random ds_read_b128 instead of 1 KiB row loads through L1 / L2.  Two ways to own a bucket:
//   LANE  one lane per bucket, 8-float slices  (2 x ds_read_b128 + 8 mul + 8 add per message and lane)
//   QUAD  four lanes per bucket, 16-float slices (1 x ds_read_b128 + 4 mul + 4 add per message and lane)
// This micro-benchmark runs ONLY the inner loop, in its most favourable form: row indices from a register LCG (no index / weight
// stream from memory at all — the real kernel re-reads 8 B per message per slice), every lane the same number of messages (no
// degree imbalance — PPI in-degrees are heavy-tailed), no output rows written, one workgroup of 8 or 16 waves per CU, all 256 CUs.
// It reports message-slices per second and what ONE C2 layer forward (1 854 895 messages x 256 floats) would take at that rate:
// if that is not well below the 93 us (warm) / 121 us (in-step) of seg_reduce_wave_kernel, the design cannot win.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/lds_gather_rate.hip -o /tmp/lds_gather_rate && /tmp/lds_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>     // MODE 0: LANE (rows of 32 B), 1: QUAD (rows of 64 B)
__global__ __launch_bounds__(64 * WAVES) void gather_loop(const float* __restrict__ table, int rows, int iters, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int ROW_BYTES = MODE == 0 ? 32 : 64;
  // stage the slice: coalesced 16-byte loads
  for (int i = threadIdx.x; i < rows * ROW_BYTES / 16; i += blockDim.x)
    reinterpret_cast<f4*>(lds)[i] = reinterpret_cast<const f4*>(table)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t state = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  if constexpr (MODE == 1) state = (blockIdx.x * blockDim.x + (threadIdx.x >> 2)) * 2654435761u + 12345u;   // a quad shares its bucket
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float w = 0.03125f + 1e-3f * lane;
#pragma unroll 8
  for (int i = 0; i < iters; ++i) {
    state = state * 1664525u + 1013904223u;
    const uint32_t r = (uint32_t)(((uint64_t)(state >> 8) * (uint32_t)rows) >> 24);           // uniform row
    if constexpr (MODE == 0) {
      const f4 a = *reinterpret_cast<const f4*>(lds + r * 32);
      const f4 b = *reinterpret_cast<const f4*>(lds + r * 32 + 16);
      acc0 += a * w;          // (-ffp-contract=off: product and add rounded separately, as the real kernel must)
      acc1 += b * w;
    } else {
      const f4 a = *reinterpret_cast<const f4*>(lds + r * 64 + (lane & 3) * 16);
      acc0 += a * w;
    }
  }
  const f4 s = acc0 + acc1;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int MODE, int WAVES>
void run(const char* name, int rows, const float* table, float* out) {
  const int iters = 4096, blocks = 256, threads = 64 * WAVES;
  const size_t lds_bytes = (size_t)rows * (MODE == 0 ? 32 : 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_loop<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    gather_loop<MODE, WAVES><<<blocks, threads, lds_bytes>>>(table, rows, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double buckets = MODE == 0 ? (double)blocks * threads : (double)blocks * threads / 4;
  const double slices = buckets * iters;                         // message-slices folded
  const double per_msg = MODE == 0 ? 256.0 / 8 : 256.0 / 16;     // slices per 256-float message
  const double c2_us = 1854895.0 * per_msg / (slices / (best * 1e-3)) * 1e6;
  printf("{\"variant\": \"%s\", \"rows_in_lds\": %d, \"lds_bytes\": %zu, \"waves_per_cu\": %d, \"ms\": %.4f, "
         "\"message_slices_per_s\": %.4g, \"lds_read_TBps\": %.2f, \"one_c2_layer_forward_at_this_rate_us\": %.1f}\n",
         name, rows, lds_bytes, WAVES, best, slices / (best * 1e-3), slices * (MODE == 0 ? 32 : 16 * 4) / (best * 1e-3) / 1e12, c2_us);
}

int main() {
  float *table, *out;
  hipMalloc(&table, 160 * 1024);
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMemset(table, 0, 160 * 1024);
  run<0, 8>("lane per bucket, 8-float slices, 8 waves/CU", 3500, table, out);
  run<0, 16>("lane per bucket, 8-float slices, 16 waves/CU", 3500, table, out);
  run<1, 8>("quad per bucket, 16-float slices, 8 waves/CU", 2245, table, out);
  run<1, 16>("quad per bucket, 16-float slices, 16 waves/CU", 2245, table, out);
  hipFree(table); hipFree(out);
  return 0;
}
