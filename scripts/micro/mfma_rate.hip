// Sustained fp32 MFMA rate by instruction shape, operands in registers (no memory traffic in the loop): what the matrix
// pipe gives on this chip under its power limit, on all-zero and on random operands.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));

template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a0 = in[t], b0 = in[t + 1], a1 = in[t + 2], b1 = in[t + 3];
  float s = 0.f;
  if constexpr (KIND == 0) {  // 32x32x2: 4096 flop / instr / wave... (32*32*2*2)
    f16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
    }
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  } else if constexpr (KIND == 1) {  // 16x16x4
    f4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c3, 0, 0, 0);
    }
    for (int k = 0; k < 4; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  } else if constexpr (KIND == 2) {  // 16x16x1, 4 blocks
    f16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, b0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, b1, c3, 0, 0, 0);
    }
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  } else {  // 32x32x1, 2 blocks
    f32v c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b1, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, b1, c1, 0, 0, 0);
    }
    for (int k = 0; k < 32; ++k) s += c0[k] + c1[k];
  }
  out[t] = s;
}

template <int KIND>
double run(const float* in, float* out, int iters, double flop_per_instr) {
  const int blocks = 256 * 8, threads = 256;       // 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<KIND><<<blocks, threads>>>(in, out, iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<KIND><<<blocks, threads>>>(in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * (threads / 64) * iters * 4.0;
  return instr * flop_per_instr / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t n = 256 * 8 * 256 + 8;
  std::vector<float> h(n);
  float *in, *out;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
  const int iters = 40000;
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t i = 0; i < n; ++i) h[i] = pass == 0 ? 0.f : (float)((rand() % 2001) - 1000) * 1e-3f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("%s operands (TFLOP/s, 8 waves/SIMD, 4 independent accumulators):\n", pass == 0 ? "all-zero" : "random");
    printf("  v_mfma_f32_32x32x2f32      %7.1f\n", run<0>(in, out, iters, 32.0 * 32 * 2 * 2));
    printf("  v_mfma_f32_16x16x4f32      %7.1f\n", run<1>(in, out, iters, 16.0 * 16 * 4 * 2));
    printf("  v_mfma_f32_16x16x1f32 (4b) %7.1f\n", run<2>(in, out, iters, 16.0 * 16 * 1 * 2 * 4));
    printf("  v_mfma_f32_32x32x1f32 (2b) %7.1f\n", run<3>(in, out, iters, 32.0 * 32 * 1 * 2 * 2));
  }
  return 0;
}
