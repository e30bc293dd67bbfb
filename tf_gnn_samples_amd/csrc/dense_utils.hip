// Small dense helpers around the hot path (task-head plumbing, not gather/segment work).
#include "common.h"

using namespace relgnn;

namespace {

// partial[blockIdx.x, c] = sum over this block's row slice of X[r, c]; 256 threads = 4 row groups x 64 columns.
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ X, int64_t rows, int32_t cols,
                                                         int64_t ld, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx;
  const int64_t chunk = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
  float acc = 0.f;
  if (c < cols) {
    // four independent loads in flight per thread (the one-load-per-iteration form was latency-bound: 19.5 us for [36 k, 121])
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t r = r0 + rg;
    for (; r + 12 < r1; r += 16) {
      a0 += X[r * ld + c];
      a1 += X[(r + 4) * ld + c];
      a2 += X[(r + 8) * ld + c];
      a3 += X[(r + 12) * ld + c];
    }
    for (; r < r1; r += 4) a0 += X[r * ld + c];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[rg][cx] = acc;
  __syncthreads();
  if (rg == 0 && c < cols) partial[(int64_t)blockIdx.x * cols + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// X[rows[i], 0 .. cols) = value: one lane group of 64 per listed row
__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ X, int64_t ld, int32_t cols, const int64_t* __restrict__ rows,
                                                        int64_t n, float value) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t r = rows[i];
  if (r < 0) return;                                   // (a list padded with -1: nothing to fill)
  float* row = X + r * ld;
  for (int c = threadIdx.x & 63; c < cols; c += 64) row[c] = value;
}

}  // namespace

extern "C" {

int relgnn_fill_rows_f32(float* X, int64_t ld, int32_t cols, const int64_t* rows, int64_t num_rows, float value, void* stream) {
  if (cols < 0 || num_rows < 0 || ld < cols) return RELGNN_EINVAL;
  if (cols == 0 || num_rows == 0) return RELGNN_OK;
  if (!X || !rows) return RELGNN_EINVAL;
  fill_rows_kernel<<<(unsigned)((num_rows + 3) / 4), 256, 0, as_stream(stream)>>>(X, ld, cols, rows, num_rows, value);
  return launch_status();
}

size_t relgnn_column_sum_workspace_bytes(int64_t rows, int32_t cols) {
  (void)rows;
  return (size_t)512 * (size_t)(cols > 0 ? cols : 1) * sizeof(float);
}

int relgnn_column_sum(const float* X, int64_t rows, int32_t cols, int64_t ld, float* out, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (rows < 0 || cols < 0 || ld < cols) return RELGNN_EINVAL;
  if (cols == 0) return RELGNN_OK;
  if (!out) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (hipMemsetAsync(out, 0, sizeof(float) * cols, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!X || !workspace) return RELGNN_EINVAL;
  if (workspace_bytes < relgnn_column_sum_workspace_bytes(rows, cols)) return RELGNN_ENOSPC;
  const unsigned nblk = (unsigned)((rows + 63) / 64 < 512 ? (rows + 63) / 64 : 512);
  dim3 grid(nblk, (unsigned)((cols + 63) / 64));
  float* partial = static_cast<float*>(workspace);
  column_sum_kernel<<<grid, 256, 0, st>>>(X, rows, cols, ld, partial);
  dim3 grid2(1, (unsigned)((cols + 63) / 64));
  column_sum_kernel<<<grid2, 256, 0, st>>>(partial, nblk, cols, cols, out);
  return launch_status();
}

}  // extern "C"
