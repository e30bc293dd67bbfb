// Batch bucketing from DATASET-RESIDENT per-graph plans (include/relgnn.h section 10).
//
// A batch is a disjoint union of graphs, and both bucketed orders of a batch are graph-major (nodes of one graph are
// contiguous, so all messages into / out of that graph are contiguous in the (node, type)-sorted orders).  If the
// whole dataset has been bucketed ONCE as one big disjoint union and its index arrays stay in HBM, the arrays of any
// batch are the per-graph slices of the dataset arrays, re-based: node ids by the graph's node offset, positions by the
// graph's message offset, original message ids through the type-major numbering of the batch.  Three streaming kernels
// replace the two radix sorts + finalise passes per batch (bit-identical result: same stable order).
#include "common.h"

using namespace relgnn;

namespace {

// largest k with table[k] <= x  (table ascending, table[0] <= x < table[n])
__device__ __forceinline__ int upper_slot(const int64_t* __restrict__ table, int n, int64_t x) {
  int lo = 0, hi = n;                  // invariant: table[lo] <= x < table[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (table[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

struct BatchTables {
  const int64_t* ids;          // [K] dataset graph id of batch slot k
  const int64_t* node_off_b;   // [K+1]
  const int64_t* msg_off_b;    // [K+1] first by-target / by-source position of slot k
  const int64_t* edge_off_b;   // [L][K+1] messages of type l before slot k (inside the type's block)
  const int64_t* type_off_b;   // [L+1]
  const int64_t* node_off_d;   // [G+1]
  const int64_t* msg_off_d;    // [G+1]
  const int64_t* edge_off_d;   // [L][G+1]
  const int64_t* type_off_d;   // [L+1]
  int32_t K, L;
  int64_t G;
};

// original message id: dataset numbering -> batch numbering (both type-major, graphs in their order inside a type)
__device__ __forceinline__ int32_t translate_message(const BatchTables& t, int32_t m_d, int l, int k, int64_t g) {
  const int64_t e = (int64_t)m_d - t.type_off_d[l] - t.edge_off_d[(int64_t)l * (t.G + 1) + g];
  return (int32_t)(t.type_off_b[l] + t.edge_off_b[(int64_t)l * (t.K + 1) + k] + e);
}

// FULL = false ("lean"): only what the fused gather kernels read per message — the source / target NODE and the scale.
// The permutations, the inverse permutation (a scattered 4-byte write per message) and the (node, type) rows are what the
// pair / attention / materialised-message paths need; a caller that passes NULL for them gets them later by calling again.
template <bool FULL>
__global__ __launch_bounds__(256) void assemble_by_target_kernel(
    BatchTables t, int64_t M, const int32_t* __restrict__ perm_d, const int32_t* __restrict__ col_d,
    int32_t* __restrict__ perm_b, int32_t* __restrict__ col_b, int32_t* __restrict__ inv_b,
    const float* __restrict__ w_t_d, int32_t* __restrict__ src_b, float* __restrict__ w_t_b) {
  // positions of one slot are contiguous: the slot of a block's first position is searched once per block (thread 0) and
  // walked forward per element, instead of a binary search per message
  __shared__ int first_slot;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < M; base += (int64_t)gridDim.x * 1024) {
    if (threadIdx.x == 0) first_slot = upper_slot(t.msg_off_b, t.K, base);
    __syncthreads();
    int k = first_slot;
    int64_t g = t.ids[k], slot_end = t.msg_off_b[k + 1];                        // per-slot constants stay in registers
    int64_t msg_delta = t.msg_off_d[g] - t.msg_off_b[k], node_delta = t.node_off_b[k] - t.node_off_d[g];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = base + threadIdx.x + 256 * e;
      if (p >= M) break;
      if (p >= slot_end) {
        do { ++k; } while (p >= t.msg_off_b[k + 1]);
        g = t.ids[k]; slot_end = t.msg_off_b[k + 1];
        msg_delta = t.msg_off_d[g] - t.msg_off_b[k]; node_delta = t.node_off_b[k] - t.node_off_d[g];
      }
      const int64_t pd = p + msg_delta;
      const uint32_t c = (uint32_t)col_d[pd];
      const uint32_t cn = c / (uint32_t)t.L;
      const int64_t src = (int64_t)cn + node_delta;
      if (src_b) src_b[p] = (int32_t)src;            // source NODE per by-target position (row of the state table)
      if (w_t_b) w_t_b[p] = w_t_d[pd];                // per-message scale: a property of the graph, not of the batch
      if constexpr (FULL) {
        const int l = (int)(c - cn * (uint32_t)t.L);
        col_b[p] = (int32_t)(src * t.L + l);
        const int32_t mb = translate_message(t, perm_d[pd], l, k, g);
        perm_b[p] = mb;
        inv_b[mb] = (int32_t)p;
      }
    }
    __syncthreads();
  }
}

template <bool FULL>
__global__ __launch_bounds__(256) void assemble_by_source_kernel(
    BatchTables t, int64_t M, const int32_t* __restrict__ perm_d, const int32_t* __restrict__ frow_d,
    const int32_t* __restrict__ pos_d, int32_t* __restrict__ perm_b, int32_t* __restrict__ frow_b,
    int32_t* __restrict__ tgt_b, int32_t* __restrict__ pos_b, const float* __restrict__ w_s_d, float* __restrict__ w_s_b) {
  __shared__ int first_slot;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < M; base += (int64_t)gridDim.x * 1024) {
    if (threadIdx.x == 0) first_slot = upper_slot(t.msg_off_b, t.K, base);
    __syncthreads();
    int k = first_slot;
    int64_t g = t.ids[k], slot_end = t.msg_off_b[k + 1];
    int64_t msg_delta = t.msg_off_d[g] - t.msg_off_b[k], node_delta = t.node_off_b[k] - t.node_off_d[g];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t q = base + threadIdx.x + 256 * e;
      if (q >= M) break;
      if (q >= slot_end) {
        do { ++k; } while (q >= t.msg_off_b[k + 1]);
        g = t.ids[k]; slot_end = t.msg_off_b[k + 1];
        msg_delta = t.msg_off_d[g] - t.msg_off_b[k]; node_delta = t.node_off_b[k] - t.node_off_d[g];
      }
      const int64_t qd = q + msg_delta;
      const uint32_t f = (uint32_t)frow_d[qd];
      const uint32_t fn = f / (uint32_t)t.L;
      const int64_t tgt = (int64_t)fn + node_delta;
      tgt_b[q] = (int32_t)tgt;
      if (w_s_b) w_s_b[q] = w_s_d[qd];
      if constexpr (FULL) {
        const int l = (int)(f - fn * (uint32_t)t.L);
        frow_b[q] = (int32_t)(tgt * t.L + l);
        perm_b[q] = translate_message(t, perm_d[qd], l, k, g);
        pos_b[q] = (int32_t)((int64_t)pos_d[qd] - msg_delta);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void assemble_rowptr_kernel(
    BatchTables t, int64_t num_buckets, int64_t M, const int32_t* __restrict__ rowptr_t_d,
    const int32_t* __restrict__ rowptr_s_d, int32_t* __restrict__ rowptr_t_b, int32_t* __restrict__ rowptr_s_b) {
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b <= num_buckets; b += (int64_t)gridDim.x * blockDim.x) {
    if (b == num_buckets) {
      rowptr_t_b[b] = (int32_t)M;
      rowptr_s_b[b] = (int32_t)M;
      continue;
    }
    const int64_t v = b / t.L;
    const int l = (int)(b - v * t.L);
    const int k = upper_slot(t.node_off_b, t.K, v);
    const int64_t g = t.ids[k];
    const int64_t bd = (v - t.node_off_b[k] + t.node_off_d[g]) * t.L + l;
    const int64_t shift = t.msg_off_b[k] - t.msg_off_d[g];
    rowptr_t_b[b] = (int32_t)((int64_t)rowptr_t_d[bd] + shift);
    rowptr_s_b[b] = (int32_t)((int64_t)rowptr_s_d[bd] + shift);
  }
}

// ---- the batch's TENSORS, gathered from the fold's flat arrays (packing rules of relgnn_batch_pack) -----------------
// Nodes of a graph are contiguous in the fold and in the batch: every copy below is K contiguous segment copies, found
// per element by a binary search over the K+1 batch offsets (K <= a few hundred; the tables sit in L1/L2).

// dst[v, :] = src[node_d(v), :]   rows of `cols` 4-byte elements.  A slot's rows are contiguous on both sides, so the copy
// is flat per slot: every thread takes 4 consecutive elements of the batch's flat payload, the slot of the block's first
// element is searched once per block and walked forward from there (a block of 1024 elements rarely crosses a slot).
__global__ __launch_bounds__(256) void gather_node_rows_kernel(BatchTables t, int64_t V, int32_t cols,
                                                               const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
  __shared__ int first_slot;
  const int64_t total = V * cols;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < total; base += (int64_t)gridDim.x * 1024) {
    if (threadIdx.x == 0) {
      int lo = 0, hi = t.K;              // invariant: node_off_b[lo] * cols <= base < node_off_b[hi] * cols
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.node_off_b[mid] * cols <= base) lo = mid; else hi = mid;
      }
      first_slot = lo;
    }
    __syncthreads();
    int k = first_slot;
    int64_t slot_end = t.node_off_b[k + 1] * cols;                               // per-slot constants stay in registers
    int64_t delta = (t.node_off_d[t.ids[k]] - t.node_off_b[k]) * cols;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = base + threadIdx.x + 256 * e;      // 256 consecutive elements per pass: coalesced
      if (i < total) {
        if (i >= slot_end) {
          do { ++k; } while (i >= t.node_off_b[k + 1] * cols);
          slot_end = t.node_off_b[k + 1] * cols;
          delta = (t.node_off_d[t.ids[k]] - t.node_off_b[k]) * cols;
        }
        dst[i] = src[delta + i];
      }
    }
    __syncthreads();
  }
}

// deg_b[l, v] = deg_d[l, node_d(v)];  node_to_graph[v] = batch slot of v   (tasks/ppi_task.py:231-237)
__global__ __launch_bounds__(256) void gather_degree_kernel(BatchTables t, int64_t V, int64_t N, const float* __restrict__ deg_d,
                                                            float* __restrict__ deg_b, int32_t* __restrict__ node_to_graph) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const int k = upper_slot(t.node_off_b, t.K, v);
    const int64_t vd = v - t.node_off_b[k] + t.node_off_d[t.ids[k]];
    for (int l = 0; l < t.L; ++l) deg_b[(int64_t)l * V + v] = deg_d[(int64_t)l * N + vd];
    node_to_graph[v] = k;
  }
}

// adj_b[p] = adj_d[message_d(p)] + node offset of the slot   (tasks/ppi_task.py:228); both lists type-major [M, 2]
__global__ __launch_bounds__(256) void gather_adjacency_kernel(BatchTables t, int64_t M, const int2* __restrict__ adj_d,
                                                               int2* __restrict__ adj_b) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < M; p += (int64_t)gridDim.x * blockDim.x) {
    const int l = upper_slot(t.type_off_b, t.L, p);
    const int64_t e = p - t.type_off_b[l];
    const int64_t* eob = t.edge_off_b + (int64_t)l * (t.K + 1);
    const int k = upper_slot(eob, t.K, e);
    const int64_t g = t.ids[k];
    const int64_t pd = t.type_off_d[l] + t.edge_off_d[(int64_t)l * (t.G + 1) + g] + (e - eob[k]);
    int2 a = adj_d[pd];
    const int off = (int)t.node_off_b[k];
    a.x += off; a.y += off;
    adj_b[p] = a;
  }
}

}  // namespace

extern "C" {

int relgnn_batch_gather(const int64_t* ids, int32_t num_batch_graphs, int32_t num_edge_types, int64_t num_dataset_graphs,
                        const int64_t* node_off_b, const int64_t* edge_off_b, const int64_t* type_off_b,
                        const int64_t* node_off_d, const int64_t* edge_off_d, const int64_t* type_off_d, int64_t num_nodes,
                        int64_t num_messages, int64_t num_dataset_nodes, int32_t n_payloads, const void* const* h_payload_d,
                        const int32_t* h_payload_cols, void* const* h_payload_b, const float* deg_d, float* deg_b,
                        const int32_t* adj_d, int32_t* adj_b, int32_t* node_to_graph, void* stream) {
  if (num_batch_graphs < 0 || num_edge_types <= 0 || num_dataset_graphs < 0 || num_nodes < 0 || num_messages < 0 ||
      n_payloads < 0 || num_dataset_nodes < 0)
    return RELGNN_EINVAL;
  if (num_batch_graphs == 0 || num_nodes == 0) return RELGNN_OK;
  if (!ids || !node_off_b || !edge_off_b || !type_off_b || !node_off_d || !edge_off_d || !type_off_d) return RELGNN_EINVAL;
  if (n_payloads > 0 && (!h_payload_d || !h_payload_cols || !h_payload_b)) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  BatchTables t{ids, node_off_b, nullptr, edge_off_b, type_off_b, node_off_d, nullptr, edge_off_d, type_off_d,
                num_batch_graphs, num_edge_types, num_dataset_graphs};
  for (int p = 0; p < n_payloads; ++p) {
    if (h_payload_cols[p] <= 0) continue;
    if (!h_payload_d[p] || !h_payload_b[p]) return RELGNN_EINVAL;
    gather_node_rows_kernel<<<flat_grid((num_nodes * h_payload_cols[p] + 3) / 4, 256), 256, 0, st>>>(
        t, num_nodes, h_payload_cols[p], (const uint32_t*)h_payload_d[p], (uint32_t*)h_payload_b[p]);
  }
  if (deg_b || node_to_graph) {
    if (!deg_d || !deg_b || !node_to_graph) return RELGNN_EINVAL;
    gather_degree_kernel<<<flat_grid(num_nodes, 256), 256, 0, st>>>(t, num_nodes, num_dataset_nodes, deg_d, deg_b, node_to_graph);
  }
  if (num_messages > 0 && adj_b) {               // adj_b == NULL: the caller does not want the adjacency lists (yet)
    if (!adj_d) return RELGNN_EINVAL;
    gather_adjacency_kernel<<<flat_grid(num_messages, 256), 256, 0, st>>>(t, num_messages, (const int2*)adj_d, (int2*)adj_b);
  }
  return launch_status();
}

int relgnn_plan_assemble(const int64_t* ids, int32_t num_batch_graphs, int32_t num_edge_types, int64_t num_dataset_graphs,
                         const int64_t* node_off_b, const int64_t* msg_off_b, const int64_t* edge_off_b,
                         const int64_t* type_off_b, const int64_t* node_off_d, const int64_t* msg_off_d,
                         const int64_t* edge_off_d, const int64_t* type_off_d, int64_t num_nodes, int64_t num_messages,
                         const int32_t* rowptr_t_d, const int32_t* perm_t_d, const int32_t* col_t_d,
                         const int32_t* rowptr_s_d, const int32_t* perm_s_d, const int32_t* frow_s_d,
                         const int32_t* pos_t_of_s_d, int32_t* rowptr_t, int32_t* perm_t, int32_t* col_t,
                         int32_t* inv_perm_t, int32_t* rowptr_s, int32_t* perm_s, int32_t* frow_s, int32_t* tgt_s,
                         int32_t* pos_t_of_s, const float* w_t_d, const float* w_s_d, int32_t* src_t, float* w_t, float* w_s,
                         void* stream) {
  if (num_batch_graphs < 0 || num_edge_types <= 0 || num_dataset_graphs < 0 || num_nodes < 0 || num_messages < 0)
    return RELGNN_EINVAL;
  const int64_t buckets = num_nodes * num_edge_types;
  if (buckets >= INT32_MAX || num_messages >= INT32_MAX) return RELGNN_EUNSUPPORTED;
  if (!rowptr_t || !rowptr_s) return RELGNN_EINVAL;
  hipStream_t st = as_stream(stream);
  if (num_batch_graphs == 0 || num_nodes == 0) {
    if (hipMemsetAsync(rowptr_t, 0, (size_t)(buckets + 1) * 4, st) != hipSuccess) return RELGNN_EHIP;
    if (hipMemsetAsync(rowptr_s, 0, (size_t)(buckets + 1) * 4, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!ids || !node_off_b || !msg_off_b || !edge_off_b || !type_off_b || !node_off_d || !msg_off_d || !edge_off_d ||
      !type_off_d || !rowptr_t_d || !rowptr_s_d)
    return RELGNN_EINVAL;
  BatchTables t{ids, node_off_b, msg_off_b, edge_off_b, type_off_b, node_off_d, msg_off_d, edge_off_d, type_off_d,
                num_batch_graphs, num_edge_types, num_dataset_graphs};
  assemble_rowptr_kernel<<<flat_grid(buckets + 1, 256), 256, 0, st>>>(t, buckets, num_messages, rowptr_t_d, rowptr_s_d,
                                                                      rowptr_t, rowptr_s);
  if (num_messages > 0) {
    // the six index arrays of the pair / attention / materialised-message paths are all-or-nothing
    const bool full = perm_t || col_t || inv_perm_t || perm_s || frow_s || pos_t_of_s;
    if (full && (!perm_t || !col_t || !inv_perm_t || !perm_s || !frow_s || !pos_t_of_s || !perm_t_d || !perm_s_d || !pos_t_of_s_d))
      return RELGNN_EINVAL;
    if (!col_t_d || !frow_s_d || !tgt_s) return RELGNN_EINVAL;
    if (!full && !src_t) return RELGNN_EINVAL;      // a lean call that produces nothing by target
    if ((w_t && !w_t_d) || (w_s && !w_s_d)) return RELGNN_EINVAL;
    const dim3 grid = flat_grid((num_messages + 3) / 4, 256);
    if (full) {
      assemble_by_target_kernel<true><<<grid, 256, 0, st>>>(t, num_messages, perm_t_d, col_t_d, perm_t, col_t, inv_perm_t,
                                                            w_t_d, src_t, w_t);
      assemble_by_source_kernel<true><<<grid, 256, 0, st>>>(t, num_messages, perm_s_d, frow_s_d, pos_t_of_s_d, perm_s, frow_s,
                                                            tgt_s, pos_t_of_s, w_s_d, w_s);
    } else {
      assemble_by_target_kernel<false><<<grid, 256, 0, st>>>(t, num_messages, nullptr, col_t_d, nullptr, nullptr, nullptr,
                                                             w_t_d, src_t, w_t);
      assemble_by_source_kernel<false><<<grid, 256, 0, st>>>(t, num_messages, nullptr, frow_s_d, nullptr, nullptr, nullptr,
                                                             tgt_s, nullptr, w_s_d, w_s);
    }
  }
  return launch_status();
}

}  // extern "C"
