// Index bookkeeping for the relational message-passing path (bit-exact integer work).
//
// Builds, once per batch, the (edge type, target)-bucketed CSR that lets the reduce
// kernels in seg_reduce.hip read source rows coalesced and accumulate without atomics.
// Reference semantics restated: gnns/rgcn.py:68-78 (message list = type-major concat of
// the adjacency lists, targets = column 1, sources = column 0).
//
// The stable sort is rocPRIM's device radix sort (LSD radix sort is stable by
// construction) restricted to the significant key bits; everything else is hand-written.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

using namespace relgnn;

namespace {

__global__ __launch_bounds__(256) void relational_keys_kernel(
    const int2* __restrict__ adj, int64_t num_edges, int32_t edge_type, int32_t L, int32_t V,
    int64_t msg_base, int32_t* __restrict__ key_t, int32_t* __restrict__ key_s,
    int32_t* __restrict__ node_t, int32_t* __restrict__ node_s, uint32_t* __restrict__ err_flag) {
  bool bad = false;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < num_edges;
       e += (int64_t)gridDim.x * blockDim.x) {
    int2 st = adj[e];  // {source, target}: one 8-byte coalesced load per edge
    int32_t s = st.x, t = st.y;
    if ((uint32_t)s >= (uint32_t)V || (uint32_t)t >= (uint32_t)V) {
      bad = true;
      s = min(max(s, 0), V - 1);
      t = min(max(t, 0), V - 1);
    }
    key_t[msg_base + e] = t * L + edge_type;
    key_s[msg_base + e] = s * L + edge_type;
    if (node_t) node_t[msg_base + e] = t;
    if (node_s) node_s[msg_base + e] = s;
  }
  // one atomic per wave at most, and only on the error path
  if (err_flag != nullptr && __any(bad)) {
    if ((threadIdx.x & (RELGNN_WAVE - 1)) == 0) atomicOr(err_flag, RELGNN_ERRFLAG_INDEX_OUT_OF_RANGE);
  }
}

// All edge types of a batch in one launch (up to kKeysTypes per launch): the per-type kernel above costs one ~8 us
// launch per edge type, 23 of them for a VarMisuse-shaped batch.
constexpr int kKeysTypes = 32;
struct KeysArgs {
  const int2* adj[kKeysTypes];
  long long base[kKeysTypes + 1];   // message offset of every type inside this launch's range; base[n] = end
};

__global__ __launch_bounds__(256) void relational_keys_all_kernel(
    KeysArgs a, int32_t n_types, int32_t first_type, int32_t L, int32_t V, long long msg_base,
    int32_t* __restrict__ key_t, int32_t* __restrict__ key_s, int32_t* __restrict__ node_t,
    int32_t* __restrict__ node_s, uint32_t* __restrict__ err_flag) {
  bool bad = false;
  const long long total = a.base[n_types];
  for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < total; m += (long long)gridDim.x * blockDim.x) {
    int t = 0;
    while (t + 1 < n_types && m >= a.base[t + 1]) ++t;
    const int2 st = a.adj[t][m - a.base[t]];
    int32_t s = st.x, tg = st.y;
    if ((uint32_t)s >= (uint32_t)V || (uint32_t)tg >= (uint32_t)V) {
      bad = true;
      s = min(max(s, 0), V - 1);
      tg = min(max(tg, 0), V - 1);
    }
    const int32_t l = first_type + t;
    key_t[msg_base + m] = tg * L + l;
    key_s[msg_base + m] = s * L + l;
    node_t[msg_base + m] = tg;
    node_s[msg_base + m] = s;
  }
  if (err_flag != nullptr && __any(bad)) {
    if ((threadIdx.x & (RELGNN_WAVE - 1)) == 0) atomicOr(err_flag, RELGNN_ERRFLAG_INDEX_OUT_OF_RANGE);
  }
}

// rowptr[s] = first sorted position whose key >= s.  Every rowptr entry is written by
// exactly one thread (the one that owns the position where the key changes), so the
// result is deterministic and needs no atomics / histogram.
__global__ __launch_bounds__(256) void rowptr_from_sorted_kernel(
    const int32_t* __restrict__ sorted_keys, int64_t n, int64_t num_segments,
    int32_t* __restrict__ rowptr) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p <= n;
       p += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = (p == 0) ? 0 : (int64_t)sorted_keys[p - 1] + 1;
    int64_t hi = (p == n) ? num_segments : (int64_t)sorted_keys[p];
    // segments lo..hi start at p (hi itself starts at p only when p < n, or is the end sentinel)
    for (int64_t s = lo; s <= hi; ++s) rowptr[s] = (int32_t)p;
  }
}

// one pass over the sorted positions: everything the two CSR orders need from the permutation
__global__ __launch_bounds__(256) void plan_finalize_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ full_key, const int32_t* __restrict__ other_key,
    int64_t n, int32_t L, int32_t* __restrict__ sorted_full, int32_t* __restrict__ col, int32_t* __restrict__ col_div,
    int32_t* __restrict__ inv_out, const int32_t* __restrict__ inv_in, int32_t* __restrict__ pos_out) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t m = perm[p];
    sorted_full[p] = full_key[m];
    const int32_t c = other_key[m];
    col[p] = c;
    if (col_div) col_div[p] = c / L;
    if (inv_out) inv_out[m] = (int32_t)p;
    if (pos_out) pos_out[p] = inv_in[m];
  }
}

__global__ __launch_bounds__(256) void gather_i32_kernel(const int32_t* __restrict__ table,
                                                         const int32_t* __restrict__ index,
                                                         int64_t n, int32_t divisor,
                                                         int32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t v = table[index[i]];
    out[i] = divisor > 1 ? v / divisor : v;
  }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ table,
                                                         const int32_t* __restrict__ index,
                                                         int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = table[index[i]];
}

__global__ __launch_bounds__(256) void invert_perm_kernel(const int32_t* __restrict__ perm,
                                                          int64_t n, int32_t* __restrict__ inv) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    inv[perm[i]] = (int32_t)i;
}

// One thread per (target, type) sub-segment: 1/(c + eps) in fp32, broadcast to its messages.
__global__ __launch_bounds__(256) void degree_scale_kernel(const float* __restrict__ deg,
                                                           const int32_t* __restrict__ rowptr,
                                                           int32_t L, int32_t V, float eps,
                                                           float* __restrict__ scale) {
  const int64_t nseg = (int64_t)L * V;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nseg;
       s += (int64_t)gridDim.x * blockDim.x) {
    int32_t b = rowptr[s], e = rowptr[s + 1];
    if (b == e) continue;
    int32_t v = (int32_t)(s / L), l = (int32_t)(s - (int64_t)v * L);
    // exactly the reference arithmetic: 1.0 / (num_incoming + SMALL_NUMBER), float32
    float inv = 1.0f / (deg[(int64_t)l * V + v] + eps);
    for (int32_t p = b; p < e; ++p) scale[p] = inv;
  }
}

__global__ __launch_bounds__(256) void counts_scale_kernel(const int32_t* __restrict__ rowptr,
                                                           int64_t num_segments, int32_t stride,
                                                           int32_t mode, const float* __restrict__ w,
                                                           float* __restrict__ scale) {
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < num_segments;
       s += (int64_t)gridDim.x * blockDim.x) {
    int32_t b = rowptr[s * stride], e = rowptr[(s + 1) * stride];
    float n = (float)max(e - b, 1);
    float f = 1.0f;
    if (mode == RELGNN_AGG_MEAN) f = 1.0f / n;
    if (mode == RELGNN_AGG_SQRT_N) f = 1.0f / sqrtf(n);
    for (int32_t p = b; p < e; ++p) scale[p] = (w ? w[p] : 1.0f) * f;
  }
}

inline int key_bits(int64_t num_segments) {
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < num_segments) ++bits;
  return bits;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

int relgnn_abi_version(void) { return RELGNN_ABI_VERSION; }

const char* relgnn_status_string(int status) {
  switch (status) {
    case RELGNN_OK: return "ok";
    case RELGNN_EINVAL: return "invalid argument";
    case RELGNN_ENOSPC: return "workspace too small";
    case RELGNN_EHIP: return "HIP runtime error";
    case RELGNN_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

int relgnn_relational_keys(const int32_t* adj, int64_t num_edges, int32_t edge_type,
                           int32_t num_edge_types, int32_t num_nodes, int64_t msg_base,
                           int32_t* key_by_target, int32_t* key_by_source, uint32_t* err_flag,
                           void* stream) {
  if (num_edges < 0 || num_edge_types <= 0 || num_nodes < 0 || msg_base < 0 || edge_type < 0 ||
      edge_type >= num_edge_types)
    return RELGNN_EINVAL;
  if ((int64_t)num_nodes * num_edge_types > INT32_MAX) return RELGNN_EUNSUPPORTED;
  if (num_edges == 0) return RELGNN_OK;  // empty edge type: tasks/ppi_task.py:248-249
  if (!adj || !key_by_target || !key_by_source || num_nodes == 0) return RELGNN_EINVAL;
  if (reinterpret_cast<uintptr_t>(adj) & 7u) return RELGNN_EINVAL;
  relational_keys_kernel<<<flat_grid(num_edges, 256), 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const int2*>(adj), num_edges, edge_type, num_edge_types, num_nodes, msg_base,
      key_by_target, key_by_source, nullptr, nullptr, err_flag);
  return launch_status();
}

int relgnn_relational_keys2(const int32_t* adj, int64_t num_edges, int32_t edge_type, int32_t num_edge_types,
                            int32_t num_nodes, int64_t msg_base, int32_t* key_by_target, int32_t* key_by_source,
                            int32_t* target_node, int32_t* source_node, uint32_t* err_flag, void* stream) {
  if (num_edges < 0 || num_edge_types <= 0 || num_nodes < 0 || msg_base < 0 || edge_type < 0 ||
      edge_type >= num_edge_types)
    return RELGNN_EINVAL;
  if ((int64_t)num_nodes * num_edge_types > INT32_MAX) return RELGNN_EUNSUPPORTED;
  if (num_edges == 0) return RELGNN_OK;
  if (!adj || !key_by_target || !key_by_source || num_nodes == 0) return RELGNN_EINVAL;
  if (reinterpret_cast<uintptr_t>(adj) & 7u) return RELGNN_EINVAL;
  relational_keys_kernel<<<flat_grid(num_edges, 256), 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const int2*>(adj), num_edges, edge_type, num_edge_types, num_nodes, msg_base,
      key_by_target, key_by_source, target_node, source_node, err_flag);
  return launch_status();
}

int relgnn_relational_keys_all(const int32_t* const* h_adj, const int64_t* h_num_edges, int32_t num_edge_types,
                               int32_t num_nodes, int32_t* key_by_target, int32_t* key_by_source, int32_t* target_node,
                               int32_t* source_node, uint32_t* err_flag, void* stream) {
  if (num_edge_types <= 0 || num_nodes < 0 || !h_adj || !h_num_edges) return RELGNN_EINVAL;
  if ((int64_t)num_nodes * num_edge_types > INT32_MAX) return RELGNN_EUNSUPPORTED;
  long long msg_base = 0;
  for (int32_t first = 0; first < num_edge_types; first += kKeysTypes) {
    const int32_t n = (num_edge_types - first < kKeysTypes) ? (num_edge_types - first) : kKeysTypes;
    KeysArgs a;
    a.base[0] = 0;
    for (int32_t t = 0; t < n; ++t) {
      const int64_t e = h_num_edges[first + t];
      if (e < 0 || (e > 0 && (!h_adj[first + t] || (reinterpret_cast<uintptr_t>(h_adj[first + t]) & 7u)))) return RELGNN_EINVAL;
      a.adj[t] = reinterpret_cast<const int2*>(h_adj[first + t]);
      a.base[t + 1] = a.base[t] + e;
    }
    const long long total = a.base[n];
    if (total > 0) {
      if (!key_by_target || !key_by_source || !target_node || !source_node || num_nodes == 0) return RELGNN_EINVAL;
      relational_keys_all_kernel<<<flat_grid(total, 256), 256, 0, as_stream(stream)>>>(
          a, n, first, num_edge_types, num_nodes, msg_base, key_by_target, key_by_source, target_node, source_node, err_flag);
    }
    msg_base += total;
  }
  return launch_status();
}

size_t relgnn_relational_plan_workspace_bytes(int64_t num_messages, int32_t num_nodes) {
  // [spare | sorted node keys | sorted full keys | rocprim temp]
  return relgnn_segment_plan_workspace_bytes(num_messages, num_nodes) + align_up((size_t)(num_messages > 0 ? num_messages : 1) * 4, 256);
}

int relgnn_relational_plan(const int32_t* sort_node, const int32_t* full_key, const int32_t* other_key,
                           int64_t num_messages, int32_t num_nodes, int32_t num_edge_types, int32_t* rowptr,
                           int32_t* perm, int32_t* col, int32_t* col_div, int32_t* inv_out, const int32_t* inv_in,
                           int32_t* pos_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (num_messages < 0 || num_nodes < 0 || num_edge_types <= 0 || !rowptr) return RELGNN_EINVAL;
  const int64_t S = (int64_t)num_nodes * num_edge_types;
  if (num_messages > INT32_MAX || S >= INT32_MAX) return RELGNN_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (num_messages == 0) {
    if (hipMemsetAsync(rowptr, 0, (size_t)(S + 1) * 4, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!sort_node || !full_key || !other_key || !perm || !col || !workspace) return RELGNN_EINVAL;
  if ((pos_out != nullptr) != (inv_in != nullptr)) return RELGNN_EINVAL;
  if (workspace_bytes < relgnn_relational_plan_workspace_bytes(num_messages, num_nodes)) return RELGNN_ENOSPC;
  char* ws = static_cast<char*>(workspace);
  const size_t seg = align_up((size_t)num_messages * 4, 256);
  int32_t* iota = reinterpret_cast<int32_t*>(ws);
  int32_t* sorted_nodes = reinterpret_cast<int32_t*>(ws + seg);
  int32_t* sorted_full = reinterpret_cast<int32_t*>(ws + 2 * seg);
  void* temp = ws + 3 * seg;
  size_t temp_bytes = workspace_bytes - 3 * seg;
  // The message list is type-major, so a STABLE sort by node id alone already yields (node, type, edge order):
  // only ceil(log2(V)) key bits go through the radix passes instead of ceil(log2(V*L)).  The values are the
  // message indices 0..M-1, fed from a counting iterator (no index array to write and read back).
  (void)iota;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, sort_node, sorted_nodes,
                                           rocprim::counting_iterator<int32_t>(0), perm, (size_t)num_messages, 0,
                                           key_bits(num_nodes), st);
  if (e != hipSuccess) return RELGNN_EHIP;
  plan_finalize_kernel<<<flat_grid(num_messages, 256), 256, 0, st>>>(perm, full_key, other_key, num_messages,
                                                                     num_edge_types, sorted_full, col, col_div, inv_out,
                                                                     inv_in, pos_out);
  rowptr_from_sorted_kernel<<<flat_grid(num_messages + 1, 256), 256, 0, st>>>(sorted_full, num_messages, S, rowptr);
  return launch_status();
}

size_t relgnn_segment_plan_workspace_bytes(int64_t num_messages, int64_t num_segments) {
  if (num_messages <= 0) return 256;
  size_t temp = 0;
  int32_t* nk = nullptr;
  rocprim::radix_sort_pairs(nullptr, temp, nk, nk, nk, nk, (size_t)num_messages, 0,
                            key_bits(num_segments), (hipStream_t)0);
  // layout: [spare | sorted keys scratch | rocprim temp]
  return align_up((size_t)num_messages * 4, 256) * 2 + align_up(temp, 256) + 256;
}

int relgnn_segment_plan(const int32_t* keys, int64_t num_messages, int64_t num_segments,
                        int32_t* rowptr, int32_t* perm, int32_t* sorted_keys, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (num_messages < 0 || num_segments < 0 || !rowptr) return RELGNN_EINVAL;
  if (num_messages > INT32_MAX || num_segments >= INT32_MAX) return RELGNN_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (num_messages == 0) {
    if (hipMemsetAsync(rowptr, 0, (size_t)(num_segments + 1) * 4, st) != hipSuccess) return RELGNN_EHIP;
    return RELGNN_OK;
  }
  if (!keys || !perm || !workspace) return RELGNN_EINVAL;
  if (workspace_bytes < relgnn_segment_plan_workspace_bytes(num_messages, num_segments))
    return RELGNN_ENOSPC;
  char* ws = static_cast<char*>(workspace);
  const size_t seg = align_up((size_t)num_messages * 4, 256);
  int32_t* iota = reinterpret_cast<int32_t*>(ws);
  int32_t* keys_out = sorted_keys ? sorted_keys : reinterpret_cast<int32_t*>(ws + seg);
  void* temp = ws + 2 * seg;
  size_t temp_bytes = workspace_bytes - 2 * seg;

  (void)iota;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, rocprim::counting_iterator<int32_t>(0),
                                           perm, (size_t)num_messages, 0, key_bits(num_segments), st);
  if (e != hipSuccess) return RELGNN_EHIP;
  rowptr_from_sorted_kernel<<<flat_grid(num_messages + 1, 256), 256, 0, st>>>(
      keys_out, num_messages, num_segments, rowptr);
  return launch_status();
}

int relgnn_gather_i32(const int32_t* table, const int32_t* index, int64_t n, int32_t* out,
                      void* stream) {
  if (n < 0) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!table || !index || !out) return RELGNN_EINVAL;
  gather_i32_kernel<<<flat_grid(n, 256), 256, 0, as_stream(stream)>>>(table, index, n, 1, out);
  return launch_status();
}

int relgnn_gather_div_i32(const int32_t* table, const int32_t* index, int64_t n, int32_t divisor,
                          int32_t* out, void* stream) {
  if (n < 0 || divisor <= 0) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!table || !index || !out) return RELGNN_EINVAL;
  gather_i32_kernel<<<flat_grid(n, 256), 256, 0, as_stream(stream)>>>(table, index, n, divisor, out);
  return launch_status();
}

int relgnn_gather_f32(const float* table, const int32_t* index, int64_t n, float* out,
                      void* stream) {
  if (n < 0) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!table || !index || !out) return RELGNN_EINVAL;
  gather_f32_kernel<<<flat_grid(n, 256), 256, 0, as_stream(stream)>>>(table, index, n, out);
  return launch_status();
}

int relgnn_invert_perm(const int32_t* perm, int64_t n, int32_t* inv, void* stream) {
  if (n < 0) return RELGNN_EINVAL;
  if (n == 0) return RELGNN_OK;
  if (!perm || !inv) return RELGNN_EINVAL;
  invert_perm_kernel<<<flat_grid(n, 256), 256, 0, as_stream(stream)>>>(perm, n, inv);
  return launch_status();
}

int relgnn_degree_scale(const float* degree_table, const int32_t* rowptr, int32_t num_edge_types,
                        int32_t num_nodes, float eps, float* scale, void* stream) {
  if (num_edge_types <= 0 || num_nodes < 0) return RELGNN_EINVAL;
  if (num_nodes == 0) return RELGNN_OK;
  if (!degree_table || !rowptr || !scale) return RELGNN_EINVAL;
  degree_scale_kernel<<<flat_grid((int64_t)num_edge_types * num_nodes, 256), 256, 0,
                        as_stream(stream)>>>(degree_table, rowptr, num_edge_types, num_nodes, eps,
                                             scale);
  return launch_status();
}

int relgnn_segment_counts_scale(const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                int32_t mode, const float* w, float* scale, void* stream) {
  if (num_segments < 0 || seg_stride <= 0) return RELGNN_EINVAL;
  if (mode != RELGNN_AGG_SUM && mode != RELGNN_AGG_MEAN && mode != RELGNN_AGG_SQRT_N)
    return RELGNN_EINVAL;
  if (num_segments == 0) return RELGNN_OK;
  if (!rowptr || !scale) return RELGNN_EINVAL;
  counts_scale_kernel<<<flat_grid(num_segments, 256), 256, 0, as_stream(stream)>>>(
      rowptr, num_segments, seg_stride, mode, w, scale);
  return launch_status();
}

}  // extern "C"
