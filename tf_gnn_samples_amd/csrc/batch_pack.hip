// Host-side disjoint-union batch builder (include/relgnn.h section 9).
//
// Replaces the numpy batching loops of the reference (tasks/ppi_task.py:209-256, tasks/qm9_task.py:212-261):
// graphs are appended in the given order, adjacency lists shifted by the running node offset, per-graph degree
// tables / node payloads concatenated along the node axis.  The dataset is flattened ONCE into a store (node and
// edge offset tables + flat arrays); a batch is then a list of graph ids, and packing is pure memcpy / offset-add
// work over a task list that a few host threads drain, writing straight into one pinned arena that goes to the GPU
// as a single copy.  No device code in this file; it lives in librelgnn.so so that one library is the whole boundary.
#include "../../include/relgnn.h"

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int64_t kAlign = 256;          // section alignment inside the arena (bytes)
constexpr int64_t kChunkBytes = 1 << 18; // task granularity

inline int64_t align_up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Task {
    int kind;        // 0 = raw copy, 1 = adjacency (+offset), 2 = fill int32
    const void* src;
    void* dst;
    int64_t n;       // bytes (kind 0) or int32 elements (kind 1, 2)
    int32_t value;   // node offset (kind 1) / fill value (kind 2)
};

void run_task(const Task& t) {
    if (t.kind == 0) {
        std::memcpy(t.dst, t.src, (size_t)t.n);
    } else if (t.kind == 1) {
        const int32_t* s = (const int32_t*)t.src;
        int32_t* d = (int32_t*)t.dst;
        const int32_t off = t.value;
        for (int64_t i = 0; i < t.n; ++i) d[i] = s[i] + off;
    } else {
        int32_t* d = (int32_t*)t.dst;
        for (int64_t i = 0; i < t.n; ++i) d[i] = t.value;
    }
}

void add_chunked(std::vector<Task>& tasks, int kind, const char* src, char* dst, int64_t n, int64_t elem, int32_t value) {
    const int64_t per = kChunkBytes / elem;
    for (int64_t at = 0; at < n; at += per) {
        const int64_t m = (n - at < per) ? (n - at) : per;
        tasks.push_back(Task{kind, src ? src + at * elem : nullptr, dst + at * elem, kind == 0 ? m * elem : m, value});
    }
}

// layout[] slots
enum { LAY_V = 0, LAY_M = 1, LAY_BYTES = 2, LAY_DEG = 3, LAY_N2G = 4, LAY_FIXED = 5 };
// then: payload offsets [n_payloads], adjacency offsets [L], adjacency edge counts [L]

}  // namespace

extern "C" {

int64_t relgnn_batch_layout_len(int32_t num_types, int32_t n_payloads) {
    if (num_types < 0 || n_payloads < 0) return -1;
    return LAY_FIXED + (int64_t)n_payloads + 2 * (int64_t)num_types;
}

int64_t relgnn_batch_count(const int64_t* h_node_off, const int64_t* h_graph_ids, int64_t n_ids, int64_t first,
                           int64_t max_nodes) {
    if (!h_node_off || !h_graph_ids || first < 0 || n_ids < 0) return -1;
    int64_t node_offset = 0, k = first;
    while (k < n_ids) {
        const int64_t g = h_graph_ids[k];
        const int64_t n = h_node_off[g + 1] - h_node_off[g];
        if (!(node_offset + n < max_nodes)) break;   // strict '<', tasks/ppi_task.py:220
        node_offset += n;
        ++k;
    }
    return k - first;
}

int relgnn_batch_layout(int32_t num_types, int64_t n_graphs, const int64_t* h_graph_ids, const int64_t* h_node_off,
                        const int64_t* const* h_edge_off, int32_t n_payloads, const int64_t* h_payload_row_bytes,
                        int64_t* h_layout) {
    if (num_types < 0 || n_graphs < 0 || n_payloads < 0 || !h_layout || !h_node_off) return RELGNN_EINVAL;
    if ((n_graphs > 0 && !h_graph_ids) || (num_types > 0 && !h_edge_off) || (n_payloads > 0 && !h_payload_row_bytes))
        return RELGNN_EINVAL;
    int64_t V = 0, M = 0;
    for (int64_t k = 0; k < n_graphs; ++k) V += h_node_off[h_graph_ids[k] + 1] - h_node_off[h_graph_ids[k]];
    if (V >= (int64_t)INT32_MAX) return RELGNN_EUNSUPPORTED;   // node ids are int32 on the device
    int64_t at = 0;
    for (int32_t p = 0; p < n_payloads; ++p) {
        if (h_payload_row_bytes[p] < 0) return RELGNN_EINVAL;
        h_layout[LAY_FIXED + p] = at;
        at = align_up(at + V * h_payload_row_bytes[p]);
    }
    h_layout[LAY_DEG] = at;
    at = align_up(at + (int64_t)num_types * V * 4);
    h_layout[LAY_N2G] = at;
    at = align_up(at + V * 4);
    for (int32_t l = 0; l < num_types; ++l) {
        int64_t E = 0;
        for (int64_t k = 0; k < n_graphs; ++k) E += h_edge_off[l][h_graph_ids[k] + 1] - h_edge_off[l][h_graph_ids[k]];
        h_layout[LAY_FIXED + n_payloads + l] = at;
        h_layout[LAY_FIXED + n_payloads + num_types + l] = E;
        at = align_up(at + E * 8);
        M += E;
    }
    if (M >= (int64_t)INT32_MAX) return RELGNN_EUNSUPPORTED;
    h_layout[LAY_V] = V;
    h_layout[LAY_M] = M;
    h_layout[LAY_BYTES] = at;
    return RELGNN_OK;
}

int relgnn_batch_pack(int32_t num_types, int64_t n_graphs, const int64_t* h_graph_ids, const int64_t* h_node_off,
                      const int64_t* const* h_edge_off, const int32_t* const* h_adj, const float* const* h_deg,
                      int32_t n_payloads, const void* const* h_payload, const int64_t* h_payload_row_bytes,
                      const int64_t* h_layout, void* h_arena, size_t arena_bytes, int32_t num_threads) {
    if (num_types < 0 || n_graphs < 0 || n_payloads < 0 || !h_layout || !h_node_off) return RELGNN_EINVAL;
    if (n_graphs > 0 && !h_graph_ids) return RELGNN_EINVAL;
    if (num_types > 0 && (!h_edge_off || !h_adj || !h_deg)) return RELGNN_EINVAL;
    if (n_payloads > 0 && (!h_payload || !h_payload_row_bytes)) return RELGNN_EINVAL;
    if ((int64_t)arena_bytes < h_layout[LAY_BYTES]) return RELGNN_ENOSPC;
    if (h_layout[LAY_BYTES] > 0 && !h_arena) return RELGNN_EINVAL;
    const int64_t V = h_layout[LAY_V];
    char* arena = (char*)h_arena;

    std::vector<Task> tasks;
    tasks.reserve((size_t)n_graphs * (size_t)(num_types * 2 + n_payloads + 1) + 16);
    std::vector<int64_t> edge_at((size_t)num_types, 0);
    int64_t node_at = 0;
    for (int64_t k = 0; k < n_graphs; ++k) {
        const int64_t g = h_graph_ids[k];
        const int64_t n0 = h_node_off[g], n = h_node_off[g + 1] - n0;
        for (int32_t p = 0; p < n_payloads; ++p) {
            const int64_t rb = h_payload_row_bytes[p];
            if (rb > 0 && n > 0)
                add_chunked(tasks, 0, (const char*)h_payload[p] + n0 * rb, arena + h_layout[LAY_FIXED + p] + node_at * rb,
                            n * rb, 1, 0);
        }
        for (int32_t l = 0; l < num_types; ++l) {
            // degree tables concatenated along the node axis (tasks/ppi_task.py:237): row l of [L, V]
            if (n > 0)
                add_chunked(tasks, 0, (const char*)(h_deg[l] + n0), arena + h_layout[LAY_DEG] + ((int64_t)l * V + node_at) * 4,
                            n * 4, 1, 0);
            const int64_t e0 = h_edge_off[l][g], E = h_edge_off[l][g + 1] - e0;
            if (E > 0)   // adjacency + node offset (tasks/ppi_task.py:228)
                add_chunked(tasks, 1, (const char*)(h_adj[l] + 2 * e0),
                            arena + h_layout[LAY_FIXED + n_payloads + l] + edge_at[l] * 8, 2 * E, 4, (int32_t)node_at);
            edge_at[l] += E;
        }
        if (n > 0)   // graph_nodes_list (tasks/qm9_task.py:238): index of the graph inside the batch
            add_chunked(tasks, 2, nullptr, arena + h_layout[LAY_N2G] + node_at * 4, n, 4, (int32_t)k);
        node_at += n;
    }
    if (node_at != V) return RELGNN_EINVAL;   // layout belongs to another id list
    for (int32_t l = 0; l < num_types; ++l)
        if (edge_at[l] != h_layout[LAY_FIXED + n_payloads + num_types + l]) return RELGNN_EINVAL;

    int nt = num_threads < 1 ? 1 : num_threads;
    if ((size_t)nt > tasks.size()) nt = (int)(tasks.empty() ? 1 : tasks.size());
    if (nt == 1) {
        for (const Task& t : tasks) run_task(t);
        return RELGNN_OK;
    }
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= tasks.size()) return;
            run_task(tasks[i]);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt - 1);
    for (int i = 1; i < nt; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    return RELGNN_OK;
}

}  // extern "C"
