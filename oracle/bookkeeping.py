"""NumPy restatement of the index bookkeeping (bit-exact integer work).

TEST INFRASTRUCTURE (oracle/__init__.py).  Two parts:
  * the reference's own batching (disjoint-union packing of graphs), restated from
    tasks/ppi_task.py:209-256 / tasks/qm9_task.py:212-261;
  * the (target, type)/(source, type) bucketing the HIP path builds on the device, restated
    with a stable NumPy argsort so the device result can be compared bit for bit.
"""
from typing import List, NamedTuple, Sequence

import numpy as np


class GraphSample(NamedTuple):
    """tasks/ppi_task.py:12-18 / tasks/qm9_task.py:13-17 (fields used by the hot path)."""
    adjacency_lists: List[np.ndarray]                  # L x [E_l, 2] int, node ids local to the graph
    type_to_node_to_num_incoming_edges: np.ndarray     # [L, V_g]
    node_features: np.ndarray                          # [V_g, F]
    node_labels: np.ndarray = None


def pack_batches(data: Sequence[GraphSample], num_edge_types: int, max_nodes_per_batch: int):
    """Disjoint-union batching, restated from tasks/ppi_task.py:209-256:
    graphs are appended while node_offset + |V_g| < max_nodes_per_batch (:220; strict '<');
    adjacency lists are shifted by node_offset (:228); degree tables concatenated on axis 1 (:237);
    an edge type without edges becomes zeros((0, 2), int32) (:248-249);
    num_edges = sum_l E_l (:244-250).  Yields dicts."""
    num_graphs = 0
    while num_graphs < len(data):
        feats, labels, gnl, degs = [], [], [], []
        adj = [[] for _ in range(num_edge_types)]
        node_offset = 0
        n_in_batch = 0
        while num_graphs < len(data) and node_offset + len(data[num_graphs].node_features) < max_nodes_per_batch:
            g = data[num_graphs]
            n = len(g.node_features)
            feats.extend(g.node_features)
            gnl.append(np.full(shape=[n], fill_value=n_in_batch, dtype=np.int32))
            for i in range(num_edge_types):
                adj[i].append(g.adjacency_lists[i] + node_offset)
            degs.append(g.type_to_node_to_num_incoming_edges)
            if g.node_labels is not None:
                labels.append(g.node_labels)
            num_graphs += 1
            n_in_batch += 1
            node_offset += n
        if n_in_batch == 0:
            # the reference would spin forever on a graph that never fits (ppi_task.py:217-220)
            raise ValueError("graph %d does not fit into max_nodes_per_batch=%d" % (num_graphs, max_nodes_per_batch))
        num_edges = 0
        merged = []
        for i in range(num_edge_types):
            if len(adj[i]) > 0:
                a = np.concatenate(adj[i])
            else:
                a = np.zeros((0, 2), dtype=np.int32)
            num_edges += a.shape[0]
            merged.append(a)
        yield dict(initial_node_features=np.array(feats),
                   type_to_num_incoming_edges=np.concatenate(degs, axis=1),
                   graph_nodes_list=np.concatenate(gnl),
                   target_labels=np.concatenate(labels, axis=0) if labels else None,
                   adjacency_lists=merged, num_graphs=n_in_batch, num_nodes=node_offset, num_edges=num_edges)


def relational_buckets(adjacency_lists, num_nodes: int):
    """What RelGraph builds on the device (tf_gnn_samples_amd/graph.py), via np.argsort(kind='stable').

    Message list = type-major concatenation of the adjacency lists (gnns/rgcn.py:78,108).
      key_by_target[m] = tgt*L + l ; key_by_source[m] = src*L + l
      perm_t = stable argsort(key_by_target); rowptr_t[s] = #messages with key < s
      col_t = key_by_source[perm_t]
      perm_s, rowptr_s likewise for key_by_source; tgt_s = target node, frow_s = key_by_target[perm_s]
      pos_t_of_s[q] = position in the by-target order of the message at by-source position q
    """
    L = len(adjacency_lists)
    adj = [np.asarray(a, dtype=np.int64).reshape(-1, 2) for a in adjacency_lists]
    types = np.concatenate([np.full(a.shape[0], l, dtype=np.int64) for l, a in enumerate(adj)]) if L else np.zeros(0, np.int64)
    src = np.concatenate([a[:, 0] for a in adj])
    tgt = np.concatenate([a[:, 1] for a in adj])
    if src.size and (min(src.min(), tgt.min()) < 0 or max(src.max(), tgt.max()) >= num_nodes):
        raise ValueError("node id out of range")
    key_t = tgt * L + types
    key_s = src * L + types
    S = num_nodes * L
    perm_t = np.argsort(key_t, kind='stable')
    perm_s = np.argsort(key_s, kind='stable')
    rowptr_t = np.concatenate([[0], np.cumsum(np.bincount(key_t, minlength=S))])
    rowptr_s = np.concatenate([[0], np.cumsum(np.bincount(key_s, minlength=S))])
    inv_t = np.empty_like(perm_t)
    inv_t[perm_t] = np.arange(perm_t.size)
    i32 = lambda a: np.asarray(a, dtype=np.int32)
    return dict(key_by_target=i32(key_t), key_by_source=i32(key_s),
                rowptr_t=i32(rowptr_t), perm_t=i32(perm_t), col_t=i32(key_s[perm_t]), src_t=i32(src[perm_t]),
                rowptr_s=i32(rowptr_s), perm_s=i32(perm_s), tgt_s=i32(tgt[perm_s]), frow_s=i32(key_t[perm_s]),
                pos_t_of_s=i32(inv_t[perm_s]), inv_perm_t=i32(inv_t))


def in_degree_table(adjacency_lists, num_nodes: int) -> np.ndarray:
    """type_to_num_incoming_edges[l, v] = #edges of type l into v (tasks/ppi_task.py:126-148), float32
    as it is fed (tasks/sparse_graph_task.py:144-145)."""
    return np.stack([np.bincount(np.asarray(a).reshape(-1, 2)[:, 1], minlength=num_nodes)
                     for a in adjacency_lists]).astype(np.float32)


def qm9_graph_to_adjacency_lists(graph, num_nodes, num_edge_types, add_self_loop_edges=True, tie_fwd_bkwd_edges=True):
    """Restatement of tasks/qm9_task.py:114-147 (raw triples (src, e, dst), e in 1..4):
    list-append in triple order, both directions when tied, self loops appended last on type 0,
    every list sorted lexicographically (:135); untied: the forward half, then one reversed list per forward type
    (:137-145).  Returns (adjacency lists, in-degree table [L, V])."""
    lists = [[] for _ in range(num_edge_types)]
    deg = np.zeros(shape=(num_edge_types, num_nodes))
    for src, e, dest in graph:
        fwd = e if add_self_loop_edges else e - 1
        lists[fwd].append((src, dest))
        deg[fwd, dest] += 1
        if tie_fwd_bkwd_edges:
            lists[fwd].append((dest, src))
            deg[fwd, src] += 1
    if add_self_loop_edges:
        for node in range(num_nodes):
            deg[0, node] = 1
            lists[0].append((node, node))
    adj = [np.array(sorted(a), dtype=np.int32) if len(a) > 0 else np.zeros(shape=(0, 2), dtype=np.int32) for a in lists]
    if not tie_fwd_bkwd_edges:                                                          # :137-145
        adj = adj[:num_edge_types // 2]
        for edge_type, a in enumerate(list(adj)):
            bwd = num_edge_types // 2 + edge_type
            adj.append(np.array(sorted((y, x) for (x, y) in a), dtype=np.int32).reshape(-1, 2))
            for (x, y) in a:
                # :145 of the reference increments the count at y, the TARGET of the forward edge, although the
                # reversed edge (y -> x) lands on x.  Restated as written: the table is fed to the model as is.
                deg[bwd][y] += 1
    return adj, deg


def ppi_graphs_from_dgl_arrays(links, node_to_features, node_to_labels, node_to_graph_id,
                               add_self_loop_edges=True, tie_fwd_bkwd_edges=False):
    """Restatement of PPI_Task.__load_data, tasks/ppi_task.py:92-162, on the already-read file contents
    ({fold}_graph.json['links'], _feats.npy, _labels.npy, _graph_id.npy): edge type 0 = forward links, then the
    self-loop type, then the backward type (:99-106); graphs in order of first appearance of their id (:112-121);
    node ids shifted by the graph's first node id (:117,139-142); an edge belongs to the graph of its SOURCE node
    (:140); self loops ascend by node id (:124-126); per-type in-degree tables counted while reading (:143-147).
    Returns [(adjacency lists, in-degree table [L, V_g], features, labels)] per graph."""
    fwd_t, n_types = 0, 1
    self_t = bkwd_t = None
    if add_self_loop_edges:
        self_t, n_types = n_types, n_types + 1
    if not tie_fwd_bkwd_edges:
        bkwd_t, n_types = n_types, n_types + 1
    graphs, offset = {}, {}
    for node_id in range(node_to_features.shape[0]):
        gid = node_to_graph_id[node_id]
        if gid not in graphs:
            graphs[gid] = dict(adj=[[] for _ in range(n_types)], deg=[[] for _ in range(n_types)], feats=[], labels=[])
            offset[gid] = node_id
        g = graphs[gid]
        g["feats"].append(node_to_features[node_id])
        g["labels"].append(node_to_labels[node_id])
        shifted = node_id - offset[gid]
        if add_self_loop_edges:
            g["adj"][self_t].append((shifted, shifted))
            g["deg"][self_t].append(1)
    for g in graphs.values():
        n = len(g["feats"])
        g["deg"][fwd_t] = np.zeros([n], np.int32)
        if not tie_fwd_bkwd_edges:
            g["deg"][bkwd_t] = np.zeros([n], np.int32)
    for edge in links:
        src, tgt = edge['source'], edge['target']
        gid = node_to_graph_id[src]
        src, tgt = src - offset[gid], tgt - offset[gid]
        g = graphs[gid]
        g["adj"][fwd_t].append((src, tgt))
        g["deg"][fwd_t][tgt] += 1
        if not tie_fwd_bkwd_edges:
            g["adj"][bkwd_t].append((tgt, src))
            g["deg"][bkwd_t][src] += 1
    return [([np.array(a) for a in g["adj"]], np.array(g["deg"]), np.array(g["feats"]), np.array(g["labels"]))
            for g in graphs.values()]
