"""Synthetic, dataset-shaped graph generators (no network / dataset on the box).

PPI-shaped (SURVEY.md 8d "C2"; statistics derived from the reference's README.md:34 —
129 362 edges and 2 245 nodes per graph over three edge types, i.e. 28.3 forward edges/node):
  per graph  V_g ~ clip(N(2245, 700), 600, 3500);  E_fwd = round(28.3 * V_g);
  sources uniform; targets drawn proportionally to log-normal(sigma=0.9) node weights
  (heavy-tailed in-degree; duplicate and self edges allowed: multigraph);
  edge types as tasks/ppi_task.py:99-106 with add_self_loop_edges=True, tie_fwd_bkwd_edges=False:
    [0] forward, [1] self loops (ascending node id), [2] backward = forward reversed, same order;
  in-degree tables as tasks/ppi_task.py:126-148; features N(0,1) [V_g, 50]; labels Bernoulli(0.3) [V_g, 121].

VarMisuse-shaped (SURVEY.md 8d "C5"; tasks/varmisuse_task.py:22-28,244-247 for the type list:
11 base edge types x {fwd, bkwd} + self loops = 23): program-graph-like, type sizes skewed
(NextToken / Child chains dominate, most other types are sparse).
"""
from typing import List, NamedTuple

import numpy as np


class GraphSample(NamedTuple):
    adjacency_lists: List[np.ndarray]
    type_to_node_to_num_incoming_edges: np.ndarray
    node_features: np.ndarray
    node_labels: np.ndarray


def _in_degrees(adj: List[np.ndarray], n: int) -> np.ndarray:
    return np.stack([np.bincount(a[:, 1], minlength=n) for a in adj]).astype(np.int32)


def make_ppi_shaped_graphs(num_graphs: int = 16, seed: int = 0, feature_size: int = 50, num_labels: int = 121,
                           mean_nodes: float = 2245.0, std_nodes: float = 700.0, min_nodes: int = 600,
                           max_nodes: int = 3500, fwd_edges_per_node: float = 28.3,
                           target_lognormal_sigma: float = 0.9) -> List[GraphSample]:
    rng = np.random.default_rng(seed)
    graphs = []
    for _ in range(num_graphs):
        n = int(np.clip(np.round(rng.normal(mean_nodes, std_nodes)), min_nodes, max_nodes))
        e = int(round(fwd_edges_per_node * n))
        src = rng.integers(0, n, size=e, dtype=np.int64)
        wts = rng.lognormal(mean=0.0, sigma=target_lognormal_sigma, size=n)
        tgt = rng.choice(n, size=e, p=wts / wts.sum())
        fwd = np.stack([src, tgt], axis=1).astype(np.int32)
        self_loops = np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.int32)
        bkwd = np.ascontiguousarray(fwd[:, ::-1])
        adj = [fwd, self_loops, bkwd]
        graphs.append(GraphSample(
            adjacency_lists=adj,
            type_to_node_to_num_incoming_edges=_in_degrees(adj, n),
            node_features=rng.standard_normal((n, feature_size)).astype(np.float32),
            node_labels=(rng.random((n, num_labels)) < 0.3).astype(np.float32)))
    return graphs


def ppi_shaped_generator_params(**overrides):
    p = dict(num_graphs=16, seed=0, feature_size=50, num_labels=121, mean_nodes=2245.0, std_nodes=700.0,
             min_nodes=600, max_nodes=3500, fwd_edges_per_node=28.3, target_lognormal_sigma=0.9)
    p.update(overrides)
    return p


VARMISUSE_BASE_EDGE_TYPES = 11  # tasks/varmisuse_task.py:22-28


def make_varmisuse_shaped_graphs(num_graphs: int, seed: int = 0, feature_size: int = 128, mean_nodes: float = 2500.0,
                                 std_nodes: float = 600.0, min_nodes: int = 500, max_nodes: int = 5000,
                                 edges_per_node: float = 4.7) -> List[GraphSample]:
    """23 edge types: base type b -> fwd index b, bkwd index 11 + b, self loops index 22."""
    rng = np.random.default_rng(seed)
    # share of (forward) edges per base type: two chain-like types dominate, the rest are sparse
    share = np.array([0.30, 0.30, 0.10, 0.08, 0.06, 0.05, 0.04, 0.03, 0.02, 0.01, 0.01])
    graphs = []
    nb = VARMISUSE_BASE_EDGE_TYPES
    for _ in range(num_graphs):
        n = int(np.clip(np.round(rng.normal(mean_nodes, std_nodes)), min_nodes, max_nodes))
        e_total = int(round(edges_per_node * n))
        fwd_lists = []
        for b in range(nb):
            e = int(round(share[b] * e_total))
            if b < 2:  # chain-like: i -> i + small offset
                s = rng.integers(0, n, size=e)
                t = np.minimum(s + rng.integers(1, 4, size=e), n - 1)
            else:
                s = rng.integers(0, n, size=e)
                t = rng.integers(0, n, size=e)
            fwd_lists.append(np.stack([s, t], axis=1).astype(np.int32))
        adj = fwd_lists + [np.ascontiguousarray(a[:, ::-1]) for a in fwd_lists]
        adj.append(np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.int32))
        graphs.append(GraphSample(
            adjacency_lists=adj,
            type_to_node_to_num_incoming_edges=_in_degrees(adj, n),
            node_features=rng.standard_normal((n, feature_size)).astype(np.float32),
            node_labels=np.zeros((n, 1), np.float32)))
    return graphs


# ---- one graph at a time, seeded per graph index -----------------------------------------------------------------------
# A data-parallel rank needs only ITS graphs: with one generator stream for the whole fold (above) every rank would have to
# build all of them to reach its own.  Here graph i draws from default_rng([seed, i]); its node count is the first draw, so the
# edge counts of the whole fold (what the by-edge sharding needs) cost one draw per graph.
def _ppi_nodes(rng, mean_nodes, std_nodes, min_nodes, max_nodes):
    return int(np.clip(np.round(rng.normal(mean_nodes, std_nodes)), min_nodes, max_nodes))


def ppi_shaped_graph_size(seed: int, index: int, mean_nodes: float = 2245.0, std_nodes: float = 700.0, min_nodes: int = 600,
                          max_nodes: int = 3500, fwd_edges_per_node: float = 28.3, **_):
    """(nodes, edges over all three types) of graph `index` without building it."""
    n = _ppi_nodes(np.random.default_rng([seed, index]), mean_nodes, std_nodes, min_nodes, max_nodes)
    return n, 2 * int(round(fwd_edges_per_node * n)) + n


def make_ppi_shaped_graph(seed: int, index: int, feature_size: int = 50, num_labels: int = 121, mean_nodes: float = 2245.0,
                          std_nodes: float = 700.0, min_nodes: int = 600, max_nodes: int = 3500,
                          fwd_edges_per_node: float = 28.3, target_lognormal_sigma: float = 0.9) -> GraphSample:
    """Same distribution as make_ppi_shaped_graphs, one graph, its own generator stream."""
    rng = np.random.default_rng([seed, index])
    n = _ppi_nodes(rng, mean_nodes, std_nodes, min_nodes, max_nodes)
    e = int(round(fwd_edges_per_node * n))
    src = rng.integers(0, n, size=e, dtype=np.int64)
    wts = rng.lognormal(mean=0.0, sigma=target_lognormal_sigma, size=n)
    tgt = rng.choice(n, size=e, p=wts / wts.sum())
    fwd = np.stack([src, tgt], axis=1).astype(np.int32)
    adj = [fwd, np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.int32), np.ascontiguousarray(fwd[:, ::-1])]
    return GraphSample(adjacency_lists=adj, type_to_node_to_num_incoming_edges=_in_degrees(adj, n),
                       node_features=rng.standard_normal((n, feature_size)).astype(np.float32),
                       node_labels=(rng.random((n, num_labels)) < 0.3).astype(np.float32))


_VM_SHARE = np.array([0.30, 0.30, 0.10, 0.08, 0.06, 0.05, 0.04, 0.03, 0.02, 0.01, 0.01])


def varmisuse_shaped_graph_size(seed: int, index: int, mean_nodes: float = 2500.0, std_nodes: float = 600.0,
                                min_nodes: int = 500, max_nodes: int = 5000, edges_per_node: float = 4.7, **_):
    n = _ppi_nodes(np.random.default_rng([seed, index]), mean_nodes, std_nodes, min_nodes, max_nodes)
    e_total = int(round(edges_per_node * n))
    return n, 2 * int(sum(int(round(s * e_total)) for s in _VM_SHARE)) + n


def make_varmisuse_shaped_graph(seed: int, index: int, feature_size: int = 128, mean_nodes: float = 2500.0,
                                std_nodes: float = 600.0, min_nodes: int = 500, max_nodes: int = 5000,
                                edges_per_node: float = 4.7) -> GraphSample:
    """Same distribution as make_varmisuse_shaped_graphs (23 edge types), one graph, its own generator stream."""
    rng = np.random.default_rng([seed, index])
    n = _ppi_nodes(rng, mean_nodes, std_nodes, min_nodes, max_nodes)
    e_total = int(round(edges_per_node * n))
    fwd_lists = []
    for b in range(VARMISUSE_BASE_EDGE_TYPES):
        e = int(round(_VM_SHARE[b] * e_total))
        s = rng.integers(0, n, size=e)
        t = np.minimum(s + rng.integers(1, 4, size=e), n - 1) if b < 2 else rng.integers(0, n, size=e)
        fwd_lists.append(np.stack([s, t], axis=1).astype(np.int32))
    adj = fwd_lists + [np.ascontiguousarray(a[:, ::-1]) for a in fwd_lists]
    adj.append(np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.int32))
    return GraphSample(adjacency_lists=adj, type_to_node_to_num_incoming_edges=_in_degrees(adj, n),
                       node_features=rng.standard_normal((n, feature_size)).astype(np.float32),
                       node_labels=(rng.random((n, 1)) < 0.3).astype(np.float32))
