"""Shared input builders for the tests (seeded, small enough for the NumPy oracle)."""
import numpy as np

from oracle import bookkeeping


def random_relational_graph(rng, num_nodes, num_edge_types, edges_per_type, empty_types=(), heavy_tail=True):
    """L adjacency lists [E_l, 2] int32 with duplicates, self edges and (optionally) empty types."""
    adj = []
    for l in range(num_edge_types):
        e = 0 if l in empty_types else int(edges_per_type if np.isscalar(edges_per_type) else edges_per_type[l])
        src = rng.integers(0, num_nodes, size=e)
        if heavy_tail and e:
            w = rng.lognormal(0.0, 1.0, size=num_nodes)
            tgt = rng.choice(num_nodes, size=e, p=w / w.sum())
        else:
            tgt = rng.integers(0, num_nodes, size=e)
        adj.append(np.stack([src, tgt], axis=1).astype(np.int32).reshape(-1, 2))
    return adj


def glorot(rng, shape):
    limit = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def rgcn_weights(rng, num_edge_types, in_dim, out_dim):
    return {"Edge_%i_Weight/kernel" % l: glorot(rng, (in_dim, out_dim)) for l in range(num_edge_types)}


def degree_table(adj, num_nodes):
    return bookkeeping.in_degree_table(adj, num_nodes)
