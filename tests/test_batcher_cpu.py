"""Native batch builder (include/relgnn.h section 9) against the NumPy restatement of the reference's batching
(oracle/bookkeeping.pack_batches <- tasks/ppi_task.py:209-256): bit-exact integer bookkeeping, CPU only (the packer
is host code; the upload half is covered by tests/test_gpu_batcher.py)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from oracle import bookkeeping
from oracle.bookkeeping import GraphSample
from tf_gnn_samples_amd.tasks.batcher import GraphStore, NativeBatcher

HERE = os.path.dirname(os.path.abspath(__file__))


def _random_graphs(rng, n_graphs, L, feat=7, labels=3, max_nodes=60):
    graphs = []
    for g in range(n_graphs):
        n = int(rng.integers(1, max_nodes))
        adj = []
        for l in range(L):
            e = 0 if (l == L - 1 and g % 2 == 0) else int(rng.integers(0, 4 * n))   # empty lists on purpose
            adj.append(np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64).reshape(-1, 2))
        deg = np.stack([np.bincount(a[:, 1], minlength=n) for a in adj])
        graphs.append(GraphSample(adjacency_lists=adj, type_to_node_to_num_incoming_edges=deg,
                                  node_features=rng.standard_normal((n, feat)).astype(np.float32),
                                  node_labels=(rng.random((n, labels)) < 0.3).astype(np.float32)))
    return graphs


PAYLOADS = {"initial_node_features": ("node_features", np.float32), "target_labels": ("node_labels", np.float32)}


def _check_batches(graphs, L, max_nodes, threads):
    store = GraphStore(graphs, L, PAYLOADS)
    nb = NativeBatcher(store, "cpu", num_threads=threads)
    ids = np.arange(len(graphs))
    got = list(nb.iterate(ids, max_nodes))
    want = list(bookkeeping.pack_batches(graphs, L, max_nodes))
    assert len(got) == len(want)
    for b, w in zip(got, want):
        assert (b.num_graphs, b.num_nodes, b.num_edges) == (w["num_graphs"], w["num_nodes"], w["num_edges"])
        assert np.array_equal(b.initial_node_features.numpy(), w["initial_node_features"])
        assert np.array_equal(b.extra["target_labels"].numpy(), w["target_labels"])
        assert np.array_equal(b.type_to_num_incoming_edges.numpy(), w["type_to_num_incoming_edges"].astype(np.float32))
        assert np.array_equal(b.graph_nodes_list.numpy(), w["graph_nodes_list"])
        for l in range(L):
            a = b.adjacency_lists[l].numpy()
            assert a.dtype == np.int32 and a.shape == (len(w["adjacency_lists"][l]), 2)
            assert np.array_equal(a, w["adjacency_lists"][l])
    return got


@pytest.mark.parametrize("threads", [1, 4])
def test_packing_matches_reference_batching_bit_for_bit(threads):
    graphs = _random_graphs(np.random.default_rng(0), 40, 4)
    _check_batches(graphs, 4, 150, threads)


def test_single_batch_and_one_graph_per_batch_limits():
    graphs = _random_graphs(np.random.default_rng(1), 12, 3, max_nodes=20)
    assert len(_check_batches(graphs, 3, 10 ** 6, 2)) == 1
    biggest = max(len(g.node_features) for g in graphs)
    _check_batches(graphs, 3, biggest + 1, 2)          # strict '<': a graph of n nodes needs max_nodes > n


def test_strict_less_than_rule_and_graph_that_never_fits():
    graphs = _random_graphs(np.random.default_rng(2), 5, 2, max_nodes=30)
    store = GraphStore(graphs, 2, PAYLOADS)
    ids = np.arange(5, dtype=np.int64)
    n0 = len(graphs[0].node_features)
    assert store.count_fitting(ids, 0, n0) == 0        # n0 < n0 is false (tasks/ppi_task.py:220)
    assert store.count_fitting(ids, 0, n0 + 1) >= 1
    with pytest.raises(ValueError):
        store.split_batches(ids, n0)
    with pytest.raises(ValueError):
        list(bookkeeping.pack_batches(graphs, 2, n0))


def test_graph_order_is_the_id_order_and_ids_may_repeat():
    graphs = _random_graphs(np.random.default_rng(3), 6, 2)
    store = GraphStore(graphs, 2, PAYLOADS)
    nb = NativeBatcher(store, "cpu", num_threads=3)
    order = [4, 1, 1, 5, 0]
    b = nb.pack(np.array(order))
    w = next(bookkeeping.pack_batches([graphs[i] for i in order], 2, 10 ** 9))
    for l in range(2):
        assert np.array_equal(b.adjacency_lists[l].numpy(), w["adjacency_lists"][l])
    assert np.array_equal(b.initial_node_features.numpy(), w["initial_node_features"])
    assert np.array_equal(b.graph_nodes_list.numpy(), w["graph_nodes_list"])


def test_empty_batch_and_edge_free_graphs():
    rng = np.random.default_rng(4)
    graphs = [GraphSample([np.zeros((0, 2), np.int64)] * 2, np.zeros((2, n), np.int64),
                          rng.standard_normal((n, 7)).astype(np.float32), np.zeros((n, 3), np.float32)) for n in (3, 1, 5)]
    got = _check_batches(graphs, 2, 100, 2)
    assert got[0].num_edges == 0 and all(a.shape == (0, 2) for a in got[0].adjacency_lists)
    store = GraphStore(graphs, 2, PAYLOADS)
    lay = store.layout(np.zeros(0, np.int64))
    assert lay[0] == 0 and lay[1] == 0


def test_arena_too_small_and_mismatched_layout_are_refused():
    graphs = _random_graphs(np.random.default_rng(5), 6, 2)
    store = GraphStore(graphs, 2, PAYLOADS)
    ids = np.arange(6, dtype=np.int64)
    lay = store.layout(ids)
    arena = np.zeros(int(lay[2]), np.uint8)
    with pytest.raises(RuntimeError):
        store.pack_into(ids, lay, arena.ctypes.data, int(lay[2]) - 1, 1)        # RELGNN_ENOSPC
    with pytest.raises(ValueError):
        store.pack_into(ids[:5], lay, arena.ctypes.data, int(lay[2]), 1)        # layout of another id list


def test_qm9_golden_batches_with_per_graph_targets():
    """Real QM9 graphs (tests/golden/qm9_valid_256.jsonl.gz) through the QM9 bookkeeping, packed natively, compared with
    the task's own numpy iterator."""
    from tf_gnn_samples_amd.tasks import QM9_Task, DataFold
    task = QM9_Task(QM9_Task.default_params())
    with gzip.open(os.path.join(HERE, "golden", "qm9_valid_256.jsonl.gz"), "rt") as f:
        raw = [json.loads(line) for line in f]
    data = task.load_raw(raw)
    store = task.make_graph_store(data)
    nb = NativeBatcher(store, "cpu", num_threads=2, constants={"out_layer_dropout_keep_prob": 1.0})
    got = list(task.make_native_minibatch_iterator(nb, DataFold.VALIDATION, 1000))
    want = list(task.make_minibatch_iterator(data, DataFold.VALIDATION, 1000))
    assert len(got) == len(want) > 2
    for b, mb in zip(got, want):
        fd = mb.feed_dict
        assert (b.num_graphs, b.num_nodes, b.num_edges) == (mb.num_graphs, mb.num_nodes, mb.num_edges)
        assert np.array_equal(b.initial_node_features.numpy(), fd["initial_node_features"])
        assert np.array_equal(b.extra["target_values"].numpy(), fd["target_values"])
        assert np.array_equal(b.graph_nodes_list.numpy(), fd["graph_nodes_list"])
        assert np.array_equal(b.type_to_num_incoming_edges.numpy(), fd["type_to_num_incoming_edges"].astype(np.float32))
        for l in range(task.num_edge_types):
            assert np.array_equal(b.adjacency_lists[l].numpy(), fd["adjacency_lists"][l])


def test_iterator_can_be_abandoned_and_reports_packing_errors():
    import threading
    graphs = _random_graphs(np.random.default_rng(7), 30, 3)
    store = GraphStore(graphs, 3, PAYLOADS)
    nb = NativeBatcher(store, "cpu", num_threads=2)
    before = threading.active_count()
    it = nb.iterate(np.arange(30), 120)
    first = next(it)
    assert first.num_graphs >= 1
    it.close()                                   # consumer walks away: the producer thread must end
    assert threading.active_count() == before
    # all batches again after an abandoned run (arenas are reused)
    assert sum(b.num_graphs for b in nb.iterate(np.arange(30), 120)) == 30
    with pytest.raises(ValueError):              # a graph that never fits is reported before any packing starts
        list(nb.iterate(np.arange(30), 1))
