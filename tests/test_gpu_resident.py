"""Data fold resident in HBM (tasks/resident.py, relgnn_plan_assemble): every batch tensor equals the host packer's and
every bucketing array equals a freshly built RelGraph of that batch, bit for bit."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from oracle.bookkeeping import GraphSample

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PLAN_ARRAYS = ("key_by_target", "key_by_source", "rowptr_t", "perm_t", "col_t", "inv_perm_t", "rowptr_s", "perm_s",
               "frow_s", "tgt_s", "pos_t_of_s")


def _graphs(rng, n_graphs, L, max_nodes=70):
    out = []
    for g in range(n_graphs):
        n = int(rng.integers(1, max_nodes))
        adj = []
        for l in range(L):
            e = 0 if (l == L - 1 and g % 2 == 0) or (g == 3) else int(rng.integers(0, 5 * n))
            adj.append(np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64).reshape(-1, 2))
        deg = np.stack([np.bincount(a[:, 1], minlength=n) for a in adj])
        out.append(GraphSample(adj, deg, rng.standard_normal((n, 9)).astype(np.float32),
                               (rng.random((n, 4)) < 0.3).astype(np.float32)))
    return out


def _same_batch(a, b):
    assert (a.num_graphs, a.num_nodes, a.num_edges) == (b.num_graphs, b.num_nodes, b.num_edges)
    assert torch.equal(a.initial_node_features, b.initial_node_features)
    assert torch.equal(a.type_to_num_incoming_edges, b.type_to_num_incoming_edges)
    assert torch.equal(a.graph_nodes_list, b.graph_nodes_list)
    for x, y in zip(a.adjacency_lists, b.adjacency_lists):
        assert x.dtype == torch.int32 and x.shape == y.shape and torch.equal(x, y)
    for k in a.extra:
        if torch.is_tensor(a.extra[k]):
            assert torch.equal(a.extra[k], b.extra[k]), k


def _same_plan(graph, adjacency_lists, V):
    from tf_gnn_samples_amd.graph import RelGraph
    fresh = RelGraph(adjacency_lists, V)
    for name in PLAN_ARRAYS + ("key_by_target", "key_by_source", "src_t"):      # keys are computed lazily; src_t is preset
        assert torch.equal(getattr(graph, name), getattr(fresh, name)), name
    assert (graph.V, graph.L, graph.M, graph.edge_counts) == (fresh.V, fresh.L, fresh.M, fresh.edge_counts)


def _same_scales(batch):
    """The per-message 1/(in-degree + 1e-7) scales copied from the fold-level arrays == what RelGraph computes from the
    batch's own degree table, in by-target and by-source order, bit for bit."""
    from tf_gnn_samples_amd import _lib
    from tf_gnn_samples_amd.graph import RelGraph
    g = batch.graph
    deg = batch.type_to_num_incoming_edges
    w = g.degree_scale(deg)                                         # preset by tasks/resident.py
    fresh = RelGraph(batch.adjacency_lists, batch.num_nodes)
    w_ref = fresh.degree_scale(deg.clone())
    assert torch.equal(w, w_ref)
    assert torch.equal(g.plan_transformed(w).w_bwd(_lib.AGG_SUM), fresh.plan_transformed(w_ref).w_bwd(_lib.AGG_SUM))
    assert torch.equal(g.w_by_source(w), fresh.w_by_source(w_ref))


def _same_pair_tables(graph, adjacency_lists, V):
    """A resident fold counts the non-empty (node, type) buckets of every graph once; a batch's RelGraph carries their sums
    (pair_counts) and graph.PairTables then builds the compact numbering without reading anything back from the device: the same
    tables, array for array, as the ones built from counts read back."""
    from tf_gnn_samples_amd.graph import PairTables, RelGraph
    fresh = RelGraph(adjacency_lists, V)
    assert getattr(fresh, "pair_counts", None) is None and graph.pair_counts is not None
    want, got = PairTables(fresh), PairTables(graph)
    for side in ("tgt", "src"):
        w, g_ = getattr(want, side), getattr(got, side)
        assert list(graph.pair_counts[0 if side == "tgt" else 1]) == w.type_counts == g_.type_counts
        assert (w.P, w.num_pairs, w.offsets, w.chunk_counts) == (g_.P, g_.num_pairs, g_.offsets, g_.chunk_counts)
        for name in ("bucket_row", "node", "node_col", "node_rowptr", "pad_rows", "chunk_type"):
            assert torch.equal(getattr(w, name), getattr(g_, name)), (side, name)
        for a_, b_ in zip(w.weight_grad_plan(4), g_.weight_grad_plan(4)):
            assert torch.equal(a_, b_)
    assert torch.equal(want.col_t, got.col_t) and torch.equal(want.frow_s, got.frow_s)
    assert fresh.wants_pair_tables() == graph.wants_pair_tables()


PAYLOADS = {"initial_node_features": ("node_features", np.float32), "target_labels": ("node_labels", np.float32)}


@pytest.mark.parametrize("L", [1, 3, 12])
def test_resident_batches_and_plans_are_bit_identical(gpu_device, L):
    from tf_gnn_samples_amd.tasks.batcher import GraphStore, NativeBatcher
    from tf_gnn_samples_amd.tasks.resident import ResidentDataset
    rng = np.random.default_rng(L)
    graphs = _graphs(rng, 30, L)
    store = GraphStore(graphs, L, PAYLOADS)
    resident = ResidentDataset(store, gpu_device)
    host = NativeBatcher(store, gpu_device, bucket=False)
    for ids in ([0], [5, 2, 2, 17], list(range(30)), [3], [29, 0, 3, 4]):          # order, repeats, an edge-free graph
        b = host.pack(np.array(ids))
        for lean in (True, False):
            a = resident.assemble(np.array(ids), lean=lean)
            assert (a.graph.__dict__.get("_complete") is not None) == lean      # lean: permutation arrays deferred
            assert torch.equal(a.graph.src_t, torch.cat([x[:, 0] for x in b.adjacency_lists])[a.graph.perm_t.long()]) \
                if a.graph.M else True
            _same_batch(a, b)
            _same_plan(a.graph, b.adjacency_lists, b.num_nodes)
            _same_scales(a)
            if L >= 8 and lean:
                _same_pair_tables(a.graph, b.adjacency_lists, b.num_nodes)
    got = [x.num_graphs for x in resident.iterate(np.arange(30), 200)]
    want = [x.num_graphs for x in host.iterate(np.arange(30), 200)]
    assert got == want and sum(got) == 30


def test_resident_qm9_with_per_graph_targets_and_training(gpu_device):
    from tf_gnn_samples_amd.models import GGNN_Model
    from tf_gnn_samples_amd.tasks import DataFold, QM9_Task
    from tf_gnn_samples_amd.tasks.batcher import NativeBatcher
    from tf_gnn_samples_amd.tasks.resident import ResidentDataset
    task = QM9_Task(QM9_Task.default_params())
    with gzip.open(os.path.join(HERE, "golden", "qm9_valid_256.jsonl.gz"), "rt") as f:
        data = task.load_raw([json.loads(line) for line in f])
    store = task.make_graph_store(data)
    resident = ResidentDataset(store, gpu_device)
    host = NativeBatcher(store, gpu_device, bucket=False)
    ids = np.array([7, 200, 13, 13, 99, 0])
    a, b = resident.assemble(ids), host.pack(ids)
    _same_batch(a, b)
    _same_plan(a.graph, b.adjacency_lists, b.num_nodes)
    # the model sees the same epoch through either pipeline
    task._loaded_data[DataFold.TRAIN] = data
    task._loaded_data[DataFold.VALIDATION] = data[:64]
    outs = []
    for mode in (True, False):
        params = GGNN_Model.default_params()
        params.update(hidden_size=32, graph_num_layers=2, max_nodes_in_batch=1500, resident_dataset=mode, random_seed=0)
        model = GGNN_Model(params, task, device=gpu_device)
        loss, res, n, *_ = model._run_epoch("valid", data[:64], DataFold.VALIDATION, quiet=True)
        outs.append((loss, n, [r['abs_err_task0'] for r in res]))
    assert outs[0] == outs[1]


def test_lean_batches_stay_lean_through_an_rgcn_step_and_train_identically(gpu_device):
    """The default (lean) assembly defers the adjacency lists and the six permutation-type arrays of the bucketing; the
    RGCN sum path must never ask for them, and must train bit-identically to the fully assembled batch."""
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    from tf_gnn_samples_amd.tasks.resident import ResidentDataset
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(6, 1, seed=2, mean_nodes=200, std_nodes=40, min_nodes=80, max_nodes=300, fwd_edges_per_node=5.0)
    store = task.make_graph_store(task._loaded_data[DataFold.TRAIN])
    resident = ResidentDataset(store, gpu_device, constants={})
    ids = np.array([4, 1, 5])
    results = []
    for lean in (True, False):
        params = RGCN_Model.default_params()
        params.update(hidden_size=64, graph_num_layers=2, graph_layer_input_dropout_keep_prob=1.0, random_seed=1)
        model = RGCN_Model(params, task, device=str(gpu_device))
        batch = task._finish_native_batch(resident.assemble(ids, lean=lean))
        losses = [float(model.train_step(batch)['loss'].detach()) for _ in range(3)]
        if lean:
            assert batch.graph.__dict__.get("_complete") is not None, "the RGCN step read a deferred array"
            assert callable(batch._adjacency), "the RGCN step read the adjacency lists"
        results.append((losses, {n: model.variables[n].detach().clone() for n in model.variables.names()}))
    assert results[0][0] == results[1][0]
    for n, v in results[0][1].items():
        assert torch.equal(v, results[1][1][n]), n
    # and the deferred pieces, once read, are the real ones
    lean_batch, full_batch = resident.assemble(ids), resident.assemble(ids, lean=False)
    for x, y in zip(lean_batch.adjacency_lists, full_batch.adjacency_lists):
        assert torch.equal(x, y)
    for name in ("perm_t", "col_t", "inv_perm_t", "perm_s", "frow_s", "pos_t_of_s"):
        assert torch.equal(getattr(lean_batch.graph, name), getattr(full_batch.graph, name)), name