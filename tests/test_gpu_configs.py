"""BASELINE.json configs as parity-test cases (the bench line is configs[1]; the others are checked here):
  golden  committed fixture tests/golden/layers_small.npz for all six layers
  C3      GGNN on REAL QM9 graphs (tests/golden/qm9_valid_256.jsonl.gz), GRU cell, mean / max aggregation
  C4      RGAT on a PPI-shaped batch, h=256, 4 heads
  C5      GNN-FiLM on a VarMisuse-shaped batch (23 edge types, h=128)
HIP path vs the NumPy oracle on identical inputs and weights.  Tolerance (helpers.assert_parity): 1e-5 ABSOLUTE on node
states for the layers whose states are bounded by construction (RGCN with 1/in-degree normalisation, RGAT, GGNN's GRU
output); for un-normalised sums that grow past O(1) the error is taken relative to max|ref|.  Max-abs and max-rel are
both recorded.  The BASELINE-size cases live in tests/test_gpu_baseline_size.py."""
import numpy as np
import pytest
import torch

from oracle import gnns as G, model as OM
from helpers import assert_parity, glorot, rgcn_weights
from test_golden_cpu import load_layers_fixture, read_qm9_fixture

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


STRICT_ABS = {"rgcn": True, "ggnn": True, "rgat": True, "film": False, "rgin": False, "edge_mlp": False}


def test_golden_fixture_all_layers(gpu_device):
    from tf_gnn_samples_amd import gnns as H
    h, adj, deg, K, w, outs = load_layers_fixture()
    D = h.shape[1]
    hd, ad, dd = _dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device)
    got = {
        "rgcn": H.sparse_rgcn_layer(hd, ad, dd, D, 2, "ReLU", "sum", weights=_dev(w["rgcn"], gpu_device)),
        "ggnn": H.sparse_ggnn_layer(hd, ad, D, 2, "gru", "tanh", "mean", weights=_dev(w["ggnn"], gpu_device)),
        "rgat": H.sparse_rgat_layer(hd, ad, D, K, 2, "tanh", weights=_dev(w["rgat"], gpu_device)),
        "film": H.sparse_gnn_film_layer(hd, ad, dd, D, 2, "ReLU", "sum", weights=_dev(w["film"], gpu_device)),
        "rgin": H.sparse_rgin_layer(hd, ad, D, 2, "ReLU", "sum", weights=_dev(w["rgin"], gpu_device)),
        "edge_mlp": H.sparse_gnn_edge_mlp_layer(hd, ad, dd, D, 2, "gelu", "sum", weights=_dev(w["edge_mlp"], gpu_device)),
    }
    for name, ref in outs.items():
        assert_parity(got[name], ref, strict_abs=STRICT_ABS[name], what="golden/" + name)


def _qm9_batch(max_nodes=3000):
    from tf_gnn_samples_amd.tasks import DataFold, QM9_Task
    task = QM9_Task(QM9_Task.default_params())
    samples = task.load_raw(read_qm9_fixture())
    mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, max_nodes))
    return task, samples, mb


@pytest.mark.parametrize("agg", ["mean", "max", "sum"])
def test_c3_ggnn_qm9_real_graphs(gpu_device, agg):
    from tf_gnn_samples_amd.gnns import sparse_ggnn_layer
    task, samples, mb = _qm9_batch()
    assert task.num_edge_types == 5 and mb.num_graphs > 100
    fd = mb.feed_dict
    rng = np.random.default_rng(0)
    D, L, V = 128, 5, mb.num_nodes
    w = rgcn_weights(rng, L, D, D)
    w.update({"gru_cell/kernel": glorot(rng, (D, 3 * D)), "gru_cell/recurrent_kernel": glorot(rng, (D, 3 * D)),
              "gru_cell/bias": (0.05 * rng.standard_normal(3 * D)).astype(np.float32)})
    h = np.tanh(fd["initial_node_features"].astype(np.float32) @ glorot(rng, (15, D)))
    adj = fd["adjacency_lists"]
    ref = G.sparse_ggnn_layer(h, adj, D, 3, "GRU", "tanh", agg, weights=w)
    out = sparse_ggnn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, 3, "GRU", "tanh", agg, weights=_dev(w, gpu_device))
    assert_parity(out, ref, strict_abs=True, what="C3 ggnn/qm9 " + agg)     # GRU states live in (-1, 1)


def test_c3_ggnn_model_with_qm9_head(gpu_device):
    """GGNN_Model + QM9 task head (per-graph unsorted_segment_sum pooling through the HIP kernel)."""
    from tf_gnn_samples_amd.models import GGNN_Model
    from tf_gnn_samples_amd.tasks import DeviceBatch
    task, samples, mb = _qm9_batch(2000)
    p = GGNN_Model.default_params()
    p.update(hidden_size=128, graph_num_layers=2, graph_rnn_cell="GRU", message_aggregation_function="mean")
    model = GGNN_Model(p, task, device=str(gpu_device))
    batch = DeviceBatch(mb, gpu_device)
    with torch.no_grad():
        final = model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists,
                                                         batch.type_to_num_incoming_edges)
        metrics = model.forward_batch(batch, training=False)
    W = {n[len("graph_model/"):]: model.variables[n].detach().cpu().numpy() for n in model.variables.names()
         if n.startswith("graph_model/")}
    fd = mb.feed_dict

    def apply(layer_idx, hcur, adj, deg, timesteps, lw):
        return G.sparse_ggnn_layer(hcur, adj, p['hidden_size'], timesteps, p['graph_rnn_cell'],
                                   p['graph_activation_function'], p['message_aggregation_function'],
                                   weights={k: v for k, v in lw.items() if k.startswith(("Edge_", "gru_cell"))})
    ref = OM.graph_propagation(fd['initial_node_features'].astype(np.float32), fd['adjacency_lists'],
                               fd['type_to_num_incoming_edges'].astype(np.float32), p, W, apply)
    assert_parity(final, ref, strict_abs=True, what="C3 ggnn model")
    # oracle restatement of the head (tasks/qm9_task.py:176-193)
    assert model._task_scope == "" and "out_layer_task0/regression/dense/kernel" in model.variables   # reference names
    scope = "out_layer_task0/"
    g = lambda n: model.variables[scope + n].detach().cpu().numpy()
    per_node = ref @ g("regression/dense/kernel") + g("regression/dense/bias")
    gate_in = np.concatenate([ref, fd['initial_node_features'].astype(np.float32)], -1)
    gate = 1.0 / (1.0 + np.exp(-(gate_in @ g("regression_gate/dense/kernel") + g("regression_gate/dense/bias"))))
    from oracle import tf_ops as T
    per_graph = T.unsorted_segment_sum((gate * per_node).astype(np.float32), fd['graph_nodes_list'], mb.num_graphs)[:, 0]
    err = per_graph - fd['target_values'][0]
    loss = np.mean(0.5 * err ** 2)
    assert abs(float(metrics['loss']) - float(loss)) < 1e-4 * max(1.0, float(loss))
    losses = [float(model.train_step(batch)['loss'].detach()) for _ in range(5)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_c4_rgat_ppi_shaped(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgat_layer
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(2, 1, seed=2, mean_nodes=700, std_nodes=100, min_nodes=400, max_nodes=1000)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    fd = mb.feed_dict
    rng = np.random.default_rng(1)
    D, L, K, V = 256, 3, 4, mb.num_nodes
    w = rgcn_weights(rng, L, D, D)
    for l in range(L):
        w["Edge_%i_Attention_Parameters" % l] = glorot(rng, (2 * D, 1))[:, 0]
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_rgat_layer(h, fd["adjacency_lists"], D, K, 1, "tanh", weights=w)
    out = sparse_rgat_layer(_dev(h, gpu_device), _dev(fd["adjacency_lists"], gpu_device), D, K, 1, "tanh",
                            weights=_dev(w, gpu_device))
    assert_parity(out, ref, strict_abs=True, what="C4 rgat small")


def test_c5_film_varmisuse_shaped(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_gnn_film_layer
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graphs
    from oracle import bookkeeping
    graphs = make_varmisuse_shaped_graphs(3, seed=0, mean_nodes=900, std_nodes=100, min_nodes=500, max_nodes=1200)
    L = 23
    assert len(graphs[0].adjacency_lists) == L
    ref_samples = [bookkeeping.GraphSample(g.adjacency_lists, g.type_to_node_to_num_incoming_edges, g.node_features, None)
                   for g in graphs]
    b = next(bookkeeping.pack_batches(ref_samples, L, 10 ** 9))
    rng = np.random.default_rng(2)
    D, V = 128, b["num_nodes"]
    w = dict(rgcn_weights(rng, L, D, D), **{"LayerNorm/gamma": np.ones(D, np.float32), "LayerNorm/beta": np.zeros(D, np.float32)})
    for l in range(L):
        w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    adj = [a.astype(np.int32) for a in b["adjacency_lists"]]
    deg = b["type_to_num_incoming_edges"].astype(np.float32)
    ref = G.sparse_gnn_film_layer(h, adj, deg, D, 1, "ReLU", "sum", False, weights=w)
    out = sparse_gnn_film_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 1, "ReLU", "sum", False,
                                weights=_dev(w, gpu_device))
    assert_parity(out, ref, strict_abs=False, what="C5 film small")


@pytest.mark.parametrize("residual_every", [2, 10000])
def test_driver_loop_dropout_sits_where_the_reference_puts_it(gpu_device, monkeypatch, residual_every):
    """models/sparse_graph_model.py:178-179: tf.nn.dropout on every layer's INPUT, before the residual average, x / keep_prob where the
    mask keeps (the shipped PPI hyper-parameters train with keep 0.8-0.9; VERDICT r04 missing 5: outside every fixture).  TF's random
    draws cannot be reproduced, so the draws are replaced on both sides by the same recorded masks: the driver loop's
    torch.nn.functional.dropout is patched to apply them, the oracle takes them as an argument.  Node states within 1e-5 abs; the
    folded activation gradients must stand aside (a dropped-out tensor is another tensor: no tag)."""
    import torch.nn.functional as F
    from oracle import model as OM
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(2, 1, seed=4, mean_nodes=300, std_nodes=40, min_nodes=100, max_nodes=400, fwd_edges_per_node=5.0)
    p = RGCN_Model.default_params()
    p.update(hidden_size=64, graph_num_layers=4, graph_residual_connection_every_num_layers=residual_every,
             graph_dense_between_every_num_gnn_layers=2, graph_layer_input_dropout_keep_prob=0.8)
    model = RGCN_Model(p, task, device=str(gpu_device))
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 6))
    batch = DeviceBatch(mb, gpu_device)
    rng = np.random.default_rng(0)
    masks = [rng.random((mb.num_nodes, 64)) < 0.8 for _ in range(4)]
    calls = []

    def recorded_dropout(x, p=0.5, training=True, inplace=False):
        assert training and abs(p - 0.2) < 1e-12
        m = torch.as_tensor(masks[len(calls)], device=x.device)
        calls.append(tuple(x.shape))
        return x / 0.8 * m

    monkeypatch.setattr(F, "dropout", recorded_dropout)
    x = batch.initial_node_features.clone().requires_grad_(True)
    final = model.compute_final_node_representations(x, batch.adjacency_lists, batch.type_to_num_incoming_edges, dropout_keep_prob=0.8)
    assert len(calls) == 4
    W = {n[len("graph_model/"):]: model.variables[n].detach().cpu().numpy() for n in model.variables.names() if n.startswith("graph_model/")}
    fd = mb.feed_dict
    want = OM.graph_propagation(fd['initial_node_features'].astype(np.float32), fd['adjacency_lists'],
                                fd['type_to_num_incoming_edges'].astype(np.float32), p, W, OM.rgcn_apply(p), dropout=(0.8, masks))
    assert float(np.abs(final.detach().cpu().numpy() - want).max()) <= 1e-5
    final.square().sum().backward()                       # and the backward runs (dropout outputs carry no activation tag)
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    # evaluation: keep-prob 1 is the exact identity (no dropout call at all)
    calls.clear()
    with torch.no_grad():
        model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists, batch.type_to_num_incoming_edges)
    assert calls == []
