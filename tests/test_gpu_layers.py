"""Layer-level parity: tf_gnn_samples_amd.gnns.sparse_*_layer (HIP path) vs the NumPy oracle on
the same seeded inputs and weights; tolerance = the north-star's 1e-5 abs on fp32 node states.
Gradients are checked against torch autograd through the reference-order torch mirror (fp64)."""
import numpy as np
import pytest
import torch

from oracle import gnns as G, model as OM, torch_ref as R
from helpers import assert_parity, degree_table, glorot, layer_norm_weights, random_relational_graph, rgcn_weights, set_switch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _close(out, ref, tol=TOL, strict_abs=False, what=""):
    """helpers.assert_parity as a predicate.  strict_abs=True (layers whose states are bounded by construction:
    1/in-degree-normalised RGCN, RGAT): north-star 1e-5 ABS whatever max|ref| is; otherwise abs where max|ref| <= 1
    and relative to max|ref| where an un-normalised sum has grown past O(1) (SURVEY.md section 7 'hard parts')."""
    try:
        assert_parity(out, ref, strict_abs=strict_abs, what=what, tol=tol)
    except AssertionError:
        return False
    return True


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


def _graph(seed, V=150, L=3, E=(900, 150, 0)):
    """type 0: random heavy-tailed edges; type 1: self loops on every node plus some random edges (so that
    every node has >= 1 incoming message: an empty max-segment is float32 lowest and would overflow in the
    next timestep's matmul in reference and port alike); type 2: empty."""
    rng = np.random.default_rng(seed)
    adj = random_relational_graph(rng, V, L, list(E))
    loops = np.stack([np.arange(V), np.arange(V)], 1).astype(np.int32)
    adj[1] = np.concatenate([loops, adj[1]]).astype(np.int32)
    return rng, adj, degree_table(adj, V)


def _grad_check(hip_fn, ref_fn, h, weights, dev, tol=TOL):
    """d(sum(out * G))/d(h, weights) for a fixed random G: HIP autograd vs fp64 torch-CPU autograd."""
    hd = torch.as_tensor(h, device=dev).requires_grad_(True)
    wd = {k: torch.as_tensor(v, device=dev).requires_grad_(True) for k, v in weights.items()}
    out = hip_fn(hd, wd)
    gout = np.random.default_rng(0).standard_normal(out.shape).astype(np.float32)
    out.backward(torch.as_tensor(gout, device=dev))
    hr = torch.as_tensor(h, dtype=torch.float64).requires_grad_(True)
    wr = {k: torch.as_tensor(v, dtype=torch.float64).requires_grad_(True) for k, v in weights.items()}
    ref = ref_fn(hr, wr)
    ref.backward(torch.as_tensor(gout, dtype=torch.float64))
    assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < tol
    pairs = [("h", hd.grad, hr.grad)] + [(k, wd[k].grad, wr[k].grad) for k in weights]
    for name, a, b in pairs:
        if b is None:
            continue
        scale = max(1.0, float(b.abs().max()))
        err = float(np.abs(a.cpu().numpy() - b.numpy()).max())
        assert err < tol * scale * 4, (name, err, scale)


@pytest.mark.parametrize("agg", ["sum", "mean", "max", "sqrt_n"])
@pytest.mark.parametrize("norm", [True, False])
def test_rgcn_layer_forward(gpu_device, agg, norm):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    rng, adj, deg = _graph(1)
    V, D = 150, 64
    w = rgcn_weights(rng, 3, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    for act in ["tanh", "ReLU", "gelu", None]:
        ref = G.sparse_rgcn_layer(h, adj, deg, D, 2, act, agg, norm, weights=w)
        out = sparse_rgcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 2, act, agg, norm,
                                weights=_dev(w, gpu_device))
        assert _close(out, ref, strict_abs=norm and agg != "max", what="rgcn %s norm=%s %s" % (agg, norm, act)), (agg, norm, act)


def test_rgcn_layer_changes_dimension(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    rng, adj, deg = _graph(2)
    V, Din, D = 150, 50, 256
    w = rgcn_weights(rng, 3, Din, D)
    h = rng.standard_normal((V, Din)).astype(np.float32)
    ref = G.sparse_rgcn_layer(h, adj, deg, D, weights=w)
    out = sparse_rgcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, weights=_dev(w, gpu_device))
    assert out.shape == (V, D) and _close(out, ref)


@pytest.mark.parametrize("agg", ["sum", "max"])
def test_rgcn_layer_gradients(gpu_device, agg):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    rng, adj, deg = _graph(3)
    V, D = 150, 32
    w = rgcn_weights(rng, 3, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    _grad_check(lambda x, ww: sparse_rgcn_layer(x, adj_d, deg_d, D, 2, "tanh", agg, weights=ww),
                lambda x, ww: R.sparse_rgcn_layer(x, adj_c, deg_c, D, 2, "tanh", agg, weights=ww),
                h, w, gpu_device)


@pytest.mark.parametrize("cell", ["GRU", "RNN"])
@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_ggnn_layer(gpu_device, cell, agg):
    from tf_gnn_samples_amd.gnns import sparse_ggnn_layer
    rng, adj, deg = _graph(4)
    V, D = 150, 128
    scope = {"GRU": "gru_cell", "RNN": "simple_rnn_cell"}[cell]
    g = 3 if cell == "GRU" else 1
    w = rgcn_weights(rng, 3, D, D)
    w.update({scope + "/kernel": glorot(rng, (D, g * D)), scope + "/recurrent_kernel": glorot(rng, (D, g * D)),
              scope + "/bias": (rng.standard_normal(g * D) * 0.1).astype(np.float32)})
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    if agg != "sum":   # keep un-normalised sums of O(degree) messages out of the 1e-5 budget
        pass
    else:
        for l in range(3):
            w["Edge_%i_Weight/kernel" % l] *= np.float32(0.2)
    ref = G.sparse_ggnn_layer(h, adj, D, 2, cell, "tanh", agg, weights=w)
    out = sparse_ggnn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, 2, cell, "tanh", agg, weights=_dev(w, gpu_device))
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    adj_d = _dev(adj, gpu_device)
    adj_c = [torch.as_tensor(a) for a in adj]
    _grad_check(lambda x, ww: sparse_ggnn_layer(x, adj_d, D, 1, cell, "tanh", agg, weights=ww),
                lambda x, ww: R.sparse_ggnn_layer(x, adj_c, D, 1, cell, "tanh", agg, weights=ww),
                h, w, gpu_device)


def test_ggnn_unknown_cell_and_lstm(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_ggnn_layer
    rng, adj, deg = _graph(5, V=10, E=(20, 5, 0))
    h = torch.zeros((10, 8), device=gpu_device)
    w = _dev(rgcn_weights(rng, 3, 8, 8), gpu_device)
    with pytest.raises(Exception, match="Unknown RNN cell type"):
        sparse_ggnn_layer(h, _dev(adj, gpu_device), 8, gated_unit_type="foo", weights=w)
    with pytest.raises(NotImplementedError):
        sparse_ggnn_layer(h, _dev(adj, gpu_device), 8, gated_unit_type="LSTM", weights=w)


def test_rgcn_model_end_to_end_vs_oracle(gpu_device):
    """RGCN_Model (driver loop + adapter) == oracle graph_propagation + PPI head; parameter count pin."""
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(3, 1, seed=3, mean_nodes=300, std_nodes=50, min_nodes=100, max_nodes=500, fwd_edges_per_node=8.0)
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3)
    model = RGCN_Model(p, task, device=str(gpu_device))
    assert model.variables.num_parameters() == 699257   # README.md:29
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 6))
    batch = DeviceBatch(mb, gpu_device)
    with torch.no_grad():
        final = model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists,
                                                         batch.type_to_num_incoming_edges)
        metrics = model.forward_batch(batch, training=False)
    W = {k[len("graph_model/"):]: v.detach().cpu().numpy() for k, v in
         ((n, model.variables[n]) for n in model.variables.names()) if k.startswith("graph_model/")}
    fd = mb.feed_dict
    ref = OM.graph_propagation(fd['initial_node_features'].astype(np.float32), fd['adjacency_lists'],
                               fd['type_to_num_incoming_edges'].astype(np.float32), p, W, OM.rgcn_apply(p))
    assert np.abs(final.cpu().numpy() - ref).max() < TOL
    loss, _ = OM.ppi_head_loss(ref, fd['target_labels'], model.variables["dense_1/kernel"].detach().cpu().numpy(),
                               model.variables["dense_1/bias"].detach().cpu().numpy())
    assert abs(float(metrics['loss']) - float(loss)) < 1e-4 * max(1.0, float(loss))
    # one training step runs and changes the weights
    before = model.variables["graph_model/gnn_layer_1/Edge_0_Weight/kernel"].detach().clone()
    m = model.train_step(batch)
    assert torch.isfinite(m['loss'])
    assert not torch.equal(before, model.variables["graph_model/gnn_layer_1/Edge_0_Weight/kernel"].detach())


# ---------------------------------------------------------------------------------------------
# RGAT / GNN-FiLM / GNN-Edge-MLP / RGIN / RGCN(use_both) — fused edge kernels
# ---------------------------------------------------------------------------------------------
LN = lambda D: layer_norm_weights(D, 2)     # identity parameters for up to two timesteps


def _mlp_weights(rng, name, d_in, d_out, hidden, scale=1.0):
    dims = [d_in] + [d_out] * hidden + [d_out]
    names = ["dense" if i == 0 else "dense_%i" % i for i in range(hidden + 1)]
    return {"%s/%s/kernel" % (name, n): glorot(rng, (dims[i], dims[i + 1])) * np.float32(scale) for i, n in enumerate(names)}


@pytest.mark.parametrize("D,K", [(256, 4), (128, 4), (320, 4), (64, 8), (32, 1)])
def test_rgat_layer(gpu_device, D, K):
    from tf_gnn_samples_amd.gnns import sparse_rgat_layer
    rng, adj, deg = _graph(11)
    V, L = 150, 3
    w = rgcn_weights(rng, L, D, D)
    for l in range(L):
        w["Edge_%i_Attention_Parameters" % l] = (rng.standard_normal(2 * D) * 0.3).astype(np.float32)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_rgat_layer(h, adj, D, K, 2, "tanh", weights=w)
    out = sparse_rgat_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, K, 2, "tanh", weights=_dev(w, gpu_device))
    assert _close(out, ref, strict_abs=True, what="rgat D=%d K=%d" % (D, K))
    adj_d, adj_c = _dev(adj, gpu_device), [torch.as_tensor(a) for a in adj]
    _grad_check(lambda x, ww: sparse_rgat_layer(x, adj_d, D, K, 1, "tanh", weights=ww),
                lambda x, ww: R.sparse_rgat_layer(x, adj_c, D, K, 1, "tanh", weights=ww), h, w, gpu_device)


def test_rgat_fused_sums_switch(gpu_device, monkeypatch):
    """RELGNN_RGAT_FUSED_SUMS=0: the two score-table gradients as separate gather-reduces over dz instead of riding along with the
    dz pass and the by-source gather — the same gradients."""
    from tf_gnn_samples_amd.gnns import sparse_rgat_layer
    rng, adj, deg = _graph(12)
    V, L, D, K = 150, 3, 256, 4
    w = rgcn_weights(rng, L, D, D)
    for l in range(L):
        w["Edge_%i_Attention_Parameters" % l] = (rng.standard_normal(2 * D) * 0.3).astype(np.float32)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    gout = torch.as_tensor(rng.standard_normal((V, D)).astype(np.float32), device=gpu_device)
    grads = []
    for flag in ("1", "0"):
        set_switch(monkeypatch, "RELGNN_RGAT_FUSED_SUMS", flag)
        hd = torch.as_tensor(h, device=gpu_device).requires_grad_(True)
        wd = {k: torch.as_tensor(v, device=gpu_device).requires_grad_(True) for k, v in w.items()}
        sparse_rgat_layer(hd, _dev(adj, gpu_device), D, K, 1, "tanh", weights=wd).backward(gout)
        grads.append([hd.grad] + [wd[k].grad for k in sorted(wd)])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


def test_rgat_nodes_without_incoming_edges(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgat_layer
    rng = np.random.default_rng(12)
    V, L, D, K = 40, 2, 64, 4
    adj = [np.array([[0, 1], [2, 1], [3, 1], [1, 0]], np.int32), np.zeros((0, 2), np.int32)]
    w = rgcn_weights(rng, L, D, D)
    for l in range(L):
        w["Edge_%i_Attention_Parameters" % l] = (rng.standard_normal(2 * D) * 0.3).astype(np.float32)
    h = rng.standard_normal((V, D)).astype(np.float32)
    ref = G.sparse_rgat_layer(h, adj, D, K, 1, None, weights=w)
    out = sparse_rgat_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, K, 1, None, weights=_dev(w, gpu_device))
    assert _close(out, ref) and float(out[5].abs().max()) == 0.0


@pytest.mark.parametrize("agg", ["sum", "mean", "sqrt_n", "max"])
@pytest.mark.parametrize("D,norm,act", [(128, False, "ReLU"), (256, True, "tanh"), (64, False, "gelu"), (320, True, "elu")])
def test_gnn_film_layer(gpu_device, agg, D, norm, act):
    from tf_gnn_samples_amd.gnns import sparse_gnn_film_layer
    rng, adj, deg = _graph(13)
    V, L = 150, 3
    w = dict(rgcn_weights(rng, L, D, D), **LN(D))
    for l in range(L):
        w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_gnn_film_layer(h, adj, deg, D, 2, act, agg, norm, weights=w)
    out = sparse_gnn_film_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 2, act, agg, norm,
                                weights=_dev(w, gpu_device))
    assert _close(out, ref, 2e-5 if act == "gelu" else TOL)
    if D <= 128:
        adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
        adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
        _grad_check(lambda x, ww: sparse_gnn_film_layer(x, adj_d, deg_d, D, 1, act, agg, norm, weights=ww),
                    lambda x, ww: R.sparse_gnn_film_layer(x, adj_c, deg_c, D, 1, act, agg, norm, weights=ww),
                    h, w, gpu_device, tol=3e-5)


@pytest.mark.parametrize("hidden", [0, 1, 2])
@pytest.mark.parametrize("use_target,norm,agg", [(True, False, "sum"), (True, True, "mean"), (False, False, "sum"),
                                                   (True, False, "max")])
def test_gnn_edge_mlp_layer(gpu_device, hidden, use_target, norm, agg):
    from tf_gnn_samples_amd.gnns import sparse_gnn_edge_mlp_layer
    rng, adj, deg = _graph(14)
    V, L, D = 150, 3, 128
    w = dict(LN(D))
    for l in range(L):
        w.update(_mlp_weights(rng, "Edge_%i_MLP" % l, 2 * D if use_target else D, D, hidden))
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_gnn_edge_mlp_layer(h, adj, deg, D, 2, "gelu", agg, norm, use_target, hidden, weights=w)
    out = sparse_gnn_edge_mlp_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 2, "gelu", agg,
                                    norm, use_target, hidden, weights=_dev(w, gpu_device))
    assert _close(out, ref, 2e-5)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    _grad_check(lambda x, ww: sparse_gnn_edge_mlp_layer(x, adj_d, deg_d, D, 1, "gelu", agg, norm, use_target, hidden, weights=ww),
                lambda x, ww: R.sparse_gnn_edge_mlp_layer(x, adj_c, deg_c, D, 1, "gelu", agg, norm, use_target, hidden, weights=ww),
                h, w, gpu_device, tol=3e-5)


@pytest.mark.parametrize("edge_hidden,aggr_hidden,use_target", [(1, None, False), (0, None, False), (None, 1, False),
                                                                 (1, 0, True), (0, None, True), (None, 1, True)])
@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_rgin_layer(gpu_device, edge_hidden, aggr_hidden, use_target, agg):
    from tf_gnn_samples_amd.gnns import sparse_rgin_layer
    rng, adj, deg = _graph(15)
    V, L, D = 150, 3, 64
    w = dict(LN(D))
    d_in = 2 * D if use_target else D
    if edge_hidden is not None:
        for l in range(L):
            w.update(_mlp_weights(rng, "Edge_%i_MLP" % l, d_in, D, edge_hidden, 0.5))
    if aggr_hidden is not None:
        w.update(_mlp_weights(rng, "Aggregation_MLP", D if edge_hidden is not None else d_in, D, aggr_hidden, 0.5))
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    kw = dict(use_target_state_as_input=use_target, num_edge_MLP_hidden_layers=edge_hidden,
              num_aggr_MLP_hidden_layers=aggr_hidden)
    ref = G.sparse_rgin_layer(h, adj, D, 2, "ReLU", agg, weights=w, **kw)
    out = sparse_rgin_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, 2, "ReLU", agg, weights=_dev(w, gpu_device), **kw)
    assert _close(out, ref, 2e-5)
    adj_d, adj_c = _dev(adj, gpu_device), [torch.as_tensor(a) for a in adj]
    _grad_check(lambda x, ww: sparse_rgin_layer(x, adj_d, D, 1, "tanh", agg, weights=ww, **kw),
                lambda x, ww: R.sparse_rgin_layer(x, adj_c, D, 1, "tanh", agg, weights=ww, **kw), h, w, gpu_device, tol=3e-5)


def test_rgin_docstring_graphs_on_gpu(gpu_device):
    """gnns/rgin.py:29-35: G1 and G2 differ only in which edge TYPE carries which edge."""
    from tf_gnn_samples_amd.gnns import sparse_rgin_layer
    rng = np.random.default_rng(1)
    D = 8
    h = rng.standard_normal((3, D)).astype(np.float32)
    w = dict(LN(D))
    for l in range(2):
        w.update(_mlp_weights(rng, "Edge_%i_MLP" % l, D, D, 1))
    g1 = [np.array([[0, 1]], np.int32), np.array([[2, 1]], np.int32)]
    g2 = [np.array([[2, 1]], np.int32), np.array([[0, 1]], np.int32)]
    o1 = sparse_rgin_layer(_dev(h, gpu_device), _dev(g1, gpu_device), D, weights=_dev(w, gpu_device))
    o2 = sparse_rgin_layer(_dev(h, gpu_device), _dev(g2, gpu_device), D, weights=_dev(w, gpu_device))
    assert _close(o1, G.sparse_rgin_layer(h, g1, D, weights=w)) and _close(o2, G.sparse_rgin_layer(h, g2, D, weights=w))
    assert float((o1[1] - o2[1]).abs().max()) > 1e-3


@pytest.mark.parametrize("agg", ["sum", "max"])
def test_rgcn_use_both_source_and_target(gpu_device, agg):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    rng, adj, deg = _graph(16)
    V, L, D = 150, 3, 64
    w = rgcn_weights(rng, L, 2 * D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_rgcn_layer(h, adj, deg, D, 2, "tanh", agg, True, True, weights=w)
    out = sparse_rgcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 2, "tanh", agg, True, True,
                            weights=_dev(w, gpu_device))
    assert _close(out, ref)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    _grad_check(lambda x, ww: sparse_rgcn_layer(x, adj_d, deg_d, D, 1, "tanh", agg, True, True, weights=ww),
                lambda x, ww: R.sparse_rgcn_layer(x, adj_c, deg_c, D, 1, "tanh", agg, True, True, weights=ww),
                h, w, gpu_device)


@pytest.mark.parametrize("model_name", ["GGNN", "RGAT", "RGIN", "GNN-FiLM", "GNN-Edge-MLP0", "GNN-Edge-MLP1"])
def test_every_model_trains_a_step(gpu_device, model_name):
    """Driver loop + adapter + HIP kernels: loss is finite and decreases over a few steps on one batch."""
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(2, 1, seed=5, mean_nodes=200, std_nodes=30, min_nodes=80, max_nodes=300, fwd_edges_per_node=6.0)
    cls, extra = name_to_model_class(model_name)
    p = cls.default_params()
    p.update(extra)
    p.update(hidden_size=64, graph_num_layers=2, learning_rate=0.01)
    model = cls(p, task, device=str(gpu_device))
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 6))
    batch = DeviceBatch(mb, gpu_device)
    losses = [float(model.train_step(batch)['loss'].detach()) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("full_state,tie", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("agg,norm", [("sum", True), ("mean", False)])
def test_rgdcn_layer(gpu_device, full_state, tie, agg, norm):
    from tf_gnn_samples_amd.gnns import sparse_rgdcn_layer
    rng, adj, deg = _graph(31)
    V, L, C, K = 150, 3, 4, 8
    D = C * K
    w = {}
    for l in range(L):
        for c in range(1 if tie else C):
            w["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, c)] = \
                (rng.standard_normal((D if full_state else K, K * K)) * 0.3).astype(np.float32)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_rgdcn_layer(h, adj, deg, C, K, 2, full_state, tie, "tanh", agg, norm, weights=w)
    out = sparse_rgdcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), C, K, 2, full_state, tie,
                             "tanh", agg, norm, weights=_dev(w, gpu_device))
    assert _close(out, ref)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    _grad_check(lambda x, ww: sparse_rgdcn_layer(x, adj_d, deg_d, C, K, 1, full_state, tie, "tanh", agg, norm, weights=ww),
                lambda x, ww: R.sparse_rgdcn_layer(x, adj_c, deg_c, C, K, 1, full_state, tie, "tanh", agg, norm, weights=ww),
                h, w, gpu_device, tol=2e-5)


@pytest.mark.parametrize("full_state,tie,norm", [(False, False, True), (True, True, False)])
def test_rgdcn_max_aggregation(gpu_device, full_state, tie, norm):
    """max does not commute with the per-target kernels: the layer evaluates the reference's per-message einsum
    (gnns/rgdcn.py:140-159 with tf.unsorted_segment_max) and must agree with the oracle, gradients included."""
    from tf_gnn_samples_amd.gnns import sparse_rgdcn_layer
    rng, adj, deg = _graph(33)
    V, L, C, K = 150, 3, 4, 8
    D = C * K
    w = {}
    for l in range(L):
        for c in range(1 if tie else C):
            w["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, c)] = \
                (rng.standard_normal((D if full_state else K, K * K)) * 0.3).astype(np.float32)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_rgdcn_layer(h, adj, deg, C, K, 2, full_state, tie, "tanh", "max", norm, weights=w)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    out = sparse_rgdcn_layer(_dev(h, gpu_device), adj_d, deg_d, C, K, 2, full_state, tie, "tanh", "max", norm,
                             weights=_dev(w, gpu_device))
    assert _close(out, ref, strict_abs=True, what="rgdcn max")
    adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    _grad_check(lambda x, ww: sparse_rgdcn_layer(x, adj_d, deg_d, C, K, 1, full_state, tie, "tanh", "max", norm, weights=ww),
                lambda x, ww: R.sparse_rgdcn_layer(x, adj_c, deg_c, C, K, 1, full_state, tie, "tanh", "max", norm, weights=ww),
                h, w, gpu_device, tol=2e-5)


@pytest.mark.parametrize("K,act", [(16, "ReLU"), (12, "tanh"), (8, "gelu")])
def test_rgdcn_kernel_and_library_paths(gpu_device, K, act):
    """K a power of two + an activation whose derivative is recoverable from the output: the HIP apply kernels
    (relgnn_rgdcn_apply_fwd/bwd); otherwise (K = 12, gelu) the same arithmetic through library ops."""
    from tf_gnn_samples_amd.gnns import sparse_rgdcn_layer
    rng, adj, deg = _graph(34)
    V, L, C = 150, 3, 3
    D = C * K
    w = {"Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, c): (rng.standard_normal((K, K * K)) * 0.3).astype(np.float32)
         for l in range(L) for c in range(C)}
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    for agg in ("sum", "mean", "sqrt_n"):
        ref = G.sparse_rgdcn_layer(h, adj, deg, C, K, 1, False, False, act, agg, True, weights=w)
        out = sparse_rgdcn_layer(_dev(h, gpu_device), adj_d, deg_d, C, K, 1, False, False, act, agg, True,
                                 weights=_dev(w, gpu_device))
        assert _close(out, ref, what="rgdcn K=%d %s %s" % (K, act, agg))
    if act != "ReLU":     # (kink at 0: finite-precision gradient comparisons are meaningful for smooth activations)
        adj_c, deg_c = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
        _grad_check(lambda x, ww: sparse_rgdcn_layer(x, adj_d, deg_d, C, K, 1, False, False, act, "mean", True, weights=ww),
                    lambda x, ww: R.sparse_rgdcn_layer(x, adj_c, deg_c, C, K, 1, False, False, act, "mean", True, weights=ww),
                    h, w, gpu_device, tol=2e-5)


def test_rgdcn_model_trains(gpu_device):
    from tf_gnn_samples_amd.models import RGDCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(2, 1, seed=5, mean_nodes=150, std_nodes=20, min_nodes=80, max_nodes=250, fwd_edges_per_node=5.0)
    p = RGDCN_Model.default_params()
    p.update(hidden_size=64, num_channels=4, graph_num_layers=2, learning_rate=0.01)
    model = RGDCN_Model(p, task, device=str(gpu_device))
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 6))
    batch = DeviceBatch(mb, gpu_device)
    losses = [float(model.train_step(batch)['loss'].detach()) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("D", [128, 256])
@pytest.mark.parametrize("layer", ["film", "edge_mlp0", "edge_mlp1"])
def test_edge_backward_regather_variant_matches(gpu_device, monkeypatch, layer, D):
    """The by-source backward has two implementations: gather-reduce of per-message gradients emitted by the by-target
    pass and the pass that re-gathers the per-bucket rows per message (relgnn_film_bwd_msg / relgnn_pair_bwd_p);
    ops._FusedEdgeMessages picks by geometry, RELGNN_EDGE_BWD=emit|regather forces one.  Same gradients from both, for
    the lane-group kernels (D = 128) and the wave kernels (D = 256, both MU variants)."""
    from tf_gnn_samples_amd.gnns import sparse_gnn_edge_mlp_layer, sparse_gnn_film_layer
    rng, adj, deg = _graph(21)
    V, L = 150, 3
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    if layer == "film":
        w = dict(rgcn_weights(rng, L, D, D), **LN(D))
        for l in range(L):
            w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
        fn = lambda x, ww, a, d: sparse_gnn_film_layer(x, a, d, D, 1, "tanh", "mean", True, weights=ww)
    elif layer == "edge_mlp0":
        w = dict({"Edge_%i_MLP/dense/kernel" % l: glorot(rng, (2 * D, D)) for l in range(L)}, **LN(D))
        fn = lambda x, ww, a, d: sparse_gnn_edge_mlp_layer(x, a, d, D, 1, "elu", "sum", True, True, 0, weights=ww)
    else:   # one hidden layer: the first Dense's pre-activation gradient (ops._PairMaterialize) has the same two routes
        w = dict({"Edge_%i_MLP/dense/kernel" % l: glorot(rng, (2 * D, D)) for l in range(L)}, **LN(D))
        w.update({"Edge_%i_MLP/dense_1/kernel" % l: glorot(rng, (D, D)) for l in range(L)})
        fn = lambda x, ww, a, d: sparse_gnn_edge_mlp_layer(x, a, d, D, 1, "gelu", "sum", True, True, 1, weights=ww)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    grads = []
    for flag in ("emit", "regather"):
        set_switch(monkeypatch, "RELGNN_EDGE_BWD", flag)
        hd = torch.as_tensor(h, device=gpu_device).requires_grad_(True)
        wd = {k: torch.as_tensor(v, device=gpu_device).requires_grad_(True) for k, v in w.items()}
        out = fn(hd, wd, adj_d, deg_d)
        out.backward(torch.as_tensor(np.random.default_rng(2).standard_normal(out.shape).astype(np.float32), device=gpu_device))
        grads.append([hd.grad] + [wd[k].grad for k in sorted(wd)])
    for a, b in zip(*grads):
        if a is None or b is None:          # the second timestep's LayerNorm scope is not touched by a 1-step layer
            assert a is None and b is None
            continue
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_edge_free_batch_through_every_rgcn_order(gpu_device, monkeypatch):
    """A batch without a single edge (tasks/ppi_task.py:248-249: every type zeros((0, 2))) is legal: every node's
    aggregate is the empty sum, the layer returns activation(0) — with and without 1/in-degree normalisation, in both
    evaluation orders, forward and backward."""
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    from tf_gnn_samples_amd.graph import clear_graph_cache
    rng = np.random.default_rng(0)
    V, D, L = 7, 64, 3
    adj = [torch.zeros((0, 2), dtype=torch.int32, device=gpu_device) for _ in range(L)]
    deg = torch.zeros((L, V), device=gpu_device)
    w = {k: torch.as_tensor(v, device=gpu_device).requires_grad_(True) for k, v in rgcn_weights(rng, L, D, D).items()}
    for order in ("aggregate_first", "transform_first"):
        set_switch(monkeypatch, "RELGNN_RGCN_ORDER", order)
        for norm in (True, False):
            clear_graph_cache()
            h = torch.randn((V, D), device=gpu_device, requires_grad=True)
            out = sparse_rgcn_layer(h, adj, deg, D, 1, "tanh", "sum", norm, weights=w)
            assert out.shape == (V, D) and float(out.abs().max()) == 0.0
            out.sum().backward()
            assert float(h.grad.abs().max()) == 0.0


@pytest.mark.parametrize("D", [192, 256])
@pytest.mark.parametrize("act", ["ReLU", "leaky_relu", None])
def test_film_sign_mask_route_matches_recompute(gpu_device, monkeypatch, act, D):
    """Piecewise-linear activations at 128 < D <= 256: the by-target backward pass leaves one sign bit per feature and
    message and the by-source pass (relgnn_film_bwd_msg_masked) gathers gamma and the target's gradient row only.  Same
    bits as the pass that re-gathers beta and recomputes the pre-activation (RELGNN_EDGE_SIGN_MASK=0): identical inputs,
    identical arithmetic, identical summation order — the gradients must agree bit for bit; and with the emit route to
    rounding."""
    from tf_gnn_samples_amd.gnns import sparse_gnn_film_layer
    rng, adj, deg = _graph(33, V=180, L=3, E=(1400, 200, 0))
    V, L = 180, 3
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    w = dict(rgcn_weights(rng, L, D, D), **LN(D))
    for l in range(L):
        w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    gout = torch.as_tensor(np.random.default_rng(4).standard_normal((V, D)).astype(np.float32), device=gpu_device)
    grads = {}
    for name, env in (("mask", {"RELGNN_EDGE_BWD": "regather", "RELGNN_EDGE_SIGN_MASK": "1"}),
                      ("recompute", {"RELGNN_EDGE_BWD": "regather", "RELGNN_EDGE_SIGN_MASK": "0"}),
                      ("emit", {"RELGNN_EDGE_BWD": "emit", "RELGNN_EDGE_SIGN_MASK": "0"})):
        for k, v in env.items():
            set_switch(monkeypatch, k, v)
        hd = torch.as_tensor(h, device=gpu_device).requires_grad_(True)
        wd = {k: torch.as_tensor(v, device=gpu_device).requires_grad_(True) for k, v in w.items()}
        out = sparse_gnn_film_layer(hd, adj_d, deg_d, D, 1, act, "sum", True, weights=wd)
        out.backward(gout)
        grads[name] = [hd.grad] + [wd[k].grad for k in sorted(wd) if wd[k].grad is not None]
    for a, b in zip(grads["mask"], grads["recompute"]):
        assert torch.equal(a, b)
    for a, b in zip(grads["mask"], grads["emit"]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
