"""Layer-level parity: tf_gnn_samples_amd.gnns.sparse_*_layer (HIP path) vs the NumPy oracle on
the same seeded inputs and weights; tolerance = the north-star's 1e-5 abs on fp32 node states.
Gradients are checked against torch autograd through the reference-order torch mirror (fp64)."""
import numpy as np
import pytest
import torch

from oracle import gnns as G, model as OM, torch_ref as R
from helpers import degree_table, glorot, random_relational_graph, rgcn_weights

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


def _graph(seed, V=150, L=3, E=(900, 150, 0)):
    rng = np.random.default_rng(seed)
    adj = random_relational_graph(rng, V, L, list(E))
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
        assert np.abs(out.cpu().numpy() - ref).max() < TOL, (agg, norm, act)


def test_rgcn_layer_changes_dimension(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    rng, adj, deg = _graph(2)
    V, Din, D = 150, 50, 256
    w = rgcn_weights(rng, 3, Din, D)
    h = rng.standard_normal((V, Din)).astype(np.float32)
    ref = G.sparse_rgcn_layer(h, adj, deg, D, weights=w)
    out = sparse_rgcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, weights=_dev(w, gpu_device))
    assert out.shape == (V, D) and np.abs(out.cpu().numpy() - ref).max() < TOL


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
