"""The oracle's primitives against INDEPENDENT implementations (PyTorch's CPU kernels), wherever the two libraries define
the same function.  This does not pin the oracle to TensorFlow (TF 1.13 cannot run here) — it removes
the possibility that a shared arithmetic slip hides in both the oracle and the HIP path, which were written by one hand:
segment reductions, segmented log-softmax, layer normalisation, the activation table, sigmoid cross-entropy, the Dense
layout, and the parts of Adam / RMSProp / SGD that TF and PyTorch share (moment updates, bias correction; the epsilon
placement, where they differ, is covered by tests/test_optimizer_cpu.py)."""
import numpy as np
import pytest
import torch

from oracle import optim as OO
from oracle import tf_ops as T

F64 = np.float64


def _ids(rng, M, S, with_negative=False):
    ids = rng.integers(0, S, M).astype(np.int32)
    ids[ids == 3] = 5                      # leave segment 3 empty
    if with_negative:
        ids[::17] = -1                     # TF drops negative ids
    return ids


@pytest.mark.parametrize("with_negative", [False, True])
def test_segment_reductions_against_torch_scatter(with_negative):
    rng = np.random.default_rng(0)
    M, S, D = 4000, 97, 13
    x = rng.standard_normal((M, D))
    ids = _ids(rng, M, S, with_negative)
    keep = ids >= 0
    xt, it = torch.as_tensor(x[keep]), torch.as_tensor(ids[keep].astype(np.int64))
    want_sum = torch.zeros(S, D, dtype=torch.float64).index_add_(0, it, xt).numpy()
    cnt = np.maximum(np.bincount(ids[keep], minlength=S), 1)[:, None]
    np.testing.assert_allclose(T.unsorted_segment_sum(x, ids, S), want_sum, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(T.unsorted_segment_mean(x, ids, S), want_sum / cnt, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(T.unsorted_segment_sqrt_n(x, ids, S), want_sum / np.sqrt(cnt), rtol=1e-12, atol=1e-12)
    want_max = torch.full((S, D), float("-inf"), dtype=torch.float64).scatter_reduce_(
        0, it[:, None].expand(-1, D), xt, reduce="amax", include_self=True).numpy()
    got_max = T.unsorted_segment_max(x, ids, S)
    empty = np.isinf(want_max)
    assert empty[3].all()
    np.testing.assert_array_equal(got_max[~empty], want_max[~empty])
    assert (got_max[empty] == np.finfo(np.float64).min).all()          # TF: lowest(), not -inf


def test_segment_log_softmax_against_torch_per_segment():
    rng = np.random.default_rng(1)
    M, S, K = 3000, 41, 4
    x = rng.standard_normal((M, K)) * 3
    ids = rng.integers(0, S, M).astype(np.int32)
    got = T.unsorted_segment_log_softmax(x, ids, S)
    for s in range(S):
        sel = np.nonzero(ids == s)[0]
        if sel.size:
            want = torch.log_softmax(torch.as_tensor(x[sel]), dim=0).numpy()
            np.testing.assert_allclose(got[sel], want, rtol=1e-12, atol=1e-12)


def test_layer_norm_against_torch():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((257, 128)) * 2 + 0.3
    gamma, beta = rng.standard_normal(128), rng.standard_normal(128)
    want = torch.nn.functional.layer_norm(torch.as_tensor(x), (128,), torch.as_tensor(gamma), torch.as_tensor(beta),
                                          eps=1e-12).numpy()
    np.testing.assert_allclose(T.layer_norm(x, gamma, beta), want, rtol=1e-11, atol=1e-11)
    x32 = x.astype(np.float32)
    want32 = torch.nn.functional.layer_norm(torch.as_tensor(x32), (128,), torch.as_tensor(gamma.astype(np.float32)),
                                            torch.as_tensor(beta.astype(np.float32)), eps=1e-12).numpy()
    np.testing.assert_allclose(T.layer_norm(x32, gamma.astype(np.float32), beta.astype(np.float32)), want32, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name,fn", [
    ("tanh", torch.tanh), ("relu", torch.relu),
    ("leaky_relu", lambda t: torch.nn.functional.leaky_relu(t, 0.2)),
    ("elu", torch.nn.functional.elu), ("selu", torch.selu),
    ("gelu", lambda t: torch.nn.functional.gelu(t, approximate="none"))])
def test_activation_table_against_torch(name, fn):
    x = np.concatenate([np.linspace(-12, 12, 4001), [0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0]])
    got = T.apply_act(T.get_activation(name), x.astype(F64))
    np.testing.assert_allclose(got, fn(torch.as_tensor(x)).numpy(), rtol=1e-12, atol=1e-13)


def test_sigmoid_cross_entropy_against_torch():
    from oracle import model as OM
    rng = np.random.default_rng(3)
    x = rng.standard_normal((500, 121)) * 6
    z = (rng.random((500, 121)) < 0.3).astype(F64)
    want = torch.nn.functional.binary_cross_entropy_with_logits(torch.as_tensor(x), torch.as_tensor(z), reduction="none").numpy()
    np.testing.assert_allclose(OM.sigmoid_cross_entropy_with_logits(x, z), want, rtol=1e-12, atol=1e-13)


def test_dense_layout_against_torch_linear():
    rng = np.random.default_rng(4)
    x, k, b = rng.standard_normal((33, 50)), rng.standard_normal((50, 17)), rng.standard_normal(17)
    want = torch.nn.functional.linear(torch.as_tensor(x), torch.as_tensor(k.T.copy()), torch.as_tensor(b)).numpy()
    np.testing.assert_allclose(T.dense(x, k, b), want, rtol=1e-12, atol=1e-12)


def test_adam_moments_and_bias_correction_against_torch_when_epsilon_vanishes():
    """TF: var -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps); torch: var -= lr/(1-b1^t) * m/(sqrt(v/(1-b2^t))+eps).
    The two coincide as eps -> 0: with eps = 1e-30 and gradients bounded away from 0 they must agree to fp64 rounding."""
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal((7, 5))
    grads = [rng.standard_normal((7, 5)) + np.sign(rng.standard_normal((7, 5))) * 0.5 for _ in range(6)]
    old = OO.F
    OO.F = np.float64                                          # the oracle's working precision is a module constant
    try:
        opt = OO.Adam([p0], 1e-3, epsilon=1e-30)
        for g in grads:
            opt.apply_gradients([g])
        got = opt.vars[0]
    finally:
        OO.F = old
    p = torch.nn.Parameter(torch.as_tensor(p0.copy()))
    topt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-30)
    for g in grads:
        p.grad = torch.as_tensor(g.copy())
        topt.step()
    np.testing.assert_allclose(got, p.detach().numpy(), rtol=1e-10, atol=1e-12)


def test_sgd_and_clip_by_norm_against_torch():
    rng = np.random.default_rng(6)
    p0, g = rng.standard_normal((9, 4)).astype(np.float32), (rng.standard_normal((9, 4)) * 5).astype(np.float32)
    clipped = OO.clip_by_norm(g, 1.0)
    tg = torch.as_tensor(g.copy())
    want = tg * (1.0 / max(float(tg.norm()), 1.0))                        # t * clip_norm / max(||t||, clip_norm)
    np.testing.assert_allclose(clipped, want.numpy(), rtol=2e-6, atol=1e-7)
    small = (g * 1e-3).astype(np.float32)
    np.testing.assert_array_equal(OO.clip_by_norm(small, 1.0), small)     # below the threshold: untouched (x * 1)
    opt = OO.GradientDescent([p0], 0.05)
    opt.apply_gradients([g])
    p = torch.nn.Parameter(torch.as_tensor(p0.copy()))
    topt = torch.optim.SGD([p], lr=0.05)
    p.grad = torch.as_tensor(g.copy())
    topt.step()
    np.testing.assert_allclose(opt.vars[0], p.detach().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("normalize", [True, False])
def test_lean_float64_rgcn_reference_equals_the_op_for_op_mirror(normalize):
    """oracle/torch_ref.py:sparse_rgcn_layer_lean (sparse products, what the BASELINE-size gradient test differentiates in
    float64) against the op-for-op mirror sparse_rgcn_layer: values and gradients (h and every kernel).  The lean form
    evaluates 1/(c + 1e-7) in float32 like the reference and widens it, the mirror evaluates it in the working dtype: at
    float64 that is a 1e-7 relative difference of the scale, which bounds the tolerance when normalising."""
    from helpers import degree_table, random_relational_graph, rgcn_weights
    from oracle import torch_model as TM, torch_ref as R
    rng = np.random.default_rng(7)
    V, D = 300, 32
    adj = random_relational_graph(rng, V, 3, [2000, 300, 1500])
    deg = degree_table(adj, V)
    w = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in rgcn_weights(rng, 3, D, D).items()}
    h = torch.tensor(np.tanh(rng.standard_normal((V, D))), dtype=torch.float64, requires_grad=True)
    adj_t, deg_t = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    a = R.sparse_rgcn_layer(h, adj_t, deg_t, D, 2, "ReLU", "sum", normalize, weights=w)
    b = R.sparse_rgcn_layer_lean(h, adj_t, deg_t, D, 2, "ReLU", "sum", normalize, weights=w)
    tol = 5e-7 if normalize else 1e-12
    assert float((a - b).abs().max()) <= tol * max(1.0, float(a.abs().max()))
    ga = torch.autograd.grad(a.square().sum(), [h] + list(w.values()))
    gb = torch.autograd.grad(b.square().sum(), [h] + list(w.values()))
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= tol * float(x.abs().max())
    # the whole-model mirror with either layer: same loss and gradients
    p = {'hidden_size': D, 'graph_num_layers': 2, 'graph_model_activation_function': 'tanh', 'graph_activation_function': 'ReLU',
         'message_aggregation_function': 'sum', 'graph_residual_connection_every_num_layers': 2,
         'graph_num_timesteps_per_layer': 1, 'graph_inter_layer_norm': False, 'graph_dense_between_every_num_gnn_layers': 1}
    W = {}
    for layer in range(2):
        for k, v in rgcn_weights(rng, 3, D, D).items():
            W["gnn_layer_%i/%s" % (layer, k)] = torch.tensor(v, dtype=torch.float64, requires_grad=True)
        W["gnn_layer_%i/Dense/kernel" % layer] = torch.tensor(rgcn_weights(rng, 1, D, D)["Edge_0_Weight/kernel"],
                                                             dtype=torch.float64, requires_grad=True)
    W["dense/kernel"] = torch.tensor(rng.standard_normal((7, D)) * 0.3, dtype=torch.float64, requires_grad=True)
    x = torch.tensor(rng.standard_normal((V, 7)), dtype=torch.float64)
    labels = torch.tensor((rng.random((V, 5)) < 0.4).astype(np.float64))
    kernel = torch.tensor(rng.standard_normal((D, 5)) * 0.2, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(5, dtype=torch.float64, requires_grad=True)
    losses, grads = [], []
    for lean in (False, True):
        final = TM.graph_propagation(x, adj_t, deg_t, p, W, TM.rgcn_apply(p, lean=lean))
        loss = TM.ppi_loss(final, labels, kernel, bias)
        losses.append(float(loss))
        grads.append(torch.autograd.grad(loss, list(W.values()) + [kernel, bias]))
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0])
    for x_, y_ in zip(*grads):
        assert float((x_ - y_).abs().max()) <= 1e-6 * max(float(x_.abs().max()), 1e-12)


def test_torch_model_mirror_equals_the_numpy_driver():
    """oracle/torch_model.py (what the gradient tests differentiate) against oracle/model.py (what the forward parity tests
    compare with), float64, residual connection and inter-layer Dense included."""
    from helpers import degree_table, random_relational_graph, rgcn_weights
    from oracle import model as OM, torch_model as TM
    rng = np.random.default_rng(8)
    V, D, F = 200, 16, 9
    adj = random_relational_graph(rng, V, 3, [900, 200, 700])
    deg = degree_table(adj, V).astype(np.float64)
    p = {'hidden_size': D, 'graph_num_layers': 4, 'graph_model_activation_function': 'tanh', 'graph_activation_function': 'ReLU',
         'message_aggregation_function': 'sum', 'graph_residual_connection_every_num_layers': 2,
         'graph_num_timesteps_per_layer': 1, 'graph_inter_layer_norm': False, 'graph_dense_between_every_num_gnn_layers': 2}
    W = {"dense/kernel": rng.standard_normal((F, D)) * 0.3}
    for layer in range(4):
        for k, v in rgcn_weights(rng, 3, D, D).items():
            W["gnn_layer_%i/%s" % (layer, k)] = v.astype(np.float64)
        W["gnn_layer_%i/Dense/kernel" % layer] = rng.standard_normal((D, D)) * 0.2
    x = rng.standard_normal((V, F))
    want = OM.graph_propagation(x, adj, deg, p, W, OM.rgcn_apply(p))
    got = TM.graph_propagation(torch.as_tensor(x), [torch.as_tensor(a) for a in adj], torch.as_tensor(deg), p,
                               {k: torch.as_tensor(v) for k, v in W.items()}, TM.rgcn_apply(p))
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-10, atol=1e-12)
