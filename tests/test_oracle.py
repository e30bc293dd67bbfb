"""The oracle against the pins that exist: known answers, hand-computed graphs,
fp64 agreement and size-independent properties.  CPU only."""
import numpy as np
import pytest

from oracle import bookkeeping, gnns as G, model as OM, tf_ops as T
from helpers import degree_table, glorot, layer_norm_weights, random_relational_graph, rgcn_weights


def test_parameter_count_known_answer():
    # README.md:29 of the reference: "Model has 699257 parameters." (RGCN / PPI, h=256, 3 layers)
    assert OM.rgcn_ppi_num_parameters() == 699257


def test_c_segment_kernels_match_numpy_at():
    rng = np.random.default_rng(0)
    data = rng.standard_normal((500, 7)).astype(np.float32)
    ids = rng.integers(0, 40, size=500).astype(np.int32)
    ref = np.zeros((40, 7), np.float32)
    np.add.at(ref, ids, data)
    np.testing.assert_array_equal(T.unsorted_segment_sum(data, ids, 40), ref)
    refm = np.full((40, 7), np.finfo(np.float32).min, np.float32)
    np.maximum.at(refm, ids, data)
    np.testing.assert_array_equal(T.unsorted_segment_max(data, ids, 40), refm)


def test_segment_semantics_empty_and_negative():
    data = np.array([[1., 2.], [3., 4.], [5., 6.]], np.float32)
    ids = np.array([2, -1, 2], np.int32)  # negative ids are dropped by TF
    s = T.unsorted_segment_sum(data, ids, 4)
    np.testing.assert_array_equal(s, [[0, 0], [0, 0], [6, 8], [0, 0]])
    m = T.unsorted_segment_max(data, ids, 4)
    assert m[0, 0] == np.float32(-3.4028235e38) and not np.isinf(m[0, 0])
    np.testing.assert_array_equal(m[2], [5, 6])
    np.testing.assert_array_equal(T.unsorted_segment_mean(data, ids, 4)[2], [3, 4])
    np.testing.assert_allclose(T.unsorted_segment_sqrt_n(data, ids, 4)[2], np.array([6, 8]) / np.sqrt(2), rtol=1e-6)
    np.testing.assert_array_equal(T.unsorted_segment_mean(data, ids, 4)[0], [0, 0])  # 0 / max(0, 1)
    with pytest.raises(IndexError):
        T.unsorted_segment_sum(data, np.array([0, 1, 9], np.int32), 4)


def test_unknown_names_raise_like_reference():
    with pytest.raises(ValueError, match="Unknown aggregation function"):
        T.get_aggregation_function("median")
    with pytest.raises(ValueError, match="Unknown activation function"):
        T.get_activation("swish")
    assert T.get_activation("linear") is None and T.get_activation(None) is None
    assert T.get_activation("ReLU") is T.relu  # case-insensitive (utils/utils.py:39)


def test_inverse_degree_fp32_arithmetic():
    # 1/(c + 1e-7) in fp32: c=1 -> 1/1.0000001, c>=2 -> exactly 1/c
    c = np.array([1, 2, 3, 28], np.float32)
    inv = np.float32(1.0) / (c + np.float32(1e-7))
    assert inv[0] == np.float32(1.0) / np.float32(1.0000001)
    np.testing.assert_array_equal(inv[1:], np.float32(1.0) / c[1:])


def test_rgcn_hand_computed_tiny_graph():
    # 3 nodes, 2 edge types; identity-like weights make the result checkable by hand
    h = np.array([[1., 0.], [0., 2.], [3., 3.]], np.float32)
    adj = [np.array([[0, 1], [2, 1]], np.int32), np.array([[1, 0]], np.int32)]
    deg = degree_table(adj, 3)
    W = {"Edge_0_Weight/kernel": np.eye(2, dtype=np.float32), "Edge_1_Weight/kernel": 2 * np.eye(2, dtype=np.float32)}
    out = G.sparse_rgcn_layer(h, adj, deg, 2, activation_function=None, weights=W)
    # node 1: (h0 + h2)/2 = (2, 1.5); node 0: 2*h1/1.0000001 = (0, ~4); node 2: nothing
    np.testing.assert_allclose(out, [[0, 4], [2, 1.5], [0, 0]], rtol=1e-6)
    out_sum = G.sparse_rgcn_layer(h, adj, deg, 2, activation_function=None, normalize_by_num_incoming=False, weights=W)
    np.testing.assert_array_equal(out_sum, [[0, 4], [4, 3], [0, 0]])
    out_max = G.sparse_rgcn_layer(h, adj, deg, 2, activation_function="relu", normalize_by_num_incoming=False,
                                  message_aggregation_function="max", weights=W)
    np.testing.assert_array_equal(out_max, [[0, 4], [3, 3], [0, 0]])  # relu(lowest) = 0 for the empty node


def test_rgin_distinguishes_docstring_graphs():
    # gnns/rgin.py:29-35: G1 = (E1={(1,2)}, E2={(3,2)}), G2 = (E1={(3,2)}, E2={(1,2)}) must differ
    rng = np.random.default_rng(1)
    D = 4
    h = rng.standard_normal((3, D)).astype(np.float32)
    w = {}
    for l in range(2):
        w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (D, D))
        w["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
    w.update(layer_norm_weights(D, 2))
    g1 = [np.array([[0, 1]], np.int32), np.array([[2, 1]], np.int32)]
    g2 = [np.array([[2, 1]], np.int32), np.array([[0, 1]], np.int32)]
    o1 = G.sparse_rgin_layer(h, g1, D, weights=w)
    o2 = G.sparse_rgin_layer(h, g2, D, weights=w)
    assert np.abs(o1[1] - o2[1]).max() > 1e-3
    # with a single shared MLP (same weights for both types) they coincide
    for k in ("dense", "dense_1"):
        w["Edge_1_MLP/%s/kernel" % k] = w["Edge_0_MLP/%s/kernel" % k]
    np.testing.assert_allclose(G.sparse_rgin_layer(h, g1, D, weights=w)[1], G.sparse_rgin_layer(h, g2, D, weights=w)[1],
                               atol=1e-6)


@pytest.mark.parametrize("agg", ["sum", "mean", "max", "sqrt_n"])
def test_rgcn_fp32_vs_fp64_and_edge_permutation(agg):
    rng = np.random.default_rng(2)
    V, D, L = 60, 16, 3
    adj = random_relational_graph(rng, V, L, 150, empty_types=(1,))
    deg = degree_table(adj, V)
    w = rgcn_weights(rng, L, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    o32 = G.sparse_rgcn_layer(h, adj, deg, D, message_aggregation_function=agg, weights=w)
    o64 = G.sparse_rgcn_layer(h.astype(np.float64), adj, deg, D, message_aggregation_function=agg, weights=w)
    assert np.abs(o32 - o64).max() < 1e-5
    # permuting the edge order inside a type changes nothing beyond fp32 rounding
    perm_adj = [a[rng.permutation(len(a))] for a in adj]
    o32p = G.sparse_rgcn_layer(h, perm_adj, deg, D, message_aggregation_function=agg, weights=w)
    assert np.abs(o32 - o32p).max() < 1e-5


def test_gru_cell_formula():
    rng = np.random.default_rng(3)
    V, u = 5, 4
    x, h = rng.standard_normal((V, u)), rng.standard_normal((V, u))
    K, U, b = rng.standard_normal((u, 3 * u)), rng.standard_normal((u, 3 * u)), rng.standard_normal(3 * u)
    out = T.gru_cell(x, h, K, U, b, np.tanh)
    hs = lambda a: np.clip(0.2 * a + 0.5, 0, 1)
    z = hs(x @ K[:, :u] + b[:u] + h @ U[:, :u])
    r = hs(x @ K[:, u:2 * u] + b[u:2 * u] + h @ U[:, u:2 * u])
    hh = np.tanh(x @ K[:, 2 * u:] + b[2 * u:] + (r * h) @ U[:, 2 * u:])
    np.testing.assert_allclose(out, z * h + (1 - z) * hh, rtol=1e-12)


def test_layer_norm_matches_definition():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((6, 10))
    g, b = rng.standard_normal(10), rng.standard_normal(10)
    ref = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-12) * g + b
    np.testing.assert_allclose(T.layer_norm(x, g, b), ref, rtol=1e-10)


def test_log_softmax_segments_sum_to_one():
    rng = np.random.default_rng(5)
    logits = rng.standard_normal(200).astype(np.float32) * 5
    ids = rng.integers(0, 17, size=200).astype(np.int32)
    p = np.exp(T.unsorted_segment_log_softmax(logits, ids, 20))
    sums = T.unsorted_segment_sum(p, ids, 20)
    present = np.bincount(ids, minlength=20) > 0
    np.testing.assert_allclose(sums[present], 1.0, rtol=1e-5)


def test_all_layers_run_and_agree_with_fp64():
    rng = np.random.default_rng(6)
    V, D, L, K = 40, 8, 3, 2
    adj = random_relational_graph(rng, V, L, 90, empty_types=(2,))
    deg = degree_table(adj, V)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ln = layer_norm_weights(D, 2)
    w_rgat = dict(rgcn_weights(rng, L, D, D))
    for l in range(L):
        w_rgat["Edge_%i_Attention_Parameters" % l] = rng.standard_normal(2 * D).astype(np.float32) * 0.3
    w_film = dict(rgcn_weights(rng, L, D, D), **ln)
    for l in range(L):
        w_film["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    w_mlp = dict(ln)
    for l in range(L):
        w_mlp["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (2 * D, D))
        w_mlp["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
    w_ggnn = dict(rgcn_weights(rng, L, D, D))
    w_ggnn.update({"gru_cell/kernel": glorot(rng, (D, 3 * D)), "gru_cell/recurrent_kernel": glorot(rng, (D, 3 * D)),
                   "gru_cell/bias": np.zeros(3 * D, np.float32)})
    calls = [
        lambda x: G.sparse_rgat_layer(x, adj, D, num_heads=K, weights=w_rgat),
        lambda x: G.sparse_gnn_film_layer(x, adj, deg, D, weights=w_film),
        lambda x: G.sparse_gnn_edge_mlp_layer(x, adj, deg, D, activation_function="gelu", weights=w_mlp),
        lambda x: G.sparse_ggnn_layer(x, adj, D, num_timesteps=2, weights=w_ggnn),
    ]
    for f in calls:
        o32, o64 = f(h), f(h.astype(np.float64))
        assert o32.shape == (V, D) and o32.dtype == np.float32 and o64.dtype == np.float64
        assert np.abs(o32 - o64).max() < 2e-5


def test_rgcn_node_side_evaluation_order_equals_op_for_op_path():
    """oracle.gnns.sparse_rgcn_layer(node_side_transform=True) — the BASELINE-size evaluation order used by
    tests/test_gpu_baseline_size.py — against the op-for-op restatement (per-edge MatMul, concat, segment sum):
    same message values, same sequential fold, so the results agree to the last bit or two of fp32 (BLAS may
    pick a different micro-kernel for the [V, D] and the [E_l, D] GEMM)."""
    from oracle import gnns as G, model as OM
    from helpers import degree_table, random_relational_graph, rgcn_weights
    rng = np.random.default_rng(5)
    V, D, L = 400, 64, 3
    adj = random_relational_graph(rng, V, L, [3000, 400, 0])
    deg = degree_table(adj, V)
    w = rgcn_weights(rng, L, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    for norm in (True, False):
        for steps in (1, 2):
            a = G.sparse_rgcn_layer(h, adj, deg, D, steps, "ReLU", "sum", norm, weights=w)
            b = G.sparse_rgcn_layer(h, adj, deg, D, steps, "ReLU", "sum", norm, weights=w, node_side_transform=True)
            assert np.abs(a - b).max() <= 4e-7 * max(1.0, np.abs(a).max())
    # aggregations the fast path does not cover fall through to the op-for-op path
    a = G.sparse_rgcn_layer(h, adj, deg, D, 1, "tanh", "max", weights=w)
    b = G.sparse_rgcn_layer(h, adj, deg, D, 1, "tanh", "max", weights=w, node_side_transform=True)
    assert np.array_equal(a, b)
