"""The route from "parity unpinned" to pinned, and what pins the TF-internal assumptions meanwhile.

(1) tests/golden/tf_layers_small.npz / tf_semantics.npz are written by scripts/dump_tf_golden.py on a machine that
    has the reference's environment (TensorFlow 1.13 + dpu_utils; it runs the UNMODIFIED gnns/*.py of the reference on
    the committed fixture).  When the files exist, the oracle (CPU) and the HIP path (GPU) are compared with what TF
    computed, 1e-5 abs.  While they do not exist — no TensorFlow can be installed in the build container or on the GPU
    box — these tests SKIP with "PARITY UNPINNED".
(2) Independent of TF: hand-derived vectors for every TF-internal semantic oracle/tf_ops.py assumes (gate order and
    recurrent activation of the Keras GRUCell, layer_norm's variance epsilon, empty segments of unsorted_segment_max /
    mean / sqrt_n, softmax over an empty segment, leaky_relu's alpha, 1/(c + 1e-7) in fp32).  Each expected number
    below is derived by hand in the comment next to it and would come out differently under the look-alike semantics
    named there; the oracle AND the package's CPU-side mirrors are checked against them."""
from pathlib import Path

import numpy as np
import pytest

from oracle import gnns as G, tf_ops as T
from test_golden_cpu import load_layers_fixture, oracle_layer_outputs

GOLDEN = Path(__file__).resolve().parent / "golden"
TF_LAYERS = GOLDEN / "tf_layers_small.npz"
TF_SEM = GOLDEN / "tf_semantics.npz"
UNPINNED = ("PARITY UNPINNED: %s is not there — run scripts/dump_tf_golden.py where TensorFlow 1.13 + dpu_utils exist "
            "and commit its output")


def test_oracle_against_tensorflow_layer_outputs():
    if not TF_LAYERS.exists():
        pytest.skip(UNPINNED % TF_LAYERS.name)
    z = np.load(TF_LAYERS)
    h, adj, deg, K, w, _ = load_layers_fixture()
    got = oracle_layer_outputs(h, adj, deg, K, w)
    for name, out in got.items():
        assert np.abs(out - z["out|" + name]).max() <= 1e-5, name


@pytest.mark.gpu
def test_hip_path_against_tensorflow_layer_outputs(gpu_device):
    if not TF_LAYERS.exists():
        pytest.skip(UNPINNED % TF_LAYERS.name)
    import torch
    from tf_gnn_samples_amd import gnns as H
    z = np.load(TF_LAYERS)
    h, adj, deg, K, w, _ = load_layers_fixture()
    D = h.shape[1]
    dev = lambda x: ({k: dev(v) for k, v in x.items()} if isinstance(x, dict) else
                     [dev(v) for v in x] if isinstance(x, list) else torch.as_tensor(x, device=gpu_device))
    hd, ad, dd = dev(h), dev(adj), dev(deg)
    got = {
        "rgcn": H.sparse_rgcn_layer(hd, ad, dd, D, 2, "ReLU", "sum", weights=dev(w["rgcn"])),
        "ggnn": H.sparse_ggnn_layer(hd, ad, D, 2, "gru", "tanh", "mean", weights=dev(w["ggnn"])),
        "rgat": H.sparse_rgat_layer(hd, ad, D, K, 2, "tanh", weights=dev(w["rgat"])),
        "film": H.sparse_gnn_film_layer(hd, ad, dd, D, 2, "ReLU", "sum", weights=dev(w["film"])),
        "rgin": H.sparse_rgin_layer(hd, ad, D, 2, "ReLU", "sum", weights=dev(w["rgin"])),
        "edge_mlp": H.sparse_gnn_edge_mlp_layer(hd, ad, dd, D, 2, "gelu", "sum", weights=dev(w["edge_mlp"])),
    }
    for name, out in got.items():
        assert np.abs(out.cpu().numpy() - z["out|" + name]).max() <= 1e-5, name


def test_oracle_against_tensorflow_single_op_vectors():
    if not TF_SEM.exists():
        pytest.skip(UNPINNED % TF_SEM.name)
    from oracle import optim as O
    z = np.load(TF_SEM)
    x, ids, probe = z["in|x"], z["in|ids"], z["in|probe"]
    close = lambda a, b, tol=1e-6: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() <= tol
    assert close(T.unsorted_segment_sum(x, ids, 6), z["seg|sum"])
    assert np.array_equal(T.unsorted_segment_max(x, ids, 6), z["seg|max"])
    assert close(T.unsorted_segment_mean(x, ids, 6), z["seg|mean"])
    assert close(T.unsorted_segment_sqrt_n(x, ids, 6), z["seg|sqrt_n"])
    assert np.array_equal(T.unsorted_segment_max(x, np.array([0, -1, 2, 5, 0, 2], np.int32), 6), z["seg|max_negative_id"])
    assert close(T.unsorted_segment_log_softmax(x[:, 0], ids, 6), z["seg|log_softmax"])
    assert close(T.leaky_relu(probe), z["act|leaky_relu"]) and close(T.elu(probe), z["act|elu"])
    assert close(T.selu(probe), z["act|selu"]) and close(T.hard_sigmoid(probe), z["act|hard_sigmoid"])
    assert close(T.gelu(probe), z["act|gelu_erf"])
    assert np.array_equal((np.float32(1.0) / (z["in|deg"] + np.float32(1e-7))), z["misc|inv_degree"])
    assert close(T.layer_norm(z["in|ln_x"], np.ones(5, np.float32), np.zeros(5, np.float32)), z["ln|out"], 2e-6)
    assert close(T.gru_cell(z["cell|gru|x"], z["cell|gru|h"], z["cell|gru|kernel"], z["cell|gru|recurrent_kernel"],
                            z["cell|gru|bias"], T.tanh), z["cell|gru|out"])
    assert close(T.simple_rnn_cell(z["cell|rnn|x"], z["cell|rnn|h"], z["cell|rnn|kernel"], z["cell|rnn|recurrent_kernel"],
                                   z["cell|rnn|bias"], T.tanh), z["cell|rnn|out"])
    for g, c in zip(z["opt|grads"], z["opt|clip"]):
        assert close(O.clip_by_norm(g, 1.0), c, 1e-7)
    for name in ("adam", "rmsprop", "sgd"):
        opt = O.make_optimizer(name, [np.ones((4, 3), np.float32)], 1e-3, decay=0.98, momentum=0.85)
        for g, want in zip(z["opt|grads"], z["opt|" + name]):
            O.train_step(opt, [g], 1.0)
            assert close(opt.vars[0], want, 1e-7), name


# ---- (2) hand-derived vectors --------------------------------------------------------------------------------------
def test_hard_sigmoid_and_leaky_relu_by_hand():
    x = np.array([-3.0, -2.5, 0.0, 1.0, 2.5, 3.0], np.float32)
    # Keras hard_sigmoid = clip(0.2 x + 0.5, 0, 1): 0.2*(-2.5)+0.5 = 0 exactly, 0.2*1+0.5 = 0.7, saturates at |x| >= 2.5
    # (a logistic sigmoid would give 0.731 at x = 1; torch's hardsigmoid clip(x/6 + 0.5) gives 0.667)
    np.testing.assert_allclose(T.hard_sigmoid(x), [0.0, 0.0, 0.5, 0.7, 1.0, 1.0], atol=1e-7)
    # tf.nn.leaky_relu default alpha = 0.2 (torch's default negative_slope is 0.01): f(-1) = -0.2, f(-2.5) = -0.5
    np.testing.assert_allclose(T.leaky_relu(np.array([-2.5, -1.0, 0.0, 3.0], np.float32)), [-0.5, -0.2, 0.0, 3.0], atol=1e-7)
    import torch
    from tf_gnn_samples_amd.utils import get_activation, hard_sigmoid
    np.testing.assert_allclose(hard_sigmoid(torch.tensor(x)).numpy(), [0.0, 0.0, 0.5, 0.7, 1.0, 1.0], atol=1e-7)
    np.testing.assert_allclose(get_activation("leaky_relu")(torch.tensor([-2.5, -1.0, 0.0, 3.0])).numpy(),
                               [-0.5, -0.2, 0.0, 3.0], atol=1e-7)


def test_gru_cell_gate_order_and_update_by_hand():
    """One unit, x = 1, h = 0.5, kernel = [k_z, k_r, k_h] = [1, -5, 0.5], recurrent kernel 0 except U_h = 2, bias 0.
       z  = hs(1*1)           = 0.7
       r  = hs(1*(-5))        = 0            (0.2*(-5) + 0.5 < 0)
       hh = tanh(0.5*1 + (r*h)*2) = tanh(0.5) = 0.46211716
       h' = z*h + (1-z)*hh    = 0.35 + 0.3*0.46211716 = 0.48863515
    Look-alikes: gate order r,z,h (cuDNN / torch.nn.GRUCell layout) gives z = 0, r = 0.7 -> tanh(0.5 + 0.7) = 0.8337;
    h' = (1-z)*h + z*hh (torch's convention) gives 0.15 + 0.7*0.4621 = 0.4735; reset_after=True applies r AFTER the
    recurrent matmul (same here since r = 0, so a second vector with r = 1 is checked too)."""
    f = np.float32
    K, U, b = np.array([[1.0, -5.0, 0.5]], f), np.array([[0.0, 0.0, 2.0]], f), np.zeros(3, f)
    out = T.gru_cell(np.array([[1.0]], f), np.array([[0.5]], f), K, U, b, T.tanh)
    assert abs(float(out[0, 0]) - 0.48863515) < 1e-7
    # second vector: k_r = +5 -> r = 1, hh = tanh(0.5 + 0.5*2) = tanh(1.5) = 0.90514825, h' = 0.35 + 0.3*0.90514825 = 0.62154448
    K2 = np.array([[1.0, 5.0, 0.5]], f)
    out2 = T.gru_cell(np.array([[1.0]], f), np.array([[0.5]], f), K2, U, b, T.tanh)
    assert abs(float(out2[0, 0]) - 0.62154448) < 1e-7
    import torch
    from tf_gnn_samples_amd.utils import get_gated_unit
    for Kx, want in ((K, 0.48863515), (K2, 0.62154448)):
        cell = get_gated_unit(1, "gru", "tanh", {"kernel": torch.tensor(Kx), "recurrent_kernel": torch.tensor(U),
                                                 "bias": torch.tensor(b)})
        got = cell(torch.tensor([[1.0]]), [torch.tensor([[0.5]])])[0]
        assert abs(float(got) - want) < 1e-7


def test_layer_norm_variance_epsilon_by_hand():
    """tf.contrib.layers.layer_norm uses variance_epsilon = 1e-12 (torch.nn.LayerNorm: 1e-5).
       row a = [0, 2e-6]: mean 1e-6, biased variance 1e-12 -> (x - mean) / sqrt(1e-12 + 1e-12) = -+1e-6 / 1.41421356e-6
               = -+0.70710678   (with eps 1e-5 the row would come out as -+3.16e-4)
       row b = constant: variance 0 -> every entry is exactly beta."""
    f = np.float32
    x = np.array([[0.0, 2e-6], [3.0, 3.0]], f)
    gamma, beta = np.array([1.0, 1.0], f), np.array([0.25, 0.25], f)
    y = T.layer_norm(x, gamma, beta)
    np.testing.assert_allclose(y[0], [0.25 - 0.70710678, 0.25 + 0.70710678], rtol=2e-5)
    np.testing.assert_allclose(y[1], [0.25, 0.25], atol=1e-6)
    import torch
    from tf_gnn_samples_amd.utils import layer_norm
    y2 = layer_norm(torch.tensor(x), torch.tensor(gamma), torch.tensor(beta)).numpy()
    np.testing.assert_allclose(y2[0], [0.25 - 0.70710678, 0.25 + 0.70710678], rtol=2e-4)


def test_empty_segments_and_dropped_ids_by_hand():
    f = np.float32
    data = np.array([[1.0, -2.0], [3.0, 5.0], [-4.0, 0.5]], f)
    ids = np.array([2, 0, 2], np.int32)                       # segment 1 empty
    # unsorted_segment_max: empty segment = numeric_limits<float>::lowest() = -3.4028235e38 (NOT -inf)
    mx = T.unsorted_segment_max(data, ids, 3)
    assert mx[1, 0] == np.finfo(np.float32).min and np.isfinite(mx[1]).all()
    np.testing.assert_array_equal(mx[[0, 2]], [[3.0, 5.0], [1.0, 0.5]])
    # mean / sqrt_n divide by max(count, 1): the empty segment is 0/1 = 0, segment 2 = (1-4)/2, (-2+0.5)/2 resp. /sqrt(2)
    np.testing.assert_allclose(T.unsorted_segment_mean(data, ids, 3), [[3.0, 5.0], [0.0, 0.0], [-1.5, -0.75]])
    np.testing.assert_allclose(T.unsorted_segment_sqrt_n(data, ids, 3),
                               [[3.0, 5.0], [0.0, 0.0], [-3.0 / 2 ** 0.5, -1.5 / 2 ** 0.5]], rtol=1e-6)
    # negative ids are dropped
    np.testing.assert_array_equal(T.unsorted_segment_sum(data, np.array([2, -1, 2], np.int32), 3),
                                  [[0.0, 0.0], [0.0, 0.0], [-3.0, -1.5]])


def test_segment_softmax_and_empty_target_by_hand():
    """unsorted_segment_log_softmax (dpu_utils) then tf.exp (gnns/rgat.py:127-130): logits [0, ln 3] in one segment ->
    softmax = [0.25, 0.75]; a node WITHOUT incoming messages owns no message at all, so its RGAT state is the empty sum
    0 -> activation(0)."""
    f = np.float32
    logits = np.array([0.0, np.log(3.0), 7.0], f)
    p = np.exp(T.unsorted_segment_log_softmax(logits, np.array([0, 0, 2], np.int32), 3))
    np.testing.assert_allclose(p, [0.25, 0.75, 1.0], rtol=1e-6)
    rng = np.random.default_rng(0)
    D, K = 8, 2
    w = {"Edge_0_Weight/kernel": rng.standard_normal((D, D)).astype(f),
         "Edge_0_Attention_Parameters": rng.standard_normal(2 * D).astype(f)}
    h = rng.standard_normal((3, D)).astype(f)
    out = G.sparse_rgat_layer(h, [np.array([[0, 1], [2, 1]], np.int32)], D, K, 1, "tanh", weights=w)
    assert np.array_equal(out[0], np.zeros(D, f)) and np.array_equal(out[2], np.zeros(D, f)) and np.abs(out[1]).max() > 0


def test_inverse_degree_in_fp32_by_hand():
    """1.0 / (c + 1e-7) evaluated in float32 (gnns/rgcn.py:100-104): c = 0 -> 1/1e-7 = 1e7 (the reference relies on a
    zero-degree target never receiving a message of that type); c = 1 -> 1/1.0000001192 = 0.99999988; c >= 2: c + 1e-7
    rounds back to c, so the scale is exactly 1/c."""
    f = np.float32
    c = np.array([0.0, 1.0, 2.0, 4.0], f)
    s = f(1.0) / (c + f(1e-7))
    assert abs(float(s[0]) - 1e7) < 1.0 and s[1] == f(0.99999988) and s[2] == f(0.5) and s[3] == f(0.25)
    got = G._inv_degree(np.array([[0.0, 1.0, 2.0, 4.0]], f), 0, np.array([0, 1, 2, 3]), f)[:, 0]
    assert np.array_equal(got, s)


def test_examples_printed_in_tensorflow_s_own_api_documentation():
    """Known answers that TensorFlow's API documentation prints next to the ops the reference calls (tf.math.unsorted_segment_sum /
    _max, tf.round, tf.nn.sigmoid_cross_entropy_with_logits' stable formula, tf.clip_by_norm's definition): small, but TensorFlow's
    own numbers rather than ours."""
    from oracle import model as OM, optim, tf_ops as T
    c = np.array([[1, 2, 3, 4], [5, 6, 7, 8], [4, 3, 2, 1]], dtype=np.float32)
    ids = np.array([0, 1, 0], dtype=np.int32)
    np.testing.assert_array_equal(T.unsorted_segment_sum(c, ids, 2), [[5, 5, 5, 5], [5, 6, 7, 8]])
    np.testing.assert_array_equal(T.unsorted_segment_max(c, ids, 2), [[4, 3, 3, 4], [5, 6, 7, 8]])
    np.testing.assert_array_equal(np.round(np.array([0.9, 2.5, 2.3, 1.5, -4.5], np.float32)), [1.0, 2.0, 2.0, 2.0, -4.0])   # tf.round
    x, z = np.array([-3.0, -0.5, 0.0, 2.0], np.float32), np.array([1.0, 0.0, 1.0, 0.0], np.float32)
    naive = z * -np.log(1 / (1 + np.exp(-x))) + (1 - z) * -np.log(1 - 1 / (1 + np.exp(-x)))        # the definition the docs start from
    np.testing.assert_allclose(OM.sigmoid_cross_entropy_with_logits(x, z), naive, rtol=2e-6)
    t = np.array([3.0, 4.0], np.float32)                                                                # l2 norm 5
    np.testing.assert_allclose(optim.clip_by_norm(t, 1.0), t / 5.0, rtol=1e-6)                          # t * clip_norm / l2norm(t)
    np.testing.assert_array_equal(optim.clip_by_norm(t, 10.0), t)                                       # below the norm: unchanged
