"""Seeded random sweep over layer types, widths (incl. widths that miss the float4 / power-of-two fast paths), numbers of
edge types (incl. the compact pair-table regime), aggregations and activations: forward parity against the NumPy oracle
within the north-star tolerance.  Catches dispatch corners the hand-picked cases of test_gpu_layers.py do not hit."""
import numpy as np
import pytest
import torch

from oracle import gnns as G
from helpers import degree_table, glorot, layer_norm_weights, rgcn_weights

pytestmark = pytest.mark.gpu

import os

# RELGNN_FUZZ=lo:hi widens the sweep for a one-off hunt (the default ranges keep the suite at a few seconds)
_lo, _hi = (int(x) for x in os.environ.get("RELGNN_FUZZ", "0:48").split(":"))
FWD_SEEDS = range(_lo, _hi)
GRAD_SEEDS = range(_hi, _hi + (_hi - _lo) // 2)

WIDTHS = [4, 12, 20, 36, 64, 100, 128, 132, 256, 260]
AGGS = ["sum", "mean", "max", "sqrt_n"]
ACTS = ["tanh", "ReLU", "leaky_relu", "elu", "selu", "gelu", None]


def _graph(rng, V, L):
    adj = []
    for l in range(L):
        kind = rng.integers(0, 4)
        if kind == 0:
            e = 0                                            # empty edge type
        elif kind == 1:
            e = int(rng.integers(1, 12))                     # a handful of edges
        else:
            e = int(rng.integers(V, 6 * V))
        src = rng.integers(0, V, e)
        tgt = rng.integers(0, max(1, V // (1 + int(kind == 3) * 7)), e)      # kind 3: few hot targets, many cold nodes
        adj.append(np.stack([src, tgt], 1).astype(np.int32).reshape(-1, 2))
    loops = np.stack([np.arange(V), np.arange(V)], 1).astype(np.int32)        # every node receives >= 1 message
    adj[0] = np.concatenate([loops, adj[0]])
    return adj, degree_table(adj, V)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    layer = ["rgcn", "ggnn", "rgat", "rgin", "film", "edge_mlp"][seed % 6]
    D = int(rng.choice(WIDTHS))
    L = int(rng.choice([1, 2, 3, 5, 12]))
    V = int(rng.integers(20, 260))
    agg = str(rng.choice(AGGS))
    act = ACTS[int(rng.integers(0, len(ACTS)))]
    return rng, layer, D, L, V, agg, act


def _build(seed, gpu_device, Ref, smooth=False):
    """(description, hip_call(h, weights), ref_call(h, weights), h, weights) for one seeded configuration; `Ref` is
    the oracle module to call (NumPy `oracle.gnns` or the autograd mirror `oracle.torch_ref`).  smooth=True swaps
    activations whose DERIVATIVE jumps at 0 (relu, leaky_relu, selu) for elu: with ~1e6 pre-activations per case one
    of them regularly lands within rounding of the kink and takes different branches in different-but-equally-valid
    fp32 evaluations (measured: the same 3.7e-3 gradient difference between HIP and torch-CPU fp32 appears and
    disappears under 1e-4 input noise), which says nothing about the kernels."""
    from tf_gnn_samples_amd import gnns as H
    rng, layer, D, L, V, agg, act = _case(seed)
    if smooth and (act is None and layer == "rgin" or (act or "").lower() in ("relu", "leaky_relu", "selu")):
        act = "elu"
    adj, deg = _graph(rng, V, L)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ln = layer_norm_weights(D, 2, rng)
    dev = lambda x: torch.as_tensor(x, device=gpu_device)
    adj_d, deg_d = [dev(a) for a in adj], dev(deg)
    if Ref is G:
        adj_r, deg_r = adj, deg
    else:
        adj_r, deg_r = [torch.as_tensor(a) for a in adj], torch.as_tensor(deg)
    steps = int(rng.integers(1, 3))
    if layer == "rgcn":
        norm = bool(rng.integers(0, 2))
        w = rgcn_weights(rng, L, D, D)
        a = act if act != "gelu" else "tanh"
        hip = lambda x, ww: H.sparse_rgcn_layer(x, adj_d, deg_d, D, steps, a, agg, norm, weights=ww)
        ref = lambda x, ww: Ref.sparse_rgcn_layer(x, adj_r, deg_r, D, steps, a, agg, norm, weights=ww)
    elif layer == "ggnn":
        w = rgcn_weights(rng, L, D, D)
        w["gru_cell/kernel"], w["gru_cell/recurrent_kernel"] = glorot(rng, (D, 3 * D)), glorot(rng, (D, 3 * D))
        w["gru_cell/bias"] = (0.1 * rng.standard_normal(3 * D)).astype(np.float32)
        ga = act if act not in ("gelu", None) else "tanh"
        hip = lambda x, ww: H.sparse_ggnn_layer(x, adj_d, D, steps, "gru", ga, agg, weights=ww)
        ref = lambda x, ww: Ref.sparse_ggnn_layer(x, adj_r, D, steps, "gru", ga, agg, weights=ww)
    elif layer == "rgat":
        K = int(rng.choice([k for k in (1, 2, 4) if D % k == 0]))
        w = rgcn_weights(rng, L, D, D)
        for l in range(L):
            w["Edge_%i_Attention_Parameters" % l] = (0.3 * rng.standard_normal(2 * D)).astype(np.float32)
        hip = lambda x, ww: H.sparse_rgat_layer(x, adj_d, D, K, steps, act, weights=ww)
        ref = lambda x, ww: Ref.sparse_rgat_layer(x, adj_r, D, K, steps, act, weights=ww)
    elif layer == "rgin":
        w = dict(ln)
        use_target = bool(rng.integers(0, 2))
        d_in = 2 * D if use_target else D
        for l in range(L):
            w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (d_in, D))
            w["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
        a = act if act is not None else "ReLU"
        hip = lambda x, ww: H.sparse_rgin_layer(x, adj_d, D, steps, a, agg, use_target, 1, None, weights=ww)
        ref = lambda x, ww: Ref.sparse_rgin_layer(x, adj_r, D, steps, a, agg, use_target, 1, None, weights=ww)
    elif layer == "film":
        norm = bool(rng.integers(0, 2))
        w = dict(rgcn_weights(rng, L, D, D), **ln)
        for l in range(L):
            w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
        hip = lambda x, ww: H.sparse_gnn_film_layer(x, adj_d, deg_d, D, steps, act, agg, norm, weights=ww)
        ref = lambda x, ww: Ref.sparse_gnn_film_layer(x, adj_r, deg_r, D, steps, act, agg, norm, weights=ww)
    else:
        norm, use_target, hidden = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 2))
        d_in = 2 * D if use_target else D
        w = dict(ln)
        for l in range(L):
            w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (d_in, D))
            if hidden:
                w["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
        hip = lambda x, ww: H.sparse_gnn_edge_mlp_layer(x, adj_d, deg_d, D, steps, act, agg, norm, use_target, hidden, weights=ww)
        ref = lambda x, ww: Ref.sparse_gnn_edge_mlp_layer(x, adj_r, deg_r, D, steps, act, agg, norm, use_target, hidden, weights=ww)
    return (layer, D, L, V, agg, act), hip, ref, h, w


@pytest.mark.parametrize("seed", GRAD_SEEDS)
def test_random_layer_gradients_match_fp64_autograd(gpu_device, seed):
    """Same sweep, gradients w.r.t. the node states and every weight against float64 autograd through the
    reference-order mirror (oracle/torch_ref.py)."""
    from oracle import torch_ref as R
    from tf_gnn_samples_amd.graph import clear_graph_cache
    desc, hip, ref, h, w = _build(seed, gpu_device, R, smooth=True)
    clear_graph_cache()
    hd = torch.as_tensor(h, device=gpu_device).requires_grad_(True)
    wd = {k: torch.as_tensor(v, device=gpu_device).requires_grad_(True) for k, v in w.items()}
    out = hip(hd, wd)
    gout = np.random.default_rng(seed).standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.as_tensor(gout, device=gpu_device))
    def mirror(dtype):
        hr = torch.as_tensor(h, dtype=dtype).requires_grad_(True)
        wr = {k: torch.as_tensor(v, dtype=dtype).requires_grad_(True) for k, v in w.items()}
        r = ref(hr, wr)
        r.backward(torch.as_tensor(gout, dtype=dtype))
        return r.detach(), hr.grad, {k: v.grad for k, v in wr.items()}

    r, gh, gw = mirror(torch.float64)
    scale_out = max(1.0, float(r.abs().max()))
    assert float(np.abs(out.detach().cpu().numpy() - r.numpy()).max()) < 2e-5 * scale_out, desc
    fp32_mirror = None
    for name, a, b in [("h", hd.grad, gh)] + [(k, wd[k].grad, gw[k]) for k in w]:
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, (desc, name)
            continue
        s = max(1.0, float(b.abs().max()))
        err = float(np.abs(a.cpu().numpy() - b.numpy()).max())
        if err < 1e-4 * s:
            continue
        # A pre-activation within fp32 rounding of a ReLU-type kink takes the other branch in float64: then the SAME
        # float32 computation on the CPU deviates from float64 just as much, and that mirror is the yardstick.
        if fp32_mirror is None:
            fp32_mirror = mirror(torch.float32)
        b32 = fp32_mirror[1] if name == "h" else fp32_mirror[2][name]
        err32 = float(np.abs(a.cpu().numpy() - b32.numpy()).max())
        assert err32 < 2e-5 * s, (desc, name, err, err32)


@pytest.mark.parametrize("seed", FWD_SEEDS)
def test_random_layer_configuration_matches_oracle(gpu_device, seed):
    from tf_gnn_samples_amd import gnns as H
    from tf_gnn_samples_amd.graph import clear_graph_cache
    rng, layer, D, L, V, agg, act = _case(seed)
    adj, deg = _graph(rng, V, L)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ln = layer_norm_weights(D, 2, rng)
    dev = lambda x: torch.as_tensor(x, device=gpu_device)
    dd = lambda w: {k: dev(v) for k, v in w.items()}
    adj_d, deg_d, h_d = [dev(a) for a in adj], dev(deg), dev(h)
    steps = int(rng.integers(1, 3))
    clear_graph_cache()
    if layer == "rgcn":
        norm = bool(rng.integers(0, 2))
        w = rgcn_weights(rng, L, D, D)
        act = act if act != "gelu" else "tanh"
        ref = G.sparse_rgcn_layer(h, adj, deg, D, steps, act, agg, norm, weights=w)
        out = H.sparse_rgcn_layer(h_d, adj_d, deg_d, D, steps, act, agg, norm, weights=dd(w))
    elif layer == "ggnn":
        w = rgcn_weights(rng, L, D, D)
        w["gru_cell/kernel"], w["gru_cell/recurrent_kernel"] = glorot(rng, (D, 3 * D)), glorot(rng, (D, 3 * D))
        w["gru_cell/bias"] = (0.1 * rng.standard_normal(3 * D)).astype(np.float32)
        ga = act if act not in ("gelu", None) else "tanh"
        ref = G.sparse_ggnn_layer(h, adj, D, steps, "gru", ga, agg, weights=w)
        out = H.sparse_ggnn_layer(h_d, adj_d, D, steps, "gru", ga, agg, weights=dd(w))
    elif layer == "rgat":
        heads = [k for k in (1, 2, 4) if D % k == 0]
        K = int(rng.choice(heads))
        w = rgcn_weights(rng, L, D, D)
        for l in range(L):
            w["Edge_%i_Attention_Parameters" % l] = (0.3 * rng.standard_normal(2 * D)).astype(np.float32)
        ref = G.sparse_rgat_layer(h, adj, D, K, steps, act, weights=w)
        out = H.sparse_rgat_layer(h_d, adj_d, D, K, steps, act, weights=dd(w))
    elif layer == "rgin":
        w = dict(ln)
        use_target = bool(rng.integers(0, 2))
        d_in = 2 * D if use_target else D
        for l in range(L):
            w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (d_in, D))
            w["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
        a = act if act is not None else "ReLU"
        ref = G.sparse_rgin_layer(h, adj, D, steps, a, agg, use_target, 1, None, weights=w)
        out = H.sparse_rgin_layer(h_d, adj_d, D, steps, a, agg, use_target, 1, None, weights=dd(w))
    elif layer == "film":
        norm = bool(rng.integers(0, 2))
        w = dict(rgcn_weights(rng, L, D, D), **ln)
        for l in range(L):
            w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
        ref = G.sparse_gnn_film_layer(h, adj, deg, D, steps, act, agg, norm, weights=w)
        out = H.sparse_gnn_film_layer(h_d, adj_d, deg_d, D, steps, act, agg, norm, weights=dd(w))
    else:
        norm, use_target, hidden = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 2))
        d_in = 2 * D if use_target else D
        w = dict(ln)
        for l in range(L):
            if hidden == 0:
                w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (d_in, D))
            else:
                w["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (d_in, D))
                w["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
        ref = G.sparse_gnn_edge_mlp_layer(h, adj, deg, D, steps, act, agg, norm, use_target, hidden, weights=w)
        out = H.sparse_gnn_edge_mlp_layer(h_d, adj_d, deg_d, D, steps, act, agg, norm, use_target, hidden, weights=dd(w))
    out = out.cpu().numpy()
    assert out.shape == ref.shape and np.isfinite(ref).all()
    scale = max(1.0, float(np.abs(ref).max()))
    tol = 2e-5 if act == "gelu" else 1e-5
    assert np.abs(out - ref).max() < tol * scale, (layer, D, L, V, agg, act, float(np.abs(out - ref).max()), scale)
