"""Hubs: buckets far longer than one wave should fold alone (ops.LONG_SEGMENT).  A validated RelGraph finds them and routes
the gather / reduce launches through chunked virtual rows (ops.SplitPlan); results must equal the unsplit kernel's up to the
re-association of the hub's sum, gradients included, for every aggregation."""
import numpy as np
import pytest
import torch

from oracle import gnns as G
from helpers import degree_table, rgcn_weights

pytestmark = pytest.mark.gpu


def _hub_graph(rng, V=3000, hub_in=30000, hub_out=20000, L=2):
    """type 0: node 0 receives hub_in edges and node 1 sends hub_out edges, plus background; type 1: sparse background"""
    bg = np.stack([rng.integers(0, V, 8000), rng.integers(0, V, 8000)], 1)
    into_hub = np.stack([rng.integers(0, V, hub_in), np.zeros(hub_in, np.int64)], 1)
    from_hub = np.stack([np.ones(hub_out, np.int64), rng.integers(0, V, hub_out)], 1)
    adj0 = np.concatenate([bg, into_hub, from_hub])
    rng.shuffle(adj0)
    adj1 = np.stack([rng.integers(0, V, 2000), rng.integers(0, V, 2000)], 1)
    return [adj0.astype(np.int32), adj1.astype(np.int32)], V


@pytest.mark.parametrize("agg", ["sum", "mean", "sqrt_n", "max"])
def test_hub_buckets_are_split_and_match(gpu_device, agg):
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(0)
    adj, V = _hub_graph(rng)
    D, L = 64, 2
    deg = degree_table(adj, V)
    w = rgcn_weights(rng, L, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    to = lambda a: torch.as_tensor(a, device=gpu_device)
    adj_d, deg_d, w_d = [to(a) for a in adj], to(deg), {k: to(v) for k, v in w.items()}

    split = RelGraph(adj_d, V, validate=True)                        # check() reads the bucket lengths back
    assert split.has_long_buckets
    assert 1 in getattr(split.rowptr_t, "_relgnn_split") and 1 in getattr(split.rowptr_s, "_relgnn_split")
    plain = RelGraph(adj_d, V, validate="deferred")                  # one wave per bucket
    assert not plain.has_long_buckets and getattr(plain.rowptr_t, "_relgnn_split", None) is None

    outs, grads = [], []
    for g in (split, plain):
        x = to(h).requires_grad_(True)
        ww = {k: v.clone().requires_grad_(True) for k, v in w_d.items()}
        out = sparse_rgcn_layer(x, g, deg_d, D, 1, "tanh", agg, weights=ww)
        (out * out).sum().backward()
        outs.append(out.detach())
        grads.append([x.grad] + [ww[k].grad for k in sorted(ww)])
    ref = G.sparse_rgcn_layer(h, adj, deg, D, 1, "tanh", agg, weights=w)
    # the hub sums 30 000 messages of O(1)/30 000 each: both orders are within fp32 rounding of the oracle
    assert np.abs(outs[0].cpu().numpy() - ref).max() < 1e-5
    assert np.abs(outs[1].cpu().numpy() - ref).max() < 1e-5
    for a, b in zip(grads[0], grads[1]):                             # hub gradients are sums of 2e4..3e4 terms
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), ((a - b).abs().max(), b.abs().max())
    # un-normalised sums too (no 1/degree): the raw kernel on the (target, type) buckets, split vs unsplit
    X = to(h)
    for mode in ("sum", "max"):
        a = ops.seg_gather_reduce(X, split.plan_untransformed(None), mode, None)
        b = ops.seg_gather_reduce(X, plain.plan_untransformed(None), mode, None)
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-3 if mode == "sum" else 0.0), (a - b).abs().max()


def test_split_plan_shapes(gpu_device):
    from tf_gnn_samples_amd.ops import SplitPlan
    rowptr = torch.tensor([0, 0, 5, 5, 10005, 10006, 10006], dtype=torch.int32, device=gpu_device)
    sp = SplitPlan(rowptr, 1, 6, chunk=4096)
    assert sp.num_virtual == 1 + 1 + 1 + 3 + 1 + 1
    assert sp.virtual_rowptr.tolist() == [0, 0, 5, 5, 4101, 8197, 10005, 10006, 10006]
    assert sp.combine_rowptr.tolist() == [0, 1, 2, 3, 6, 7, 8]
    merged = SplitPlan(rowptr, 2, 3, chunk=4096)                      # buckets merged two by two: [0,5), [5,10005), [10005,10006)
    assert merged.virtual_rowptr.tolist() == [0, 5, 4101, 8197, 10005, 10006]
    assert merged.combine_rowptr.tolist() == [0, 1, 4, 5]


def test_hub_speedup(gpu_device):
    """the point of splitting: a 200 000-message bucket must not take one wave's sequential time"""
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(1)
    V, E = 4096, 200000
    adj = [torch.as_tensor(np.stack([rng.integers(0, V, E), np.zeros(E, np.int64)], 1).astype(np.int32), device=gpu_device)]
    X = torch.rand(V, 256, device=gpu_device)
    split, plain = RelGraph(adj, V, validate=True), RelGraph(adj, V, validate="deferred")

    def t(g):
        plan = g.plan_untransformed(None)
        for _ in range(3):
            ops.seg_gather_reduce(X, plan, "sum", None)
        best = float("inf")
        for _ in range(4):                      # best of four rounds: one allocator / clock hiccup must not decide a test
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                out = ops.seg_gather_reduce(X, plan, "sum", None)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 5)
        return best, out
    ts, a = t(split)
    tp, b = t(plain)
    assert torch.allclose(a, b, rtol=1e-4, atol=0.5)
    assert ts < 0.5 * tp, (ts, tp)


def _layer_case(name, rng, L, D):
    """(hip layer call, oracle layer call, weights) for one of the layers whose fused kernels give a whole target to one wave"""
    from helpers import glorot, layer_norm_weights
    from tf_gnn_samples_amd import gnns as HG
    w = rgcn_weights(rng, L, D, D)
    if name == "film":
        for l in range(L):
            w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
        w.update(layer_norm_weights(D, 1, rng))
        hip = lambda x, adj, deg, ww: HG.sparse_gnn_film_layer(x, adj, deg, D, 1, "ReLU", "sum", True, weights=ww)
        ora = lambda x, adj, deg, ww: G.sparse_gnn_film_layer(x, adj, deg, D, 1, "ReLU", "sum", True, weights=ww)
    elif name == "edge_mlp0":
        w = {"Edge_%i_MLP/dense/kernel" % l: glorot(rng, (2 * D, D)) for l in range(L)}
        w.update(layer_norm_weights(D, 1, rng))
        hip = lambda x, adj, deg, ww: HG.sparse_gnn_edge_mlp_layer(x, adj, deg, D, 1, "gelu", "sum", True, True, 0, weights=ww)
        ora = lambda x, adj, deg, ww: G.sparse_gnn_edge_mlp_layer(x, adj, deg, D, 1, "gelu", "sum", True, True, 0, weights=ww)
    else:
        for l in range(L):
            w["Edge_%i_Attention_Parameters" % l] = glorot(rng, (2 * D, 1))[:, 0]
        hip = lambda x, adj, deg, ww: HG.sparse_rgat_layer(x, adj, D, 4, 1, "tanh", weights=ww)
        ora = lambda x, adj, deg, ww: G.sparse_rgat_layer(x, adj, D, 4, 1, "tanh", weights=ww)
    return hip, ora, w


@pytest.mark.parametrize("layer", ["film", "edge_mlp0", "rgat"])
def test_hub_targets_in_the_one_wave_per_target_layers(gpu_device, layer):
    """GNN-FiLM, GNN-Edge-MLP (pair kernels) and RGAT give a whole target node to one wave / lane group.  On a graph KNOWN to
    hold hubs (validated RelGraph: 30 000 messages into one node) they take their materialised routes, whose reductions go
    through the gather-reduce kernel with chunked virtual rows: values against the oracle, gradients against the fused one-wave
    route on the same graph (validate="deferred": the hub walked by a single wave — slow, but the same function)."""
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(3)
    adj, V = _hub_graph(rng, V=2000, hub_in=30000, hub_out=20000)
    D, L = 64, 2
    deg = degree_table(adj, V)
    hip, ora, w = _layer_case(layer, rng, L, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    to = lambda a: torch.as_tensor(a, device=gpu_device)
    adj_d, deg_d = [to(a) for a in adj], to(deg)
    # float64 oracle: the hub's un-normalised FiLM sum (30 000 terms relu(beta + O(1e-5)), ~1e4 in total, then a layer norm
    # across features) carries ~1e-4..1e-3 of float32 summation-order noise in EITHER order, the sequential oracle included
    ref = ora(h.astype(np.float64), adj, deg.astype(np.float64), {k: v.astype(np.float64) for k, v in w.items()})
    tol = 3e-3 if layer == "film" else 2e-5
    hub_safe = RelGraph(adj_d, V, validate=True)
    one_wave = RelGraph(adj_d, V, validate="deferred")
    assert hub_safe.has_long_buckets and not one_wave.has_long_buckets
    outs, grads = [], []
    for g in (hub_safe, one_wave):
        x = to(h).requires_grad_(True)
        ww = {k: to(v).requires_grad_(True) for k, v in w.items()}
        out = hip(x, g, deg_d, ww)
        (out * out).sum().backward()
        outs.append(out.detach().cpu().numpy())
        grads.append([x.grad] + [ww[k].grad for k in sorted(ww)])
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0] - ref).max() < tol * scale, np.abs(outs[0] - ref).max()
    assert np.abs(outs[1] - ref).max() < tol * scale, np.abs(outs[1] - ref).max()
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 25 * tol * max(1.0, float(b.abs().max())), (float((a - b).abs().max()), float(b.abs().max()))
