"""How much of the 1e-5 ABSOLUTE budget each evaluation order of sparse_rgcn_layer uses, at BASELINE size, in the regimes
where a re-associated sum breaks first.

Three ways the HIP path can evaluate gnns/rgcn.py:84-114 (all the same function up to float32 rounding):
  aggregate_first_f32   default: bucket sums A_l[v] = sum 1/(c+1e-7) h_u (sequential float32), then ONE GEMM [V, L*D] @ [L*D, D]
  aggregate_first_f64   RELGNN_AGG_ACC=f64: the same with float64 bucket accumulators, rounded once (isolates the GEMM's share)
  transform_first       RELGNN_RGCN_ORDER=transform_first: T = H [W_0|..|W_L-1], then the sequential float32 fold of scaled rows
                        of T in the reference's message order (round 1's order; differs from the reference only by the order
                        inside each K = D dot product)
against (i) the float32 oracle in the reference's op order and (ii) the same function evaluated in float64 ("truth"); the
oracle's own distance from the truth is recorded next to them: where the oracle itself is 3e-6 from the truth, 1e-5 against the
oracle is a statement about two float32 roundings, not about the implementation.

The three orders run on whatever route RELGNN_GEMM selects for the Dense products (import-time switch of dense.py): the default
limb route (six bf16 MFMA products per fp32 product, csrc/limb_gemm.hip) or, with RELGNN_GEMM=lib, the exact-fp32 library GEMM.

Everything is written to gpurun_out/parity_margin.json (committed as profiles/r03_parity_margin.json for the default route and
profiles/r03_parity_margin_exact_fp32_gemm.json for RELGNN_GEMM=lib)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gnns as G
from helpers import PARITY_TOL, parity_errors, rgcn_weights, set_switch
from tf_gnn_samples_amd import config

pytestmark = pytest.mark.gpu

D = 256
_ROWS = []
VARIANTS = {
    "aggregate_first_f64": {"RELGNN_RGCN_ORDER": "aggregate_first", "RELGNN_AGG_ACC": "f64"},
    "aggregate_first_f32": {"RELGNN_RGCN_ORDER": "aggregate_first", "RELGNN_AGG_ACC": "f32"},
    "transform_first": {"RELGNN_RGCN_ORDER": "transform_first", "RELGNN_AGG_ACC": "f32"},
}
DEFAULT = "aggregate_first_f32"


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


@pytest.fixture(scope="module")
def c2_graph():
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    fd = mb.feed_dict
    return fd["adjacency_lists"], fd["type_to_num_incoming_edges"].astype(np.float32), mb.num_nodes


def _hub_graph(rng):
    """One 20 000-node graph with 40 targets of in-degree ~500 and 10 of ~2000 on top of a uniform background
    ([fwd, self, bkwd] like the PPI task, tasks/ppi_task.py:99-106)."""
    V = 20000
    src = [rng.integers(0, V, 200000)]
    tgt = [rng.integers(0, V, 200000)]
    hubs = rng.choice(V, 50, replace=False)
    for i, hnode in enumerate(hubs):
        deg = 2000 if i < 10 else 500
        src.append(rng.integers(0, V, deg))
        tgt.append(np.full(deg, hnode))
    fwd = np.stack([np.concatenate(src), np.concatenate(tgt)], axis=1).astype(np.int32)
    loops = np.stack([np.arange(V), np.arange(V)], axis=1).astype(np.int32)
    adj = [fwd, loops, np.ascontiguousarray(fwd[:, ::-1])]
    deg = np.stack([np.bincount(a[:, 1], minlength=V) for a in adj]).astype(np.float32)
    return adj, deg, V


def _run_case(monkeypatch, dev, name, adj, deg, V, h, w, normalize, bounded):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    ref32 = G.sparse_rgcn_layer(h, adj, deg, D, 1, "ReLU", "sum", normalize, weights=w, node_side_transform=True)
    truth = G.sparse_rgcn_layer(h.astype(np.float64), adj, deg.astype(np.float64), D, 1, "ReLU", "sum", normalize,
                                weights={k: v.astype(np.float64) for k, v in w.items()}, node_side_transform=True)
    row = {"case": name, "nodes": int(V), "messages": int(sum(len(a) for a in adj)), "normalize_by_num_incoming": bool(normalize),
           "max_abs_state_in": float(np.abs(h).max()), "max_abs_out": float(np.abs(truth).max()),
           "max_in_degree": int(deg.max()), "oracle_f32_vs_f64_truth_abs": parity_errors(ref32, truth)[0], "variants": {}}
    h_d, adj_d, deg_d, w_d = _dev(h, dev), _dev(adj, dev), _dev(deg, dev), _dev(w, dev)
    for vname, env in VARIANTS.items():
        for k, v in env.items():
            set_switch(monkeypatch, k, v)
        out = sparse_rgcn_layer(h_d, adj_d, deg_d, D, 1, "ReLU", "sum", normalize, weights=w_d).cpu().numpy()
        a, r = parity_errors(out, ref32)
        row["variants"][vname] = {"abs_vs_oracle_f32": a, "rel_vs_oracle_f32": r, "abs_vs_f64_truth": parity_errors(out, truth)[0]}
    _ROWS.append(row)
    d = row["variants"][DEFAULT]
    if bounded:       # the north-star regime: 1/in-degree-normalised sums of O(1) states
        assert d["abs_vs_oracle_f32"] <= PARITY_TOL, row
    # everywhere: the default order is as close to the float64 truth as the reference's own order is, up to a small factor
    assert d["abs_vs_f64_truth"] <= 3.0 * row["oracle_f32_vs_f64_truth_abs"] + 1e-6, row
    assert d["rel_vs_oracle_f32"] <= PARITY_TOL, row
    return row


def test_margin_c2_uniform_states(gpu_device, monkeypatch, c2_graph):
    adj, deg, V = c2_graph
    rng = np.random.default_rng(10)
    h = (rng.random((V, D), dtype=np.float32) * 2 - 1)
    _run_case(monkeypatch, gpu_device, "C2 batch, U(-1,1) states, Glorot weights", adj, deg, V, h, rgcn_weights(rng, 3, D, D),
              True, bounded=True)


def test_margin_c2_post_relu_states(gpu_device, monkeypatch, c2_graph):
    adj, deg, V = c2_graph
    rng = np.random.default_rng(11)
    h = np.minimum(np.abs(rng.standard_normal((V, D)) * 1.3), 4.0).astype(np.float32)     # non-negative, |h| up to 4
    _run_case(monkeypatch, gpu_device, "C2 batch, post-ReLU-chain-like states (>= 0, up to 4), Glorot weights", adj, deg, V, h,
              rgcn_weights(rng, 3, D, D), True, bounded=False)


def test_margin_c2_large_weights(gpu_device, monkeypatch, c2_graph):
    adj, deg, V = c2_graph
    rng = np.random.default_rng(12)
    h = (rng.random((V, D), dtype=np.float32) * 2 - 1)
    w = {k: (3.0 * v).astype(np.float32) for k, v in rgcn_weights(rng, 3, D, D).items()}
    _run_case(monkeypatch, gpu_device, "C2 batch, U(-1,1) states, weights at 3x Glorot scale", adj, deg, V, h, w, True, bounded=False)


def test_margin_hub_targets(gpu_device, monkeypatch):
    rng = np.random.default_rng(13)
    adj, deg, V = _hub_graph(rng)
    h = (rng.random((V, D), dtype=np.float32) * 2 - 1)
    _run_case(monkeypatch, gpu_device, "hub graph: 40 targets of in-degree 500, 10 of in-degree 2000", adj, deg, V, h,
              rgcn_weights(rng, 3, D, D), True, bounded=True)


def test_margin_c2_unnormalised(gpu_device, monkeypatch, c2_graph):
    adj, deg, V = c2_graph
    rng = np.random.default_rng(14)
    h = (rng.random((V, D), dtype=np.float32) * 2 - 1)
    _run_case(monkeypatch, gpu_device, "C2 batch, normalize_by_num_incoming=False (sums of up to ~300 messages)", adj, deg, V, h,
              rgcn_weights(rng, 3, D, D), False, bounded=False)


def test_zz_write_margin_report():
    for r in _ROWS:
        print("%s: max|out| %.3g, oracle-vs-truth %.2e" % (r["case"], r["max_abs_out"], r["oracle_f32_vs_f64_truth_abs"]))
        for v, e in r["variants"].items():
            print("    %-22s abs vs oracle %.2e  rel %.2e  abs vs truth %.2e" % (v, e["abs_vs_oracle_f32"], e["rel_vs_oracle_f32"],
                                                                                e["abs_vs_f64_truth"]))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_margin.json", "w") as f:
            json.dump({"tolerance_abs": PARITY_TOL, "default_variant": DEFAULT, "dense_product_route": config.settings.gemm + "/" + config.settings.limb,
                   "cases": _ROWS}, f, indent=1)
