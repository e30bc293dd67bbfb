"""Parity at the sizes BASELINE.json states, every output row, at the north-star tolerance (1e-5 ABSOLUTE on fp32
node states for the normalised layers; max-abs and max-rel recorded for all):

  C2  the full synthetic PPI-shaped batch of the bench (16 graphs, 32 203 nodes, 1 854 895 messages, 3 edge types,
      h = 256): three chained sparse_rgcn_layer calls and the whole 3-layer RGCN_Model forward, HIP vs the C-backed
      oracle (oracle.gnns.sparse_rgcn_layer(node_side_transform=True): the reference's message values and
      sequential fp32 fold in the reference's message order; tests/test_oracle.py pins it to the op-for-op path)
  C4  sparse_rgat_layer (h = 256, 4 heads) on the same C2 batch
  C5  sparse_gnn_film_layer on a VarMisuse-shaped batch of one rank's share (~1.04 M messages, 23 edge types, h = 128)
  C3  sparse_ggnn_layer (GRU, mean / max) on one 50 000-node batch of real QM9 molecules (153 206 messages, 5 edge types)

The oracle needs a few seconds per layer at these sizes (host BLAS + the sequential C fold)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bookkeeping, gnns as G, model as OM
from helpers import assert_parity, glorot, parity_log, rgcn_weights

pytestmark = pytest.mark.gpu


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


@pytest.fixture(scope="module")
def c2():
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes == 32203 and mb.num_edges == 1854895          # the batch the bench line and DESIGN.md quote
    return task, mb


def test_c2_full_batch_three_rgcn_layers_every_row(gpu_device, c2):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    _, mb = c2
    fd = mb.feed_dict
    rng = np.random.default_rng(0)
    D = 256
    adj, deg = fd["adjacency_lists"], fd["type_to_num_incoming_edges"].astype(np.float32)
    adj_d, deg_d = _dev(adj, gpu_device), _dev(deg, gpu_device)
    h_ref = (rng.random((mb.num_nodes, D), dtype=np.float32) * 2 - 1)     # post-tanh-like states
    h_hip = _dev(h_ref, gpu_device)
    for layer in range(3):
        w = rgcn_weights(rng, 3, D, D)
        h_ref = G.sparse_rgcn_layer(h_ref, adj, deg, D, 1, "ReLU", "sum", weights=w, node_side_transform=True)
        h_hip = sparse_rgcn_layer(h_hip, adj_d, deg_d, D, 1, "ReLU", "sum", weights=_dev(w, gpu_device))
        assert h_hip.shape == h_ref.shape
        assert_parity(h_hip, h_ref, strict_abs=True, what="C2 full batch rgcn layer %d (chained)" % layer)
    assert float(np.abs(h_ref).max()) > 0.05        # the chain did not collapse to zero


def test_c2_full_batch_rgcn_model_forward(gpu_device, c2):
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DeviceBatch
    task, mb = c2
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3, graph_activation_function="ReLU", message_aggregation_function="sum",
             graph_layer_input_dropout_keep_prob=1.0)                      # README.md:32 of the reference
    model = RGCN_Model(p, task, device=str(gpu_device))
    batch = DeviceBatch(mb, gpu_device)
    with torch.no_grad():
        final = model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists,
                                                         batch.type_to_num_incoming_edges)
    W = {n[len("graph_model/"):]: model.variables[n].detach().cpu().numpy() for n in model.variables.names()
         if n.startswith("graph_model/")}
    fd = mb.feed_dict
    ref = OM.graph_propagation(fd['initial_node_features'].astype(np.float32), fd['adjacency_lists'],
                               fd['type_to_num_incoming_edges'].astype(np.float32), p, W,
                               OM.rgcn_apply(p, node_side_transform=True))
    assert_parity(final, ref, strict_abs=True, what="C2 full batch RGCN_Model forward (3 layers + dense)")


def test_c2_full_batch_rgcn_layer_against_the_op_for_op_oracle(gpu_device, c2):
    """One layer on the full C2 batch against the oracle in the REFERENCE's op order (gnns/rgcn.py:84-112: gather the source
    rows of every edge, per-edge [E_l, D] @ [D, D], scale, concat, sequential segment sum — 2 GB of messages, ~30 s of host
    time) — not the node-side evaluation order the other C2 cases use.  The two oracle orders must agree BIT FOR BIT at this
    size too (a row of a matrix product does not depend on which other rows are in the batch; tests/test_oracle.py pins the
    same on small graphs), which is what lets the chained / whole-model cases use the cheap one."""
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    _, mb = c2
    fd = mb.feed_dict
    rng = np.random.default_rng(0)
    D = 256
    adj, deg = fd["adjacency_lists"], fd["type_to_num_incoming_edges"].astype(np.float32)
    h = (rng.random((mb.num_nodes, D), dtype=np.float32) * 2 - 1)
    w = rgcn_weights(rng, 3, D, D)
    ref = G.sparse_rgcn_layer(h, adj, deg, D, 1, "ReLU", "sum", weights=w, node_side_transform=False)
    cheap = G.sparse_rgcn_layer(h, adj, deg, D, 1, "ReLU", "sum", weights=w, node_side_transform=True)
    assert np.array_equal(ref, cheap), "the oracle's node-side order deviates from the op-for-op order at full size"
    out = sparse_rgcn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 1, "ReLU", "sum",
                            weights=_dev(w, gpu_device))
    assert_parity(out, ref, strict_abs=True, what="C2 full batch rgcn layer vs op-for-op oracle (per-edge matmul order)")


GRADIENT_ROUTES = {"pair": dict(gemm="limb", limb="pair"), "triple": dict(gemm="limb", limb="triple"), "lib": dict(gemm="lib")}
_GRADIENT_ROWS = []


@pytest.mark.parametrize("route", list(GRADIENT_ROUTES))
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_c2_full_batch_model_gradients(gpu_device, c2, monkeypatch, seed, route):
    """d loss / d (every variable) and d loss / d (input features) of the 3-layer RGCN + PPI head on the FULL C2 batch:
    the HIP path's backward (by-source gather-reduce, GEMMs with the stacked kernels, streaming / split-K weight
    gradients, fused loss) against float64 autograd through the torch mirror of the driver (oracle/torch_model.py) with
    the layer evaluated as sparse products (oracle/torch_ref.py:sparse_rgcn_layer_lean, pinned to the op-for-op mirror
    in tests/test_oracle_crosscheck_cpu.py) — for five model initialisations on each arithmetic of the Dense products
    (two fp16 limbs, three bf16 limbs, exact fp32).

    FLIP-AWARE: the float64 mirror evaluates every ReLU with the branch the HIP forward took (oracle/torch_ref.py:
    sparse_rgcn_layer_lean, relu_mask).  A ReLU unit whose float32 pre-activation lies within the forward error (~3e-6) of
    zero takes the other branch than in float64 — a few of the 25 M units, which ones is chance — and its gradient path (a
    rank-one term ~1e-6 .. 1e-5 in every weight gradient upstream) then exists in one run and not in the other: round 3's
    comparison sat at 9.4e-6 of its 1e-5 bar for some draws on either arithmetic for that reason alone
    (profiles/r03_c_gradient_parity_by_seed.txt).  With the branches prescribed both runs differentiate the same
    piecewise-linear function and the assert measures arithmetic: bar 5e-6 abs, half the north-star tolerance.  The number of
    units whose branch differs, the largest |pre-activation| among them (must be within the forward error: otherwise it is a
    bug, not a knife edge) and the plain comparison's numbers are recorded next to it (profiles/r04_gradient_parity_by_seed.json)."""
    from oracle import torch_model as TM
    from tf_gnn_samples_amd import config, ops
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DeviceBatch
    task, mb = c2
    for name, value in GRADIENT_ROUTES[route].items():
        monkeypatch.setattr(config.settings, name, value)
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3, graph_activation_function="ReLU", message_aggregation_function="sum",
             graph_layer_input_dropout_keep_prob=1.0, random_seed=seed)
    model = RGCN_Model(p, task, device=str(gpu_device))
    batch = DeviceBatch(mb, gpu_device)
    batch.wait_ready()
    x_hip = batch.initial_node_features.detach().clone().requires_grad_(True)
    batch.initial_node_features = x_hip
    model.optimizer.zero_grad()
    layer_outputs = []                                        # the HIP forward's post-ReLU states, layer by layer
    real = ops.aggregate_then_transform

    def recording(*args, **kwargs):
        out = real(*args, **kwargs)
        layer_outputs.append(out.detach())
        return out
    monkeypatch.setattr(ops, "aggregate_then_transform", recording)
    metrics = model.forward_batch(batch, training=True)
    monkeypatch.setattr(ops, "aggregate_then_transform", real)
    metrics['loss'].backward()
    torch.cuda.synchronize()
    assert len(layer_outputs) == 3
    masks = [(o > 0).cpu() for o in layer_outputs]

    names = list(model.variables.names())
    fd = mb.feed_dict
    adj = [torch.as_tensor(a) for a in fd['adjacency_lists']]
    deg = torch.as_tensor(fd['type_to_num_incoming_edges'].astype(np.float32))

    def float64_gradients(relu_masks):
        W = {n[len("graph_model/"):]: model.variables[n].detach().cpu().double().requires_grad_(True)
             for n in names if n.startswith("graph_model/")}
        head = {n: model.variables[n].detach().cpu().double().requires_grad_(True) for n in names if not n.startswith("graph_model/")}
        x = torch.as_tensor(fd['initial_node_features']).double().requires_grad_(True)
        pre = []
        final = TM.graph_propagation(x, adj, deg, p, W, TM.rgcn_apply(p, lean=True, relu_masks=relu_masks, pre_activations=pre))
        kernel = next(v for k, v in head.items() if k.endswith("kernel"))
        bias = next(v for k, v in head.items() if k.endswith("bias"))
        loss = TM.ppi_loss(final, torch.as_tensor(fd['target_labels']).double(), kernel, bias)
        loss.backward()
        grads = {n: (W[n[len("graph_model/"):]] if n.startswith("graph_model/") else head[n]).grad.numpy() for n in names}
        grads["initial_node_features"] = x.grad.numpy()
        return float(loss), grads, pre

    loss64, ref, pre = float64_gradients(masks)
    assert abs(float(metrics['loss']) - loss64) <= 1e-5 * max(1.0, abs(loss64))
    # the units whose branch differs between the two arithmetics: how many, and how far from zero the float64 pre-activation is
    flipped, knife = 0, 0.0
    for z, m in zip(pre, masks):
        diff = (z > 0) != m
        flipped += int(diff.sum())
        if bool(diff.any()):
            knife = max(knife, float(z[diff].abs().max()))
    assert knife <= 2e-5, "a ReLU unit %g away from zero took the other branch: not a rounding knife edge" % knife

    got = {n: model.variables[n].grad.detach().cpu().numpy().astype(np.float64) for n in names}
    got["initial_node_features"] = x_hip.grad.detach().cpu().numpy().astype(np.float64)
    aware = {n: float(np.abs(got[n] - ref[n]).max()) for n in ref}
    fro = {n: float(np.linalg.norm(got[n] - ref[n]) / max(np.linalg.norm(ref[n]), 1e-300)) for n in ref}
    row = {"seed": seed, "route": route, "relu_units_with_other_branch": flipped, "largest_abs_preactivation_among_them": knife,
           "flip_aware_max_abs": max(aware.values()), "flip_aware_worst_gradient": max(aware, key=aware.get),
           "flip_aware_max_rel_frobenius": max(fro.values()),
           "flip_aware_max_abs_per_gradient": {k.split("/", 1)[-1]: "%.1e" % v for k, v in aware.items()}}
    # the plain comparison (float64 decides its own branches), for the record — what round 3 asserted on; it doubles the host time of
    # the test, so it runs on request (RELGNN_TEST_PLAIN_GRADIENTS=1: profiles/r04_a_gradient_parity_by_seed.json has all 15 cases:
    # up to 4.8e-5 on the EXACT-fp32 route, seed 2, where the flip-aware comparison says 1.9e-7)
    if flipped and os.environ.get("RELGNN_TEST_PLAIN_GRADIENTS") == "1":
        _, plain_ref, _ = float64_gradients(None)
        plain = {n: float(np.abs(got[n] - plain_ref[n]).max()) for n in plain_ref}
        row["plain_max_abs"], row["plain_worst_gradient"] = max(plain.values()), max(plain, key=plain.get)
    elif not flipped:
        row["plain_max_abs"], row["plain_worst_gradient"] = row["flip_aware_max_abs"], row["flip_aware_worst_gradient"]
    _GRADIENT_ROWS.append(row)
    print(row)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gradient_parity_by_seed.json", "w") as f:
        json.dump({"what": "C2 full batch, 3-layer RGCN + PPI head: max |HIP gradient - float64 gradient| over every variable and "
                           "the input features; flip_aware = the float64 mirror takes the HIP forward's ReLU branches",
                   "bar_flip_aware": 5e-6, "rows": _GRADIENT_ROWS}, f, indent=1)
    for n, e in aware.items():
        assert e <= 5e-6, ("C2 full batch d loss / d %s (%s, seed %d)" % (n, route, seed), e)
    assert max(fro.values()) <= 2e-3


def test_c4_rgat_on_the_c2_batch(gpu_device, c2):
    from tf_gnn_samples_amd.gnns import sparse_rgat_layer
    _, mb = c2
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
    assert_parity(out, ref, strict_abs=True, what="C4 rgat on the full C2 batch")


def test_c5_film_rank_share(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_gnn_film_layer
    from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graphs
    graphs = make_varmisuse_shaped_graphs(40, seed=0)      # ~100 k nodes, ~1.04 M messages: 1/8 of BASELINE's 10 M
    L = 23
    samples = [bookkeeping.GraphSample(g.adjacency_lists, g.type_to_node_to_num_incoming_edges, g.node_features, None)
               for g in graphs]
    b = next(bookkeeping.pack_batches(samples, L, 10 ** 9))
    rng = np.random.default_rng(2)
    D, V = 128, b["num_nodes"]
    M = sum(len(a) for a in b["adjacency_lists"])
    assert 0.9e6 < M < 1.2e6, M
    w = dict(rgcn_weights(rng, L, D, D), **{"LayerNorm/gamma": np.ones(D, np.float32), "LayerNorm/beta": np.zeros(D, np.float32)})
    for l in range(L):
        w["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    adj = [a.astype(np.int32) for a in b["adjacency_lists"]]
    deg = b["type_to_num_incoming_edges"].astype(np.float32)
    ref = G.sparse_gnn_film_layer(h, adj, deg, D, 1, "ReLU", "sum", False, weights=w)
    out = sparse_gnn_film_layer(_dev(h, gpu_device), _dev(adj, gpu_device), _dev(deg, gpu_device), D, 1, "ReLU", "sum", False,
                                weights=_dev(w, gpu_device))
    # un-normalised sum followed by layer norm (outputs up to |7.5|): measured 6.0e-6 abs = 7.9e-7 relative — inside the north-star
    # 1e-5 ABSOLUTE budget, so that is what is asserted (round 5; a relative budget before)
    assert_parity(out, ref, strict_abs=True, what="C5 film, one rank's share (%d messages)" % M)


@pytest.mark.parametrize("agg", ["mean", "max"])
def test_c3_ggnn_qm9_full_batch(gpu_device, agg):
    """BASELINE configs[2] at its stated size: one max_nodes_in_batch = 50 000 batch of REAL QM9 molecules (the 256
    committed molecules repeated 11x: 2 768 graphs, 49 986 nodes, 153 206 messages, 5 edge types), GGNN with the GRU
    cell, D = 128, three timesteps, mean / max aggregation."""
    from test_golden_cpu import read_qm9_fixture
    from tf_gnn_samples_amd.gnns import sparse_ggnn_layer
    from tf_gnn_samples_amd.tasks import DataFold, QM9_Task
    task = QM9_Task(QM9_Task.default_params())
    samples = task.load_raw(read_qm9_fixture() * 11)
    mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, 50000))
    assert mb.num_nodes == 49986 and mb.num_edges == 153206 and task.num_edge_types == 5
    fd = mb.feed_dict
    rng = np.random.default_rng(0)
    D, L = 128, 5
    w = rgcn_weights(rng, L, D, D)
    w.update({"gru_cell/kernel": glorot(rng, (D, 3 * D)), "gru_cell/recurrent_kernel": glorot(rng, (D, 3 * D)),
              "gru_cell/bias": (0.05 * rng.standard_normal(3 * D)).astype(np.float32)})
    h = np.tanh(fd["initial_node_features"].astype(np.float32) @ glorot(rng, (15, D)))
    adj = fd["adjacency_lists"]
    ref = G.sparse_ggnn_layer(h, adj, D, 3, "GRU", "tanh", agg, weights=w)
    out = sparse_ggnn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, 3, "GRU", "tanh", agg, weights=_dev(w, gpu_device))
    assert_parity(out, ref, strict_abs=True, what="C3 ggnn/qm9 full batch %s" % agg)     # GRU states live in (-1, 1)


def test_zz_report_parity_numbers():
    """Not a check: prints the recorded max-abs / max-rel of this module's cases (and writes them under gpurun_out/)."""
    import json
    import os
    rows = [r for r in parity_log() if r["what"].startswith(("C2", "C3", "C4", "C5"))]
    for r in rows:
        print("%-60s abs %.3e  rel %.3e  max|ref| %.3g" % (r["what"], r["abs"], r["rel"], r["max_ref"]))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_baseline_size.json", "w") as f:
            json.dump(rows, f, indent=1)
