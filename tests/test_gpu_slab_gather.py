"""relgnn_slab_gather_f32 (csrc/slab_gather.hip): the aggregate-first gather with the gathered table tiled through LDS must be
BIT-IDENTICAL to relgnn_seg_reduce_fwd on the same batch — both fold every bucket sequentially in bucket order, product and sum
rounded separately — in both directions (by target: forward; by source: input gradient), with and without the 1/in-degree
scales, and its per-piece magnitudes must combine to the L2 kernel's row magnitudes.  Then a whole RGCN training step."""
import numpy as np
import pytest
import torch

from oracle.bookkeeping import GraphSample

pytestmark = pytest.mark.gpu
PAYLOADS = {"initial_node_features": ("node_features", np.float32), "target_labels": ("node_labels", np.float32)}


def _graphs(rng, n_graphs, L, max_nodes, heavy=False):
    out = []
    for g in range(n_graphs):
        n = int(rng.integers(1, max_nodes))
        adj = []
        for l in range(L):
            e = 0 if (g == 3 or (l == L - 1 and g % 4 == 0)) else int(rng.integers(0, (30 if heavy else 6) * n))
            src = rng.integers(0, n, e)
            tgt = (rng.lognormal(0, 1.0, e) * n / 8).astype(np.int64) % n if heavy else rng.integers(0, n, e)
            adj.append(np.stack([src, tgt], 1).astype(np.int64).reshape(-1, 2))
        deg = np.stack([np.bincount(a[:, 1], minlength=n) for a in adj])
        out.append(GraphSample(adj, deg, rng.standard_normal((n, 9)).astype(np.float32), (rng.random((n, 4)) < 0.3).astype(np.float32)))
    return out


@pytest.mark.parametrize("L,D,n_graphs,max_nodes,heavy", [(3, 256, 12, 300, True), (1, 8, 5, 40, False), (5, 64, 30, 70, False),
                                                          (3, 128, 3, 4700, True), (2, 1024, 4, 90, False)])
def test_slab_gather_is_bit_identical_to_the_l2_kernel(gpu_device, L, D, n_graphs, max_nodes, heavy):
    from tf_gnn_samples_amd import _lib, config, ops
    from tf_gnn_samples_amd.tasks.batcher import GraphStore
    from tf_gnn_samples_amd.tasks.resident import ResidentDataset
    rng = np.random.default_rng(L * 100 + D)
    graphs = _graphs(rng, n_graphs, L, max_nodes, heavy)
    store = GraphStore(graphs, L, PAYLOADS)
    resident = ResidentDataset(store, gpu_device)
    picks = [list(range(n_graphs)), [n_graphs - 1, 0, 0, 2], [1]]
    with config.override(gather="lds"):
        for ids in picks:
            b = resident.assemble(np.array(ids))
            g = b.graph
            assert getattr(g, "slab", None) is not None
            V = g.V
            w_t = g.degree_scale(b.type_to_num_incoming_edges)
            X = torch.randn((V, D), device=gpu_device)
            X[0, :3] = torch.tensor([float("inf"), float("nan"), -0.0], device=gpu_device)[:min(3, D)]
            for by_source in (False, True):
                plan = g.plan_transformed(w_t)
                for weighted in (True, False):
                    w = (plan.w_bwd(_lib.AGG_SUM) if by_source else w_t) if weighted else None
                    route = ops.slab_route(g, X, w, by_source)
                    assert route is not None
                    got, gmax = ops.slab_gather(g, X, route, weighted, True)
                    rowmax = torch.empty(V * L, device=gpu_device)
                    if by_source:
                        want = ops._seg_reduce_raw(_lib.AGG_SUM, X, g.rowptr_s, 1, g.tgt_s, w, V * L,
                                                   rowmax=rowmax if D > 128 and D <= 1024 else None)
                    else:
                        want = ops._seg_reduce_raw(_lib.AGG_SUM, X, g.rowptr_t, 1, g.src_t, w, V * L,
                                                   rowmax=rowmax if D > 128 and D <= 1024 else None)
                    same = (got == want) | (torch.isnan(got) & torch.isnan(want))
                    assert bool(same.all()), (ids, by_source, weighted, int((~same).sum()))
                    assert torch.equal(torch.signbit(got), torch.signbit(want))
                    if 128 < D <= 1024:
                        assert torch.equal(gmax.view(V * L, D // 8).amax(1), rowmax)
            # a weight tensor the lists do not stand for: the route must decline
            assert ops.slab_route(g, X, w_t.clone(), False) is None
            assert ops.slab_route(g, X[:, :D - 4] if D > 8 else X.double(), None, False) is None
    b = resident.assemble(np.array(picks[0]))
    assert getattr(b.graph, "slab", None) is None                      # default switch value: nothing attached


def test_a_training_step_on_the_lds_gather_equals_the_l2_one(gpu_device):
    """3-layer RGCN + PPI head on PPI-shaped batches of a resident fold, pair and triple arithmetic: same loss, same gradients —
    bit for bit on the triple (the gathers are bit-identical and nothing else changes), within the two-limb scales' rounding on the
    pair (its row scales come from 8-column pieces instead of whole bucket rows: the same power of two unless a piece is empty)."""
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.graph import clear_graph_cache
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(6, 1, seed=5)
    data = task._loaded_data[DataFold.TRAIN]

    def run(**switches):
        with config.override(**switches):
            clear_graph_cache()
            p = RGCN_Model.default_params()
            p.update(hidden_size=256, graph_num_layers=3, graph_layer_input_dropout_keep_prob=1.0, random_seed=0,
                     max_nodes_in_batch=7500)
            model = RGCN_Model(p, task, device=str(gpu_device))
            batch = next(iter(model._batches(data, DataFold.TRAIN)))
            assert batch.num_nodes >= 4096 and (getattr(batch.graph, "slab", None) is not None) == (switches.get("gather") == "lds")
            model.optimizer.zero_grad()
            m = model.forward_batch(batch, training=True)
            m['loss'].backward()
            torch.cuda.synchronize()
            return float(m['loss'].detach()), [model.variables[n].grad.detach().clone() for n in model.variables.names()]

    for limb in ("triple", "pair"):
        loss0, g0 = run(limb=limb, gather="l2")
        loss1, g1 = run(limb=limb, gather="lds")
        if limb == "triple":
            assert loss0 == loss1
            for a, b in zip(g0, g1):
                assert torch.equal(a, b)
        else:
            assert abs(loss0 - loss1) <= 1e-6 * max(1.0, abs(loss0))
            for a, b in zip(g0, g1):
                assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-12)
