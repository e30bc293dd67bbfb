#!/usr/bin/env python
"""The BASELINE.json configs other than the headline (configs[2..4]) on ONE MI355X, bounded to ~20 s, run by bench.py as a
child process after its timed region (rank 0, N = 1) so that their numbers land in the driver's record:

  C3  GGNN on real QM9 molecules (the 256 committed molecules tiled to one 50 000-node batch: 153 206 messages, 5 edge types),
      GRU cell, D = 128, 6 layers, mean and max aggregation
  C4  RGAT on the C2 PPI-shaped batch, D = 256, 4 heads, 3 layers
  C5  GNN-FiLM on one rank's share of the VarMisuse-shaped batch (42 graphs, ~1.0 M messages, 23 edge types), D = 128, 10 layers

Per config one JSON line: training-step time (forward + backward + clip + Adam on a fixed resident batch, bucketing rebuilt per
step), edges/s, and the algorithmic-bytes rate of the config's gather kernel(s) timed with HIP events on the launch stream
(same protocol as bench_roofline.py: SURVEY.md 8d byte model / average launch time)."""
import gzip
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _quiet(fn):
    so = sys.stdout
    sys.stdout = sys.stderr
    try:
        return fn()
    finally:
        sys.stdout = so


def _time_steps(model, batch, prime, steps):
    from tf_gnn_samples_amd.graph import clear_graph_cache
    import gc
    gc.collect()
    gc.freeze()                    # (the loaded graphs out of the collector's sight: bench.py says why)

    def step():
        clear_graph_cache()
        return model.train_step(batch)
    for _ in range(prime):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - t0) / steps * 1e3
    with torch.no_grad():
        clear_graph_cache(); model.forward_batch(batch, training=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            clear_graph_cache(); model.forward_batch(batch, training=False)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / steps * 1e3
    return train_ms, fwd_ms


def _time_kernel(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


def _time_captured_step(model, batch, steps):
    """The same training step (fixed resident batch, its bucketing part of the batch) recorded as ONE hipGraph and replayed:
    what a launch-bound config costs without the host's ~150 enqueues per step (models/sparse_graph_model.py: capture_train_step)."""
    try:
        cap = model.capture_train_step(batch)
        for _ in range(3):
            cap.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cap.replay()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / steps * 1e3, 3)
    except Exception as e:                 # the eager number stands on its own
        return "capture failed: %r" % (e,)


def _line(name, mb, train_ms, fwd_ms, kernel, **extra):
    row = {"config": name, "nodes": mb.num_nodes, "edges": mb.num_edges, "graphs": mb.num_graphs,
           "train_ms": round(train_ms, 3), "train_edges_per_s": round(mb.num_edges / train_ms * 1e3),
           "fwd_ms": round(fwd_ms, 3), "fwd_edges_per_s": round(mb.num_edges / fwd_ms * 1e3),
           "dominant_gather_kernel": kernel}
    row.update(extra)
    print(json.dumps(row), flush=True)


def _kernel_record(what, alg_bytes, ms):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"kernel": what, "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(ms, 4),
            "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3),
            "note": "HIP events on the launch stream, warm (back-to-back on one resident table, as inside a training step)"}


def run_c3(dev):
    from tf_gnn_samples_amd import _lib, ops
    from tf_gnn_samples_amd.graph import as_rel_graph
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, QM9_Task
    with gzip.open(ROOT / "tests" / "golden" / "qm9_valid_256.jsonl.gz", "rt") as f:
        raw = [json.loads(line) for line in f]
    task = QM9_Task(QM9_Task.default_params())
    samples = task.load_raw(raw * 11)
    mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, 50000))
    batch = DeviceBatch(mb, dev)
    g = as_rel_graph(batch.adjacency_lists, mb.num_nodes)
    D, L, V, M = 128, g.L, g.V, g.M
    T = torch.rand((V * L, D), device=dev) * 2 - 1
    plan = g.plan_transformed(None)
    for agg in ("mean", "max"):
        cls, extra = name_to_model_class("GGNN")
        p = cls.default_params(); p.update(hidden_size=D, graph_num_layers=6, graph_rnn_cell="GRU", message_aggregation_function=agg)
        model = _quiet(lambda: cls(p, task, device=str(dev)))
        train_ms, fwd_ms = _time_steps(model, batch, 6, 24)       # (4.5 ms steps: eight of them are a 36 ms window, which one host hiccup doubles)
        graph_ms = _time_captured_step(model, batch, 20)
        mode = ops.aggregation_mode_id(agg)
        ms = _time_kernel(lambda: ops._seg_reduce_raw(mode, T, plan.rowptr, plan.stride, plan.col, None, plan.num_out))
        k = _kernel_record("seg_reduce_group_kernel<32> (gather rows of the [V*L, 128] transformed table + segment-%s)" % agg,
                           M * (4 * D + 4) + V * 4 * D + 4 * (V * L + 1), ms)
        k["note"] += "; 80 MB per launch: launch-latency bound (SURVEY.md 8d), not a bandwidth statement"
        extra = {"train_ms_hipgraph": graph_ms,
                 "train_ms_hipgraph_what": "the same step on the same fixed batch (bucketing kept with the batch) as one captured "
                                           "hipGraph, replayed 20 times"}
        if isinstance(graph_ms, float):
            extra["train_edges_per_s_hipgraph"] = round(mb.num_edges / graph_ms * 1e3)
        _line("C3 GGNN / QM9 (real molecules), GRU, %s aggregation, D=128, 6 layers" % agg, mb, train_ms, fwd_ms, k, **extra)
        del model


def run_c4(dev):
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.graph import as_rel_graph
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, dev)
    cls, extra = name_to_model_class("RGAT")
    p = cls.default_params(); p.update(extra); p.update(hidden_size=256, graph_num_layers=3)
    model = _quiet(lambda: cls(p, task, device=str(dev)))
    train_ms, fwd_ms = _time_steps(model, batch, 6, 24)
    g = as_rel_graph(batch.adjacency_lists, mb.num_nodes)
    D, K, L, V, M = 256, 4, g.L, g.V, g.M
    T = torch.rand((V * L, D), device=dev) * 2 - 1
    s_src, s_tgt = torch.rand((V * L, K), device=dev), torch.rand((V * L, K), device=dev)
    with torch.no_grad():
        ms = _time_kernel(lambda: ops.rgat_attention(T, s_src, s_tgt, g, K))
    k = _kernel_record("rgat_alpha_kernel + headw_reduce_kernel (segmented softmax over all incoming messages + per-head weighted "
                       "gather-reduce: gnns/rgat.py:98-136 forward)", M * (4 * D + 8 + 4 * K) + V * (4 * D + 8 * K), ms)
    k["note"] += "; the 99 MB table lives in L2 + Infinity Cache at this size (cf. roofline.sizes c2): an L2-side rate"
    _line("C4 RGAT on the C2 PPI-shaped batch, D=256, 4 heads, 3 layers", mb, train_ms, fwd_ms, k)


def c5_task_and_graphs(num_graphs, seed=0, first_index=0):
    """The synthetic VarMisuse-shaped stand-in: 23 edge types, 128 features, one label per node through the PPI head (the
    VarMisuse task's own data pipeline and candidate head are out of scope)."""
    from tf_gnn_samples_amd.tasks import PPI_Task
    from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graph
    task = PPI_Task(PPI_Task.default_params())
    graphs = [make_varmisuse_shaped_graph(seed, first_index + i) for i in range(num_graphs)]
    g0 = graphs[0]
    task.restore_from_metadata({'num_edge_types': len(g0.adjacency_lists), 'initial_node_feature_size': g0.node_features.shape[1],
                                'num_labels': g0.node_labels.shape[1]})
    return task, graphs


def c5_model(task, dev):
    from tf_gnn_samples_amd.models import name_to_model_class
    cls, extra = name_to_model_class("GNN-FiLM")
    p = cls.default_params(); p.update(extra)
    p.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1,
             graph_residual_connection_every_num_layers=2)        # tasks/default_hypers/VarMisuse_GNN-FiLM.json
    return _quiet(lambda: cls(p, task, device=str(dev))), p


def run_c5(dev):
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.graph import as_rel_graph
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch
    task, graphs = c5_task_and_graphs(42)
    mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, dev)
    model, _ = c5_model(task, dev)
    # (6 + 8 steps: from a cold process one of the first ten steps of this model is a ~60 ms one, which a 3 + 5 window turned into
    #  "40 ms per step": scripts/exp_c5_steps.py)
    train_ms, fwd_ms = _time_steps(model, batch, 6, 8)
    g = as_rel_graph(batch.adjacency_lists, mb.num_nodes)
    pairs = g.pair_tables()
    D, V, M = 128, g.V, g.M
    T = torch.rand((pairs.P_s, D), device=dev) * 2 - 1
    film = torch.rand((pairs.P_t, 2 * D), device=dev)
    with torch.no_grad():
        ms = _time_kernel(lambda: ops.film_messages_reduce(T, film, g, None, "sum", "relu", pairs))
    k = _kernel_record("edge_fwd_kernel<FILM> (gather + FiLM modulate + ReLU + segment-sum: gnns/gnn_film.py:92-116 forward)",
                       M * (4 * D + 8) + pairs.tgt.num_pairs * 8 * D + V * 4 * D, ms)
    _line("C5 GNN-FiLM, one rank's share of the VarMisuse-shaped batch (23 edge types), D=128, 10 layers", mb, train_ms, fwd_ms, k)


def main():
    if not torch.cuda.is_available():
        raise SystemExit("bench_other.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    which = sys.argv[1:] or ["C3", "C4", "C5"]
    for name, fn in (("C3", run_c3), ("C4", run_c4), ("C5", run_c5)):
        if name in which:
            try:
                fn(dev)
            except Exception as e:            # one config failing must not hide the others
                print(json.dumps({"config": name, "error": repr(e)}), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
