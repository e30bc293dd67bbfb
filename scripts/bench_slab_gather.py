#!/usr/bin/env python
"""The aggregate-first gather of one C2 batch (16 PPI-shaped graphs, ~1.86 M messages, D = 256): seg_reduce_wave_kernel (one 1 KiB row
load per message through L1 / L2) against slab_gather_kernel (8-column slices of every graph's slab in LDS, sliced-ELL lists), by
target (forward) and by source (input gradient), with the 1/in-degree scales, warm (back-to-back) and behind a cache-evicting fill.
One JSON line per case; outputs compared bit for bit."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def timed(fn, evict=None, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(iters):
        if evict is not None:
            evict.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms)) * 1e3


def main():
    from tf_gnn_samples_amd import _lib, config, ops
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    from tf_gnn_samples_amd.tasks.resident import ResidentDataset
    dev = torch.device("cuda:0")
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    store = task.make_graph_store(task._loaded_data[DataFold.TRAIN])
    resident = ResidentDataset(store, dev)
    with config.override(gather="lds"):
        b = resident.assemble(np.arange(16))
    g = b.graph
    V, L, D = g.V, g.L, 256
    w_t = g.degree_scale(b.type_to_num_incoming_edges)
    plan = g.plan_transformed(w_t)
    w_s = plan.w_bwd(_lib.AGG_SUM)
    X = torch.rand((V, D), device=dev) * 2 - 1
    evict = torch.empty(1 << 28, device=dev)            # 1 GiB streaming write: L2 + Infinity Cache evicted
    fold = g.slab.fold
    for name, by_source, w, rowptr, col, d in (("by target (forward)", False, w_t, g.rowptr_t, g.src_t, fold.by_target),
                                                ("by source (input gradient)", True, w_s, g.rowptr_s, g.tgt_s, fold.by_source)):
        rowmax = torch.empty(V * L, device=dev)
        with config.override(gather="lds"):
            route = ops.slab_route(g, X, w, by_source)
            got, _ = ops.slab_gather(g, X, route, True, True)
        want = ops._seg_reduce_raw(_lib.AGG_SUM, X, rowptr, 1, col, w, V * L, rowmax=rowmax)
        row = {"case": name, "messages": g.M, "nodes": V, "D": D, "bit_identical": bool(torch.equal(got, want)),
               "ell_entries": d.entries, "ell_padding": round(d.entries / max(d.messages, 1) - 1.0, 4), "fold_padding": round(d.lane_steps / max(d.messages, 1) - 1.0, 4), "longest_bucket": d.max_bucket}
        for label, ev in (("warm", None), ("cold", evict)):
            row["l2_kernel_us_" + label] = round(timed(lambda: ops._seg_reduce_raw(_lib.AGG_SUM, X, rowptr, 1, col, w, V * L, rowmax=rowmax), ev), 1)
            row["lds_kernel_us_" + label] = round(timed(lambda: ops.slab_gather(g, X, route, True, True), ev), 1)
            row["lds_kernel_no_rowmax_us_" + label] = round(timed(lambda: ops.slab_gather(g, X, route, True, False), ev), 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
