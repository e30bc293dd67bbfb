#!/usr/bin/env python
"""Why does the aggregate-first layer's forward product take 78 us inside a training step and 56 us in a benchmark loop?
Times relgnn_limb16_gemm_xf32 on the C2 batch (a) right behind the gather that produced its left operand (what a step does),
(b) a second time on the same operand (what a benchmark loop does), (c) behind a cache-evicting fill; same for the triple."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    from tf_gnn_samples_amd import _lib, config, dense as DN, ops
    from tf_gnn_samples_amd.graph import RelGraph
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    dev = torch.device("cuda:0")
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, dev)
    g = RelGraph(batch.adjacency_lists, mb.num_nodes)
    V, L, D = g.V, g.L, 256
    w = g.degree_scale(batch.type_to_num_incoming_edges)
    H = torch.rand((V, D), device=dev) * 2 - 1
    kernels = [(torch.rand((D, D), device=dev) * 2 - 1) * 0.1 for _ in range(L)]
    evict = torch.empty(1 << 28, device=dev)
    for limb in ("pair", "triple"):
        with config.override(limb=limb):
            rows = {"limbs": limb}
            for name, n_gather, do_evict in (("behind_the_gather", 1, False), ("second_call_same_operand", 0, False), ("behind_a_1GiB_fill", 0, True)):
                ts = []
                for it in range(25):
                    amax = torch.empty(V * L, device=dev) if limb == "pair" else None
                    if n_gather or it == 0:
                        agg = ops._seg_reduce_raw(_lib.AGG_SUM, H, g.rowptr_t, 1, g.src_t, w, V * L, rowmax=amax).view(V, L * D)
                        keep_amax = amax
                    if do_evict:
                        evict.fill_(1.0)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    DN.grouped_nn_gemm(agg, kernels, relu=True, xmax=keep_amax, xgroups=L)
                    b.record()
                    torch.cuda.synchronize()
                    if it >= 5:
                        ts.append(a.elapsed_time(b) * 1e3)
                rows[name + "_us"] = round(float(np.median(ts)), 1)
            print(json.dumps(rows), flush=True)


if __name__ == "__main__":
    main()
