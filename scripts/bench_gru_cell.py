#!/usr/bin/env python
"""The GRU cell of a GGNN layer at C3's size (49 986 nodes, 128 units): the one-kernel forward and backward (relgnn_gru_cell_fwd_xf32,
relgnn_gru_cell_bwd_xf32) against the composition they replace (three limb products + gru.hip's gate and output kernels, and the
same in reverse); forward in the training form (z, r, r * h and the candidate kept) and the inference form; the backward with its
weight gradients on the same stream.  HIP events, median of 7 x 20 launches."""
import json, statistics, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tf_gnn_samples_amd import config, ops, utils
dev = torch.device("cuda:0")
ops.handover_word(dev)
U = 128
g = torch.Generator(device="cpu").manual_seed(0)
K = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
R = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
b = torch.zeros(3 * U, device=dev).requires_grad_(True)


def timed(fn, reps=7, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    return round(statistics.median(out), 1)


for V in (49986, 16416, 200000):
    x = (torch.rand((V, U), generator=g) * 2 - 1).to(dev).requires_grad_(True)
    h = (torch.rand((V, U), generator=g) * 2 - 1).to(dev).requires_grad_(True)
    gout = torch.randn((V, U), generator=g).to(dev)
    row = {"nodes": V}
    for sw in ("0", "1"):
        with config.override(gru_cell=sw):
            fwd = timed(lambda: utils._GRUCellFn.apply(x, h, K, R, b, 1))
            row["train_form_us_gru_cell_" + sw] = fwd
            with torch.no_grad():
                row["inference_form_us_gru_cell_" + sw] = timed(lambda: utils._GRUCellFn.apply(x, h, K, R, b, 1))

            def both():
                for t in (x, h, K, R, b):
                    t.grad = None
                utils._GRUCellFn.apply(x, h, K, R, b, 1).backward(gout)
            row["forward_backward_us_gru_cell_" + sw] = timed(both)
    row["handover_status"] = ops.handover_status()
    print(json.dumps(row), flush=True)
