#!/usr/bin/env python
"""One shape of relgnn_panel_gemm_f32 under RELGNN_PANEL_ABLATE (set by the caller, read once per process): where the
k-loop's time goes.  bit 0: no DMA after the prologue, bit 1: no fragment reads in the loop, bit 2: no waits / barrier."""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
shapes = [(36096, 768, 256), (36096, 256, 256), (36864, 768, 256)]
for (V, K, N) in shapes:
    a = torch.rand((V, K), device=dev, generator=gen) * 2 - 1
    b = (torch.rand((K, N), device=dev, generator=gen) * 2 - 1) * 0.1
    for _ in range(3):
        DN.panel_gemm(0, a, b)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            DN.panel_gemm(0, a, b)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    ts.sort()
    print(json.dumps({"ablate": int(os.environ.get("RELGNN_PANEL_ABLATE", "0")), "shape": [V, K, N], "median_us": round(ts[4], 1),
                      "min_us": round(ts[0], 1), "TFLOPs": round(2.0 * V * K * N / ts[4] / 1e6, 1)}), flush=True)
