#!/usr/bin/env python
"""relgnn_panel_gemm_f32 NN [256*16*u, K] x [K, 256] for u = 2..10 units per panel (every CU one panel of exactly u 16-row units):
how the kernel time splits into a part per k-tile that does not depend on the panel height and a part per unit."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
for K in (768, 256):
    b = (torch.rand((K, 256), device=dev, generator=gen) * 2 - 1) * 0.1
    rows = {}
    for u in range(2, 11):
        M = 256 * 16 * u
        a = torch.rand((M, K), device=dev, generator=gen) * 2 - 1
        for _ in range(3):
            DN.panel_gemm(0, a, b)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                DN.panel_gemm(0, a, b)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        ts.sort()
        lib = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                DN.lib_gemm(0, a, b)
            e1.record(); torch.cuda.synchronize()
            lib.append(e0.elapsed_time(e1) / 5 * 1e3)
        lib.sort()
        print(json.dumps({"K": K, "units_per_panel": u, "M": M, "panel_us": round(ts[3], 1), "lib_us": round(lib[2], 1),
                          "panel_TFLOPs": round(2.0 * M * K * 256 / ts[3] / 1e6, 1), "lib_TFLOPs": round(2.0 * M * K * 256 / lib[2] / 1e6, 1)}), flush=True)
