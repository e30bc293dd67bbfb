#!/usr/bin/env python
"""C3's message-transform weight gradient [V, 128]^T @ [V, 640] (5 edge types x 128): library split-K (what runs today, 95 us in the
step's trace) against the streaming MFMA kernel on the whole product and on 256-column pieces."""
import json, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tf_gnn_samples_amd import config, dense as DN
dev = torch.device("cuda:0")
V = 49986
a = torch.rand((V, 128), device=dev) * 2 - 1
g = (torch.rand((V, 640), device=dev) * 2 - 1) * 0.05
truth = a.double().t() @ g.double()

def timed(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def pieces():
    out = torch.empty((128, 640), device=dev)
    for c0 in (0, 256, 512):
        out[:, c0:c0 + 256] = DN.tn_stream_gemm(a, g[:, c0:min(c0 + 256, 640)])
    return out

rows = {}
for name, fn in (("library_split_k (current)", lambda: DN.matmul_tn_splitk(a, g)), ("stream_whole", lambda: DN.tn_stream_gemm(a, g)), ("stream_256_column_pieces", pieces)):
    try:
        out = fn()
        rows[name] = {"us": round(timed(fn), 1), "err_vs_f64": float((out.double() - truth).abs().max())}
    except Exception as e:
        rows[name] = {"error": repr(e)}
print(json.dumps(rows))
