#!/usr/bin/env python
"""relgnn_limb_dense_sel_f32 against relgnn_panel_gemm_f32 (exact fp32) on the typed transforms of C5 (737 k gathered rows, 23 edge
types, 512-row tiles, D = 128) and against the library on the plain D = 128 products of C3."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, reps=7, inner=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]


L, tiles, V = 23, 1440, 120000
P = tiles * 512
tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
node = torch.randint(0, V, (P,), generator=g).to(torch.int32).to(dev)
H = (torch.rand((V, 128), generator=g) * 2 - 1).to(dev)
for Dout in (128, 256):
    W = ((torch.rand((L, 128, Dout), generator=g) * 2 - 1) * 0.1).to(dev)
    t_l = timed(lambda: DN.limb_dense_sel(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512))
    t_p = timed(lambda: DN.panel_gemm(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512))
    print(json.dumps({"what": "typed forward [%d gathered rows, 128] @ W_type[128, %d]" % (P, Dout), "limb_us": round(t_l, 1), "panel_f32_us": round(t_p, 1)}), flush=True)
    gY = (torch.rand((P, Dout), generator=g) * 2 - 1).to(dev)
    t_l = timed(lambda: DN.limb_dense_sel(DN.GEMM_NT, gY, W, b_select=tile_type, rows_per_select=512))
    t_p = timed(lambda: DN.panel_gemm(DN.GEMM_NT, gY, W, b_select=tile_type, rows_per_select=512, dims=(P, 128, Dout)))
    print(json.dumps({"what": "typed input gradient [%d, %d] @ W_type^T -> [.., 128]" % (P, Dout), "limb_us": round(t_l, 1), "panel_f32_us": round(t_p, 1)}), flush=True)
    del gY
for (M, N, K) in [(50000, 640, 128), (50000, 384, 128), (50000, 128, 128), (50000, 128, 640), (100000, 128, 128)]:
    a = (torch.rand((M, K), generator=g) * 2 - 1).to(dev)
    W = ((torch.rand((K, N), generator=g) * 2 - 1) * 0.1).to(dev)
    t_l = timed(lambda: DN.limb_dense_sel(DN.GEMM_NN, a, W))
    t_t = timed(lambda: torch.mm(a, W))
    print(json.dumps({"what": "plain [%d, %d] @ [%d, %d]" % (M, K, K, N), "limb_us": round(t_l, 1), "library_f32_us": round(t_t, 1)}), flush=True)
