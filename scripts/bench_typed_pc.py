#!/usr/bin/env python
"""relgnn_limb_gemm_sel_pc_xf32 (wave roles) against the panel kernels (relgnn_limb_gemm_sel_xf32) on the typed products of C5:
737 k table rows, 23 edge types, 512-row tiles, D = 128.  One JSON line per shape: time (HIP events), bit identity, status word."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import config, dense as DN, ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, reps=7, inner=5):
    for _ in range(6):
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
    return round(sorted(ts)[len(ts) // 2], 1)


L, tiles, V = 23, 1440, 100000
P = tiles * 512
tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
node = torch.randint(0, V, (P,), generator=g).to(torch.int32).to(dev)
H = (torch.rand((V, 128), generator=g) * 2 - 1).to(dev)
for Dout in (128, 256):
    Ws = [((torch.rand((128, Dout), generator=g) * 2 - 1) * 0.1).to(dev) for _ in range(L)]
    gY = (torch.rand((P, Dout), generator=g) * 2 - 1).to(dev)
    for what, layout, a, args in (("forward gather(H) @ W_type [128, %d]" % Dout, DN.GEMM_NN, H, dict(a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)),
                                  ("input gradient [P, %d] @ W_type^T" % Dout, DN.GEMM_NT, gY, dict(b_select=tile_type, rows_per_select=512))):
        row, res = {"what": what, "P": P}, {}
        for pc in ("0", "1"):
            with config.override(typed_pc=pc):
                im = DN.sel_image(Ws, layout)
                res[pc] = DN.limb_dense_sel(layout, a, Ws, image=im, **args)
                row["panel_us" if pc == "0" else "roles_us"] = timed(lambda: DN.limb_dense_sel(layout, a, Ws, image=im, **args))
        row["bit_identical"] = bool(torch.equal(res["0"], res["1"]))
        row["status_word"] = ops.handover_status()
        n = Dout if layout == DN.GEMM_NN else 128
        k = 128 if layout == DN.GEMM_NN else Dout
        row["bytes"] = P * 4 * (k + n)
        row["roles_TBps"] = round(row["bytes"] / row["roles_us"] / 1e6, 2)
        print(json.dumps(row), flush=True)
    del gY
