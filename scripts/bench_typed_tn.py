#!/usr/bin/env python
"""Typed weight-gradient partials of a C5-sized pair table, isolated (HIP events, no co-running kernels): the gathered three-limb TN
kernel (relgnn_limb_gemm_tn_tiles_f32) against the exact-fp32 panel kernel it replaces in ops.typed_linear's backward.
    python scripts/bench_typed_tn.py >> gpurun_out/typed_tn.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tf_gnn_samples_amd import dense as DN  # noqa: E402


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, chunk = 96000, 512
    for tiles in (1400, 600):
        P = tiles * chunk
        rows = torch.randint(0, N, (P,), device=dev, dtype=torch.int32)
        rows = torch.sort(rows.view(tiles, chunk), dim=1).values.contiguous().view(-1)      # ascending node ids inside a tile
        rows[torch.rand(P, device=dev) < 0.05] = -1
        for J, C in ((128, 128), (128, 256)):
            A = torch.randn(N, J, device=dev)
            G = torch.randn(P, C, device=dev)
            t_limb = timed(lambda: DN.limb_gemm_tn_tiles(A, G, rows, chunk))
            t_panel = timed(lambda: DN.panel_gemm(DN.GEMM_TN, A, G, a_rows=rows, batch=tiles, strides=(0, chunk * C, J * C),
                                                  dims=(J, C, chunk)))
            flops = 2.0 * P * J * C
            print(json.dumps({"tiles": tiles, "P": P, "J": J, "C": C, "limb_tn_tiles_us": round(t_limb, 1), "panel_tn_us": round(t_panel, 1),
                              "limb_TFLOPs_fp32_equivalent": round(flops / t_limb * 1e-6, 1),
                              "panel_TFLOPs": round(flops / t_panel * 1e-6, 1)}), flush=True)


if __name__ == "__main__":
    main()
