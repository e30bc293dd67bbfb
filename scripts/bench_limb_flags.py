#!/usr/bin/env python
"""A/B of relgnn_limb_gemm_tuning flags inside one process (interleaved rounds, same buffers): xf32 product time per flag value."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import _lib, dense as DN
dev = torch.device("cuda:0")
lib = _lib.load_library()
flags = [int(x) for x in (sys.argv[1:] or ["0", "1"])]
for (M, N, K) in [(40960, 256, 768), (36096, 256, 768), (36096, 768, 256), (32768, 256, 768)]:
    a = torch.rand((M, K), device=dev) * 2 - 1
    wl = DN.limb_split((torch.rand((N, K), device=dev) * 2 - 1) * 0.1)
    out = torch.empty((M, N), device=dev)
    for _ in range(800):
        DN.limb_gemm_xf32(a, wl, out=out)
    res = {f: [] for f in flags}
    for _ in range(9):
        for f in flags:
            lib.relgnn_limb_gemm_tuning(f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                DN.limb_gemm_xf32(a, wl, out=out)
            e1.record(); torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.relgnn_limb_gemm_tuning(0)
    print("[%d,%d]x[%d,%d]^T " % (M, K, N, K) + "  ".join("flags=%d: %.1f us" % (f, sorted(v)[4]) for f, v in res.items()), flush=True)
