#!/usr/bin/env python
"""relgnn_limb_gemm_xf32_pc (producer / matrix wave roles) against relgnn_limb_gemm_xf32 on the C2 layer's product shapes: bit
identity and time (HIP events on the launch stream, median of 7 x 10 launches).  One JSON line per shape."""
import ctypes, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import _lib, config, dense as DN
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=7, inner=10):
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
    ts.sort()
    return ts[len(ts) // 2]


for name, M, K, N in (("forward  [V, 768] @ [768, 256]", 36096, 768, 256), ("input gradient  [V, 256] @ [256, 768]", 36096, 256, 768),
                      ("Dense  [V, 256] @ [256, 256]", 36096, 256, 256), ("forward, V = 32203", 32203, 768, 256),
                      ("input gradient, V = 32203", 32203, 256, 768)):
    a = torch.rand((M, K), device=dev, generator=gen) * 2 - 1
    w = [(torch.rand((N, 256), device=dev, generator=gen) * 2 - 1) * 0.1 for _ in range(K // 256)]
    y = (torch.rand((M, N), device=dev, generator=gen) * 2 - 1).relu_()
    out = torch.empty((M, N), device=dev)
    row = {"shape": name, "M": M, "K": K, "N": N}
    res = {}
    for pc in ("0", "1"):
        with config.override(limb_pc=pc):
            res[pc] = DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_RELU).clone()
            row["plain_us" if pc == "0" else "roles_us"] = timed(lambda: DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_RELU, out=out))
            row["plain_dact_us" if pc == "0" else "roles_dact_us"] = timed(
                lambda: DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_LINEAR, out=out, dact=_lib.ACT_RELU, dy=y))
    row["bit_identical"] = bool(torch.equal(res["0"], res["1"]))
    from tf_gnn_samples_amd import ops
    row["status_word"] = ops.handover_status()
    print(json.dumps(row), flush=True)
