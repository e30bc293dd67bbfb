#!/usr/bin/env python
"""Where the k-loop time of relgnn_limb_gemm_f32 goes: the library built with -DRELGNN_LIMB_ABLATE, RELGNN_LIMB_ABLATE=bits in
the environment (bit 0 no DMA after the prologue, bit 1 no fragment reads in the loop, bit 2 no waits / barrier)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
M, N, K = 256 * 160, 256, 768
a = DN.limb_split(torch.rand((M, K), device=dev) * 2 - 1)
w = DN.limb_split((torch.rand((N, K), device=dev) * 2 - 1) * 0.1)
out = torch.empty((M, N), device=dev)
for _ in range(1500):            # ~150 ms: the clock has to come up before anything is timed
    DN.limb_gemm(a, w, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        DN.limb_gemm(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
ts.sort()
print("ablate=%s  [%d,%d]x[%d,%d]^T  %.1f us" % (os.environ.get("RELGNN_LIMB_ABLATE", "0"), M, K, N, K, ts[4]), flush=True)
