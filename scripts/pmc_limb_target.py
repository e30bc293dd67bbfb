#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes for the limb GEMM kernels: a few launches of the C2 layer products on the limb route
(forward [V, 768] x [768, 256], input gradient [V, 256] x [768, 256]^T, weight gradient [V, 768]^T x [V, 256]) and of the exact-fp32
library GEMM on the forward shape, V = 36 096."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
V = 36096
a = torch.rand((V, 768), device=dev) * 2 - 1
g = (torch.rand((V, 256), device=dev) * 2 - 1) * 0.05
W = (torch.rand((768, 256), device=dev) * 2 - 1) * 0.1
for _ in range(6):
    DN.limb_dense(DN.GEMM_NN, a, W)
    DN.limb_dense(DN.GEMM_NT, g, W)
    DN.limb_gemm_tn(a, g)
torch.cuda.synchronize()
for _ in range(6):
    torch.mm(a, W)
torch.cuda.synchronize()
