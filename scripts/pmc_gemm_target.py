"""PMC target: the two forward GEMM shapes of a C2 step, hand-written MFMA kernel and library, 10 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense as D
dev = torch.device("cuda:0")
V = 36411
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.rand(s, device=dev, generator=g) * 2 - 1
for a, b in ((r(V, 256), r(256, 768)), (r(V, 768), r(768, 256))):
    for _ in range(10):
        D.own_gemm(D.GEMM_NN, a, b)
    for _ in range(10):
        a @ b
torch.cuda.synchronize()
