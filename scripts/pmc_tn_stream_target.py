"""PMC target: the streaming weight-gradient kernel at the three small-Dense shapes of a C2 step, 10 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense as D
dev = torch.device("cuda:0")
V = 36411
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.rand(s, device=dev, generator=g) * 2 - 1
for M, N in ((256, 256), (256, 121), (50, 256)):
    a, b = r(V, M), r(V, N)
    for _ in range(10):
        D.tn_stream_gemm(a, b)
torch.cuda.synchronize()
