"""Weight-gradient GEMM dW = X^T @ G at the shapes of a C2 step: streaming kernel (csrc/gemm_tn_stream.hip) vs the library
split-K path; checks the result against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense as D
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.rand(s, device=dev, generator=g) * 2 - 1
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for V in (36411, 30011, 1000003, 777):
    print("V=%d" % V)
    for M, N in ((256, 256), (256, 121), (50, 256), (768, 256), (256, 768), (121, 50), (128, 128)):
        if V > 500000 and M * N > 256 * 256: continue
        a, b = r(V, M), r(V, N)
        want = a.double().t() @ b.double()
        got = D.tn_stream_gemm(a, b)
        err = (got.double() - want).abs().max().item() / max(1.0, want.abs().max().item())
        ts = t(lambda: D.tn_stream_gemm(a, b))
        tl = t(lambda: D.matmul_tn_splitk(a, b))
        own = ""
        D._OWN_GEMM = True
        D._OWN_GEMM = False
        fl = 2.0 * V * M * N
        print("  [V,%3d]^T@[V,%3d]  stream %7.1f us %6.1f TF (rel err %.1e) | library split-K %7.1f us %6.1f TF %s"
              % (M, N, ts, fl / ts / 1e6, err, tl, fl / tl / 1e6, own))
