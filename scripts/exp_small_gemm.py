"""hipBLASLt efficiency on the tall-skinny node-side GEMMs of D=128 models (V=100k)."""
import torch, time
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for V in (100108, 32203, 49986):
    for (K, N) in ((128, 128), (128, 256), (128, 384), (256, 256), (256, 121), (50, 256)):
        x = torch.randn(V, K, device=dev); w = torch.randn(K, N, device=dev); g = torch.randn(V, N, device=dev)
        b = torch.randn(N, device=dev)
        fl = 2 * V * K * N
        a = t(lambda: x @ w); c = t(lambda: torch.addmm(b, x, w)); d = t(lambda: g @ w.t()); e = t(lambda: x.t() @ g)
        print("V=%6d K=%3d N=%3d  NN %.0f us (%.0f TF)  addmm %.0f us  NT(dX) %.0f us (%.0f TF)  TN(dW) %.0f us (%.0f TF)" % (
            V, K, N, a, fl / a / 1e6, c, d, fl / d / 1e6, e, fl / e / 1e6))
