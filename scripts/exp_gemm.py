"""Experiment: the K=V (tall-skinny reduction) weight-gradient GEMMs  dW = A^T @ B, A [V, M], B [V, N]."""
import torch, time
dev = torch.device("cuda")
V = 32203

def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def splitk(A, B, S):
    V, M = A.shape
    N = B.shape[1]
    c = V // S
    main = torch.bmm(A[:c * S].view(S, c, M).transpose(1, 2), B[:c * S].view(S, c, N)).sum(0)
    if c * S < V:
        main = main + A[c * S:].t() @ B[c * S:]
    return main

for (M, N) in [(256, 768), (256, 256), (50, 256), (256, 121)]:
    A = torch.randn(V, M, device=dev); B = torch.randn(V, N, device=dev)
    ref = A.t() @ B
    gf = 2 * V * M * N / 1e9
    t = bench(lambda: A.t() @ B)
    print("M=%d N=%d  A.t()@B: %.1f us (%.1f TF/s)" % (M, N, t, gf / t * 1e3 / 1e3))
    t = bench(lambda: (B.t() @ A).t())
    print("           (B.t()@A).t(): %.1f us" % t)
    At = A.t().contiguous()
    t = bench(lambda: At @ B)
    print("           At_contig@B: %.1f us (+transpose copy %.1f us)" % (t, bench(lambda: A.t().contiguous())))
    for S in (8, 16, 32, 64):
        t = bench(lambda: splitk(A, B, S))
        err = (splitk(A, B, S) - ref).abs().max().item()
        print("           splitK bmm S=%d: %.1f us (%.1f TF/s) err %.2e" % (S, t, gf / t * 1e3 / 1e3, err))
# forward-type GEMM for reference
H = torch.randn(V, 256, device=dev); W = torch.randn(256, 768, device=dev)
t = bench(lambda: H @ W); print("H@Wcat: %.1f us (%.1f TF/s)" % (t, 2 * V * 256 * 768 / 1e9 / t * 1e3 / 1e3))
dT = torch.randn(V, 768, device=dev)
t = bench(lambda: dT @ W.t()); print("dT@Wcat^T: %.1f us" % t)
# bias-gradient reductions
g = torch.randn(V, 121, device=dev)
print("g.sum(0): %.1f us;  ones@g: %.1f us" % (bench(lambda: g.sum(0)), bench(lambda: torch.ones(1, V, device=dev) @ g)))
ones = torch.ones(1, V, device=dev)
print("ones(pre)@g: %.1f us" % bench(lambda: ones @ g))
