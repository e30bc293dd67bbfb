"""Feasibility data for a split-bf16 (bf16x3 / bf16x6) replacement of the fp32 node-side GEMMs:
library bf16 GEMM rates on the K-concatenated shapes, the cost of splitting, and the accuracy of the emulation."""
import torch
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
V, K, N = 32203, 256, 768
x = torch.randn(V, K, device=dev); w = torch.randn(K, N, device=dev) * 0.06
ref = (x.double() @ w.double())
f32 = t(lambda: x @ w)
print("fp32 GEMM [%d,%d]@[%d,%d]: %.0f us (%.0f TF/s), max err vs fp64 %.2e" % (V, K, K, N, f32, 2 * V * K * N / f32 / 1e6, float((x @ w - ref).abs().max())))
def split(a, terms):
    out, r = [], a
    for _ in range(terms):
        h = r.to(torch.bfloat16); out.append(h); r = r - h.float()
    return out
for terms, pairs in ((2, [(0, 0), (0, 1), (1, 0)]), (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)])):
    xs, ws = split(x, terms), split(w, terms)
    A = torch.cat([xs[i] for i, j in pairs], dim=1).contiguous()          # [V, len(pairs)*K]
    B = torch.cat([ws[j] for i, j in pairs], dim=0).contiguous()          # [len(pairs)*K, N]
    try:
        out = torch.mm(A, B, out_dtype=torch.float32)
        g = t(lambda: torch.mm(A, B, out_dtype=torch.float32))
        how = "bf16 in, fp32 out"
    except Exception as e:
        out = (A @ B).float(); g = t(lambda: A @ B); how = "bf16 out (no out_dtype: %s)" % type(e).__name__
    sp = t(lambda: torch.cat([h for h in split(x, terms)], dim=1))
    print("bf16x%d: GEMM [%d,%d]@[%d,%d] %.0f us (%.0f TF/s raw, %s); splitting x with library ops %.0f us; max err vs fp64 %.2e"
          % (len(pairs), V, A.shape[1], A.shape[1], N, g, 2 * V * A.shape[1] * N / g / 1e6, how, sp, float((out.double() - ref).abs().max())))
