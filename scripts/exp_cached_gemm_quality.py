"""GPU time of the cached-solution library GEMM (relgnn_blaslt_gemm_f32: solution picked for the FIRST node count of a
4096-row bucket) vs torch.mm (heuristic for the exact shape), at the C2 shapes, for several V of one bucket."""
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
for V in (32800, 34567, 36411, 36800):
    print("V=%d" % V)
    for name, layout, a, b, ref in (
            ("fwd  [V,768]@[768,256]", D.GEMM_NN, r(V, 768), r(768, 256), lambda a, b: a @ b),
            ("dH   [V,768]@[768,256] (stacked W^T)", D.GEMM_NN, r(V, 768), r(768, 256), lambda a, b: a @ b),
            ("fwd  [V,256]@[256,256]", D.GEMM_NN, r(V, 256), r(256, 256), lambda a, b: a @ b),
            ("fwd  [V,50]@[50,256]", D.GEMM_NN, r(V, 50), r(50, 256), lambda a, b: a @ b),
            ("out  [V,256]@[256,121]", D.GEMM_NN, r(V, 256), r(256, 121), lambda a, b: a @ b),
            ("dX   [V,256]@[256,256]^T", D.GEMM_NT, r(V, 256), r(256, 256), lambda a, b: a @ b.t()),
            ("dX   [V,121]@[256,121]^T", D.GEMM_NT, r(V, 121), r(256, 121), lambda a, b: a @ b.t())):
        print("  %-40s cached %7.1f us | torch %7.1f us" % (name, t(lambda: D.lib_gemm(layout, a, b)), t(lambda: ref(a, b))))
    for M, N in ((768, 256), (256, 256), (256, 121), (50, 256)):
        a, b = r(V, M), r(V, N)
        D._CACHED_LIB_GEMM = True; c = t(lambda: D.matmul_tn_splitk(a, b))
        D._CACHED_LIB_GEMM = False; o = t(lambda: D.matmul_tn_splitk(a, b))
        D._CACHED_LIB_GEMM = True
        print("  dW   [V,%d]^T@[V,%d] split-K %-14s cached %7.1f us | torch %7.1f us" % (M, N, "", c, o))
