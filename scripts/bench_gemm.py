"""The node-side GEMM shapes of a C2 step: hand-written exact-fp32 MFMA kernel (csrc/gemm_f32.hip) vs the library
(hipBLASLt through torch.mm, default solution = what a distinct-batch epoch gets), TFLOP/s each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense as D

dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 36411


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.rand(s, device=dev, generator=g) * 2 - 1
for name, layout, a, b, lib in (
        ("fwd  [V,256]@[256,768]", D.GEMM_NN, r(V, 256), r(256, 768), lambda a, b: a @ b),
        ("fwd  [V,256]@[256,256]", D.GEMM_NN, r(V, 256), r(256, 256), lambda a, b: a @ b),
        ("dX   [V,768]@[256,768]^T", D.GEMM_NT, r(V, 768), r(256, 768), lambda a, b: a @ b.t()),
        ("dX   [V,256]@[256,256]^T", D.GEMM_NT, r(V, 256), r(256, 256), lambda a, b: a @ b.t()),
        ("fwd  [V,768]@[768,256]  (aggregate-first)", D.GEMM_NN, r(V, 768), r(768, 256), lambda a, b: a @ b),
        ("dX   [V,256]@[768,256]^T (aggregate-first dA)", D.GEMM_NT, r(V, 256), r(768, 256), lambda a, b: a @ b.t()),
        ("dW   [V,768]^T@[V,256]  (aggregate-first)", D.GEMM_TN, r(V, 768), r(V, 256), None),
        ("dW   [V,256]^T@[V,768]", D.GEMM_TN, r(V, 256), r(V, 768), None),
        ("dW   [V,256]^T@[V,256]", D.GEMM_TN, r(V, 256), r(V, 256), None)):
    M, N, K = (a.shape[0], b.shape[1], a.shape[1]) if layout == D.GEMM_NN else \
              (a.shape[0], b.shape[0], a.shape[1]) if layout == D.GEMM_NT else (a.shape[1], b.shape[1], a.shape[0])
    flop = 2.0 * M * N * K
    own = t(lambda: D.own_gemm(layout, a, b))
    if lib is None:
        os.environ  # library path of the weight gradient = the split-K bmm + sum of dense.matmul_tn_splitk
        D._OWN_GEMM = False
        libt = t(lambda: D.matmul_tn_splitk(a, b))
        single = t(lambda: a.t() @ b)
        D._OWN_GEMM = True
        print("%-48s own %7.1f us %6.1f TF | lib split-K %7.1f us %6.1f TF | lib single %7.1f us" % (name, own, flop / own / 1e6, libt, flop / libt / 1e6, single))
    else:
        libt = t(lambda: lib(a, b))
        print("%-48s own %7.1f us %6.1f TF | lib %7.1f us %6.1f TF" % (name, own, flop / own / 1e6, libt, flop / libt / 1e6))
