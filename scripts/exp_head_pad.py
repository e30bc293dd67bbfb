"""PPI head GEMMs with 121 label columns vs padded to 128 (library first-guess solutions): forward [V,256]@[256,N],
input gradient [V,N]@[256,N]^T, weight gradient [V,256]^T@[V,N] (streaming kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense

dev = torch.device("cuda:0")
V = 36000
h = torch.randn(V, 256, device=dev)


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N in (121, 128):
    w = torch.randn(256, N, device=dev) * 0.1
    b = torch.zeros(N, device=dev)
    g = torch.randn(V, N, device=dev)
    print("N=%d  fwd %.1f us   dX %.1f us   dW %.1f us" % (
        N, t(lambda: dense.lib_gemm(dense.GEMM_NN, h, w, b)), t(lambda: dense.lib_gemm(dense.GEMM_NT, g, w)),
        t(lambda: dense.matmul_tn_splitk(h, g))))
# a [V, 128]-strided view of 121 columns: what the forward would hand to the loss if only the weight were padded
w = torch.randn(256, 128, device=dev) * 0.1
out = torch.empty(V, 128, device=dev)
print("N=128 forward into a padded buffer: %.1f us" % t(lambda: dense.lib_gemm(dense.GEMM_NN, h, w, None, out=out)))
