"""Host-side cost of a library GEMM call when the node dimension changes from call to call (every batch of a shuffled
epoch has its own V) vs when it repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_gnn_samples_amd import dense as D
dev = torch.device("cuda:0")
w = torch.rand(768, 256, device=dev)
xs = [torch.rand(30000 + 37 * i, 768, device=dev) for i in range(60)]
def host_us(fn, args_list):
    torch.cuda.synchronize()
    t = []
    for a in args_list:
        t0 = time.perf_counter(); fn(a); t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    t.sort()
    return 1e6 * t[len(t) // 2], 1e6 * t[-1]
for _ in range(3): xs[0] @ w
print("library, NEW shape each call      median %.0f us  max %.0f us" % host_us(lambda a: a @ w, xs[1:]))
print("library, same shapes again         median %.0f us  max %.0f us" % host_us(lambda a: a @ w, xs[1:]))
print("library, ONE shape repeated        median %.0f us  max %.0f us" % host_us(lambda a: a @ w, [xs[0]] * 50))
print("relgnn_blaslt_gemm_f32, new shapes median %.0f us  max %.0f us" % host_us(lambda a: D.lib_gemm(D.GEMM_NN, a, w), [torch.rand(32000 + 43 * i, 768, device=dev) for i in range(50)]))
g = torch.rand(30000, 256, device=dev)
print("relu_ (elementwise)                median %.0f us  max %.0f us" % host_us(lambda a: a.relu_(), [g] * 50))

for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    ys = [torch.rand(33000 + 29 * i + (7 if lib == "cublas" else 0), 768, device=dev) for i in range(50)]
    print("preferred_blas_library=%s: NEW shapes median %.0f us max %.0f us" % ((lib,) + host_us(lambda a: a @ w, ys)), end="")
    a = ys[0]
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): a @ w
    e1.record(); torch.cuda.synchronize()
    print("   GPU %.1f us per [33k,768]@[768,256]" % (e0.elapsed_time(e1) / 20 * 1e3))
