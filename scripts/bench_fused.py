"""Micro-benchmark: fused aggregate->MFMA kernel (csrc/agg_transform.hip) vs (library GEMM + seg_reduce) on the C2 batch,
one RGCN layer: forward only, and forward + backward (dH, dW)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_gnn_samples_amd import _lib, ops
from tf_gnn_samples_amd.dense import dense
from tf_gnn_samples_amd.graph import RelGraph
dev = torch.device("cuda:0")
task, mb, batch, gen, local = bench.build_local_batch(0, 1, dev)
g = RelGraph(batch.adjacency_lists, mb.num_nodes)
w = g.degree_scale(batch.type_to_num_incoming_edges)
V, L, D = g.V, g.L, 256
gen_ = torch.Generator(device=dev).manual_seed(0)
H = (torch.rand((V, D), device=dev, generator=gen_) * 2 - 1).requires_grad_(True)
W = ((torch.rand((L, D, D), device=dev, generator=gen_) * 2 - 1) * 0.1).requires_grad_(True)
GO = torch.rand((V, D), device=dev, generator=gen_) - 0.5
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
plan = g.plan_transformed(w)
def unfused():
    wcat = W.permute(1, 0, 2).reshape(D, L * D)
    T = dense(H, wcat).view(V * L, D)
    return ops.seg_gather_reduce(T, plan, "sum", "relu")
def fused():
    return ops.fused_aggregate_transform(H, W, g, w, "sum", "relu")
def train(fn):
    def run():
        H.grad = None; W.grad = None
        fn().backward(GO)
    return run
a, b = unfused(), fused()
print("max abs diff fused vs unfused:", float((a - b).abs().max()), "scale", float(a.abs().max()))
train(unfused)(); ga, gw = H.grad.clone(), W.grad.clone()
train(fused)()
print("grad diff dH %.3e (scale %.3e)  dW %.3e (scale %.3e)" % (float((H.grad - ga).abs().max()), float(ga.abs().max()),
                                                                float((W.grad - gw).abs().max()), float(gw.abs().max())))
with torch.no_grad():
    print("forward only  : unfused %.1f us   fused %.1f us" % (timeit(unfused), timeit(fused)))
print("forward+backward: unfused %.1f us   fused %.1f us" % (timeit(train(unfused)), timeit(train(fused))))
def agg_first():
    return ops.aggregate_then_transform(H, W, g, w, "sum", "relu")
with torch.no_grad():
    print("aggregate-then-transform (two kernels): forward %.1f us" % timeit(agg_first))
print("aggregate-then-transform (two kernels): forward+backward %.1f us" % timeit(train(agg_first)))
ops.check_agg_transform_errors()
print("variant:", os.environ.get("RELGNN_AGG_VARIANT", "ring"))
