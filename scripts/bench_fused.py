"""Micro-benchmark: fused aggregate->MFMA kernel vs (GEMM + seg_reduce) on the C2 batch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_gnn_samples_amd import _lib, ops
from tf_gnn_samples_amd.graph import RelGraph
dev = torch.device("cuda:0")
task, mb, batch, gen, local = bench.build_local_batch(0, 1, dev)
g = RelGraph(batch.adjacency_lists, mb.num_nodes)
w = g.degree_scale(batch.type_to_num_incoming_edges)
V, L, D = g.V, g.L, 256
gen_ = torch.Generator(device=dev).manual_seed(0)
H = torch.rand((V, D), device=dev, generator=gen_) * 2 - 1
W = (torch.rand((L, D, D), device=dev, generator=gen_) * 2 - 1) * 0.1
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
plan = g.plan_transformed(w)
wcat = W.permute(1, 0, 2).reshape(D, L * D).contiguous()
def unfused():
    T = (H @ wcat).view(V * L, D)
    return ops._seg_reduce_raw(_lib.AGG_SUM, T, plan.rowptr, plan.stride, plan.col, plan.w, plan.num_out, _lib.ACT_RELU)
def fused():
    return ops.fused_aggregate_transform(H, W, g, w, "sum", "relu")
a, b = unfused(), fused()
print("max abs diff fused vs unfused:", float((a - b).abs().max()), "scale", float(a.abs().max()))
print("unfused us %.1f   fused us %.1f  (ablate=%s)" % (timeit(unfused), timeit(fused), os.environ.get("RELGNN_FUSED_ABLATE", "0")))
