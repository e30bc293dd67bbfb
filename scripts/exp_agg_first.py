"""Experiment: order of the two halves of an RGCN layer on the C2 batch (forward, one layer, D = 256):
  (a) transform-then-aggregate  T = H @ [W_0|W_1|W_2] (library GEMM, writes [V, 768]), gather T rows + reduce -> [V, 256]
  (b) aggregate-then-transform  A[v, l] = sum_{p in (v,l)} w_p H[src_p] (gather from the 3x smaller [V, 256] table, writes
      [V, 768]), out = relu(A @ [W_0;W_1;W_2]) (library GEMM, K = 768)
  (c) the fused kernel, with the MFMA / gather phases ablated (RELGNN_AGG_ABLATE) to see which side binds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_gnn_samples_amd import _lib, ops
from tf_gnn_samples_amd.graph import GatherReducePlan, RelGraph
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
plan_t = g.plan_transformed(w)
wcat = W.permute(1, 0, 2).reshape(D, L * D).contiguous()
wstack = W.reshape(L * D, D).contiguous()
bucket = GatherReducePlan(rowptr=g.rowptr_t, stride=1, col=g.src_t, w=w, num_out=V * L, num_rows_x=V, rowptr_b=g.rowptr_s,
                          stride_b=L, col_b=g.frow_s, pos_b=g.pos_t_of_s, num_messages=g.M)
gemm_a = lambda: H @ wcat
red_a = lambda T: ops._seg_reduce_raw(_lib.AGG_SUM, T.view(V * L, D), plan_t.rowptr, plan_t.stride, plan_t.col, plan_t.w, V, _lib.ACT_RELU)
red_b = lambda: ops._seg_reduce_raw(_lib.AGG_SUM, H, bucket.rowptr, 1, bucket.col, bucket.w, V * L)
gemm_b = lambda A: torch.relu(A.view(V, L * D) @ wstack)
T = gemm_a(); A = red_b()
oa, ob = red_a(T), gemm_b(A)
print("max diff (a) vs (b): %.3e" % float((oa - ob).abs().max()))
ta, tb = timeit(gemm_a), timeit(lambda: red_a(T))
tc, td = timeit(red_b), timeit(lambda: gemm_b(A))
print("(a) GEMM [V,256]@[256,768] %.1f us + gather/reduce from T %.1f us = %.1f us" % (ta, tb, ta + tb))
print("(b) gather/reduce from H into [V*L,256] %.1f us + GEMM [V,768]@[768,256] + relu %.1f us = %.1f us" % (tc, td, tc + td))
from tf_gnn_samples_amd.ops import _agg_transform, _pack_agg_weights
packed = _pack_agg_weights(W, False)
fused = lambda: _agg_transform(H, g.rowptr_t, V, L, g.src_t, w, packed, D, D, _lib.AGG_SUM, _lib.ACT_RELU, False)[0]
print("(c) fused kernel %.1f us (RELGNN_AGG_ABLATE=%s)" % (timeit(fused), os.environ.get("RELGNN_AGG_ABLATE", "0")))
fused_agg = lambda: _agg_transform(H, g.rowptr_t, V, L, g.src_t, w, packed, D, D, _lib.AGG_SUM, _lib.ACT_RELU, True)[0]
print("(c') fused kernel + aggregated-row output %.1f us" % timeit(fused_agg))
ops.check_agg_transform_errors()
print("max diff fused vs (a): %.3e   variant %s" % (float((fused() - oa).abs().max()), os.environ.get("RELGNN_AGG_VARIANT", "ring")))
