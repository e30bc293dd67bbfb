"""Experiment: does one gather/reduce pass PER EDGE TYPE (working set = one type's slab of the graph, which fits the
4 MiB per-XCD L2) beat the single pass over all types (75 % L2 hit rate)?  C2 batch, D = 256."""
import os, sys
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
X = torch.rand((V * L, D), device=dev) * 2 - 1
plan = g.plan_transformed(w)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
single = lambda: ops._seg_reduce_raw(_lib.AGG_SUM, X, plan.rowptr, plan.stride, plan.col, plan.w, plan.num_out, _lib.ACT_LINEAR)
# per-type sub-plans
rp = g.rowptr_t.long()
counts = (rp[1:] - rp[:-1]).view(V, L)
pos = torch.arange(g.M, device=dev)
bucket_of_pos = torch.repeat_interleave(torch.arange(V * L, device=dev), counts.view(-1))
type_of_pos = bucket_of_pos % L
subs = []
for l in range(L):
    sel = type_of_pos == l
    rowptr_l = torch.zeros(V + 1, dtype=torch.int32, device=dev)
    rowptr_l[1:] = torch.cumsum(counts[:, l], 0).to(torch.int32)
    subs.append((rowptr_l, g.col_t[sel].contiguous(), w[sel].contiguous()))
def per_type():
    outs = [ops._seg_reduce_raw(_lib.AGG_SUM, X, r, 1, c, ww, V, _lib.ACT_LINEAR) for r, c, ww in subs]
    return outs
a = single(); b = per_type()
print("max diff", float((a - (b[0] + b[1] + b[2])).abs().max()))
print("single pass %.1f us" % timeit(single))
print("3 per-type passes (no accumulate) %.1f us" % timeit(per_type))
for l, (r, c, ww) in enumerate(subs):
    print("  type %d (%d msgs): %.1f us" % (l, c.numel(), timeit(lambda: ops._seg_reduce_raw(_lib.AGG_SUM, X, r, 1, c, ww, V, _lib.ACT_LINEAR))))
