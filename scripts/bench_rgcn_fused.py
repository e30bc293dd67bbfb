#!/usr/bin/env python
"""relgnn_rgcn_fused_fwd against the two launches it replaces (relgnn_seg_reduce_fwd + relgnn_limb_gemm_xf32) on the C2 batch:
bit identity of output and bucket sums, the hand-over status word, and the time of each form (HIP events on the launch stream,
median of 7 x 10 launches).  One JSON line per row into gpurun_out/rgcn_fused.jsonl."""
import ctypes, json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from tf_gnn_samples_amd import _lib, dense as DN, ops
from tf_gnn_samples_amd.graph import RelGraph

dev = torch.device("cuda:0")
D = 256


def timed(fn, reps=7, inner=10):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def status():
    from tf_gnn_samples_amd import ops
    return ops.handover_status()


def c2_batch():
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    fd = mb.feed_dict
    adj = [torch.as_tensor(a, device=dev) for a in fd["adjacency_lists"]]
    g = RelGraph(adj, mb.num_nodes)
    w = g.degree_scale(torch.as_tensor(fd["type_to_num_incoming_edges"].astype(np.float32), device=dev))
    return g, w


rows = []
out_path = Path(os.environ.get("GRAFT_REPO_ROOT", ".")) / "gpurun_out" / "rgcn_fused.jsonl"
out_path.parent.mkdir(exist_ok=True)
graph, w = c2_batch()
V, L = graph.V, graph.L
gen = torch.Generator(device="cpu").manual_seed(0)
H = torch.randn((V, D), generator=gen).to(dev)
kernels = [((torch.rand((D, D), generator=gen) * 2 - 1) * 0.08).to(dev) for _ in range(L)]
print("C2 batch: V = %d, L = %d, M = %d" % (V, L, graph.M), flush=True)


def two_kernels(relu=True):
    agg = ops._seg_reduce_raw(_lib.AGG_SUM, H, graph.rowptr_t, 1, graph.src_t, w, V * L).view(V, L * D)
    out = DN.limb_gemm_weight(agg, kernels, DN.WEIGHT_NN, None, _lib.ACT_RELU if relu else _lib.ACT_LINEAR)
    return agg, out


agg_ref, out_ref = two_kernels()
agg, out = ops._rgcn_fused(H, graph, w, kernels, True, True)
torch.cuda.synchronize()
st = status()
same_out, same_agg = bool(torch.equal(out, out_ref)), bool(torch.equal(agg, agg_ref))
diff = float((out - out_ref).abs().max()) if not same_out else 0.0
print("status %d, out identical %s (max diff %.3g), bucket sums identical %s" % (st, same_out, diff, same_agg), flush=True)
row = {"workload": "C2 layer forward [V=%d, L=%d, M=%d], 256 -> 256" % (V, L, graph.M), "status_word": st,
       "out_bit_identical": same_out, "bucket_sums_bit_identical": same_agg, "max_abs_diff": diff}
if st == 0:
    row["gather_us"] = timed(lambda: ops._seg_reduce_raw(_lib.AGG_SUM, H, graph.rowptr_t, 1, graph.src_t, w, V * L))
    row["product_us"] = timed(lambda: DN.limb_gemm_weight(agg_ref, kernels, DN.WEIGHT_NN, None, _lib.ACT_RELU))
    row["two_kernels_us"] = timed(two_kernels)
    row["fused_with_sums_us"] = timed(lambda: ops._rgcn_fused(H, graph, w, kernels, True, True))
    row["fused_no_sums_us"] = timed(lambda: ops._rgcn_fused(H, graph, w, kernels, True, False))
    row["status_after_timing"] = status()
print(json.dumps(row), flush=True)
with open(out_path, "a") as f:
    f.write(json.dumps(row) + "\n")
