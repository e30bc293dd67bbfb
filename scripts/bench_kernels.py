"""Kernel-level micro-benchmarks (HIP events): achieved algorithmic GB/s of the gather/segment kernels on the
PPI-shaped (C2) and VarMisuse-shaped (C5) batches for several D."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf_gnn_samples_amd import _lib, ops
from tf_gnn_samples_amd.graph import RelGraph
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graphs

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def batches():
    task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    yield "C2-ppi", mb
    graphs = make_varmisuse_shaped_graphs(42, seed=0)
    task = PPI_Task(PPI_Task.default_params())
    task._PPI_Task__num_edge_types = 23
    mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
    yield "C5-varmisuse", mb


for name, mb in batches():
    batch = DeviceBatch(mb, dev)
    g = RelGraph(batch.adjacency_lists, mb.num_nodes)
    V, L, M = g.V, g.L, g.M
    w = g.degree_scale(batch.type_to_num_incoming_edges)
    nonempty = int((g.rowptr_t[1:] != g.rowptr_t[:-1]).sum())
    for D in (128, 256):
        gen = torch.Generator(device=dev).manual_seed(0)
        T = torch.rand((V * L, D), device=dev, generator=gen) * 2 - 1
        H = torch.rand((V, D), device=dev, generator=gen) * 2 - 1
        film = torch.rand((V * L, 2 * D), device=dev, generator=gen) * 2 - 1
        plan = g.plan_transformed(w)
        res = {"batch": name, "V": V, "L": L, "M": M, "D": D, "nonempty_buckets": nonempty}
        ms = timeit(lambda: ops._seg_reduce_raw(_lib.AGG_SUM, T, plan.rowptr, plan.stride, plan.col, plan.w, plan.num_out, _lib.ACT_RELU))
        alg = M * (4 * D + 8) + V * 4 * D
        res["seg_reduce_fwd_ms"] = round(ms, 4); res["seg_reduce_fwd_GBs"] = round(alg / ms / 1e6)
        gout = torch.rand((V, D), device=dev, generator=gen)
        ms = timeit(lambda: ops._seg_reduce_raw(_lib.AGG_SUM, gout, plan.rowptr_b, plan.stride_b, plan.col_b, plan.w_bwd(_lib.AGG_SUM), plan.num_rows_x))
        alg_b = M * (4 * D + 8) + V * L * 4 * D
        res["seg_reduce_bwd_ms"] = round(ms, 4); res["seg_reduce_bwd_GBs"] = round(alg_b / ms / 1e6)
        for act in ("relu", "gelu"):
            ms = timeit(lambda: ops.film_messages_reduce(T, film, g, None, "sum", act))
            alg_f = M * (4 * D + 8) + nonempty * 8 * D + V * 4 * D
            res["film_fwd_%s_ms" % act] = round(ms, 4); res["film_fwd_%s_GBs" % act] = round(alg_f / ms / 1e6)
            ms = timeit(lambda: ops.pair_messages_reduce_fused(T, T, g, None, "sum", act))
            alg_p = M * (4 * D + 8) + nonempty * 4 * D + V * 4 * D
            res["pair_fwd_%s_ms" % act] = round(ms, 4); res["pair_fwd_%s_GBs" % act] = round(alg_p / ms / 1e6)
        K = 4
        s_src = torch.rand((V * L, K), device=dev, generator=gen); s_tgt = torch.rand((V * L, K), device=dev, generator=gen)
        ms = timeit(lambda: ops.rgat_attention(T, s_src, s_tgt, g, K))
        alg_r = M * (4 * D + 8 + 4 * K) + V * (4 * D + 8 * K)
        res["rgat_fwd_ms"] = round(ms, 4); res["rgat_fwd_GBs"] = round(alg_r / ms / 1e6)
        print(json.dumps(res), flush=True)
