#!/usr/bin/env python
"""Roofline measurement of the dominant kernel (gather + 1/deg scale + segment-sum + ReLU of one RGCN
layer forward, csrc/seg_reduce.hip) at four working-set sizes, imported by bench.py and runnable on its own
(the rocprofv3 passes of bench.py and scripts/gpu_profile_r02.sh launch exactly this file).

Why four sizes: the kernel gathers one D-float row per MESSAGE (M*(4D+8) + V*4D + 4(VL+1)
algorithmic bytes per launch), but every source row is gathered ~28 times (the mean out-degree), so whether those
bytes cross HBM depends on whether the gathered table stays in the 4 MiB-per-XCD L2 / 256 MiB Infinity Cache:

  c2      BASELINE.json configs[1]: 16 PPI-shaped graphs, table [V*L, 256] = 99 MB -> lives in L2 + Infinity
          Cache; the algorithmic rate is bounded by the L2 -> CU path (34.5 TB/s aggregate, MI355X_MICROARCH.md),
          NOT by HBM.  HBM only sees the compulsory bytes (each gathered row once + indices + output).
  ppi256  256 PPI-shaped graphs (scripts/exp_c2_big.py): table 1.8 GB, past the Infinity Cache as a whole, but a
          batch is a disjoint union and each XCD walks one graph's ~7 MB slab at a time -> same regime as c2.
  c2h     the same C2 batch in the order the RGCN layer uses since round 2 (aggregate, then transform): rows of the
          [V, 256] node-state table (33 MB; one graph's 2.3 MB slab fits the 4 MiB L2) are gathered into the V*L
          (target, type) buckets; same messages, 3x the output rows.
  ppi256h the 256-graph union in that same aggregate-first order (table [V, 256] = 0.6 GB).
  giant   ONE graph with PPI degree statistics and 2^20 nodes: table 3.2 GB.  Forward-type sources are uniform over
          the table, but the backward type's sources are the forward TARGETS, drawn proportionally to log-normal(0.9)
          weights: its hottest rows could stay in the 256 MiB Infinity Cache, and FETCH_SIZE counts at the L2's fabric
          side, Infinity-Cache hits included.  `mall_hit_upper_bound` is the share of all gathers that go to the
          most-gathered rows that fit 256 MiB (perfect retention of exactly the hottest rows: an upper bound).
  giant_uniform  the same construction with UNIFORM targets and 2^21 nodes: table 6.4 GB = 25x the Infinity Cache, every
          row equally likely at every gather -> at most 256 MiB / 6.4 GB = 4 % of the gathers can hit any cache, the
          algorithmic bytes ARE HBM bytes to within that.  This is the size the `roofline` object of the bench line is
          quoted on (frac <= 1).

Protocols: `warm` = back-to-back launches on the same table (what a training step sees: the table was just
written by the GEMM); `cold` = every timed launch is preceded by a 1 GiB streaming write that evicts L2 and
the Infinity Cache, and launches rotate over COPIES distinct tables where memory allows.  Timing: HIP events on
the launch stream around each single launch.

  python bench_roofline.py [--only c2,ppi256,giant] [--iters N] [--json]
"""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s float4-copy measured)
HBM_COPY_GBS = 6290.0      # measured float4 copy ceiling of the same guide
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth, same guide
# (the workload the bench line quotes comes first, on the fresh process's first allocations: behind `giant` the same launches
#  measured 2 % slower on one box — 23.06 vs 22.59 ms, twice each — which is placement of the 25 GB of tables, not the kernel)
WORKLOADS = ("giant_uniform", "giant", "c2", "c2h", "ppi256", "ppi256h")
MALL_BYTES = 256 << 20     # Infinity Cache (MI355X_MICROARCH.md)
HIDDEN = 256
KERNEL_NAME = "seg_reduce_wave_kernel"


def _ppi_union(num_graphs, seed, device):
    """Disjoint union of PPI-shaped graphs (tasks/synthetic.py) as device adjacency lists + degree table."""
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(num_graphs, 1, seed=seed)
    graphs = task._loaded_data[DataFold.TRAIN]
    mb = next(task.make_minibatch_iterator(graphs, DataFold.VALIDATION, 10 ** 9))
    b = DeviceBatch(mb, device)
    return b.adjacency_lists, b.type_to_num_incoming_edges, mb.num_nodes


def _giant(device, log2_nodes=20, fwd_edges_per_node=28.3, sigma=0.9, seed=0):
    """ONE PPI-statistics graph with 2^log2_nodes nodes, built on the device: sources uniform, targets drawn
    proportionally to log-normal(sigma) node weights, edge types [fwd, self loops, bkwd = fwd reversed]."""
    gen = torch.Generator(device=device).manual_seed(seed)
    n = 1 << log2_nodes
    e = int(round(fwd_edges_per_node * n))
    src = torch.randint(0, n, (e,), device=device, generator=gen, dtype=torch.int64)
    wts = torch.exp(sigma * torch.randn(n, device=device, generator=gen, dtype=torch.float64))
    cdf = torch.cumsum(wts, 0)
    u = torch.rand(e, device=device, generator=gen, dtype=torch.float64) * cdf[-1]
    tgt = torch.searchsorted(cdf, u).clamp_(max=n - 1)
    fwd = torch.stack([src, tgt], 1).to(torch.int32).contiguous()
    ar = torch.arange(n, device=device, dtype=torch.int32)
    adj = [fwd, torch.stack([ar, ar], 1).contiguous(), fwd.flip(1).contiguous()]
    deg = torch.stack([torch.bincount(a[:, 1].long(), minlength=n) for a in adj]).to(torch.float32)
    return adj, deg, n


def build_workload(name, device):
    from tf_gnn_samples_amd.graph import RelGraph
    if name in ("c2", "c2h"):
        adj, deg, V = _ppi_union(16, 0, device)
    elif name in ("ppi256", "ppi256h"):
        adj, deg, V = _ppi_union(256, 0, device)
    elif name == "giant":
        adj, deg, V = _giant(device)
    elif name == "giant_uniform":
        adj, deg, V = _giant(device, log2_nodes=21, sigma=0.0)
    else:
        raise ValueError(name)
    g = RelGraph(adj, V)
    w = g.degree_scale(deg)
    L, M, D = g.L, g.M, HIDDEN
    if name in ("c2h", "ppi256h"):
        from tf_gnn_samples_amd.graph import GatherReducePlan
        plan = GatherReducePlan(rowptr=g.rowptr_t, stride=1, col=g.src_t, w=w, num_out=V * L, num_rows_x=V,
                                rowptr_b=g.rowptr_s, stride_b=L, col_b=g.frow_s, pos_b=g.pos_t_of_s, num_messages=M)
        unique_rows = int(((g.rowptr_s[L::L] - g.rowptr_s[:-1:L]) > 0).sum())     # source NODES with any out-edge
        table_rows, out_rows = V, V * L
    else:
        plan = g.plan_transformed(w)
        unique_rows = int((g.rowptr_s[1:] > g.rowptr_s[:-1]).sum())     # non-empty (source, type) buckets
        table_rows, out_rows = V * L, V
    table_bytes = table_rows * D * 4
    # share of the gathers that go to the most-gathered rows fitting the Infinity Cache (upper bound on its hit rate)
    counts = torch.bincount(plan.col.long(), minlength=table_rows)
    hot = min(table_rows, MALL_BYTES // (D * 4))
    mall_ub = float(torch.topk(counts, hot).values.sum()) / max(1, M) if hot < table_rows else 1.0
    del counts
    free = torch.cuda.mem_get_info(device)[0]
    copies = int(max(1, min(4, (free - (6 << 30)) // table_bytes)))
    gen = torch.Generator(device=device).manual_seed(0)
    tables = [torch.rand((table_rows, D), device=device, generator=gen) * 2 - 1 for _ in range(copies)]
    return {
        "name": name, "plan": plan, "tables": tables, "graph": g, "V": V, "L": L, "M": M, "D": D,
        "unique_rows": unique_rows, "table_bytes": table_bytes, "mall_hit_upper_bound": mall_ub,
        # per message one D-float row + (col, w); per node one D-float output row; row pointers
        "algorithmic_bytes": M * (4 * D + 8) + out_rows * 4 * D + 4 * (V * L + 1),
        # what HBM must move at least once: every distinct gathered row, the index/weight streams, rowptr, output
        "compulsory_bytes": unique_rows * 4 * D + M * 8 + out_rows * 4 * D + 4 * (V * L + 1),
    }


def launch(wl, i=0):
    from tf_gnn_samples_amd import _lib, ops
    p = wl["plan"]
    X = wl["tables"][i % len(wl["tables"])]
    return ops._seg_reduce_raw(_lib.AGG_SUM, X, p.rowptr, p.stride, p.col, p.w, p.num_out, _lib.ACT_RELU)


def time_workload(wl, iters, device, cold_only=False):
    """cold_only: every launch of the process (warm-up included) runs behind the cache-evicting fill, so that the average
    duration in a `rocprofv3 --kernel-trace --stats` summary of the process IS the cold figure (scripts/gpu_profile_r02.sh)."""
    scratch = torch.empty(1 << 28, dtype=torch.float32, device=device)     # 1 GiB: > L2 + Infinity Cache
    for i in range(3):
        if cold_only:
            scratch.fill_(float(i))
        launch(wl, i)
    torch.cuda.synchronize()

    def run(cold):
        evs = []
        for i in range(iters):
            if cold:
                scratch.fill_(float(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            launch(wl, i if cold else 0)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    cold = run(True)
    warm = cold if cold_only else run(False)
    del scratch
    return float(np.mean(warm)), float(np.mean(cold)), float(np.min(warm)), float(np.min(cold))


def summarize(wl, warm_ms, cold_ms, warm_min, cold_min):
    alg, comp = wl["algorithmic_bytes"], wl["compulsory_bytes"]
    gbps = lambda b, ms: b / (ms * 1e-3) / 1e9
    return {
        "workload": wl["name"], "nodes": wl["V"], "edge_types": wl["L"], "messages": wl["M"], "hidden": wl["D"],
        "table_bytes": wl["table_bytes"], "distinct_tables_rotated": len(wl["tables"]),
        "unique_gathered_rows": wl["unique_rows"], "mall_hit_upper_bound": wl["mall_hit_upper_bound"],
        # HBM-proper rate if the Infinity Cache kept exactly the hottest rows (it cannot do better): a LOWER bound
        "hbm_GBps_lower_bound_cold": gbps(alg, cold_ms) * (1.0 - min(1.0, wl["mall_hit_upper_bound"])),
        "algorithmic_bytes": alg, "compulsory_bytes": comp,
        "warm_ms": warm_ms, "cold_ms": cold_ms, "warm_ms_min": warm_min, "cold_ms_min": cold_min,
        "algorithmic_GBps_warm": gbps(alg, warm_ms), "algorithmic_GBps_cold": gbps(alg, cold_ms),
        "compulsory_GBps_cold": gbps(comp, cold_ms),
        "frac_of_hbm_peak_algorithmic_cold": gbps(alg, cold_ms) / HBM_PEAK_GBS,
        "frac_of_hbm_peak_compulsory_cold": gbps(comp, cold_ms) / HBM_PEAK_GBS,
        "frac_of_l2_peak_algorithmic_warm": gbps(alg, warm_ms) / L2_PEAK_GBS,
        "edge_layers_per_sec_warm": wl["M"] / (warm_ms * 1e-3), "edge_layers_per_sec_cold": wl["M"] / (cold_ms * 1e-3),
    }


def measure(names, iters, device, cold_only=False):
    out = []
    for n in names:
        wl = build_workload(n, device)
        it = iters if not n.startswith("giant") else max(8, iters // 2)
        res = summarize(wl, *time_workload(wl, it, device, cold_only))
        if cold_only:
            res["protocol"] = "cold only: warm_* fields repeat the cold figures"
        out.append(res)
        del wl
        torch.cuda.empty_cache()
    return out


# ---- PMC target mode: launch every workload's kernel a few times, print the launch order ---------------------------
def pmc_target(names, iters, device):
    """Run under `rocprofv3 --pmc ...`: per workload `iters` cold launches.  The launch ORDER is printed so that the
    parent can attribute the seg_reduce rows of the counter CSV (dispatch order) to workloads."""
    order = []
    scratch = torch.empty(1 << 28, dtype=torch.float32, device=device)
    for n in names:
        wl = build_workload(n, device)
        torch.cuda.synchronize()
        for i in range(iters):
            scratch.fill_(float(i))
            launch(wl, i)
            order.append(n)
        torch.cuda.synchronize()
        del wl
        torch.cuda.empty_cache()
    print("PMC_LAUNCH_ORDER " + ",".join(order), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(WORKLOADS))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pmc-target", action="store_true")
    ap.add_argument("--cold-only", action="store_true", help="every launch behind the cache-evicting fill (for kernel traces)")
    args = ap.parse_args()
    names = [n for n in args.only.split(",") if n]
    if not torch.cuda.is_available():
        raise SystemExit("bench_roofline.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if args.pmc_target:
        pmc_target(names, args.iters, device)
        return
    for r in measure(names, args.iters, device, args.cold_only):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
