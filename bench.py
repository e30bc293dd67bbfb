#!/usr/bin/env python
"""bench.py — the reference's headline metric on MI355X.

Metric (BASELINE.json): edges/sec of RGCN on PPI-shaped batches, hidden_size=256, 3 layers, sum aggregation,
where "edges" = sum over edge types of the adjacency-list lengths of a batch, counted once per step whatever the
layer count (tasks/ppi_task.py:244-250, models/sparse_graph_model.py:285,310 of the reference).

A step = ONE training step of the reference's model the way the reference's epoch loop runs it
(models/sparse_graph_model.py:263-311): the next DISTINCT batch of a shuffled epoch is assembled (disjoint union of
~16 PPI-shaped graphs ~ 2 M edges = BASELINE.json configs[1], "C2"), bucketed by (target, type) / (source, type),
3-layer RGCN forward, PPI head + loss, backward, per-variable gradient clipping, Adam update, and the step's
metrics are fetched to the host (one step late, so the fetch never idles the GPU).  The data fold is resident in
HBM when the timed region starts (tasks/resident.py); nothing is cached across steps.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Either launched through torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in
  the environment) or, when WORLD_SIZE is unset, bench.py re-launches itself through torch.distributed.run with N
  ranks.  Graphs are sharded across ranks by edge count (weak scaling: 64 graphs per rank), message passing needs
  no collective, ONE RCCL all-reduce of the flat gradient per step.

Prints ONE JSON line on rank 0 with `roofline` (the gather/segment-reduce kernel, timed live with HIP events on the
launch stream at four working-set sizes, HBM bytes from live rocprofv3 PMC passes) and `cpu_baseline` (the
reference-order CPU restatement timed on this box's host cores on a bounded sample of the same batch).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

GRAPHS_PER_BATCH = 16         # BASELINE.json configs[1]: ~2 M edges per batch


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the kernel roofline section (rank 0, N=1 only)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (same-batch step, forward only, H2D)")
    ap.add_argument("--config", default="C2", choices=["C2", "C5"],
                    help="C2 (default): BASELINE configs[1], RGCN on PPI-shaped batches — the headline metric.  C5: BASELINE "
                         "configs[4], GNN-FiLM on VarMisuse-shaped batches (23 edge types, h=128, 10 layers), ~1.2 M edges per rank "
                         "and step, sharded by graph with ONE all-reduce of the ~46 MB FiLM gradient per step")
    ap.add_argument("--cpu-sample-graphs", type=int, default=GRAPHS_PER_BATCH,
                    help="graphs of the bench batch the CPU baseline's TRAINING leg runs on (default: all %d = the whole C2 batch, "
                         "~11 s per step on 16 host threads)" % GRAPHS_PER_BATCH)
    ap.add_argument("--model-param-overrides", default=None,
                    help="JSON dict of model hyper-parameters laid over the config's (the reference's train.py:38-59 layering: class "
                         "defaults -> the config's values -> this), e.g. '{\"graph_num_layers\": 4}'.  A run with overrides is NOT "
                         "the BASELINE workload any more: the line says so in config.model_param_overrides")
    ap.add_argument("--task-param-overrides", default=None,
                    help="JSON dict laid over the synthetic fold's generator parameters (graphs_per_rank, mean_nodes, ...)")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--no-allreduce-compare", action="store_true",
                    help="N > 1: skip the second loop that times the other form of the gradient all-reduce")
    ap.add_argument("--no-detail", action="store_true",
                    help="do not write the sidecar bench_detail.json (child runs of this file must not overwrite the parent's)")
    ap.add_argument("--no-step-trace", action="store_true",
                    help="skip the rocprofv3 kernel trace / PMC passes of the timed loop (roofline.c2 = null)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: re-launch through torch.distributed.run
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: run N ranks of this same file through
    torch.distributed.run (one process per GPU, RCCL).  Rank 0 of the child job prints the JSON line on our stdout."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("RELGNN_BENCH_SHARE_GPU") != "1":
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, n_dev), file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
# data
# ------------------------------------------------------------------------------------------------------------------
CONFIGS = {
    # graphs per rank and epoch, graphs per batch (sizes the node budget of a batch)
    "C2": {"graphs_per_rank": 64, "graphs_per_batch": 16},
    "C5": {"graphs_per_rank": 200, "graphs_per_batch": 50},
}


def build_local_fold(rank, world, config="C2", overrides=None):
    """This rank's shard of the fold: graphs_per_rank * world synthetic graphs, graph i drawn from its own generator stream
    default_rng([seed, i]) (tasks/synthetic.py), sharded over ranks by edge count (LPT).  The sizes of ALL graphs cost one
    draw each (the node count is the stream's first draw), so a rank builds only the graphs it owns — at 8 ranks that is 1/8 of
    the host work and memory of generating the whole fold everywhere."""
    from tf_gnn_samples_amd.parallel import shard_graphs_by_edges
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    from tf_gnn_samples_amd.tasks import synthetic as S
    overrides = dict(overrides or {})
    n_graphs = int(overrides.pop("graphs_per_rank", CONFIGS[config]["graphs_per_rank"])) * world
    seed = int(overrides.pop("seed", 0))
    if config == "C2":
        gen = S.ppi_shaped_generator_params(num_graphs=n_graphs, seed=seed)
        unknown = sorted(k for k in overrides if k not in gen)
        if unknown:
            raise SystemExit("--task-param-overrides: unknown generator parameter(s) %s (known: graphs_per_rank, seed, %s)"
                             % (unknown, ", ".join(sorted(gen))))
        gen.update(overrides)
        size_of, make = S.ppi_shaped_graph_size, S.make_ppi_shaped_graph
        kw = {k: gen[k] for k in ("mean_nodes", "std_nodes", "min_nodes", "max_nodes", "fwd_edges_per_node")}
        make_kw = dict(kw, feature_size=gen["feature_size"], num_labels=gen["num_labels"],
                       target_lognormal_sigma=gen["target_lognormal_sigma"])
    else:
        gen = {"num_graphs": n_graphs, "seed": seed, "shape": "VarMisuse-like program graphs: 23 edge types (11 base types x "
               "fwd/bkwd + self loops, tasks/varmisuse_task.py:22-28,244-247), ~2500 nodes, 4.7 forward edges per node, two "
               "chain-like types carry 60 % of the edges", "feature_size": 128}
        if overrides:
            raise SystemExit("--task-param-overrides: the C5 generator takes graphs_per_rank and seed only")
        size_of, make, kw, make_kw = S.varmisuse_shaped_graph_size, S.make_varmisuse_shaped_graph, {}, {}
    gen["per_graph_streams"] = "numpy default_rng([seed, graph_index])"
    edge_counts = [size_of(seed, i, **kw)[1] for i in range(n_graphs)]
    shard = shard_graphs_by_edges(edge_counts, world)[rank]
    local = [make(seed, i, **make_kw) for i in shard]
    task = PPI_Task(PPI_Task.default_params())
    g0 = local[0]
    # (the PPI head — per-node sigmoid cross-entropy — also stands in for the VarMisuse task's candidate head, which with its
    # data pipeline is outside the path)
    task.restore_from_metadata({'num_edge_types': len(g0.adjacency_lists), 'initial_node_feature_size': g0.node_features.shape[1],
                                'num_labels': g0.node_labels.shape[1]})
    task._loaded_data[DataFold.TRAIN] = local
    return task, local, gen


def build_local_batch(rank, world, device):
    """The round-1 C2 batch (16 * world PPI-shaped graphs, seed 0, sharded by edge count; this rank's shard as ONE
    resident batch).  Used by scripts/ (kernel experiments); the bench line itself runs distinct batches."""
    from tf_gnn_samples_amd.parallel import shard_graphs_by_edges
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    from tf_gnn_samples_amd.tasks.synthetic import ppi_shaped_generator_params
    gen = ppi_shaped_generator_params(num_graphs=GRAPHS_PER_BATCH * world, seed=0)
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(gen["num_graphs"], 1, seed=gen["seed"])
    graphs = task._loaded_data[DataFold.TRAIN]
    edge_counts = [sum(len(a) for a in g.adjacency_lists) for g in graphs]
    local = [graphs[i] for i in shard_graphs_by_edges(edge_counts, world)[rank]]
    mb = next(task.make_minibatch_iterator(local, DataFold.VALIDATION, 10 ** 9))
    return task, mb, DeviceBatch(mb, device), gen, local


def c2_batch(task, graphs, device, n=GRAPHS_PER_BATCH):
    """The first n graphs of the fold as ONE resident batch (secondary same-batch figures, CPU baseline)."""
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch
    mb = next(task.make_minibatch_iterator(list(graphs[:n]), DataFold.VALIDATION, 10 ** 9))
    return mb, DeviceBatch(mb, device)


# ------------------------------------------------------------------------------------------------------------------
# roofline: live kernel timing (bench_roofline.py) + live PMC passes
# ------------------------------------------------------------------------------------------------------------------
def pmc_passes(names, iters=3, timeout_s=300):
    """HBM-side bytes per launch from rocprofv3 PMC counters, collected NOW, in their own runs (one --pmc pass per
    counter group, --kernel-trace only, as MI355X_MICROARCH.md prescribes): FETCH_SIZE and WRITE_SIZE are KiB at the
    L2's fabric side; gfx950 reports HALF of a wide coalesced read, so bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.
    Infinity-Cache hits are INCLUDED in FETCH_SIZE (upper bound on HBM traffic; equal to it when the working set is
    past the 256 MiB cache).  Returns {workload: {...}} or {"error": ...}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {n: {} for n in names}
    tmp = tempfile.mkdtemp(prefix="relgnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    try:
        for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
            d = os.path.join(tmp, "_".join(group))
            cmd = [exe, "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "k", "--",
                   sys.executable, str(ROOT / "bench_roofline.py"), "--pmc-target", "--only", ",".join(names),
                   "--iters", str(iters)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=timeout_s, text=True)
            order = None
            for line in r.stdout.splitlines():
                if line.startswith("PMC_LAUNCH_ORDER "):
                    order = line.split(" ", 1)[1].split(",")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or order is None or not files:
                return {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(group), r.returncode, r.stdout[-300:])}
            per_counter = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if "seg_reduce_wave_kernel" in row.get("Kernel_Name", ""):
                        per_counter.setdefault(row["Counter_Name"], []).append(
                            (int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
            for cname, vals in per_counter.items():
                vals.sort()
                if len(vals) != len(order):
                    return {"error": "%s: %d kernel rows for %d launches" % (cname, len(vals), len(order))}
                for n in names:
                    v = [x for (_, x), o in zip(vals, order) if o == n]
                    out[n][cname] = float(np.mean(v))
        for n in names:
            c = out[n]
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                c["hbm_side_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        return out
    except subprocess.TimeoutExpired:
        return {"error": "rocprofv3 pass timed out after %d s" % timeout_s}
    except Exception as e:   # reporting only
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_roofline_sizes(iters, timeout_s=600):
    """Run `bench_roofline.py` (HIP-event timing of the dominant kernel at the four working-set sizes) as a child process on
    this rank's GPU and parse its JSON lines.  A child, not a call: the figure must be the kernel's, not this process's —
    after the training loop the same 3.2 GB random gather measured 10.1-10.5 ms in-process against 9.2-9.4 ms in a fresh
    process on the same box (the kernel traces under profiles/ are fresh processes too, so the two now agree)."""
    import bench_roofline as R
    env = dict(os.environ)
    env.setdefault("LOCAL_RANK", "0")
    r = subprocess.run([sys.executable, str(ROOT / "bench_roofline.py"), "--iters", str(iters)], cwd=str(ROOT), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
    sizes = []
    for line in r.stdout.splitlines():
        line = line.strip()
        if line.startswith("{"):
            sizes.append(json.loads(line))
    if r.returncode != 0 or [s["workload"] for s in sizes] != list(R.WORKLOADS):
        raise RuntimeError("bench_roofline.py failed (rc %d): %s" % (r.returncode, r.stderr[-400:]))
    return sizes


def roofline_section(device, iters, with_pmc):
    import bench_roofline as R
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                 # hand this process's cached blocks back before the child needs 14 GB
    sizes = measure_roofline_sizes(iters)
    pmc = pmc_passes(list(R.WORKLOADS), iters=2) if with_pmc else {"error": "skipped (--no-pmc)"}
    for s in sizes:
        c = pmc.get(s["workload"]) if "error" not in pmc else None
        s["pmc"] = c
        if c and "hbm_side_bytes" in c:
            s["hbm_side_GBps_cold"] = c["hbm_side_bytes"] / (s["cold_ms"] * 1e-3) / 1e9
            s["hbm_side_over_compulsory"] = c["hbm_side_bytes"] / s["compulsory_bytes"]
    big = next(s for s in sizes if s["workload"] == "giant_uniform")
    skew = next(s for s in sizes if s["workload"] == "giant")
    c2 = next(s for s in sizes if s["workload"] == "c2")
    traffic = big["pmc"]["hbm_side_bytes"] if big.get("pmc") and "hbm_side_bytes" in big["pmc"] else None
    roof = {
        "kernel": "seg_reduce_wave_kernel<1,false,true> (gather + 1/deg scale + segment-sum + ReLU = one RGCN layer forward)",
        "bound": "hbm",
        "workload": "giant_uniform: ONE graph, 2^21 nodes, 3 edge types [fwd, self, bkwd], sources AND targets uniform, D=256 — "
                    "gathered table 6.4 GB = 25x the 256 MiB Infinity Cache with every row equally likely at every gather, so at "
                    "most %.1f %% of the gathers can hit any cache (mall_hit_upper_bound) and the algorithmic bytes (SURVEY.md 8d) "
                    "are HBM bytes to within that; cold protocol (caches evicted before every launch, tables rotated)"
                    % (100.0 * big["mall_hit_upper_bound"]),
        "achieved": big["algorithmic_GBps_cold"], "peak": R.HBM_PEAK_GBS, "unit": "GB/s",
        "frac": big["frac_of_hbm_peak_algorithmic_cold"],
        "frac_hbm_proper_lower_bound": big["hbm_GBps_lower_bound_cold"] / R.HBM_PEAK_GBS,
        "frac_of_measured_copy_ceiling": big["algorithmic_GBps_cold"] / R.HBM_COPY_GBS,
        "traffic": traffic,
        "traffic_source": ("live rocprofv3 --pmc passes of bench_roofline.py --pmc-target inside this run "
                           "(2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch; fabric-side of the L2: Infinity-Cache hits "
                           "are counted, which is why the cache-hostile workload is the one quoted)" if traffic else
                           "null: " + str(pmc.get("error", "no counters"))),
        "avg_kernel_ms": big["cold_ms"], "algorithmic_bytes_per_launch": big["algorithmic_bytes"],
        "messages_per_launch": big["messages"],
        "skewed_giant": {
            "what": "round 2's roofline workload (2^20 nodes, PPI degree statistics: the bkwd type gathers the forward TARGETS, "
                    "log-normal(0.9)-skewed, so its hottest rows may sit in the Infinity Cache): HBM-proper rate is between "
                    "algorithmic x (1 - mall_hit_upper_bound) and algorithmic",
            "algorithmic_GBps": skew["algorithmic_GBps_cold"], "mall_hit_upper_bound": skew["mall_hit_upper_bound"],
            "frac_range": [skew["hbm_GBps_lower_bound_cold"] / R.HBM_PEAK_GBS, skew["frac_of_hbm_peak_algorithmic_cold"]],
            "avg_kernel_ms": skew["cold_ms"]},
        "c2_note": "at the C2 size the same kernel runs %.1f GB/s algorithmic = %.2f of the %.0f GB/s aggregate-L2 peak: "
                   "the 99 MB table lives in L2 + Infinity Cache, HBM only moves the compulsory bytes (%.0f GB/s cold)"
                   % (c2["algorithmic_GBps_warm"], c2["frac_of_l2_peak_algorithmic_warm"], R.L2_PEAK_GBS,
                      c2["compulsory_GBps_cold"])
                   + "".join("; in the order the RGCN layer runs since round 2 (c2h: rows of the [V, 256] state table into the "
                             "V*L buckets) %.1f GB/s = %.2f of the L2 peak" % (h["algorithmic_GBps_warm"],
                                                                                 h["frac_of_l2_peak_algorithmic_warm"])
                             for h in sizes if h["workload"] == "c2h"),
        "sizes": sizes,
    }
    return roof


def c2_in_step_section(edges_per_step, nodes_per_step, L, D, with_pmc=True, steps=12, warmup=4, timeout_s=240):
    """The dominant kernel INSIDE the headline's timed loop (BASELINE.json configs[1]): `rocprofv3 --kernel-trace --stats` of this
    file's own loop (child process, default switches, no extras) gives the average duration of seg_reduce_wave_kernel over all its
    launches in training steps (3 forward gathers by target + 3 backward gathers by source per step: same messages, same row
    width); algorithmic bytes per launch from the mean batch of the timed region (SURVEY.md 8d in the aggregate-first order:
    M (4 D + 8) + V L 4 D + 4 (V L + 1)).  At this size the [V, D] table (33-37 MB) lives in L2 + Infinity Cache, so the bound is
    the aggregate L2 rate (34.5 TB/s, MI355X_MICROARCH.md), not HBM: `frac_of_l2_peak`.  HBM side: 2 * FETCH_SIZE * 1024 +
    WRITE_SIZE * 1024 per launch from two PMC passes of the same loop, over the compulsory bytes (every table row once + index /
    weight streams + output rows + row pointers)."""
    import bench_roofline as R
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    env = dict(os.environ, TMPDIR="/tmp")
    env.setdefault("LOCAL_RANK", "0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    child = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
             "--no-roofline", "--no-extras", "--no-cpu-baseline", "--no-detail"]
    tmp = tempfile.mkdtemp(prefix="relgnn_step_", dir="/tmp")
    M, V = float(edges_per_step), float(nodes_per_step)
    alg = M * (4 * D + 8) + V * L * 4 * D + 4 * (V * L + 1)
    compulsory = V * 4 * D + M * 8 + V * L * 4 * D + 4 * (V * L + 1)
    out = {"kernel": R.KERNEL_NAME, "what": c2_in_step_section.__doc__.split("\n")[0].strip(),
           "launches_per_step": 6, "messages_per_launch": M, "algorithmic_bytes_per_launch": alg,
           "compulsory_hbm_bytes_per_launch": compulsory, "l2_peak_GBps": R.L2_PEAK_GBS, "trace_steps": steps + warmup}
    try:
        d = os.path.join(tmp, "trace")
        r = subprocess.run([exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "k", "--", *child],
                           cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s, text=True)
        files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return dict(out, error="rocprofv3 --kernel-trace of the timed loop failed (rc %d): %s" % (r.returncode, r.stdout[-300:]))
        rows = list(csv.DictReader(open(files[0])))
        keep = os.environ.get("RELGNN_BENCH_KEEP_TRACE")           # a directory: the stats table is copied there (profiles/)
        if keep:
            os.makedirs(keep, exist_ok=True)
            shutil.copy(files[0], os.path.join(keep, "timed_loop_kernel_stats.csv"))
        total_ns = sum(float(x["TotalDurationNs"]) for x in rows)
        seg = [x for x in rows if R.KERNEL_NAME in x["Name"]]
        calls = sum(int(x["Calls"]) for x in seg)
        seg_ns = sum(float(x["TotalDurationNs"]) for x in seg)
        if not calls:
            return dict(out, error="no %s rows in the kernel trace" % R.KERNEL_NAME)
        avg_ms = seg_ns / calls * 1e-6
        out.update(avg_kernel_ms_in_step=avg_ms, launches_traced=calls, share_of_kernel_time=seg_ns / max(total_ns, 1.0),
                   kernel_ms_per_step_all_kernels=total_ns * 1e-6 / (steps + warmup),
                   algorithmic_GBps=alg / (avg_ms * 1e-3) / 1e9)
        out["frac_of_l2_peak"] = out["algorithmic_GBps"] / R.L2_PEAK_GBS
        out["frac_of_hbm_peak_algorithmic"] = out["algorithmic_GBps"] / R.HBM_PEAK_GBS      # (> 1: the table is cache resident)
        top = sorted(rows, key=lambda x: -float(x["TotalDurationNs"]))[:8]
        out["top_kernels"] = [{"name": _short_kernel(x["Name"]), "calls": int(x["Calls"]),
                               "avg_us": round(float(x["AverageNs"]) * 1e-3, 2),
                               "share": round(float(x["TotalDurationNs"]) / max(total_ns, 1.0), 4)} for x in top]
        if with_pmc:
            sums = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "k", "--",
                                    *child], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=timeout_s, text=True)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    out["hbm_side_error"] = "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stdout[-200:])
                    break
                vals = [float(row["Counter_Value"]) for f in files for row in csv.DictReader(open(f))
                        if R.KERNEL_NAME in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter]
                if not vals:
                    out["hbm_side_error"] = "no %s rows for %s" % (counter, R.KERNEL_NAME)
                    break
                sums[counter] = float(np.mean(vals))
            if len(sums) == 2:
                out["hbm_side_bytes_per_launch"] = 2.0 * sums["FETCH_SIZE"] * 1024.0 + sums["WRITE_SIZE"] * 1024.0
                out["hbm_side_over_compulsory"] = out["hbm_side_bytes_per_launch"] / compulsory
            out["mfma"] = mfma_pass(exe, child, env, tmp, rows, timeout_s)
        return out
    except subprocess.TimeoutExpired:
        return dict(out, error="rocprofv3 pass of the timed loop timed out after %d s" % timeout_s)
    except Exception as e:   # reporting only
        return dict(out, error=repr(e))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


BF16_DENSE_PEAK_PFLOPS = 2.5      # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
MFMA_FLOPS_32x32x16 = 2 * 32 * 32 * 16


def _short_kernel(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def mfma_pass(exe, child, env, tmp, stat_rows, timeout_s):
    """Matrix-pipe counters of the Dense-product kernels that actually run in the timed loop (one more --pmc pass of the same child):
    busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) — the guide's MfmaUtil —, and the bf16 rate
    SQ_INSTS_MFMA x 32768 flop (every MFMA of these kernels is v_mfma_f32_32x32x16_{bf16,f16}) / the kernel's average duration in the
    kernel trace of the same loop.  Returns {kernels: [...], + the kernel with the largest share of MFMA time hoisted}."""
    group = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"]
    d = os.path.join(tmp, "mfma")
    try:
        r = subprocess.run([exe, "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "k", "--", *child],
                           cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "rocprofv3 --pmc (MFMA counters) timed out after %d s" % timeout_s}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        return {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(group), r.returncode, r.stdout[-200:])}
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            acc.setdefault(_short_kernel(row.get("Kernel_Name", "")), {}).setdefault(row.get("Counter_Name"), []).append(
                float(row["Counter_Value"]))
    avg_ns = {_short_kernel(x["Name"]): (float(x["AverageNs"]), float(x["TotalDurationNs"]), int(x["Calls"])) for x in stat_rows}
    kernels = []
    for name, c in acc.items():
        if not all(k in c for k in group):
            continue
        insts, busy, gui = (float(np.mean(c[k])) for k in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
        if insts <= 0 or name not in avg_ns or gui <= 0:
            continue
        ns, total_ns, calls = avg_ns[name]
        pf = insts * MFMA_FLOPS_32x32x16 / (ns * 1e-9) / 1e15
        kernels.append({"kernel": name, "calls": calls, "avg_kernel_us": round(ns * 1e-3, 2), "mfma_insts_per_launch": insts,
                        "busy_frac": round(busy / (gui / 8.0 * 1024.0), 4), "bf16_PFLOPs": round(pf, 4),
                        "frac_of_bf16_peak": round(pf / BF16_DENSE_PEAK_PFLOPS, 4), "_total_ns": total_ns})
    if not kernels:
        return {"error": "no kernel with MFMA instructions in the counter rows"}
    kernels.sort(key=lambda k: -k["_total_ns"])
    for k in kernels:
        k["share_of_mfma_kernel_time"] = round(k.pop("_total_ns") / sum(avg_ns[x["kernel"]][1] for x in kernels), 4)
    top = dict(kernels[0])
    top["kernels"] = kernels
    top["what"] = mfma_pass.__doc__.split("\n")[0].strip()
    top["caveat"] = ("rates assume 32x32x16 16-bit MFMAs (true of limb_gemm*; the exact-fp32 panel / streaming kernels issue "
                     "v_mfma_f32_32x32x2_f32: read their busy_frac, not their PFLOPs)")
    return top


def route_leg(route_env, what, steps, warmup, timeout_s=180):
    """The same timed loop in a child process on another arithmetic of the node-side Dense products (RELGNN_GEMM / RELGNN_LIMB are
    the initial values of tf_gnn_samples_amd.config.settings): reported next to `value` so that the gain of each limb arithmetic
    is a driver-observable number and anybody who rules one of them out has the figure without it.
      RELGNN_GEMM=lib      exact fp32 through the library (v_mfma_f32_32x32x2_f32, an fmaf chain bit for bit)
      RELGNN_LIMB=triple   every limb product from three bf16 limbs per value (the EXACT split: hi + mid + lo == x), six products"""
    env = dict(os.environ)
    env.update(route_env)
    env.setdefault("LOCAL_RANK", "0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
                            "--no-roofline", "--no-extras", "--no-cpu-baseline", "--no-detail"], cwd=str(ROOT), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"what": what, "switches": route_env, "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                "final_loss": d.get("final_loss")}
    except Exception as e:
        return {"error": repr(e)}


def other_configs_section(timeout_s=240):
    """BASELINE.json configs[2..4] (C3 GGNN/QM9 mean + max, C4 RGAT, C5 GNN-FiLM rank share) on this GPU through bench_other.py
    (a child process, like the roofline: its own allocator state, bounded by a timeout): step time, edges/s and the
    algorithmic-bytes rate of each config's gather kernel from HIP events."""
    env = dict(os.environ)
    env.setdefault("LOCAL_RANK", "0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench_other.py")], cwd=str(ROOT), env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout_s, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "bench_other.py timed out after %d s" % timeout_s}
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if r.returncode != 0 and not rows:
        return {"error": "bench_other.py failed (rc %d): %s" % (r.returncode, r.stderr[-300:])}
    return {"what": "the other BASELINE.json configs on one MI355X: training step on a fixed resident batch (bucketing rebuilt per "
                    "step, fwd + bwd + clip + Adam), forward-only pass, and the config's gather kernel timed with HIP events",
            "configs": rows}


# ------------------------------------------------------------------------------------------------------------------
# secondary host/transfer figures
# ------------------------------------------------------------------------------------------------------------------
def time_h2d(mb, device, iters=7):
    """Host->device time of one batch's feed (pinned staging buffers, one copy per tensor), the copy the reference
    pays inside every sess.run (models/sparse_graph_model.py:293).  Median; reported next to the HBM-resident number,
    never inside it."""
    from tf_gnn_samples_amd.tasks import DeviceBatch
    DeviceBatch(mb, device, pin=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        DeviceBatch(mb, device, pin=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def time_native_batch(task, graphs, device, iters=11):
    """Host-side cost of one batch through the C++ builder (tasks/batcher.py, include/relgnn.h section 9):
    pack = relgnn_batch_pack into a pinned arena (host threads), upload = the single H2D copy of that arena."""
    from tf_gnn_samples_amd.tasks.batcher import NativeBatcher
    nb = NativeBatcher(task.make_graph_store(graphs), device)
    ids = np.arange(len(graphs))
    nb.pack(ids); nb.pack(ids)                     # both arenas allocated + pinned
    torch.cuda.synchronize()
    pack_s, upload_s = [], []
    for i in range(iters):
        t0 = time.perf_counter()
        packed = nb.pack_host(ids, i % 2)
        t1 = time.perf_counter()
        nb.upload(packed, i % 2)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pack_s.append(t1 - t0)
        upload_s.append(t2 - t1)
    return float(np.median(pack_s)) * 1e3, float(np.median(upload_s)) * 1e3, int(packed[3]), nb.num_threads


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(batch_graphs, sample_n, params):
    """Reference-order CPU restatement (oracle/torch_ref.py: gather -> per-edge [E,D]@[D,D] -> 1/deg scale
    -> concat -> index_add -> ReLU) on this box's host cores, SURVEY.md 8d's protocol:
      forward-only leg (the reference's validation pass): the WHOLE bench batch (all GRAPHS_PER_BATCH graphs = C2),
      training leg (fwd + bwd through autograd): the first `sample_n` graphs of that batch — by default ALL of them, the same batch
      (~11 s per step on 16 threads: one warm-up + three timed steps) — nothing is scaled, each leg's value = its own edges / its
      own median step time.  The thread count is the fastest of {quota, quota / 2, quota / 4} on a short probe (4 graphs)."""
    from oracle import torch_ref as R
    from tf_gnn_samples_amd.parallel import effective_cpu_count
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    quota = effective_cpu_count()          # the cgroup quota, not the host's hardware threads
    task = PPI_Task(PPI_Task.default_params())
    h = params['hidden_size']
    gen = torch.Generator().manual_seed(0)

    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return ((torch.rand((i, o), generator=gen) * 2 - 1) * lim).requires_grad_(True)

    def load(graphs):
        mb = next(PPI_Task.make_minibatch_iterator(task, list(graphs), DataFold.VALIDATION, 10 ** 9))
        fd = mb.feed_dict
        return {"mb": mb, "x": torch.as_tensor(fd['initial_node_features'], dtype=torch.float32),
                "adj": [torch.as_tensor(a) for a in fd['adjacency_lists']],
                "deg": torch.as_tensor(fd['type_to_num_incoming_edges'], dtype=torch.float32),
                "labels": torch.as_tensor(fd['target_labels'])}

    sample_n = max(1, min(int(sample_n), len(batch_graphs)))
    full = load(batch_graphs)
    sample = full if sample_n == len(batch_graphs) else load(batch_graphs[:sample_n])
    small = load(batch_graphs[:min(4, len(batch_graphs))])                 # (the thread probe's input)
    F, n_labels = sample["x"].shape[1], sample["labels"].shape[1]
    W = {"in": glorot(F, h), "dense0": glorot(h, h), "out": glorot(h, n_labels), "bias": torch.zeros(n_labels, requires_grad=True)}
    layers = [{"Edge_%i_Weight/kernel" % l: glorot(h, h) for l in range(3)} for _ in range(params['graph_num_layers'])]

    def forward(d):
        cur = torch.tanh(d["x"] @ W["in"])
        for i, lw in enumerate(layers):
            cur = R.sparse_rgcn_layer(cur, d["adj"], d["deg"], h, 1, "ReLU", "sum", weights=lw)
            if i == 0:
                cur = torch.tanh(cur @ W["dense0"])
        logits = cur @ W["out"] + W["bias"]
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, d["labels"], reduction='sum') / d["labels"].shape[0]

    def train_step():
        forward(sample).backward()

    def fwd_step(d):
        with torch.no_grad():
            forward(d)

    def timed(fn, warmups, reps, budget_s):
        t_begin = time.time()
        for _ in range(warmups):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
            if time.time() - t_begin > budget_s and len(ts) >= 3:
                break
        return float(np.median(ts)), len(ts)

    # thread-count probe (an oversubscribed OpenMP pool can be slower than half of it): forward pass of the sample, one warm-up +
    # one timed run per setting
    probe = {}
    for n in sorted({quota, max(1, quota // 2), max(1, quota // 4)}, reverse=True):
        torch.set_num_threads(n)
        probe[n] = timed(lambda: fwd_step(small), 1, 1, 0.0)[0]
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    dt_fwd, n_fwd = timed(lambda: fwd_step(full), 1, 3, 40.0)          # the whole batch
    train_warmups = 1 if sample_n > 8 else 2
    dt, n = timed(train_step, train_warmups, 3 if sample_n > 8 else 5, 60.0)   # the whole batch by default
    torch.set_num_threads(max(1, quota // 2))
    smb, fmb = sample["mb"], full["mb"]
    return {"value": smb.num_edges / dt, "unit": "edges/sec", "cores": cores, "kind": "port",
            "host": "%d hardware threads visible, cgroup CPU quota %d" % (os.cpu_count() or 1, quota),
            "thread_probe_forward_4_graphs_s": {str(k): round(v, 3) for k, v in probe.items()},
            "sample": "training leg (`value`): the first %d of the bench batch's %d graphs (%d edges, %d nodes), full step fwd + bwd, "
                      "median of %d after 1-2 warm-ups; forward-only leg (`forward_only_value`): the WHOLE batch (%d edges, %d nodes = "
                      "C2), median of %d after 1 warm-up.  Nothing is scaled: each value = that leg's edges / that leg's median step "
                      "time.  torch-CPU fp32 restatement of gnns/rgcn.py op order incl. the per-edge matmul, %d threads (the fastest "
                      "of the probed counts; the CPU quota is %d)"
                      % (sample_n, len(batch_graphs), smb.num_edges, smb.num_nodes, n, fmb.num_edges, fmb.num_nodes, n_fwd, cores,
                         quota),
            "sample_short": "train leg: %d of %d graphs of the C2 batch (%d edges), fwd+bwd, median of %d; fwd leg: whole batch; "
                            "torch-CPU port, %d threads" % (sample_n, len(batch_graphs), smb.num_edges, n, cores),
            "ms_per_step": dt * 1e3,
            "forward_only_value": fmb.num_edges / dt_fwd, "forward_only_ms": dt_fwd * 1e3,
            "forward_only_sample": "whole batch"}


# ------------------------------------------------------------------------------------------------------------------
# the printed line: scalars + short structured fields; everything else goes to the sidecar
# ------------------------------------------------------------------------------------------------------------------
LINE_LIMIT_BYTES = 6144          # the driver parses the line out of a bounded tail of stdout (round 5's 20 KB line did not parse)
DETAIL_NAME = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _cut(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1].rstrip() + "…"


def compact_line(full):
    """The ONE line bench.py prints, derived from the full record (which goes to bench_detail.json): every scalar of the driver's
    contract, `config` (workload + sizes, no prose), `roofline` and `cpu_baseline` with the fields the contract names, the three
    arithmetic legs and the other BASELINE configs as scalars.  Pure function of the record (tests/test_bench_contract.py runs it
    over committed records); always below LINE_LIMIT_BYTES."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    line["metric"] = _cut(line.get("metric", ""), 120)
    cfg = full.get("config", {})
    c = _pick(cfg, ("graphs_per_rank", "max_nodes_in_batch", "input_pipeline", "parallelism", "edges_all_ranks_timed_region",
                    "nodes_all_ranks_timed_region"))
    c = dict({"workload": _cut(cfg.get("workload", ""), 160)}, **c)
    for k in ("model_param_overrides", "task_param_overrides"):
        if cfg.get(k):
            c[k] = cfg[k]
    line["config"] = c
    r = full.get("roofline")
    if isinstance(r, dict):
        if "error" in r and "achieved" not in r:
            line["roofline"] = {"error": _cut(r["error"], 200)}
        else:
            rr = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "algorithmic_bytes_per_launch",
                           "messages_per_launch", "frac_of_measured_copy_ceiling"))
            rr["kernel"] = _cut(r.get("kernel", ""), 100)
            rr["workload"] = _cut(r.get("workload", ""), 80)
            c2 = r.get("c2")
            if isinstance(c2, dict):
                rr["c2"] = _pick(c2, ("avg_kernel_ms_in_step", "launches_traced", "algorithmic_bytes_per_launch", "frac_of_l2_peak",
                                      "hbm_side_over_compulsory", "share_of_kernel_time"))
                if "error" in c2:
                    rr["c2"]["error"] = _cut(c2["error"], 160)
            m = r.get("mfma")
            if isinstance(m, dict):
                rr["mfma"] = (_pick(m, ("kernel", "busy_frac", "bf16_PFLOPs", "frac_of_bf16_peak", "avg_kernel_us"))
                              if "error" not in m else {"error": _cut(m["error"], 160)})
            line["roofline"] = rr
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ms_per_step", "forward_only_value", "forward_only_ms"))
        if "sample" in cb:
            line["cpu_baseline"]["sample"] = _cut(cb.get("sample_short") or cb["sample"], 140)
        if "error" in cb:
            line["cpu_baseline"]["error"] = _cut(cb["error"], 160)
    for k in ("fp32_exact_split_ms_per_step", "fp32_exact_split_value", "exact_fp32_lib_ms_per_step", "exact_fp32_lib_value",
              "pair_route_ms_per_step", "pair_route_value"):
        if k in full:
            line[k] = full[k]
    oc = full.get("other_configs")
    if isinstance(oc, dict):
        o = {}
        for row in oc.get("configs", []):
            name = str(row.get("config", "?"))
            key = name[:2] + ("_max" if name.startswith("C3") and "max aggregation" in name else "")
            if "error" in row:
                o[key + "_error"] = _cut(row["error"], 100)
                continue
            o[key + "_train_ms"] = row.get("train_ms")
            if isinstance(row.get("train_ms_hipgraph"), (int, float)):
                o[key + "_train_ms_hipgraph"] = row["train_ms_hipgraph"]
        if "error" in oc:
            o["error"] = _cut(oc["error"], 160)
        line["other_configs"] = o
    for k in ("world_size", "nranks", "backend", "per_rank_edges", "allreduce", "allreduce_ms", "gpu_step_ms",
              "gpu_step_ms_slowest_over_fastest_rank", "edge_imbalance", "gradient_allreduce_bytes", "final_loss",
              "handover_status", "host_blocked_on_gpu_ms_per_step", "peak_device_bytes"):
        if k in full:
            line[k] = full[k]
    for k in sorted(full):                       # N > 1: both all-reduce forms as scalars
        if k.startswith("allreduce_") and k not in line and not isinstance(full[k], (dict, list)):
            line[k] = full[k]
    pr = full.get("per_rank")
    if isinstance(pr, dict):
        line["per_rank"] = _pick(pr, ("gpu_step_ms_median", "allreduce_ms_mean"))
    sb = full.get("same_batch")
    if isinstance(sb, dict) and "ms_per_step" in sb:
        line["same_batch_ms_per_step"], line["forward_only_ms"] = sb["ms_per_step"], sb.get("forward_only_ms")
    line["detail"] = full.get("detail", DETAIL_NAME)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT_BYTES:             # cannot happen with the bounded fields above; never lose the number over it
        for k in ("per_rank", "other_configs", "per_rank_edges"):
            line.pop(k, None)
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT_BYTES, len(text)
    return line


def write_detail(full):
    """The full record (per-size roofline rows, PMC blocks, top kernels, prose) next to bench.py and, on a gpurun box, under
    gpurun_out/ so that it travels back.  Returns the path the line names."""
    text = json.dumps(full, indent=1)
    paths = [ROOT / DETAIL_NAME]
    if (ROOT / "gpurun_out").is_dir():
        paths.append(ROOT / "gpurun_out" / DETAIL_NAME)
    written = None
    for p in paths:
        try:
            p.write_text(text)
            written = written or p
        except OSError:
            pass
    return str(written.relative_to(ROOT)) if written else None


# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout must carry exactly ONE line (the JSON): park the real fd 1 and point fd 1 at stderr so that neither
    # Python prints nor C-level chatter of libraries (gloo/RCCL/hipBLASLt) can land on it.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from tf_gnn_samples_amd.parallel import GradientAllReducer, effective_cpu_count, init_distributed
    # torch's CPU thread pool defaults to the host's hardware threads (256); the container may be capped far below
    # that, and a burst of busy threads beyond the quota stalls the whole process for tens of milliseconds
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    torch.set_num_threads(max(1, effective_cpu_count() // (2 * max(1, local_world))))
    share_gpu = os.environ.get("RELGNN_BENCH_SHARE_GPU") == "1"     # debug: all ranks on cuda:0 over gloo (1-GPU box)
    if share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    backend = os.environ.get("RELGNN_DIST_BACKEND") or ("gloo" if share_gpu else None)
    rank, local_rank, world = init_distributed(backend)
    if world != args.gpus:
        print("bench.py: --gpus %d but the job has WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sys.stdout = sys.stderr if rank == 0 else open(os.devnull, "w")

    from tf_gnn_samples_amd.graph import check_pending_graph_errors, clear_graph_cache
    from tf_gnn_samples_amd.models import RGCN_Model, name_to_model_class
    from tf_gnn_samples_amd.models.sparse_graph_model import MetricsReadback
    from tf_gnn_samples_amd.tasks import DataFold
    from tf_gnn_samples_amd import config as route_config
    cfg = CONFIGS[args.config]
    model_overrides = json.loads(args.model_param_overrides) if args.model_param_overrides else {}
    task_overrides = json.loads(args.task_param_overrides) if args.task_param_overrides else {}
    if not isinstance(model_overrides, dict) or not isinstance(task_overrides, dict):
        raise SystemExit("--model-param-overrides / --task-param-overrides take a JSON object")
    task, fold, gen_params = build_local_fold(rank, world, args.config, task_overrides)
    if args.config == "C2":
        model_cls = RGCN_Model
        params = model_cls.default_params()
        params.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1,
                      message_aggregation_function="sum", graph_activation_function="ReLU",
                      graph_layer_input_dropout_keep_prob=1.0)   # README.md:32 of the reference
    else:
        model_cls, extra = name_to_model_class("GNN-FiLM")
        params = model_cls.default_params()
        params.update(extra)
        params.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1,
                      graph_residual_connection_every_num_layers=2,          # tasks/default_hypers/VarMisuse_GNN-FiLM.json
                      graph_layer_input_dropout_keep_prob=1.0)
    unknown = sorted(k for k in model_overrides if k not in params)
    if unknown:
        raise SystemExit("--model-param-overrides: unknown hyper-parameter(s) %s (known: %s)" % (unknown, sorted(params)))
    params.update(model_overrides)
    nodes = sorted(len(g.node_features) for g in fold)
    params['max_nodes_in_batch'] = int(sum(nodes) / max(1, len(fold) // cfg["graphs_per_batch"])) + nodes[-1]
    model = model_cls(params, task, device=str(device))
    # The fold is millions of long-lived Python objects (graphs, arrays, tensors): moved out of the cyclic collector's sight, so that a
    # full collection inside the timed loop scans the step's own objects only and a step's cyclic garbage (autograd graphs hold device
    # tensors) is not kept waiting by CPython's "a quarter of all tracked objects" rule.  Hygiene, not a cure: the distinct-batch C5
    # loop shows rare 75-290 ms steps (0-4 per 60 steps, different per run and per box; none when every step is synchronised:
    # scripts/exp_c5_outliers.py) with and without this — 6 clean runs and 3 with such steps with it, 0 clean of 3 without.
    import gc
    gc.collect()
    gc.freeze()
    # The gradient all-reduce (N > 1): one flat collective behind the backward (`flat`, the default) or buckets that leave during the
    # backward (`overlap`, parallel.py).  The timed region runs the form config.settings.allreduce names; at N > 1 the OTHER form is
    # timed right behind it over the same batch sequence (same shuffling seed), so that one run on an N-GPU node compares the two.
    primary_kind = route_config.settings.allreduce

    def make_reducer(kind):
        if world == 1:
            return None
        if kind == "overlap":
            from tf_gnn_samples_amd.parallel import OverlappedGradientAllReducer
            return OverlappedGradientAllReducer(model.optimizer.params)
        return GradientAllReducer(model.optimizer.params)

    cur_stream = torch.cuda.current_stream(device)

    def timed_loop(kind, steps, warmup):
        """`warmup` untimed steps, barrier + synchronize, exactly `steps` timed steps, barrier + synchronize.  Returns the wall time
        of this rank and its per-step records."""
        reducer = make_reducer(kind)
        overlap_reduce = kind == "overlap"
        np.random.seed(20240924 + rank)           # the epoch shuffling (tasks/sparse_graph_task.py): same batch sequence per call

        def batch_stream():
            """Shuffled epochs over the HBM-resident fold, forever (models/sparse_graph_model.py:263-311)."""
            while True:
                for b in model._batches(fold, DataFold.TRAIN):
                    yield b

        stream = batch_stream()
        state = {"upcoming": next(stream), "pending": None, "edges": 0, "nodes": 0, "graphs": 0, "loss": 0.0, "fetched": 0,
                 "host_wait": 0.0}

        def fetch(pending):
            m, b = pending
            t_wait = time.perf_counter()
            m = m.get()                               # the host sync of sess.run's fetch (:293), one step late: waits for the
            state["host_wait"] += time.perf_counter() - t_wait   # D2H copy enqueued right behind THAT step (MetricsReadback)
            state["loss"] = m['loss']
            m['f1_score']
            state["fetched"] += 1

        marks = {"step_end": [], "reduce": [], "step_edges": []}     # HIP events on the launch stream (GPU-side durations)

        def reduce_hook(batch):
            def hook(_params):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur_stream)
                if overlap_reduce:
                    reducer.finish()                  # (what is left of the collective behind the backward)
                else:
                    reducer(float(batch.num_nodes))
                e1.record(cur_stream)
                marks["reduce"].append((e0, e1))
            return hook

        def one_step():
            batch = state["upcoming"]
            m = model.train_step(batch, grad_hook=reduce_hook(batch) if reducer is not None else None,
                                 pre_backward=(lambda: reducer.arm(float(batch.num_nodes))) if reducer is not None and overlap_reduce
                                 else None)
            readback = MetricsReadback(m)             # async D2H of this step's metrics into pinned memory
            end = torch.cuda.Event(enable_timing=True)
            end.record(cur_stream)
            marks["step_end"].append(end)
            marks["step_edges"].append(batch.num_edges)
            state["upcoming"] = next(stream)          # assembly + bucketing of the next batch, enqueued behind this step
            if state["pending"] is not None:
                fetch(state["pending"])
            state["pending"] = (readback, batch)
            state["edges"] += batch.num_edges
            state["nodes"] += batch.num_nodes
            state["graphs"] += batch.num_graphs

        for _ in range(warmup):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        state.update(edges=0, nodes=0, graphs=0, host_wait=0.0)
        for k in marks:
            marks[k].clear()
        start_mark = torch.cuda.Event(enable_timing=True)
        start_mark.record(cur_stream)
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        fetch(state["pending"])
        check_pending_graph_errors()   # deferred device-side index validation of every step's bucketing
        # per-step GPU-side durations of THIS rank (end-of-step event to end-of-step event), its all-reduce time, host wait
        ends = [start_mark] + marks["step_end"]
        step_ms = np.array([ends[i].elapsed_time(ends[i + 1]) for i in range(len(ends) - 1)]) if len(ends) > 1 else np.zeros(1)
        reduce_ms = np.array([a.elapsed_time(b) for a, b in marks["reduce"]]) if marks["reduce"] else np.zeros(1)
        local_counts = [float(state["edges"]), float(state["nodes"]), float(state["graphs"]), float(step_ms.min()),
                        float(np.median(step_ms)), float(step_ms.max()), float(reduce_ms.mean()),
                        state["host_wait"] / max(1, steps) * 1e3, elapsed * 1e3 / max(1, steps)]
        step_edges = np.array(marks["step_edges"], dtype=np.float64)
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            per_rank = torch.zeros((world, len(local_counts)), dtype=torch.float64, device=device)
            per_rank[rank] = torch.tensor(local_counts, dtype=torch.float64, device=device)
            dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
            edges_rs = torch.zeros((world, len(step_edges)), dtype=torch.float64, device=device)
            edges_rs[rank] = torch.as_tensor(step_edges, device=device)
            dist.all_reduce(edges_rs, op=dist.ReduceOp.SUM)
            elapsed = float(tmax[0])
            per_rank = per_rank.cpu().numpy()
            edges_rs = edges_rs.cpu().numpy()
        else:
            per_rank = np.array([local_counts])
            edges_rs = step_edges[None, :]
        info = {"nbytes": reducer.nbytes if reducer is not None else 0,
                "buckets": len(reducer.buckets) if overlap_reduce and reducer is not None else (1 if reducer is not None else 0)}
        if reducer is not None and hasattr(reducer, "close"):
            reducer.close()                     # (the bucketed form's post-accumulate hooks must not outlive it)
        return {"elapsed": elapsed, "per_rank": per_rank, "edges_rs": edges_rs, "state": state, "reducer": info,
                "step_ms": [round(float(x), 3) for x in step_ms]}

    run = timed_loop(primary_kind, args.steps, args.warmup)
    elapsed, per_rank, edges_rs, state, reducer_info = run["elapsed"], run["per_rank"], run["edges_rs"], run["state"], run["reducer"]
    overlap_reduce = primary_kind == "overlap"
    other_run = None
    if world > 1 and not args.no_allreduce_compare:
        other_kind = "flat" if overlap_reduce else "overlap"
        other_run = (other_kind, timed_loop(other_kind, args.steps, min(args.warmup, 3)))
    total_edges, total_nodes, total_graphs = (float(x) for x in per_rank[:, :3].sum(0))
    # the ranks meet once per step (the all-reduce): a step lasts as long as its largest shard
    imbalance = edges_rs.max(0) / np.maximum(edges_rs.mean(0), 1.0) if edges_rs.size else np.ones(1)
    pipeline = type(next(iter(model._native_batchers.values()))[1]).__name__ if model._native_batchers else "numpy iterator"

    result = {
        "metric": ("edges/sec (whole node), RGCN PPI h=256 training, distinct batches (the reference's epoch-loop definition)"
                   if args.config == "C2" else
                   "edges/sec (whole node), GNN-FiLM VarMisuse-shaped h=128 training (BASELINE configs[4]), distinct batches"),
        "value": total_edges / elapsed,
        "unit": "edges/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # fp32 values everywhere in HBM and in every result.  The tall node-side Dense products are evaluated on the 16-bit matrix
        # pipe from the EXACT three-bf16-limb split of every fp32 operand (hi + mid + lo == x bit for bit; six limb products, fp32
        # accumulation; the dropped terms are below 2^-23 of a product): fp32 semantics, the default route of the package.
        "dtype": "f32" if (route_config.settings.gemm != "limb" or route_config.settings.limb == "triple")
                 else "f32 storage; 2 x fp16-limb products (22-bit operands) in the K = 768 layer products",
        "dense_products": {
            "route": route_config.settings.gemm,
            "limbs": route_config.settings.limb,
            "switches": route_config.current(),
            "triple": "each fp32 operand as three bf16 limbs (hi + mid + lo == x exactly), the six limb products of weight >= 2^-16 "
                      "on v_mfma_f32_32x32x16_bf16 (each exact in fp32), fp32 accumulation; dropped terms < 2^-23 of a product. "
                      "Measured against float64 at [36 k, 768] x [768, 256]: 4.0e-6 max abs (exact-fp32 library GEMM: 5.3e-6); "
                      "C2 layer vs the fp32 oracle 3.8e-6 abs (library route 6.2e-6): profiles/r03_parity_margin.json.  THE DEFAULT "
                      "since round 5 (`value` / `ms_per_step` are measured on it)",
            "pair": "opt-in (RELGNN_LIMB=pair): the aggregate-first layer's products (forward, input gradient, weight gradient: "
                    "K = 768) from TWO fp16 limbs per value (22 significant bits per operand instead of 24) behind exact "
                    "power-of-two scales, three v_mfma_f32_32x32x16_f16 products per fp32 product — narrower than fp32, so it is "
                    "NOT the headline; timed in `fp16_pair_limb_route` (scalars: pair_route_ms_per_step / pair_route_value)",
            "lib": "exact fp32 on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32 through hipBLASLt): timed in `exact_fp32_gemm_route` "
                   "(scalars: exact_fp32_lib_ms_per_step / exact_fp32_lib_value)"},
        "data": "synthetic",
        "config": {
            "workload": ("C2: RGCN on synthetic PPI-shaped batches (~%d graphs, ~%.2f M edges, ~%d k nodes each), 3 edge types "
                         "[fwd,self,bkwd], h=256, 3 layers, sum aggregation, 1/in-degree normalisation, F=50 -> 121 labels; "
                         if args.config == "C2" else
                         "C5: GNN-FiLM on synthetic VarMisuse-shaped batches (~%d graphs, ~%.2f M edges, ~%d k nodes per rank and "
                         "step), 23 edge types, h=128, 10 layers, residual every 2, Dense between all layers, sum aggregation, "
                         "per-node sigmoid head (stand-in for the task head); ")
                        % (cfg["graphs_per_batch"], total_edges / world / args.steps / 1e6, total_nodes / world / args.steps / 1e3)
                        + "step = next distinct batch of a shuffled epoch assembled from the HBM-resident fold + bucketing + fwd + "
                          "bwd + (N > 1: one all-reduce of the flat gradient) + clip + Adam + metrics fetch (one step late)",
            "graphs_per_rank": len(fold), "max_nodes_in_batch": params['max_nodes_in_batch'],
            "input_pipeline": pipeline, "generator": gen_params, "parallelism": "dp%d-by-graph" % world,
            "edges_all_ranks_timed_region": int(total_edges), "nodes_all_ranks_timed_region": int(total_nodes),
            "model_param_overrides": model_overrides, "task_param_overrides": task_overrides,
        },
        "world_size": world,
        "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none (1 rank)",
        "per_rank_edges": [int(x) for x in per_rank[:, 0]],
        # what a bad scaling curve would have to be explained with: GPU-side step durations per rank (HIP events at the end of
        # every step on the launch stream), the all-reduce (pack + scale + collective + unscale, GPU-side), each rank's host wait
        # for its one-step-late metrics copy (~0: that rank's host is its bottleneck), and the per-step edge imbalance between
        # ranks (a step lasts as long as its largest shard)
        "per_rank": {
            "gpu_step_ms_min": [round(float(x), 4) for x in per_rank[:, 3]],
            "gpu_step_ms_median": [round(float(x), 4) for x in per_rank[:, 4]],
            "gpu_step_ms_max": [round(float(x), 4) for x in per_rank[:, 5]],
            "allreduce_ms_mean": [round(float(x), 4) for x in per_rank[:, 6]],
            "host_blocked_on_gpu_ms_per_step": [round(float(x), 4) for x in per_rank[:, 7]],
            "wall_ms_per_step": [round(float(x), 4) for x in per_rank[:, 8]],
        },
        "step_edge_imbalance_max_over_mean": {"mean": float(imbalance.mean()), "max": float(imbalance.max())},
        # rank 0's GPU-side duration of every timed step (sidecar only): a mean far above the median is a few long steps
        "gpu_step_ms_rank0": run["step_ms"],
        "gradient_allreduce_bytes": reducer_info["nbytes"],
        "gradient_allreduce": ("none" if world == 1 else
                               "%d buckets launched from the backward's post-accumulate hooks" % reducer_info["buckets"] if overlap_reduce
                               else "one flat collective behind the backward"),
        # scalars a reader of the driver's record needs to explain an N > 1 number (the lists are in `per_rank`)
        "allreduce": primary_kind,
        "allreduce_ms": round(float(per_rank[:, 6].max()), 4) if world > 1 else 0.0,
        "gpu_step_ms": round(float(per_rank[:, 4].max()), 4),
        "gpu_step_ms_slowest_over_fastest_rank": round(float(per_rank[:, 4].max() / max(per_rank[:, 4].min(), 1e-9)), 4),
        "edge_imbalance": round(float(imbalance.mean()), 4),
        "nranks": int(dist.get_world_size()) if world > 1 else 1,
        "gemm_autotuned": False,
        "final_loss": state["loss"],
        "handover_status": __import__("tf_gnn_samples_amd.ops", fromlist=["handover_status"]).handover_status(),        # 0: every LDS hand-over of the wave-role kernels completed (ops.handover_status)
        # time rank 0's host spent blocked on the (one step late) metrics copy: ~0 = the host is the bottleneck,
        # large = the GPU is
        "host_blocked_on_gpu_ms_per_step": state["host_wait"] / args.steps * 1e3,
        # rank 0's peak device memory over fold + model + timed loop (a batch's arrays must die with the step)
        "peak_device_bytes": int(torch.cuda.max_memory_allocated(device)),
    }

    if other_run is not None:
        kind, o = other_run
        o_edges = float(o["per_rank"][:, 0].sum())
        result["allreduce_compare"] = {
            "what": "the same loop right behind the timed region with the OTHER form of the gradient all-reduce, same batch sequence "
                    "(same shuffling seed per rank), %d steps after %d warm-up steps" % (args.steps, min(args.warmup, 3)),
            "allreduce": kind, "ms_per_step": o["elapsed"] / args.steps * 1e3, "value": o_edges / o["elapsed"],
            "allreduce_ms_mean_per_rank": [round(float(x), 4) for x in o["per_rank"][:, 6]],
            "gpu_step_ms_median_per_rank": [round(float(x), 4) for x in o["per_rank"][:, 4]],
            "buckets": o["reducer"]["buckets"]}
        result["allreduce_%s_ms_per_step" % primary_kind] = result["ms_per_step"]
        result["allreduce_%s_ms_per_step" % kind] = o["elapsed"] / args.steps * 1e3
        result["allreduce_%s_ms" % kind] = round(float(o["per_rank"][:, 6].max()), 4)
    # ---- secondary figures (rank 0, single GPU): same-batch step, forward only, transfers ---------------------------
    if rank == 0 and world == 1 and not args.no_extras and args.config == "C2":
        try:
            mb, batch = c2_batch(task, fold, device)
            from tf_gnn_samples_amd.graph import RelGraph
            side = torch.cuda.Stream(device=device)
            st = {"g": None}

            def same_batch_step():
                # every step pays for its own bucketing (one RelGraph build per step, one step ahead on a side stream)
                batch.graph = st["g"] if st["g"] is not None else RelGraph.build_on_stream(batch.adjacency_lists, batch.num_nodes, side)
                st["g"] = RelGraph.build_on_stream(batch.adjacency_lists, batch.num_nodes, side)
                return model.train_step(batch)

            for _ in range(5):
                same_batch_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                same_batch_step()
            torch.cuda.synchronize()
            same_ms = (time.perf_counter() - t1) / 20 * 1e3
            batch.graph = None
            with torch.no_grad():
                for _ in range(2):
                    clear_graph_cache(); model.forward_batch(batch, training=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    clear_graph_cache(); model.forward_batch(batch, training=False)
                torch.cuda.synchronize()
                fwd_ms = (time.perf_counter() - t1) / 20 * 1e3
            result["same_batch"] = {
                "what": "ONE HBM-resident C2 batch (%d edges) re-run: bucketing + fwd + bwd + clip + Adam, no batch assembly, "
                        "no metrics fetch (round 1's headline protocol, library GEMMs untuned)" % mb.num_edges,
                "ms_per_step": same_ms, "edges_per_sec": mb.num_edges / (same_ms * 1e-3),
                "forward_only_ms": fwd_ms, "forward_only_edges_per_sec": mb.num_edges / (fwd_ms * 1e-3)}
            h2d_ms = time_h2d(mb, device)
            pack_ms, up_ms, nbytes, nthreads = time_native_batch(task, list(fold[:GRAPHS_PER_BATCH]), device)
            result["host_feed"] = {
                "h2d_ms_per_batch_pinned": h2d_ms, "native_pack_ms_median": pack_ms, "native_upload_ms_median": up_ms,
                "arena_bytes": nbytes, "host_threads": nthreads, "upload_GBps": nbytes / (up_ms * 1e-3) / 1e9,
                "what": "PCIe-inclusive cost of ONE C2 batch when the fold is NOT resident (tasks/batcher.py): never inside `value`"}
            del batch
        except Exception as e:
            result["same_batch"] = {"error": repr(e)}
    del model
    torch.cuda.empty_cache()
    if rank == 0 and not args.no_roofline:          # (at N > 1 the other ranks wait at the final barrier meanwhile)
        try:
            result["roofline"] = roofline_section(device, args.kernel_iters, not args.no_pmc)
        except Exception as e:
            result["roofline"] = {"error": repr(e)}
    exact_split = route_config.settings.limb_gemm and route_config.settings.limb == "triple"
    if exact_split:
        result["fp32_exact_split_ms_per_step"], result["fp32_exact_split_value"] = result["ms_per_step"], result["value"]
    if (rank == 0 and world == 1 and not args.no_extras and args.config == "C2" and not model_overrides and not task_overrides
            and route_config.settings.limb_gemm):
        # the same loop on the other arithmetics of the tall Dense products, each in a child process; their step times and rates
        # are hoisted into scalar top-level keys so that a record that keeps only scalars still holds all three
        result["exact_fp32_gemm_route"] = route_leg(
            {"RELGNN_GEMM": "lib"}, "the same loop with RELGNN_GEMM=lib (exact-fp32 library GEMMs on the fp32 matrix pipe for every "
            "Dense product)", args.steps, args.warmup)
        result["exact_fp32_lib_ms_per_step"] = result["exact_fp32_gemm_route"].get("ms_per_step")
        result["exact_fp32_lib_value"] = result["exact_fp32_gemm_route"].get("value")
        if exact_split:
            result["fp16_pair_limb_route"] = route_leg(
                {"RELGNN_LIMB": "pair"}, "the same loop with RELGNN_LIMB=pair (the K = 768 layer products from two fp16 limbs per "
                "value: 22-bit operands, three MFMA products per fp32 product — reduced precision, opt-in, NOT the headline)",
                args.steps, args.warmup)
            result["pair_route_ms_per_step"] = result["fp16_pair_limb_route"].get("ms_per_step")
            result["pair_route_value"] = result["fp16_pair_limb_route"].get("value")
        else:
            result["bf16_triple_limb_route"] = route_leg(
                {"RELGNN_LIMB": "triple"}, "the same loop with RELGNN_LIMB=triple (every limb product from three bf16 limbs per "
                "value: the exact split, six MFMA products per fp32 product)", args.steps, args.warmup)
            result["fp32_exact_split_ms_per_step"] = result["bf16_triple_limb_route"].get("ms_per_step")
            result["fp32_exact_split_value"] = result["bf16_triple_limb_route"].get("value")
    if (rank == 0 and world == 1 and not args.no_roofline and not args.no_step_trace and args.config == "C2"
            and isinstance(result.get("roofline"), dict) and "error" not in result["roofline"]):
        # the BASELINE configuration's own roofline figure: the gather kernel inside the timed loop (kernel trace + PMC passes of
        # this same file), not only the isolated launches above
        edges_per_step = total_edges / max(1, args.steps)
        nodes_per_step = total_nodes / max(1, args.steps)
        result["roofline"]["c2"] = c2_in_step_section(edges_per_step, nodes_per_step, task.num_edge_types,
                                                      params['hidden_size'], with_pmc=not args.no_pmc)
    if rank == 0 and world == 1 and not args.no_extras:
        result["other_configs"] = other_configs_section()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "C2":
        try:
            result["cpu_baseline"] = cpu_baseline(fold[:GRAPHS_PER_BATCH], args.cpu_sample_graphs, params)
        except Exception as e:  # the baseline is reporting only; never lose the GPU number over it
            result["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        # the mfma block of the in-step trace belongs to the roofline (north_star: "MFMA utilisation" of the Dense products)
        c2 = result.get("roofline", {}).get("c2") if isinstance(result.get("roofline"), dict) else None
        if isinstance(c2, dict) and "mfma" in c2:
            result["roofline"]["mfma"] = c2.pop("mfma")
        if not args.no_detail:
            result["detail"] = write_detail(dict(result, detail=DETAIL_NAME))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(compact_line(result)) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
