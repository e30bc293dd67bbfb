"""bench.py's one-line JSON contract: keys, types and internal consistency — on the committed line of the last profile run
(CPU) and on a short live run (GPU, subprocess exactly as the driver starts it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def _check_line(d, full):
    for k, t in TOP.items():
        assert k in d, k
        assert isinstance(d[k], t) or (t is float and isinstance(d[k], int)), (k, type(d[k]))
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this exact metric
    assert d["unit"] == "edges/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    # ("f32" on the default switches; RELGNN_LIMB=pair says what it multiplies with: "f32 storage; 2 x fp16-limb products ...")
    assert d["dtype"].startswith("f32") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    edges = d["config"]["edges_all_ranks_timed_region"]
    assert abs(d["value"] - edges / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert len(d["per_rank_edges"]) == d["world_size"] == d["n_gpus"]
    if not full:
        return
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    if r["traffic"] is not None:                                     # PMC bytes of an HBM-bound launch ~ its algorithmic bytes
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == "edges/sec" and c["cores"] >= 1 and c["value"] > 0


LINE_LIMIT = 6144      # the driver parses the line out of a bounded tail of stdout: round 5's 20 KB line came back `parsed: null`


def _compact(d):
    sys.path.insert(0, ROOT)
    import bench
    assert bench.LINE_LIMIT_BYTES == LINE_LIMIT
    line = bench.compact_line(d)
    text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    assert max(len(v) for v in _strings(line)) <= 160            # no prose in the line
    return line


def _strings(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, list):
        for v in x:
            yield from _strings(v)


def test_the_printed_line_is_small_and_keeps_the_contract_on_every_committed_record():
    """bench.py prints compact_line(record) and writes the record itself to bench_detail.json: over the full records of round 5
    (N = 1 with every section: 20 KB; C5; 2 and 8 ranks) the line stays below 6 KB, keeps every key of the driver's contract with
    `roofline` + `cpu_baseline` and the internal arithmetic, and names its sidecar."""
    for name, full in (("r05_bench.json", True), ("r05_bench_c5.json", False), ("r05_bench_2ranks_one_gpu_gloo.json", False),
                       ("r05_bench_8ranks_one_gpu_gloo.json", False)):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        line = _compact(d)
        _check_line(line, full=full)
        assert line["detail"] == "bench_detail.json"
        for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
            assert line[k] == d[k]
        if d["n_gpus"] > 1:
            assert line["allreduce_flat_ms_per_step"] == d["ms_per_step"] and line["allreduce_overlap_ms_per_step"] > 0
            assert len(line["per_rank"]["gpu_step_ms_median"]) == d["n_gpus"] == len(line["per_rank_edges"])
    line = _compact(json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json"))))
    assert set(line["roofline"]["c2"]) >= {"avg_kernel_ms_in_step", "frac_of_l2_peak", "hbm_side_over_compulsory"}
    assert set(line["other_configs"]) >= {"C3_train_ms", "C3_max_train_ms", "C4_train_ms", "C5_train_ms"}
    for k in ("fp32_exact_split_ms_per_step", "pair_route_ms_per_step", "exact_fp32_lib_ms_per_step"):
        assert line[k] > 0


def test_round6_line_and_sidecar_when_committed():
    """The round-6 evidence run: profiles/r06_bench.json is the LINE as printed, profiles/r06_bench_detail.json the sidecar."""
    lp, dp = (os.path.join(ROOT, "profiles", n) for n in ("r06_bench.json", "r06_bench_detail.json"))
    if not (os.path.exists(lp) and os.path.exists(dp)):
        pytest.skip("no round-6 evidence run committed yet")
    raw = open(lp).read().strip()
    assert len(raw) < LINE_LIMIT and "\n" not in raw
    line, detail = json.loads(raw), json.load(open(dp))
    _check_line(line, full=True)
    assert _compact(detail) == line                       # the line IS the compaction of the sidecar
    assert len(detail["roofline"]["sizes"]) >= 4 and "top_kernels" in detail["roofline"]["c2"]
    m = line["roofline"]["mfma"]
    assert "error" not in m and 0.0 < m["busy_frac"] <= 1.0 and 0.0 < m["frac_of_bf16_peak"] <= 1.0


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    _check_line(d, full=True)
    r = d["roofline"]
    # the line is quoted on the workload no cache can help; the upper bound on Infinity-Cache hits rides along
    assert "giant_uniform" in r["workload"] and 0.0 < r["frac_hbm_proper_lower_bound"] <= r["frac"]
    assert r["frac_of_measured_copy_ceiling"] <= 1.0
    lo, hi = r["skewed_giant"]["frac_range"]
    assert 0.0 < lo < hi <= 1.0
    # round 5: the BASELINE configuration's own figure is a structured field — the gather inside the timed loop against the L2 peak
    c2 = r["c2"]
    assert "error" not in c2, c2
    assert c2["launches_traced"] == 6 * c2["trace_steps"] and 0.0 < c2["frac_of_l2_peak"] < 1.0
    assert abs(c2["algorithmic_GBps"] - c2["algorithmic_bytes_per_launch"] / (c2["avg_kernel_ms_in_step"] * 1e-3) / 1e9) < 1e-6 * c2["algorithmic_GBps"]
    assert abs(c2["frac_of_l2_peak"] - c2["algorithmic_GBps"] / c2["l2_peak_GBps"]) < 1e-9
    assert 1.0 <= c2["hbm_side_over_compulsory"] < 2.0 and c2["top_kernels"][0]["name"].startswith("seg_reduce_wave_kernel")
    # round 5: the headline is the exact-split (fp32) arithmetic; the other two arithmetics are scalar top-level keys
    assert d["dense_products"]["route"] == "limb" and d["dense_products"]["limbs"] == "triple"
    assert d["fp32_exact_split_ms_per_step"] == d["ms_per_step"] and d["fp32_exact_split_value"] == d["value"]
    for k in ("pair_route_ms_per_step", "pair_route_value", "exact_fp32_lib_ms_per_step", "exact_fp32_lib_value"):
        assert isinstance(d[k], float) and d[k] > 0, k
    assert d["pair_route_ms_per_step"] < d["ms_per_step"] < d["exact_fp32_lib_ms_per_step"]
    for k in ("allreduce", "allreduce_ms", "gpu_step_ms", "edge_imbalance", "nranks"):
        assert k in d and not isinstance(d[k], (dict, list)), k
    # every BASELINE config is in the driver-visible record, each with its gather kernel's rate
    cfgs = d["other_configs"]["configs"]
    assert {c["config"][:2] for c in cfgs} == {"C3", "C4", "C5"} and len(cfgs) == 4
    for c in cfgs:
        assert c["train_ms"] > 0 and c["fwd_ms"] > 0 and c["dominant_gather_kernel"]["algorithmic_GBps"] > 0
    # per-rank statistics (one entry per rank)
    for k in ("gpu_step_ms_min", "gpu_step_ms_median", "gpu_step_ms_max", "allreduce_ms_mean", "host_blocked_on_gpu_ms_per_step"):
        assert len(d["per_rank"][k]) == d["world_size"]
    assert d["step_edge_imbalance_max_over_mean"]["max"] >= 1.0 - 1e-9
    c = d["cpu_baseline"]
    # the training leg runs on the WHOLE bench batch (round 5), like the forward-only leg
    assert "the first 16 of the bench batch's 16 graphs" in c["sample"] and c["forward_only_value"] > c["value"]


def test_committed_multi_rank_rehearsal_times_both_forms_of_the_all_reduce():
    """bench.py --gpus N (here: 2 and 8 ranks sharing ONE MI355X over gloo — a launch-path rehearsal, not a scaling number): the
    timed region runs the default (flat) all-reduce, the same loop right behind it the bucketed one, and the line carries both as
    scalars next to the per-rank lists."""
    for n in (2, 8):
        d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_%dranks_one_gpu_gloo.json" % n)))
        _check_line(d, full=False)
        assert d["n_gpus"] == d["nranks"] == n and d["allreduce"] == "flat" and d["backend"] == "gloo"
        assert d["allreduce_flat_ms_per_step"] == d["ms_per_step"] and d["allreduce_overlap_ms_per_step"] > 0
        assert d["allreduce_compare"]["allreduce"] == "overlap" and d["allreduce_compare"]["buckets"] >= 2
        assert len(d["per_rank"]["allreduce_ms_mean"]) == n and d["allreduce_ms"] == max(d["per_rank"]["allreduce_ms_mean"])
        assert 1.0 <= d["edge_imbalance"] <= 1.10


def test_bench_refuses_to_run_without_a_gpu_or_with_too_few():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode != 0 and not p.stdout.strip().startswith("{")


@pytest.mark.gpu
def test_live_short_run_prints_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--no-roofline", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, cwd=ROOT,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT
    d = json.loads(lines[0])
    _check_line(d, full=False)
    assert d["steps"] == 4 and d["warmup"] == 2 and d["n_gpus"] == 1
    # the sidecar holds the full record and the line is its compaction
    detail = json.load(open(os.path.join(ROOT, d["detail"])))
    assert _compact(detail) == d and "dense_products" in detail and "generator" in detail["config"]


@pytest.mark.gpu
def test_live_multi_rank_line_is_small_too():
    """8 ranks sharing this box's GPU over gloo (launch-path rehearsal): the line obeys the same size bound."""
    env = dict(os.environ, RELGNN_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--no-roofline", "--no-cpu-baseline", "--no-extras", "--no-detail",
                        "--task-param-overrides", '{"graphs_per_rank": 16}'], capture_output=True, text=True, cwd=ROOT,
                       timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT
    d = json.loads(lines[0])
    _check_line(d, full=False)
    assert d["n_gpus"] == d["nranks"] == 8 and len(d["per_rank"]["gpu_step_ms_median"]) == 8


@pytest.mark.gpu
def test_asking_for_more_gpus_than_the_box_has_is_an_error():
    import torch
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                        "--no-roofline", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, cwd=ROOT,
                       timeout=900, env={k: v for k, v in os.environ.items() if k != "RELGNN_BENCH_SHARE_GPU"})
    assert p.returncode != 0 and not any(l.startswith("{") for l in p.stdout.splitlines())
