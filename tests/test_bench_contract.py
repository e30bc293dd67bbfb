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
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
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


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_c_bench.json")))
    _check_line(d, full=True)
    r = d["roofline"]
    # the line is quoted on the workload no cache can help; the upper bound on Infinity-Cache hits rides along
    assert "giant_uniform" in r["workload"] and 0.0 < r["frac_hbm_proper_lower_bound"] <= r["frac"]
    assert r["frac_of_measured_copy_ceiling"] <= 1.0
    lo, hi = r["skewed_giant"]["frac_range"]
    assert 0.0 < lo < hi <= 1.0
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
    assert "median of 5" in c["sample"] and "2 warm-ups" in c["sample"] and c["forward_only_value"] > c["value"]


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
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_line(d, full=False)
    assert d["steps"] == 4 and d["warmup"] == 2 and d["n_gpus"] == 1


@pytest.mark.gpu
def test_asking_for_more_gpus_than_the_box_has_is_an_error():
    import torch
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                        "--no-roofline", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, cwd=ROOT,
                       timeout=900, env={k: v for k, v in os.environ.items() if k != "RELGNN_BENCH_SHARE_GPU"})
    assert p.returncode != 0 and not any(l.startswith("{") for l in p.stdout.splitlines())
