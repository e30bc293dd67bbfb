#!/bin/bash
# Round-2 GPU round trip A: parity suite, new DP test, bench line (N=1 with roofline + live PMC), 2-rank self-spawn on one GPU.
set -u
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
O=gpurun_out/r02a
echo "== nproc $(nproc); GPUs: $(python -c 'import torch;print(torch.cuda.device_count())')"
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
echo "== bench N=1"
( time timeout 900 python bench.py --steps 60 --warmup 12 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -4
tail -5 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02a/bench.json"))
print("value %.4g edges/s  ms/step %.3f" % (d["value"], d["ms_per_step"]))
r = d.get("roofline", {})
print("roofline frac", r.get("frac"), "achieved", r.get("achieved"), "traffic", r.get("traffic"), r.get("traffic_source", "")[:80])
for s in r.get("sizes", []):
    print(s["workload"], "warm %.3f ms cold %.3f ms alg %.0f/%.0f GB/s comp %.0f GB/s pmc %s" % (
        s["warm_ms"], s["cold_ms"], s["algorithmic_GBps_warm"], s["algorithmic_GBps_cold"], s["compulsory_GBps_cold"], s.get("pmc")))
print("same_batch", d.get("same_batch"))
print("cpu", d.get("cpu_baseline"))
PY
echo "== bench N=2 on one GPU (gloo, shared device: launch-path test only)"
RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 2>$O/bench2.err >$O/bench2.json; echo "rc=$?"
tail -3 $O/bench2.err; python -c "
import json; d=json.load(open('$O/bench2.json')); print({k:d[k] for k in ('value','n_gpus','world_size','backend','per_rank_edges','ms_per_step')})"
echo "== bench --gpus 2 without share (must fail loudly on a 1-GPU box)"
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1; echo "rc=$?"
