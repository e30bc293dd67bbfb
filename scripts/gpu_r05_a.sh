#!/bin/bash
# round 5, GPU call A: full GPU suite on the new defaults + bench with the new sections + C5 typed-TN A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_a/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_a/gpu_tests.txt
tail -5 gpurun_out/r05_a/gpu_tests.txt
RELGNN_BENCH_KEEP_TRACE=gpurun_out/r05_a timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a/bench.json 2> gpurun_out/r05_a/bench.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_a/bench.json"))
    keep = {k: v for k, v in d.items() if not isinstance(v, (dict, list))}
    print(json.dumps(keep)[:3000])
    print("c2:", json.dumps(d.get("roofline", {}).get("c2"))[:2500])
    print("cpu:", json.dumps(d.get("cpu_baseline"))[:600])
except Exception as e:
    print("bench parse failed", e)
PY
for t in limb panel; do
  RELGNN_TYPED_TN=$t timeout 600 python bench.py --config C5 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline --no-extras > gpurun_out/r05_a/bench_c5_$t.json 2> gpurun_out/r05_a/bench_c5_$t.err
  echo "C5 typed_tn=$t rc $?"; python -c "
import json; d=json.load(open('gpurun_out/r05_a/bench_c5_$t.json')); print(d['ms_per_step'], d['value'], d['final_loss'])"
done
