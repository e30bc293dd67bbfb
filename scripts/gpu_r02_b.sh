#!/bin/bash
# Round-2 GPU round trip B: full parity suite (incl. BASELINE-size cases), pipeline-step breakdown, kernel trace of the
# bench loop, RGDCN / Edge-MLP1 timings.
set -u
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
echo "== pytest -m gpu"
( time timeout 1500 python -m pytest tests -m gpu -x -q -rs 2>&1 | tail -25 ) 2>&1 | tee $O/pytest_gpu.log
echo "== pipeline breakdown"
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
echo "== kernel trace of the bench loop"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02b/bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6)
for r in rows[:28]:
    print("%6.2f%% %8.1f us x %5s  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
find $O/trace -name "*kernel_trace.csv" -delete
cat $O/bench_under_rocprof.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under rocprof: ms/step', d['ms_per_step'])"
echo "== other configs"
timeout 600 python scripts/bench_configs.py RGDCN MLP1 C4 2>/dev/null | tee $O/other_configs.jsonl
