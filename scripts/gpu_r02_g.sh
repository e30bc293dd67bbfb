#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02g
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tee $O/pytest_gpu.log
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
RELGNN_RGCN_ORDER=transform_first timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown_transform_first.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02g/bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6, "per step (52 steps)", tot / 1e6 / 52)
for r in rows[:40]:
    print("%6.2f%% %8.1f us x %5s  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
find $O/trace -name "*kernel_trace.csv" -delete
