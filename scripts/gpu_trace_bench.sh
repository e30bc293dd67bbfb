#!/bin/bash
# kernel trace + stats of the bench loop (no roofline / extras): where the GPU time of a C2 step goes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_bench; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o bench -- \
    python $R/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
rm -rf $O/t
python - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_bench"
rows = list(csv.DictReader(open(O + "/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%5.1f%% %7.1f us x %5s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
lib = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("Cijk"))
print("total %.1f ms; Cijk_* share %.1f%%" % (tot / 1e6, 100 * lib / tot))
PY
