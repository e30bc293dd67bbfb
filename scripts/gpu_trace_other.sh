#!/bin/bash
# kernel trace + stats of one of the other BASELINE configs (bench_other.py C3 | C4 | C5): where the GPU time of its step goes
export TMPDIR=/tmp
C=${1:-C3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$C; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o k -- python $R/bench_other.py $C > $O/lines.jsonl 2> $O/err.txt
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
rm -rf $O/t
cat $O/lines.jsonl | cut -c1-300
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%5.1f%% %7.1f us x %5s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"].replace("(anonymous namespace)::", "")[:110]))
print("total %.1f ms" % (tot / 1e6))
PY
