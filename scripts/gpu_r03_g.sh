#!/bin/bash
# round 3, GPU session G: kernel traces of the C3 and C4 steps
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; rm -rf $O; mkdir -p $O; cd /tmp
for c in C3 C4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o t -- python $R/bench_other.py $c > $O/$c.jsonl 2> $O/$c.err
  f=$(find $O/trace_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace_C3 $O/trace_C4
python - <<'PY'
import csv, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03g"
for c in ("C3","C4"):
    rows=list(csv.DictReader(open(O+"/%s_kernel_stats.csv"%c)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(c, "total ms", tot/1e6)
    for r in rows[:22]:
        print("  %5.1f%% %5s calls avg %8.1f us %s"%(float(r['Percentage']), r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:95]))
PY
