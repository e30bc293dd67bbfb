#!/bin/bash
# rocprofv3 kernel trace of one scripts/bench_configs.py configuration ($1 = C3|C4|C5|RGIN|MLP0|MLP1).  Run through gpurun.
set -u
export TMPDIR=/tmp
CFG=${1:-C4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$CFG
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- \
    python $R/scripts/bench_configs.py $CFG > $O/out.json 2> $O/err.log
cd $R
cat $O/out.json
python - "$CFG" <<'PY'
import csv, glob, os, sys
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_" + sys.argv[1]
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6, "launches", sum(int(r["Calls"]) for r in rows))
    for r in rows[:30]:
        print(r["Name"][:100].ljust(100), r["Calls"], "%.2f ms" % (float(r["TotalDurationNs"]) / 1e6), "%.1f us" % (float(r["AverageNs"]) / 1e3))
    os.system("cp %s %s/kernel_stats.csv" % (f, O))
PY
