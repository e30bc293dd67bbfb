#!/bin/bash
# rocprofv3 kernel trace of the C5 (GNN-FiLM, VarMisuse-shaped) step.  Run through gpurun.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c5
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c5 -- \
    python $R/scripts/bench_configs.py C5 > $O/c5.json 2> $O/c5.err
cd $R
cat $O/c5.json
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_c5"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6, "kernels", sum(int(r["Calls"]) for r in rows))
    for r in rows[:25]:
        print(r["Name"][:110].ljust(110), r["Calls"], "%.2f ms" % (float(r["TotalDurationNs"]) / 1e6), "%.1f us" % (float(r["AverageNs"]) / 1e3))
    os.system("cp %s %s/c5_kernel_stats.csv" % (f, O))
PY
