#!/bin/bash
# Ordered kernel timeline of ONE steady-state training step of bench.py (which op costs what, in program order).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O; cd /tmp
# usage: gpu_step_timeline.sh [command ...]   (default: the bench loop); a step ends with the Adam kernel
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python $R/bench.py --steps 30 --warmup 10 --no-roofline --no-extras --no-cpu-baseline"; fi
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- $CMD > $O/bench.json 2> $O/err.txt
cd $R
python - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/timeline"
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step boundary = the Adam kernel; take the step between the 20th and 21st occurrence
# a step ends with the LAST optimizer kernel of a run of them (large models take several multi-tensor launches)
idx = [i for i, r in enumerate(rows[:-1]) if "mt_adam_clip" in r["Kernel_Name"] and "mt_" not in rows[i + 1]["Kernel_Name"]]
k = min(24, len(idx) - 2)
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["End_Timestamp"])
out = []
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("Cijk"): name = name[:14] + name[name.find("_MT"):name.find("_MT") + 14]
    out.append("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:110]))
open(O + "/step_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
print("step span us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)
PY
find $O -name "*kernel_trace.csv" -delete
