#!/bin/bash
# rocprofv3 evidence for profiles/ (round 2).  Run through gpurun; outputs under gpurun_out/r02_profile/.
#  (1) the bench line (its roofline section runs the live PMC passes itself)
#  (2) kernel trace + stats of the bench command's timed loop
#  (3) kernel trace + stats of the roofline workload alone (HBM-bound 'giant' size): the seg_reduce average duration the
#      bench line's roofline.avg_kernel_ms must agree with (--cold-only: every launch of that process runs the bench line's
#      cold protocol)
#  (4) PMC passes of the four roofline workloads written out as a CSV (same passes bench.py runs live)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_profile
rm -rf $O; mkdir -p $O
cd $R
( time timeout 900 python bench.py --steps 60 --warmup 12 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o bench -- \
    python $R/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_giant -o giant -- \
    python $R/bench_roofline.py --only giant --iters 20 --cold-only > $O/roofline_giant.jsonl 2> $O/roofline_giant.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -o c2 -- \
    python $R/bench_roofline.py --only c2 --iters 40 > $O/roofline_c2.jsonl 2> $O/roofline_c2.err
cd $R
for n in bench giant c2; do
  f=$(find $O/trace_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02_profile"
d = json.load(open(O + "/bench.json"))
rows = ["workload,counter,mean_per_launch"]
for s in d.get("roofline", {}).get("sizes", []):
    for k, v in (s.get("pmc") or {}).items():
        rows.append("%s,%s,%r" % (s["workload"], k, v))
open(O + "/seg_reduce_pmc.csv", "w").write("\n".join(rows) + "\n")
print("value %.4g edges/s, %.3f ms/step; roofline frac %.3f achieved %.0f GB/s avg_kernel_ms %.3f traffic %s" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["avg_kernel_ms"], d["roofline"]["traffic"]))
PY
head -3 $O/giant_kernel_stats.csv | cut -c1-200
