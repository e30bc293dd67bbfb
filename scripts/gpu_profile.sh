#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel trace + stats of the bench command, (2) PMC passes (HBM
# FETCH_SIZE / WRITE_SIZE, separate runs) of the dominant kernel alone.  Run through gpurun.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- \
    python $R/bench.py --steps 10 --warmup 3 --prime 10 --no-cpu-baseline --no-gemm-tuning > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o k -- \
      python $R/scripts/kernel_only.py 10 > $O/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_L2 -o k -- \
      python $R/scripts/kernel_only.py 10 > $O/pmc_L2.log 2>&1
cd $R
find $O -type f | head -40
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof2"
lines = ["pass,counter,launches,mean_per_launch,min,max"]
for c in ("FETCH_SIZE", "WRITE_SIZE", "L2"):
    for f in glob.glob(O + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "seg_reduce_wave_kernel" in r.get("Kernel_Name", "")]
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            print(c, k, "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
            lines.append("%s,%s,%d,%.3f,%.3f,%.3f" % (c, k, len(v), sum(v) / len(v), min(v), max(v)))
open(O + "/seg_reduce_pmc.csv", "w").write("\n".join(lines) + "\n")
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    os.system("cp %s %s/bench_kernel_stats.csv" % (f, O))
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)          # hundreds of MB of per-launch rows; the stats summaries are what gets committed
PY
