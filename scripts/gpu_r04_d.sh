#!/bin/bash
# round 4, trip d: kernel traces of the bench loop on the pair and the triple arithmetic (where did the pair's 4 % go?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd /tmp
for v in pair triple; do
  RELGNN_LIMB=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -o bench -- \
      python $R/bench.py --steps 50 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$v.csv
  rm -rf $O/t_$v
  python - "$O/kernel_stats_$v.csv" "$O/bench_$v.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.load(open(sys.argv[2]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1].split("_")[-1], "ms/step", round(d["ms_per_step"], 4), "GPU kernel time per step (60 steps traced) ms", round(tot / 60 / 1e6, 4))
for r in rows[:26]:
    print("%5.1f%% %7.1f us x %5s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"],
                                         r["Name"].replace("(anonymous namespace)::", "")[:110]))
PY
done
