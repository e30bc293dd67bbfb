#!/bin/bash
# round 4, trip i: the cheaper split (packed fp32 scale / subtraction, no redundant selects): limb tests, per-shape times, A/B of the step, trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_extreme_values.py tests/test_gpu_baseline_size.py -q --tb=short -x 2>&1 | tail -4
timeout 300 python scripts/bench_limb16.py 2>/dev/null | cut -c1-300 | tail -6
timeout 300 python scripts/bench_limb_gemm.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    x = json.loads(l)
    print(x['shape'], {k: v for k, v in x.items() if k.endswith('_us')})"
for i in 1 2; do
  for v in pair triple; do
    RELGNN_LIMB=$v timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${v}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'])"
  done
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o bench -- \
    python $R/bench.py --steps 50 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_pair.csv; rm -rf $O/t
python - "$O/kernel_stats_pair.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU kernel time per step ms", round(tot / 60 / 1e6, 4))
for r in rows[:8]:
    print("%5.1f%% %7.1f us x %5s  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"],
                                         r["Name"].replace("(anonymous namespace)::", "")[:100]))
PY
