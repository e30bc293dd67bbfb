#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3).  Run through gpurun; outputs under gpurun_out/r03_profile/.
#  (1) the full GPU test suite (its parity reports land in gpurun_out/*.json)
#  (2) the bench line (roofline section with live PMC passes, other_configs, cpu_baseline)
#  (3) kernel trace + stats of the bench command's timed loop
#  (4) kernel trace + stats of the roofline workload the line is quoted on (giant_uniform, cold protocol only): the average
#      duration of seg_reduce_wave_kernel there must agree with roofline.avg_kernel_ms
#  (5) kernel trace + stats of the C5 step (GNN-FiLM, VarMisuse-shaped) and of `bench.py --config C5`
#  (6) matrix-pipe counters of the panel GEMM and of the limb GEMM kernels next to the library GEMM on the C2 layer shapes
#  (8) counters of the typed K = 128 product kernels (scripts/gpu_pmc_typed.sh)
#  (7) the limb GEMM against the exact-fp32 products per shape (time + error vs float64); the parity-margin test once more with
#      RELGNN_GEMM=lib (the default run above is the limb route)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_profile
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests.txt
cp gpurun_out/parity_margin.json gpurun_out/parity_baseline_size.json $O/ 2>/dev/null
RELGNN_GEMM=lib timeout 600 python -m pytest tests/test_gpu_parity_margin.py -q > $O/margin_lib_tests.txt 2>&1
cp gpurun_out/parity_margin.json $O/parity_margin_exact_fp32_gemm.json 2>/dev/null
cp $O/parity_margin.json gpurun_out/parity_margin.json 2>/dev/null
timeout 600 python scripts/bench_limb_gemm.py > $O/limb_gemm.jsonl 2> $O/limb_gemm.err
timeout 600 python scripts/bench_limb_typed.py > $O/limb_typed.jsonl 2>> $O/limb_gemm.err
( time timeout 900 python bench.py 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -3
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
RELGNN_LIMB=triple timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_limb_triple.json 2>> $O/bench.err
timeout 300 python scripts/bench_limb16.py > $O/limb16.jsonl 2>> $O/limb_gemm.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o bench -- \
    python $R/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_giant_uniform -o g -- \
    python $R/bench_roofline.py --only giant_uniform --iters 16 --cold-only > $O/roofline_giant_uniform.jsonl 2> $O/roofline_giant_uniform.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o c5 -- \
    python $R/bench_other.py C5 > $O/other_c5.jsonl 2> $O/other_c5.err
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  d=$O/pmc_gemm_$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o k -- python $R/scripts/bench_panel_gemm.py pmc > /dev/null 2>> $O/pmc.err
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d ${d}_limb -o k -- python $R/scripts/pmc_limb_target.py > /dev/null 2>> $O/pmc.err
done
cd $R
for n in bench giant_uniform c5; do
  f=$(find $O/trace_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace_bench $O/trace_giant_uniform $O/trace_c5
python - <<'PY'
import csv, glob, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_profile"
d = json.load(open(O + "/bench.json"))
rows = ["workload,counter,mean_per_launch"]
for s in d.get("roofline", {}).get("sizes", []):
    for k, v in (s.get("pmc") or {}).items():
        rows.append("%s,%s,%r" % (s["workload"], k, v))
open(O + "/seg_reduce_pmc.csv", "w").write("\n".join(rows) + "\n")
r = d["roofline"]
print("value %.4g edges/s, %.3f ms/step; roofline frac %.3f achieved %.0f GB/s avg_kernel_ms %.3f traffic %s" % (
    d["value"], d["ms_per_step"], r["frac"], r["achieved"], r["avg_kernel_ms"], r["traffic"]))
# matrix-pipe counters per kernel: mean over the dispatches of the GEMM kernels
agg = {}
for f in glob.glob(O + "/pmc_gemm_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "panel_gemm" in k or "limb_gemm" in k or k.startswith("Cijk"):
            agg.setdefault((k.replace("void (anonymous namespace)::", "")[:70], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
with open(O + "/gemm_pmc.txt", "w") as f:
    for (k, c), v in sorted(agg.items()):
        f.write("%-72s %-30s n=%3d mean %.4g\n" % (k, c, len(v), sum(v) / len(v)))
PY

bash scripts/gpu_pmc_typed.sh > /dev/null 2>&1; cp gpurun_out/pmc_typed/summary.txt $O/typed_pmc.txt 2>/dev/null
tail -3 $O/gpu_tests.txt; head -4 $O/giant_uniform_kernel_stats.csv | cut -c1-200; cat $O/gemm_pmc.txt | head -30; cut -c1-400 $O/bench_c5.json
