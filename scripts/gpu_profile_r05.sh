#!/bin/bash
# rocprofv3 evidence for profiles/ (round 5).  Run through gpurun; outputs under gpurun_out/r05_profile/.
#  (1) smoke + the full GPU test suite on the default switches (exact split), then with RELGNN_LIMB=pair and with RELGNN_GEMM=lib
#  (2) the bench line: headline on the exact split, pair / exact-lib legs, roofline with live PMC passes, the in-step C2 roofline
#      (kernel trace + PMC passes of the timed loop; its stats table is kept), other_configs, cpu_baseline on the whole batch
#  (3) bench.py --config C5 (distinct batches)
#  (4) kernel trace + stats of the roofline workload the line is quoted on (giant_uniform, cold protocol only)
#  (5) kernel trace + stats of the C5 step (bench_other.py C5)
#  (6) PMC rows (FETCH_SIZE / WRITE_SIZE) of the FiLM edge kernels (C5) and the RGAT kernels (C4) on the final code
#  (7) limb kernels per shape (time + error vs float64), typed TN isolated
#  (8) the fused layer kernel (opt-in): bit identity + time against gather + product, matrix-pipe / VALU / LDS counters
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_profile
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests.txt
cp gpurun_out/parity_margin.json gpurun_out/parity_baseline_size.json gpurun_out/gradient_parity_by_seed.json gpurun_out/limb16_tn_column_range.json $O/ 2>/dev/null
RELGNN_LIMB=pair timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_limb_pair.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests_limb_pair.txt
RELGNN_GEMM=lib timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_gemm_lib.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests_gemm_lib.txt
cp $O/parity_margin.json $O/parity_baseline_size.json $O/gradient_parity_by_seed.json $O/limb16_tn_column_range.json gpurun_out/ 2>/dev/null
timeout 300 python scripts/parity_margins.py 2> /dev/null | sed -n '/^{/,$p' > $O/parity_margins.json
( time RELGNN_BENCH_KEEP_TRACE=$O timeout 1200 python bench.py 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -3
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_giant_uniform -o g -- \
    python $R/bench_roofline.py --only giant_uniform --iters 16 --cold-only > $O/roofline_giant_uniform.jsonl 2> $O/roofline_giant_uniform.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o c5 -- \
    python $R/bench_other.py C5 > $O/other_c5.jsonl 2> $O/other_c5.err
cd $R
for n in giant_uniform c5; do
  f=$(find $O/trace_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace_giant_uniform $O/trace_c5
GRAFT_REPO_ROOT=$R timeout 900 bash scripts/profile_edge_pmc.sh > $O/edge_pmc.log 2>&1; cp gpurun_out/prof_edge_pmc/edge_kernels_pmc.csv $O/ 2>/dev/null
timeout 600 python scripts/bench_limb_gemm.py > $O/limb_gemm.jsonl 2> $O/limb_gemm.err
timeout 300 python scripts/bench_typed_tn.py > $O/typed_tn.jsonl 2> $O/typed_tn.err
timeout 300 python scripts/bench_rgcn_fused.py > $O/rgcn_fused.txt 2>&1
GRAFT_REPO_ROOT=$R timeout 900 bash scripts/gpu_pmc_rgcn_fused.sh > $O/rgcn_fused_pmc.log 2>&1; cp gpurun_out/pmc_rgcn_fused/summary.txt $O/rgcn_fused_pmc.txt 2>/dev/null
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_profile"
d = json.load(open(O + "/bench.json"))
rows = ["workload,counter,mean_per_launch"]
for s in d.get("roofline", {}).get("sizes", []):
    for k, v in (s.get("pmc") or {}).items():
        rows.append("%s,%s,%r" % (s["workload"], k, v))
open(O + "/seg_reduce_pmc.csv", "w").write("\n".join(rows) + "\n")
r = d["roofline"]
print("value %.4g edges/s, %.4f ms/step (dtype %s, limbs %s); roofline frac %.3f achieved %.0f GB/s avg_kernel_ms %.3f traffic %s" % (
    d["value"], d["ms_per_step"], d["dtype"], d["dense_products"]["limbs"], r["frac"], r["achieved"], r["avg_kernel_ms"], r["traffic"]))
print("pair", d.get("pair_route_ms_per_step"), "exact lib", d.get("exact_fp32_lib_ms_per_step"))
c2 = r.get("c2", {}); print("c2 in-step:", {k: c2.get(k) for k in ("avg_kernel_ms_in_step", "frac_of_l2_peak", "hbm_side_over_compulsory", "error")})
for c in d.get("other_configs", {}).get("configs", []):
    print(c.get("config", "")[:50], c.get("train_ms"), c.get("train_ms_hipgraph"))
c = d.get("cpu_baseline", {}); print("cpu", c.get("value"), c.get("forward_only_value"), c.get("cores"))
PY
tail -2 $O/gpu_tests.txt; tail -2 $O/gpu_tests_limb_pair.txt; tail -2 $O/gpu_tests_gemm_lib.txt; head -3 $O/giant_uniform_kernel_stats.csv | cut -c1-200; cut -c1-300 $O/bench_c5.json; cat $O/typed_tn.jsonl; tail -5 $O/edge_kernels_pmc.csv
