#!/bin/bash
# rocprofv3 evidence for profiles/ (round 6).  Run through gpurun; outputs under gpurun_out/r06_profile/.
#  (1) smoke + the full GPU test suite on the default switches (exact split), then with RELGNN_LIMB=pair and with RELGNN_GEMM=lib
#  (2) the bench line AS THE DRIVER RUNS IT (python bench.py --gpus 1 --steps 20 --warmup 5): the printed line (bench.json, < 6 KB) and
#      the sidecar (bench_detail.json); the stats table of the timed loop is kept
#  (3) bench.py --config C5 (distinct batches)
#  (4) kernel trace + stats of the roofline workload the line is quoted on (giant_uniform, cold protocol only)
#  (5) kernel trace + stats + step sequence of the C5 and C3 steps (bench_other.py)
#  (6) matrix-pipe counters of the product kernels in the C5 and C3 steps (the C2 timed loop's are in the bench line: roofline.mfma)
#  (7) PMC rows (FETCH_SIZE / WRITE_SIZE) of the FiLM edge kernels (C5) and the RGAT kernels (C4)
#  (8) launch-path rehearsals: 2 and 8 ranks sharing this GPU over gloo (line size, both all-reduce forms)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_profile
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests.txt
cp gpurun_out/parity_margin.json gpurun_out/parity_baseline_size.json gpurun_out/gradient_parity_by_seed.json gpurun_out/adam_outliers.json $O/ 2>/dev/null
RELGNN_LIMB=pair timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests_limb_pair.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests_limb_pair.txt
RELGNN_GEMM=lib timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests_gemm_lib.txt 2>&1; echo "gpu tests rc=$?" >> $O/gpu_tests_gemm_lib.txt
( time RELGNN_BENCH_KEEP_TRACE=$O timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -3
cp bench_detail.json $O/bench_detail.json; wc -c $O/bench.json
timeout 300 python bench.py --config C5 --steps 24 --warmup 8 --no-roofline --no-extras --no-cpu-baseline --no-detail > $O/bench_c5.json 2>> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_giant_uniform -o g -- \
    python $R/bench_roofline.py --only giant_uniform --iters 16 --cold-only > $O/roofline_giant_uniform.jsonl 2> $O/roofline_giant_uniform.err
for c in C5 C3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o t -- python $R/bench_other.py $c > $O/other_$c.jsonl 2> $O/other_$c.err
  f=$(find $O/trace_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/scripts/step_sequence.py $f 1400 > $O/sequence_$c.txt 2>&1
  f=$(find $O/trace_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
  timeout 900 python $R/scripts/pmc_mfma_table.py python $R/bench_other.py $c > $O/gemm_pmc_$c.txt 2> $O/gemm_pmc_$c.err
done
f=$(find $O/trace_giant_uniform -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/giant_uniform_kernel_stats.csv
cd $R
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace_giant_uniform $O/trace_C5 $O/trace_C3
GRAFT_REPO_ROOT=$R timeout 900 bash scripts/profile_edge_pmc.sh > $O/edge_pmc.log 2>&1; cp gpurun_out/prof_edge_pmc/edge_kernels_pmc.csv $O/ 2>/dev/null
for n in 2 8; do
  RELGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $n --steps 6 --warmup 3 --no-roofline --no-cpu-baseline --no-extras --no-detail \
      --task-param-overrides '{"graphs_per_rank": 32}' > $O/bench_${n}ranks_one_gpu_gloo.json 2> $O/bench_${n}ranks.err
  wc -c $O/bench_${n}ranks_one_gpu_gloo.json
done
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_profile"
line = json.load(open(O + "/bench.json")); d = json.load(open(O + "/bench_detail.json"))
rows = ["workload,counter,mean_per_launch"]
for s in d.get("roofline", {}).get("sizes", []):
    for k, v in (s.get("pmc") or {}).items():
        rows.append("%s,%s,%r" % (s["workload"], k, v))
open(O + "/seg_reduce_pmc.csv", "w").write("\n".join(rows) + "\n")
r = line["roofline"]
print("value %.4g edges/s, %.4f ms/step (dtype %s); roofline frac %.3f achieved %.0f GB/s avg_kernel_ms %.3f traffic %s" % (
    line["value"], line["ms_per_step"], line["dtype"], r["frac"], r["achieved"], r["avg_kernel_ms"], r["traffic"]))
print("pair", line.get("pair_route_ms_per_step"), "exact lib", line.get("exact_fp32_lib_ms_per_step"))
print("c2 in-step:", r.get("c2")); print("mfma:", r.get("mfma")); print("other:", line.get("other_configs")); print("cpu:", line.get("cpu_baseline"))
PY
tail -2 $O/gpu_tests.txt; tail -2 $O/gpu_tests_limb_pair.txt; tail -2 $O/gpu_tests_gemm_lib.txt; head -3 $O/giant_uniform_kernel_stats.csv | cut -c1-200; cut -c1-300 $O/bench_c5.json; head -12 $O/gemm_pmc_C5.txt; tail -5 $O/edge_kernels_pmc.csv
