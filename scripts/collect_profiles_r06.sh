#!/bin/bash
# copy the summaries of scripts/gpu_profile_r06.sh (gpurun_out/r06_profile/) into profiles/ under round-6 names
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_profile; P=profiles
cp $O/bench.json $P/r06_bench.json; cp $O/bench_detail.json $P/r06_bench_detail.json
cp $O/bench_c5.json $P/r06_bench_c5.json
cp $O/timed_loop_kernel_stats.csv $P/r06_bench_kernel_stats.csv 2>/dev/null
cp $O/giant_uniform_kernel_stats.csv $P/r06_giant_uniform_kernel_stats.csv
cp $O/roofline_giant_uniform.jsonl $P/r06_roofline_giant_uniform.jsonl
cp $O/C5_kernel_stats.csv $P/r06_c5_kernel_stats.csv; cp $O/C3_kernel_stats.csv $P/r06_c3_kernel_stats.csv
cp $O/sequence_C5.txt $P/r06_c5_step_sequence.txt; cp $O/sequence_C3.txt $P/r06_c3_step_sequence.txt
cp $O/other_C5.jsonl $P/r06_other_c5.jsonl; cp $O/other_C3.jsonl $P/r06_other_c3.jsonl
( echo "# matrix-pipe counters of the product kernels that run (scripts/pmc_mfma_table.py); the C2 timed loop's rows: r06_bench_detail.json roofline.mfma.kernels"; cat $O/gemm_pmc_C5.txt; echo; cat $O/gemm_pmc_C3.txt ) > $P/r06_gemm_pmc.txt
cp $O/edge_kernels_pmc.csv $P/r06_edge_kernels_pmc.csv 2>/dev/null
cp $O/seg_reduce_pmc.csv $P/r06_seg_reduce_pmc.csv
cp $O/bench_2ranks_one_gpu_gloo.json $P/r06_bench_2ranks_one_gpu_gloo.json; cp $O/bench_8ranks_one_gpu_gloo.json $P/r06_bench_8ranks_one_gpu_gloo.json
cp $O/parity_margin.json $P/r06_parity_margin.json 2>/dev/null; cp $O/parity_baseline_size.json $P/r06_parity_baseline_size.json 2>/dev/null
cp $O/gradient_parity_by_seed.json $P/r06_gradient_parity_by_seed.json 2>/dev/null; cp $O/adam_outliers.json $P/r06_adam_outliers.json 2>/dev/null
{
  echo "# default switches"; tail -3 $O/gpu_tests.txt
  echo "# RELGNN_LIMB=pair"; tail -3 $O/gpu_tests_limb_pair.txt
  echo "# RELGNN_GEMM=lib"; grep "^FAILED" $O/gpu_tests_gemm_lib.txt; tail -3 $O/gpu_tests_gemm_lib.txt
  echo "# smoke"; tail -1 $O/smoke.txt
} > $P/r06_gpu_tests_summary.txt
ls -la $P | grep r06_ | wc -l
