#!/bin/bash
# round 3, trip k: the weights' limb images split once per step from the per-edge-type kernels as they are
# (relgnn_limb_split_multi_f32, dense.weight_limbs): tests, the C2 step A/B/A/B/A/B (cache off / on), a kernel trace of the step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_streams_graphs.py tests/test_gpu_layers.py tests/test_gpu_baseline_size.py tests/test_gpu_dp.py -x -q 2>&1 | tail -4
for i in 1 2 3; do
  for c in 0 1; do
    RELGNN_WEIGHT_LIMB_CACHE=$c timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_cache${c}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_cache${c}_$i.json'));print('cache $c run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1))"
  done
done
tail -5 $O/err.txt
bash scripts/gpu_trace_bench.sh 2>&1 | tail -36
