#!/bin/bash
# round 3, trip p: the aggregate-first layer's forward / input-gradient products from two fp16 limbs behind row scales
# (RELGNN_LIMB=pair): tests, BASELINE-size parity + margin sweep on that route, the C2 step alternated triple / pair
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_seg_reduce.py -x -q 2>&1 | tail -4
RELGNN_LIMB=pair timeout 1200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_baseline_size.py tests/test_gpu_parity_margin.py tests/test_gpu_streams_graphs.py tests/test_gpu_dp.py -x -q 2>&1 | tail -4
cp gpurun_out/parity_margin.json $O/parity_margin_pair.json 2>/dev/null; cp gpurun_out/parity_baseline_size.json $O/parity_baseline_size_pair.json 2>/dev/null
for i in 1 2 3; do
  for v in triple pair; do
    RELGNN_LIMB=$v timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${v}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'])"
  done
done
tail -3 $O/err.txt
