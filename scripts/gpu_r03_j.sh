#!/bin/bash
# round 3, trip j: the C2 step with the forward / input-gradient GEMMs on the limb kernel vs the library (A/B/A/B), and the
# BASELINE-size parity + margin tests with the limb route on
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_limb_gemm.py -x -q 2>&1 | tail -4
for i in 1 2; do
  for g in lib limb; do
    RELGNN_GEMM=$g timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${g}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${g}_$i.json'));print('$g $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['per_rank'])"
  done
done
RELGNN_GEMM=limb timeout 900 python -m pytest tests/test_gpu_baseline_size.py tests/test_gpu_parity_margin.py -x -q 2>&1 | tail -8
cp gpurun_out/parity_margin.json $O/parity_margin_limb.json 2>/dev/null
cp gpurun_out/parity_baseline_size.json $O/parity_baseline_size_limb.json 2>/dev/null
tail -5 $O/err.txt
