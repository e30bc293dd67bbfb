#!/bin/bash
# round 6, call B: hand-over status block tests, Adam bar, CU-mask experiment
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_b; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_rgcn_fused.py tests/test_gpu_reference_run.py tests/test_gpu_streams_graphs.py -m gpu -q -x > $O/tests.txt 2>&1; tail -15 $O/tests.txt
cp gpurun_out/adam_outliers.json $O/ 2>/dev/null
timeout 600 python scripts/exp_cu_mask.py > $O/cu_mask.jsonl 2> $O/cu_mask.err; tail -3 $O/cu_mask.err; cat $O/cu_mask.jsonl
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
