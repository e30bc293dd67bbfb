#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02i
O=$GRAFT_REPO_ROOT/gpurun_out/r02i
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fuzz_model.py tests/test_gpu_dp.py -q -x 2>&1 | tail -3
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
RELGNN_OVERLAP_DW=0 timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown_no_overlap.txt
