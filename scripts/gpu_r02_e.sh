#!/bin/bash
set -u
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
timeout 600 python -m pytest tests/test_gpu_layers.py -x -q -k "fused or rgcn" 2>&1 | tail -8
timeout 300 python scripts/bench_fused.py 2>&1 | grep -v amdgpu.ids | tee $O/fused_bench.txt
timeout 900 python -m pytest tests/test_gpu_baseline_size.py tests/test_gpu_fuzz_model.py tests/test_gpu_dp.py -x -q 2>&1 | tail -8
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
RELGNN_FUSED_MFMA=0 timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown_unfused.txt
