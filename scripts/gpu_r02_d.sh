#!/bin/bash
set -u
mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02d
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -5
timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
