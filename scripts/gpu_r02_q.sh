#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02q
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -2
timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q/gemm_bench2.txt
