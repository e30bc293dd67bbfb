#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02u
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_gemm.py -q -x 2>&1 | tail -15
timeout 600 python bench.py --steps 60 --warmup 12 --no-roofline --no-cpu-baseline 2>gpurun_out/r02u/bench.err | tee gpurun_out/r02u/bench.json | cut -c1-400
