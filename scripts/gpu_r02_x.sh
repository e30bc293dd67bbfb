#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02x
timeout 600 python -m pytest tests/test_gpu_lib_gemm.py tests/test_gpu_baseline_size.py -q -x 2>&1 | tail -5
timeout 600 python bench.py --steps 60 --warmup 12 --no-roofline --no-cpu-baseline 2>gpurun_out/r02x/bench.err | tee gpurun_out/r02x/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_blocked_on_gpu_ms_per_step'], d['value'], d['same_batch']['ms_per_step'])"
