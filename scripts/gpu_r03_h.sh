#!/bin/bash
# round 3, trip h: the limb GEMM (fp32 from three bf16 limbs) — tests, then time + error against the exact-fp32 products
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py -x -q 2>&1 | tail -15 > gpurun_out/r03h/pytest.log
cat gpurun_out/r03h/pytest.log
timeout 600 python scripts/bench_limb_gemm.py > gpurun_out/r03h/limb_gemm.jsonl 2> gpurun_out/r03h/limb_gemm.err
cat gpurun_out/r03h/limb_gemm.jsonl; tail -5 gpurun_out/r03h/limb_gemm.err
