#!/bin/bash
# Round-2 GPU round trip C: parity suite with the MFMA GEMM + device batch gather, GEMM micro-benchmark, pipeline step.
set -u
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c
echo "== pytest -m gpu (gemm + resident first)"
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_resident.py -x -q 2>&1 | tail -15
echo "== pytest -m gpu (all)"
( time timeout 1500 python -m pytest tests -m gpu -q -rs 2>&1 | tail -25 ) 2>&1 | tee $O/pytest_gpu.log
echo "== gemm bench"
timeout 300 python scripts/bench_gemm.py 2>&1 | tee $O/gemm_bench.txt
echo "== pipeline breakdown (own GEMM)"
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
echo "== pipeline breakdown (library GEMM)"
RELGNN_GEMM=lib timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown_lib.txt
echo "== bench (short)"
timeout 600 python bench.py --steps 60 --warmup 12 --no-roofline --no-cpu-baseline 2>$O/bench.err | tee $O/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g ms/step %.3f' % (d['value'], d['ms_per_step'])); print(d.get('same_batch'))"
