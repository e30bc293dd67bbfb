#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02j
O=$GRAFT_REPO_ROOT/gpurun_out/r02j
timeout 120 python -m pytest tests/test_gpu_layers.py -q -x -k "fused" 2>&1 | tail -3
for a in 0 1 2; do RELGNN_AGG_ABLATE=$a timeout 120 python scripts/exp_agg_first.py 2>&1 | grep -E "fused|\(b\)" | tee -a $O/agg_ring.txt; done
timeout 120 python scripts/bench_fused.py 2>&1 | grep -v -E "amdgpu.ids|Warning|detach|print" | tee $O/fused_bench.txt
