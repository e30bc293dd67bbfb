#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02f
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
for a in 0 1 2; do RELGNN_AGG_ABLATE=$a timeout 300 python scripts/exp_agg_first.py 2>&1 | grep -v amdgpu.ids | tee -a $O/agg_first.txt; done
