#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
for a in 0 1 2 3; do RELGNN_AGG_ABLATE=$a timeout 120 python scripts/exp_agg_first.py 2>&1 | grep -E "\(c\)" | tee -a $O/agg_ring.txt; done
