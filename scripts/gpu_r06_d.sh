#!/bin/bash
# round 6, call D: new tests (folding sweep, multi-rank launch), tanh epilogue on the 128-column route, C3 with / without the side stream
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_d; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_rccl.py tests/test_gpu_reference_run.py tests/test_gpu_configs.py tests/test_gpu_layers.py -m gpu -q -x > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for ov in auto 0; do
  echo "== C3 RELGNN_BWD_OVERLAP=$ov"; RELGNN_BWD_OVERLAP=$ov timeout 300 python bench_other.py C3 2>/dev/null | cut -c1-330
done
echo "== C5"; timeout 300 python bench_other.py C5 2>/dev/null | cut -c1-260
echo "== C5 no overlap"; RELGNN_BWD_OVERLAP=0 timeout 300 python bench_other.py C5 2>/dev/null | cut -c1-260
