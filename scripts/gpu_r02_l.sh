#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02l
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
timeout 300 python -m pytest tests/test_gpu_streams_graphs.py tests/test_gpu_train_utils.py -q -x 2>&1 | tail -15
timeout 600 python scripts/bench_configs.py C3 C4 2>$O/configs.err | tee $O/configs_graph.jsonl
tail -3 $O/configs.err
