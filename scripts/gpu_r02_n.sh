#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02n
O=$GRAFT_REPO_ROOT/gpurun_out/r02n
timeout 300 python -m pytest tests/test_gpu_streams_graphs.py -q -x -k captured 2>&1 | tail -2
RELGNN_TUNE_GEMMS=1 timeout 600 python scripts/bench_configs.py C3 2>$O/configs.err | tee $O/configs_graph_tuned.jsonl
