#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_streams_graphs.py -q -x -k captured 2>&1 | grep -E "Mismatch|Max abs|Max rel|err_msg|graph_model|dense|ACTUAL|DESIRED|passed|failed|x:|y:" | head -30
