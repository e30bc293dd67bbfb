#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_resident.py tests/test_gpu_layers.py -q -x -k "resident or edge_free" 2>&1 | tail -12
