#!/bin/bash
# round 6: the one-kernel GRU cell forward — its tests and everything that runs a GGNN, the cell alone, C3 with and without it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gru; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_gru_cell.py tests/test_gpu_layers.py tests/test_gpu_reference_run.py tests/test_gpu_baseline_size.py tests/test_gpu_streams_graphs.py tests/test_gpu_fuzz_model.py tests/test_gpu_limb_gemm.py tests/test_gpu_configs.py tests/test_gpu_extreme_values.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python scripts/bench_gru_cell.py 2>&1 | tee $O/cell.jsonl
for sw in 1 0; do RELGNN_GRU_CELL=$sw timeout 600 python bench_other.py C3 2>/dev/null | cut -c1-330 | tee -a $O/c3_gru_cell_$sw.jsonl; done
