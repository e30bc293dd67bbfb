#!/bin/bash
# round 3, GPU session B: panel-GEMM ablations, typed (C5) shapes, TN, other configs with the typed transform on the panel kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_panel_gemm.py -x -q > $O/t_panel.txt 2>&1; echo "panel tests rc=$?" >> $O/t_panel.txt
timeout 120 python scripts/bench_panel_ablate.py >> $O/ablate.jsonl 2>> $O/ablate.err
timeout 600 python scripts/bench_panel_gemm.py dense typed > $O/panel_typed.jsonl 2> $O/panel_typed.err; echo "rc=$?" >> $O/panel_typed.err
timeout 300 python -m pytest tests/test_gpu_pair_tables.py tests/test_gpu_layers.py tests/test_gpu_fuzz_edge_layers.py -x -q > $O/t_typed_layers.txt 2>&1; echo "rc=$?" >> $O/t_typed_layers.txt
timeout 300 python bench_other.py C3 C4 C5 > $O/other.jsonl 2> $O/other.err; echo "rc=$?" >> $O/other.err
RELGNN_TYPED=bmm timeout 200 python bench_other.py C5 > $O/other_bmm.jsonl 2>> $O/other.err
tail -2 $O/t_panel.txt; cat $O/ablate.jsonl; cut -c1-400 $O/panel_typed.jsonl; tail -3 $O/panel_typed.err; tail -3 $O/t_typed_layers.txt; cut -c1-700 $O/other.jsonl; cut -c1-300 $O/other_bmm.jsonl; tail -5 $O/other.err
