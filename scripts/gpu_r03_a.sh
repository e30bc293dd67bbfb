#!/bin/bash
# round 3, GPU session A: counter list, the new parity / panel-GEMM tests, panel-GEMM benchmark, full suite, bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; rm -rf $O; mkdir -p $O; cd $R
(cd /tmp && timeout 120 rocprofv3 -L > $O/counters_all.txt 2>&1); grep -i "dram\|mall\|hbm\|TCC_EA\|TCC_BUBBLE\|RDREQ\|WRREQ" $O/counters_all.txt | head -80 > $O/counters_mem.txt
timeout 600 python -m pytest tests/test_gpu_panel_gemm.py -x -q > $O/t_panel.txt 2>&1; echo "panel tests rc=$?" >> $O/t_panel.txt
timeout 600 python scripts/bench_panel_gemm.py dense typed > $O/panel_gemm.jsonl 2> $O/panel_gemm.err; echo "rc=$?" >> $O/panel_gemm.err
timeout 900 python -m pytest tests/test_gpu_seg_reduce.py tests/test_gpu_parity_margin.py tests/test_gpu_baseline_size.py -q -s > $O/t_parity.txt 2>&1; echo "parity tests rc=$?" >> $O/t_parity.txt
cp gpurun_out/parity_margin.json gpurun_out/parity_baseline_size.json $O/ 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity_margin.py --deselect tests/test_gpu_baseline_size.py --deselect tests/test_gpu_panel_gemm.py > $O/t_all.txt 2>&1; echo "suite rc=$?" >> $O/t_all.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -3 $O/t_panel.txt; tail -3 $O/t_parity.txt; tail -3 $O/t_all.txt; cut -c1-600 $O/bench.json; tail -2 $O/bench.err; head -c 1500 $O/panel_gemm.jsonl
