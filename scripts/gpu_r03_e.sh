#!/bin/bash
# round 3, GPU session E: edge kernels with preloaded bucket boundaries (C5), layer tests, C5 kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_pair_tables.py tests/test_gpu_layers.py tests/test_gpu_fuzz_edge_layers.py tests/test_gpu_long_buckets.py tests/test_gpu_fuzz.py -x -q > $O/t_layers.txt 2>&1; echo "rc=$?" >> $O/t_layers.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c5 -- python $R/bench_other.py C5 > $O/c5.jsonl 2> $O/c5.err)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace
tail -3 $O/t_layers.txt; head -12 $O/c5_kernel_stats.csv | cut -c1-160; cut -c1-330 $O/c5.jsonl
