#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02z; rm -rf $O; mkdir -p $O; cd /tmp
RELGNN_CAPTURE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c3 -- python $R/scripts/bench_configs.py ${CFG:-C3} > $O/c3.jsonl 2> $O/err.txt
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
cat $O/c3.jsonl | cut -c1-200
