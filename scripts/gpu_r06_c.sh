#!/bin/bash
# round 6, call C: kernel sequences of the C5 and C3 training steps (fixed batch, bench_other.py)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c; rm -rf $O; mkdir -p $O; cd /tmp
for c in C5 C3; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -o t -- python $R/bench_other.py $c > $O/other_$c.jsonl 2> $O/err_$c.txt
  f=$(find $O/trace_$c -name "*kernel_trace.csv" | head -1)
  python $R/scripts/step_sequence.py $f 1400 > $O/sequence_$c.txt 2>&1
  rm -rf $O/trace_$c
  cut -c1-250 $O/other_$c.jsonl
  head -60 $O/sequence_$c.txt
done
