#!/bin/bash
# round 6, call E: the launch diet (cached panel weight images, dense_multi) — full default suite, C3 / C5 / C4 step times, a short bench
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_e; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 600 python bench_other.py 2>/dev/null | tee $O/other.jsonl | cut -c1-420
timeout 600 python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-extras --no-detail 2>/dev/null | cut -c1-600
cd /tmp
for c in C3 C5; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -o t -- python $R/bench_other.py $c > /dev/null 2> $O/err_$c.txt
  f=$(find $O/trace_$c -name "*kernel_trace.csv" | head -1)
  python $R/scripts/step_sequence.py $f 10 2>&1 | head -4
  python $R/scripts/step_sequence.py $f 1400 > $O/sequence_$c.txt 2>&1
  rm -rf $O/trace_$c
done
