#!/bin/bash
# rocprofv3 kernel stats of the non-headline configs (scripts/bench_configs.py): bash scripts/gpu_profile_configs.sh C5 FILM ...
set -u
export TMPDIR=/tmp RELGNN_CAPTURE=0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_configs
rm -rf $O; mkdir -p $O
cd /tmp
for c in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$c -o $c -- python $R/scripts/bench_configs.py $c > $O/$c.jsonl 2> $O/$c.err
  f=$(find $O/t_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
  find $O/t_$c -name "*kernel_trace.csv" -delete
  cat $O/$c.jsonl | cut -c1-300
  head -25 $O/${c}_kernel_stats.csv | cut -c1-160
done
