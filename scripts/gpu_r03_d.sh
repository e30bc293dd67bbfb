#!/bin/bash
# round 3, GPU session D: the C2 step with the forward / input-gradient GEMMs on the panel kernel vs the library (A/B/A/B)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2; do
  for g in lib panel; do
    RELGNN_GEMM=$g timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${g}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${g}_$i.json'));print('$g $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['per_rank'])"
  done
done
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5.json 2>> $O/err.txt; cut -c1-900 $O/bench_c5.json
tail -5 $O/err.txt
