#!/bin/bash
# round 3, GPU session C: panel GEMM with one loader wave per SIMD; kernel trace of the C5 step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_panel_gemm.py -x -q > $O/t_panel.txt 2>&1; echo "panel tests rc=$?" >> $O/t_panel.txt
timeout 120 python scripts/bench_panel_ablate.py >> $O/ablate.jsonl 2>> $O/ablate.err
timeout 600 python scripts/bench_panel_gemm.py dense typed > $O/panel.jsonl 2> $O/panel.err; echo "rc=$?" >> $O/panel.err
RELGNN_PANEL_NC=256 timeout 300 python scripts/bench_panel_gemm.py typed > $O/panel_nc256.jsonl 2>> $O/panel.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c5 -- python $R/bench_other.py C5 > $O/c5.jsonl 2> $O/c5.err)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace
tail -2 $O/t_panel.txt; cat $O/ablate.jsonl; python - <<'PY'
import json,sys
for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03c/panel.jsonl"):
    d=json.loads(l); print(d['what'], {k:v for k,v in d.items() if k.endswith('_us')})
PY
head -25 $O/c5_kernel_stats.csv | cut -c1-150; cut -c1-300 $O/c5.jsonl
