#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_baseline_size.py tests/test_gpu_streams_graphs.py tests/test_gpu_pair_tables.py -m gpu -q -x > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for rep in 1 2; do for on in 0 1; do
  echo "== cache $on (rep $rep)"; timeout 300 python scripts/ab_sel_cache.py $on C3 C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d.get('config','')[:40], 'train', d.get('train_ms'), 'graph', d.get('train_ms_hipgraph'), 'fwd', d.get('fwd_ms'), d.get('error',''))"
done; done
