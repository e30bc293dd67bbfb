#!/bin/bash
# A/B of RELGNN_TYPED_PC on the C5 step (bench_other.py C5), alternated
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 fwd; do
  echo -n "typed_pc=$v rep $rep: "; RELGNN_TYPED_PC=$v timeout 300 python bench_other.py C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('train', d.get('train_ms'), 'fwd', d.get('fwd_ms'), d.get('error',''))"
done; done
