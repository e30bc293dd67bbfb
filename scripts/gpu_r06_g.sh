#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for on in 0 1; do
  echo "== cache $on (rep $rep)"; timeout 300 python scripts/ab_sel_cache.py $on C3 C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d.get('config','')[:40], 'train', d.get('train_ms'), 'graph', d.get('train_ms_hipgraph'), 'fwd', d.get('fwd_ms'), d.get('error',''))"
done; done
