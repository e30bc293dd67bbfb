#!/bin/bash
# the headline loop from two trees on one box, alternating (a worktree of an older commit under _ab_old/, built there): ms per step and
# the per-step GPU times of each run
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for tree in _ab_old .; do
    ( cd $tree; timeout 300 python bench.py --gpus 1 --steps 40 --warmup 8 --no-roofline --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=json.load(open('bench_detail.json'))['gpu_step_ms_rank0']
print('$tree', round(d['ms_per_step'],4), 'slow steps (> 2.1 ms):', [(i, x) for i, x in enumerate(s) if x > 2.1], 'median', sorted(s)[len(s)//2])" )
  done
done
