#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1 | tee $O/pytest_gpu.log
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
timeout 600 python bench.py --steps 60 --warmup 12 --no-roofline --no-cpu-baseline 2>$O/bench.err | tee $O/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g ms/step %.3f loss %.3f' % (d['value'], d['ms_per_step'], d['final_loss'])); print(d.get('same_batch'))"
