#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02o
O=$GRAFT_REPO_ROOT/gpurun_out/r02o
timeout 300 python -m pytest tests/test_gpu_resident.py tests/test_gpu_dp.py -q -x 2>&1 | tail -3
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | tee $O/pytest_gpu.log
timeout 300 python scripts/exp_pipeline_breakdown.py 2>/dev/null | tee $O/pipeline_breakdown.txt
timeout 600 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-pmc 2>$O/bench.err | tee $O/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g ms/step %.3f loss %.3f' % (d['value'], d['ms_per_step'], d['final_loss']))
for s in d['roofline']['sizes']: print(s['workload'], 'warm %.3f cold %.3f ms  alg %.0f GB/s (L2 frac %.2f)' % (s['warm_ms'], s['cold_ms'], s['algorithmic_GBps_warm'], s['frac_of_l2_peak_algorithmic_warm']))"
