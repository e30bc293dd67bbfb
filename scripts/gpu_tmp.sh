#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_limb_gemm.py -x -q 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python -c "import json;d=json.load(open('/tmp/b.json'));print('step', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['per_rank']['gpu_step_ms_median'])"
done
