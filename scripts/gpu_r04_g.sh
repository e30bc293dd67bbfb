#!/bin/bash
# round 4, trip g: pair tables without host round trips on resident folds: tests, C5 distinct-batch step, bench_other C5
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_pair_tables.py tests/test_gpu_configs.py tests/test_gpu_layers.py tests/test_gpu_memory.py -q --tb=short -x 2>&1 | tail -8
for i in 1 2; do
  timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5_$i.json 2>> $O/err.txt
  python -c "import json;d=json.load(open('$O/bench_c5_$i.json'));print('C5 distinct', round(d['ms_per_step'],3), round(d['value']/1e6,2), 'host blocked', round(d['host_blocked_on_gpu_ms_per_step'],3), d['per_rank']['gpu_step_ms_median'])"
done
timeout 300 python bench_other.py C5 2>> $O/err.txt | tee $O/other_c5.jsonl | cut -c1-260
tail -2 $O/err.txt
