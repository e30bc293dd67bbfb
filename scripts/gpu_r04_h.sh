#!/bin/bash
# round 4, trip h: joins of the weight-gradient side stream deferred behind the whole backward: tests + A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_streams_graphs.py tests/test_gpu_dp.py tests/test_gpu_switches.py tests/test_gpu_train_utils.py tests/test_gpu_memory.py -q --tb=short -x 2>&1 | tail -5
for i in 1 2 3; do
  for v in layer deferred; do
    RELGNN_TMP_JOIN=$v timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${v}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'])"
  done
done
tail -2 $O/err.txt
