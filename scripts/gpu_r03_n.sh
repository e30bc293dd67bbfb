#!/bin/bash
# round 3, trip n: the gradient all-reduce in buckets during the backward (RELGNN_ALLREDUCE=overlap): the 2-rank HIP-path test,
# then two ranks sharing ONE MI355X over gloo (a launch-path check with a slow host-side collective: what the overlap hides of it)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -3
for v in flat overlap flat overlap; do
  RELGNN_ALLREDUCE=$v RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-extras > $O/c2_$v.json 2>> $O/err.txt
  python -c "import json;d=json.load(open('$O/c2_$v.json'));print('C2 2 ranks one GPU gloo', '$v', round(d['ms_per_step'],3), d['per_rank']['allreduce_ms_mean'], d['gradient_allreduce'])"
done
for v in flat overlap; do
  RELGNN_ALLREDUCE=$v RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --config C5 --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-extras > $O/c5_$v.json 2>> $O/err.txt
  python -c "import json;d=json.load(open('$O/c5_$v.json'));print('C5 2 ranks one GPU gloo', '$v', round(d['ms_per_step'],3), d['per_rank']['allreduce_ms_mean'], d['gradient_allreduce'])"
done
tail -3 $O/err.txt
