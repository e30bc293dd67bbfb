#!/bin/bash
# round 4, trip b: the fixes of trip a's failures (ldexp unscale, tiny-value exactness bound, strided col_absmax), the switch tests,
# A/B of the C2 step pair / triple with A's column maxima taken next to the forward product, the LDS gather upper bound,
# the 8-rank launch-path rehearsal on one GPU (gloo), C3 through the captured step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_extreme_values.py tests/test_gpu_limb_gemm.py tests/test_gpu_switches.py tests/test_gpu_dp.py \
  tests/test_gpu_pair_tables.py tests/test_gpu_resident.py tests/test_gpu_streams_graphs.py tests/test_gpu_layers.py -q --tb=short 2>&1 | tail -60 > $O/tests.txt
tail -15 $O/tests.txt
for i in 1 2; do
  for v in pair triple; do
    RELGNN_LIMB=$v timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_${v}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'])"
  done
done
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result scripts/micro/lds_gather_rate.hip -o /tmp/lds_gather_rate 2>/dev/null && /tmp/lds_gather_rate | tee $O/lds_gather_rate.jsonl
timeout 300 python bench_other.py C3 2>> $O/err.txt | tee $O/other_c3.jsonl | cut -c1-600
RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 8 --steps 10 --warmup 3 --no-roofline --no-extras --no-cpu-baseline > $O/bench_8ranks_one_gpu_gloo.json 2> $O/bench8.err
python -c "
import json;d=json.load(open('$O/bench_8ranks_one_gpu_gloo.json'))
print('8 ranks C2', d['world_size'], d['backend'], round(d['ms_per_step'],3), d['per_rank']['host_blocked_on_gpu_ms_per_step'], d['per_rank']['allreduce_ms_mean'], d['peak_device_bytes'])"
tail -3 $O/bench8.err
RELGNN_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --config C5 --steps 6 --warmup 2 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5_8ranks_one_gpu_gloo.json 2> $O/bench8c5.err
python -c "
import json;d=json.load(open('$O/bench_c5_8ranks_one_gpu_gloo.json'))
print('8 ranks C5', d['world_size'], d['backend'], round(d['ms_per_step'],3), d['per_rank']['host_blocked_on_gpu_ms_per_step'], d['per_rank']['allreduce_ms_mean'], d['peak_device_bytes'], d['gradient_allreduce_bytes'])"
tail -3 $O/bench8c5.err
