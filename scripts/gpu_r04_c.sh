#!/bin/bash
# round 4, trip c: the LDS-tiled gather (csrc/slab_gather.hip) — bit-exactness tests, kernel A/B on the C2 batch, the C2 step with
# gather = l2 / lds; the fixed switch tests; pair vs triple after the per-edge-type scales
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_slab_gather.py tests/test_gpu_switches.py tests/test_gpu_limb_gemm.py tests/test_gpu_extreme_values.py tests/test_gpu_resident.py -q --tb=short -x 2>&1 | tail -40 > $O/tests.txt
tail -12 $O/tests.txt
timeout 300 python scripts/bench_slab_gather.py 2> $O/slab.err | tee $O/slab_gather.jsonl
tail -3 $O/slab.err
for i in 1 2; do
  for v in "pair l2" "pair lds" "triple l2" "triple lds"; do
    set -- $v
    RELGNN_LIMB=$1 RELGNN_GATHER=$2 timeout 300 python bench.py --steps 60 --warmup 12 --no-roofline --no-extras --no-cpu-baseline > $O/bench_$1_$2_$i.json 2>> $O/err.txt
    python -c "import json;d=json.load(open('$O/bench_$1_$2_$i.json'));print('$1 $2 run $i', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'])"
  done
done
tail -3 $O/err.txt
