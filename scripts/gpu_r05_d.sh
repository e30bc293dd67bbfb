#!/bin/bash
# round 5, GPU call D: targeted tests + switch A/B on the headline loop (gather warm-up, overlap, assembly stream)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_limb_gemm.py tests/test_gpu_pair_tables.py tests/test_gpu_reference_run.py tests/test_gpu_switches.py tests/test_gpu_streams_graphs.py tests/test_gpu_lib_gemm.py -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
short="--steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-extras"
for v in "" "RELGNN_GATHER_WARM=1" "" "RELGNN_GATHER_WARM=1" "RELGNN_BWD_OVERLAP=0" "RELGNN_BWD_OVERLAP=0 RELGNN_GATHER_WARM=1" "RELGNN_ASSEMBLE_STREAM=side" "RELGNN_LIMB=pair RELGNN_GATHER_WARM=1" "RELGNN_LIMB=pair"; do
  env $v timeout 300 python bench.py $short > $O/b.json 2> $O/b.err; echo "[$v] rc $? $(python -c "
import json; d=json.load(open('$O/b.json')); print(round(d['ms_per_step'],4), round(d['final_loss'],5), round(d['host_blocked_on_gpu_ms_per_step'],3))")"
done
RELGNN_GATHER_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/warm -o k -- python bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-extras > /dev/null 2> $O/warm_trace.err
cp $(find /tmp/warm -name "*kernel_stats.csv" | head -1) $O/warm_kernel_stats.csv; head -8 $O/warm_kernel_stats.csv | cut -c1-140; grep l2_warm $O/warm_kernel_stats.csv | cut -c1-200
