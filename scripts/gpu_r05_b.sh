#!/bin/bash
# round 5, GPU call B: full suite with the folded activation gradients + typed TN, parity margins, fusion A/B, C5 traces
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
timeout 300 python scripts/parity_margins.py > $O/parity_margins.json 2> $O/parity_margins.err; echo "margins rc $?"
short="--steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-extras"
for v in "" "RELGNN_ACT_FUSION=0" "RELGNN_LIMB=pair" "RELGNN_LIMB=pair RELGNN_ACT_FUSION=0" ""; do
  env $v timeout 300 python bench.py $short > $O/b.json 2> $O/b.err; echo "[$v] rc $? $(python -c "
import json; d=json.load(open('$O/b.json')); print(round(d['ms_per_step'],4), round(d['final_loss'],5))")"
done
for t in limb panel; do
  RELGNN_TYPED_TN=$t timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_$t -o k -- python bench.py --config C5 --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-extras > $O/c5_$t.json 2> $O/c5_$t.err
  echo "C5 trace typed_tn=$t rc $?"
  cp $(find /tmp/c5_$t -name "*kernel_stats.csv" | head -1) $O/c5_${t}_kernel_stats.csv
  head -12 $O/c5_${t}_kernel_stats.csv | cut -c1-150
done
