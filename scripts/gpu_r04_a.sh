#!/bin/bash
# round 4, trip a: extreme-value / non-finite tests on every Dense route, flip-aware gradient parity (5 seeds x 3 routes), per-column
# scales of the two-limb weight gradient, the full -m gpu suite, 200-step trajectories per route, limb kernel timings, a bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/gpu_tests.txt
cp gpurun_out/gradient_parity_by_seed.json gpurun_out/limb16_tn_column_range.json gpurun_out/parity_*.json $O/ 2>/dev/null
timeout 300 python scripts/exp_trajectory_routes.py 200 > $O/trajectory_routes.json 2> $O/trajectory.err; tail -2 $O/trajectory.err
timeout 300 python scripts/bench_limb_gemm.py > $O/limb_gemm.jsonl 2> $O/limb_gemm.err; tail -4 $O/limb_gemm.jsonl | cut -c1-400
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench.json'))
print('bench', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['final_loss'], 'roof', d.get('roofline',{}).get('frac'))
for k in ('exact_fp32_gemm_route','bf16_triple_limb_route'): print(k, d.get(k,{}).get('ms_per_step'))
print('cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k!='sample'})
print('other', [(r.get('config'), r.get('train_ms')) for r in d.get('other_configs',{}).get('rows',[])])
"
tail -3 $O/bench.err
