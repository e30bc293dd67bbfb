#!/bin/bash
# round 6, call A: the printed line as the driver runs it (size + sidecar), bench contract tests live
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_a; rm -rf $O; mkdir -p $O; cd $R
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err >$O/bench.json ) 2>&1 | tail -3
cp bench_detail.json $O/bench_detail.json
wc -c $O/bench.json
timeout 1500 python -m pytest tests/test_bench_contract.py -m gpu -q -x > $O/contract_tests.txt 2>&1; tail -3 $O/contract_tests.txt
cut -c1-3000 $O/bench.json
tail -5 $O/bench.err
