#!/bin/bash
# round 5, GPU call C: suite after the relu-flag fix, backward margins, typed TN isolated
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
timeout 300 python scripts/parity_margins.py 2> $O/parity_margins.err | sed -n '/^{/,$p' > $O/parity_margins.json; echo "margins rc $?"
timeout 300 python scripts/bench_typed_tn.py > $O/typed_tn.jsonl 2> $O/typed_tn.err; echo "typed tn rc $?"; cat $O/typed_tn.jsonl
