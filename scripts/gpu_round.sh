#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprofv3 kernel trace.  Run through gpurun.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx950|Compute Unit" | head -6
echo "== nproc: $(nproc)"
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
if [ "${PROFILE:-1}" = "1" ]; then
  echo "== rocprofv3 kernel trace"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof -type f | head -20; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
fi
