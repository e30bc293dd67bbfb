#!/bin/bash
# Where the HOST time of a distinct-batch training step goes (cProfile of bench.py's timed loop; rank 0, one GPU).
export TMPDIR=/tmp
mkdir -p gpurun_out
python - <<'PY' 2>/dev/null
import cProfile, pstats, sys, io, runpy
sys.argv = ['bench.py', '--steps', '100', '--warmup', '12', '--no-roofline', '--no-cpu-baseline', '--no-extras']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
open('gpurun_out/host_profile.txt', 'w').write(out.getvalue())
PY
head -70 gpurun_out/host_profile.txt | cut -c1-180
