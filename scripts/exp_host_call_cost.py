"""Host time spent inside individual entry points during bench.py's loop (wall-clock accumulators around the Python
wrappers; no cProfile, so the loop runs at its normal speed)."""
import os, sys, time, runpy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_gnn_samples_amd import dense as D, ops as O, _lib
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[mod.__name__.split(".")[-1] + "." + name]; e[0] += 1; e[1] += time.perf_counter() - t
    setattr(mod, name, w)
for mod, names in ((D, ("lib_gemm", "tn_stream_gemm", "matmul_tn_splitk", "column_sum")), (O, ("_seg_reduce_raw",))):
    for n in names:
        wrap(mod, n)
lib = _lib.load_library()
real = lib.relgnn_blaslt_gemm_f32
def c_call(*a):
    t = time.perf_counter()
    try:
        return real(*a)
    finally:
        e = acc["C relgnn_blaslt_gemm_f32"]; e[0] += 1; e[1] += time.perf_counter() - t
lib.relgnn_blaslt_gemm_f32 = c_call
sys.argv = ['bench.py', '--steps', '100', '--warmup', '12', '--no-roofline', '--no-cpu-baseline', '--no-extras']
try:
    runpy.run_path(os.path.join(os.path.dirname(__file__), '..', 'bench.py'), run_name='__main__')
except SystemExit:
    pass
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-34s calls %6d  total %.3f s  per call %.1f us  per step %.3f ms" % (k, n, t, t / max(n, 1) * 1e6, t / 112 * 1e3), file=sys.stderr)
