#!/usr/bin/env python
"""Where the waves of relgnn_rgcn_fused_fwd spend their cycles (library variant built by scripts/build_timing_variant.sh rgcn_fused
RELGNN_FUSED_TIMING: s_memtime stamps).  Per role (matrix waves 0-7, gather waves 8-15), mean over the workgroups of the C2 batch:
total, time inside polls, polls that had to wait, and for the gather waves the row-load wait / fold / prepare / issue segments."""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from tf_gnn_samples_amd import _lib
_lib.LIB_PATH = ROOT / "tf_gnn_samples_amd" / "build" / "librelgnn_rgcn_fused_timing.so"
from tf_gnn_samples_amd import ops                                      # noqa: E402
sys.argv = [sys.argv[0]]
import importlib.util                                                   # noqa: E402
spec = importlib.util.spec_from_file_location("bench_rgcn_fused", ROOT / "scripts" / "bench_rgcn_fused.py")

dev = torch.device("cuda:0")
lib = _lib.load_library()
lib.relgnn_rgcn_fused_timing_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((256, 16, 8), dtype=torch.int64, device=dev)
lib.relgnn_rgcn_fused_timing_buffer(buf.data_ptr())
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)                                            # runs the bit-identity check and the timings once
graph, w, H, kernels = mod.graph, mod.w, mod.H, mod.kernels
for sums in (True, False):
    for _ in range(5):
        ops._rgcn_fused(H, graph, w, kernels, True, sums)
    torch.cuda.synchronize()
    t = buf.cpu().double()                                              # [workgroup, wave, slot]
    names = ["total", "in polls", "polls that waited", "row-load wait", "fold", "prepare", "issue | k-loop", "-"]
    print("bucket sums stored: %s   (s_memtime ticks; mean / max over workgroups)" % sums)
    for role, sl in (("matrix waves", slice(0, 8)), ("gather waves", slice(8, 16))):
        r = t[:, sl, :].reshape(-1, 8)
        print("  %-13s" % role + "  ".join("%s %.0f / %.0f" % (n, r[:, i].mean(), r[:, i].max()) for i, n in enumerate(names[:7])))
    tot = t[:, :, 0].max(dim=1).values
    print("  slowest wave per workgroup: min %.0f  mean %.0f  max %.0f" % (tot.min(), tot.mean(), tot.max()))
