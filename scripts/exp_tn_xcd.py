#!/usr/bin/env python
"""The streaming weight-gradient kernel by the number of waves it is planned for (RELGNN_TN_WAVES) on C3's and C2's shapes; the
round-6 runs also compared a grid that keeps a chunk's tile groups on one XCD (RELGNN_TN_XCD, no longer in the kernel: no
difference) and found the plan launching 260 workgroups for [128, 640] (profiles/r06_tn_stream_plan.jsonl).  One process per
setting (the variable is read once)."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, str(ROOT))
    from tf_gnn_samples_amd import dense as DN
    dev = torch.device("cuda:0")
    def timed(fn, iters=40):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / iters * 1e3, 1)
    out = {}
    for V, M, N in ((49986, 128, 640), (49986, 128, 384), (49986, 128, 256), (49986, 128, 128), (32203, 256, 256), (32203, 256, 121),
                    (32203, 50, 256), (20000, 128, 640)):
        a = torch.rand((V, M), device=dev) * 2 - 1
        g = (torch.rand((V, N), device=dev) * 2 - 1) * 0.05
        out["%dx%dx%d" % (V, M, N)] = timed(lambda: DN.tn_stream_gemm(a, g))
    print(json.dumps(out))
else:
    for waves in ("768", "1024", "1536", "2048"):
        if True:
            env = dict(os.environ, RELGNN_TN_WAVES=waves)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(json.dumps({"waves": waves, "us": json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-300:]}), flush=True)
