#!/usr/bin/env python
"""Where the waves of relgnn_limb_gemm_sel_pc_xf32 spend their cycles (library variant built by
`scripts/build_timing_variant.sh limb_gemm_pc_typed RELGNN_PCT_TIMING`: s_memtime stamps), per role, mean over the workgroups, on the
four typed products of a C5-sized table."""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from tf_gnn_samples_amd import _lib
_lib.LIB_PATH = ROOT / "tf_gnn_samples_amd" / "build" / "librelgnn_limb_gemm_pc_typed_timing.so"
from tf_gnn_samples_amd import config, dense as DN                     # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load_library()
lib.relgnn_pct_timing_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((256, 16, 8), dtype=torch.int64, device=dev)
lib.relgnn_pct_timing_buffer(buf.data_ptr())
g = torch.Generator(device="cpu").manual_seed(0)
L, tiles, V = 23, 1440, 100000
P = tiles * 512
tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
sorted_ids = len(sys.argv) > 1 and sys.argv[1] == "sorted"
node = torch.randint(0, V, (P,), generator=g)
if sorted_ids:                                                          # ascending inside a tile run, like a real pair table
    node = torch.sort(node.view(tiles, 512), dim=1).values.view(-1)
node = node.to(torch.int32).to(dev)
H = (torch.rand((V, 128), generator=g) * 2 - 1).to(dev)
names_m = ["total", "in polls", "polls that waited", "k-loops", "stores"]
names_p = ["total", "in polls", "polls that waited", "row wait", "split+write(+poll)", "issue"]
for Dout in (128, 256):
    Ws = [((torch.rand((128, Dout), generator=g) * 2 - 1) * 0.1).to(dev) for _ in range(L)]
    gY = (torch.rand((P, Dout), generator=g) * 2 - 1).to(dev)
    for what, layout, a, args in (("forward N=%d" % Dout, DN.GEMM_NN, H, dict(a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)),
                                  ("input gradient K=%d" % Dout, DN.GEMM_NT, gY, dict(b_select=tile_type, rows_per_select=512))):
        with config.override(typed_pc="1"):
            im = DN.sel_image(Ws, layout)
            for _ in range(3):
                DN.limb_dense_sel(layout, a, Ws, image=im, **args)
            torch.cuda.synchronize()
        t = buf.cpu().double()
        print(what, "(s_memtime ticks of 10 ns; mean over workgroups)" + (" sorted ids" if sorted_ids else ""))
        for role, sl, names in (("matrix waves", slice(0, 8), names_m), ("producer waves", slice(8, 16), names_p)):
            r = t[:, sl, :].reshape(-1, 8)
            print("  %-15s" % role + "  ".join("%s %.0f" % (n, r[:, i].mean()) for i, n in enumerate(names)))
    del gY
