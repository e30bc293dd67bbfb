#!/usr/bin/env python
"""Where the waves of relgnn_gru_cell_fwd_xf32 spend their cycles (library variant built by
`scripts/build_timing_variant.sh gru_cell RELGNN_GRU_TIMING`: s_memtime stamps), per role, mean over the workgroups."""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from tf_gnn_samples_amd import _lib
_lib.LIB_PATH = ROOT / "tf_gnn_samples_amd" / "build" / "librelgnn_gru_cell_timing.so"
from tf_gnn_samples_amd import ops, utils                              # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load_library()
lib.relgnn_gru_timing_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((256, 16, 8), dtype=torch.int64, device=dev)
lib.relgnn_gru_timing_buffer(buf.data_ptr())
ops.handover_word(dev)
U = 128
g = torch.Generator(device="cpu").manual_seed(0)
K = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
R = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
b = torch.zeros(3 * U, device=dev).requires_grad_(True)
names = {"z waves": ["total", "in polls", "polls that waited", "k-loop halves", "gate epilogue", "blend epilogue"],
         "r waves": ["total", "in polls", "polls that waited", "k-loop halves", "gate epilogue + r*h write"],
         "producers": ["total", "in polls", "polls that waited", "-", "-", "row wait", "split + write (+ poll)"]}
for V in (49986, 200000):
    x = (torch.rand((V, U), generator=g) * 2 - 1).to(dev)
    h = (torch.rand((V, U), generator=g) * 2 - 1).to(dev)
    for _ in range(3):
        utils._GRUCellFn.apply(x, h, K, R, b, 1)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    print("nodes", V, "(s_memtime ticks, ~2 per ns; mean over workgroups and the role's waves)")
    for role, sl in (("z waves", slice(0, 4)), ("r waves", slice(4, 8)), ("producers", slice(8, 16))):
        r = t[:, sl, :].reshape(-1, 8)
        print("  %-10s" % role + "  ".join("%s %.0f" % (n, r[:, i].mean()) for i, n in enumerate(names[role]) if n != "-"))
