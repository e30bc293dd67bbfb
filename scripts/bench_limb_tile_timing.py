#!/usr/bin/env python
"""Per-wave s_memtime totals of limb_gemm_tile_kernel's segments (library built with -DRELGNN_LIMB_TIMING) on the C5 typed forward."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import _lib, dense as DN
dev = torch.device("cuda:0")
lib = ctypes.CDLL(str(_lib.LIB_PATH))
buf = torch.zeros((64, 8, 8), dtype=torch.int64, device=dev)
lib.relgnn_limb_timing_buffer.argtypes = [ctypes.c_void_p]
lib.relgnn_limb_timing_buffer(buf.data_ptr())
g = torch.Generator(device="cpu").manual_seed(0)
L, tiles, V = 23, 1440, 120000
P = tiles * 512
tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
node = torch.randint(0, V, (P,), generator=g).to(torch.int32).to(dev)
H = (torch.rand((V, 128), generator=g) * 2 - 1).to(dev)
W = ((torch.rand((L, 128, 128), generator=g) * 2 - 1) * 0.1).to(dev)
for _ in range(50):
    DN.limb_dense_sel(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
torch.cuda.synchronize()
t = buf.cpu().double()
m = t.mean(0)
print("s_memtime ticks (100 MHz: 1 tick = 10 ns), mean over 64 workgroups; per super-tile columns are totals / 16")
print("wave  prologue   split  loads+mfma  store_panel(total/4)  lgkm-wait  barrier   total   S")
for wv in range(8):
    r = m[wv].tolist()
    print("  %d  %8.0f %8.1f %8.1f %10.1f %12.1f %8.1f %8.0f %4.0f" % (wv, r[0], r[1] / (4 * r[7]), r[2] / (4 * r[7]), r[3] / r[7], r[4] / (4 * r[7]), r[5] / (4 * r[7]), r[6], r[7]))
