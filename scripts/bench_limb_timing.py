#!/usr/bin/env python
"""Per-wave cycle totals of the limb kernel's k-loop segments (library built with -DRELGNN_LIMB_TIMING: s_memtime stamps around the
split work, the MFMA stretch, the DMA wait, the LDS wait, the barrier and the last row tile), [40960, 768] x [256, 768]^T."""
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
M, N, K = 256 * 160, 256, 768
a = torch.rand((M, K), device=dev) * 2 - 1
w = DN.limb_split((torch.rand((N, K), device=dev) * 2 - 1) * 0.1)
al = DN.limb_split(a)
for name, fn in (("xf32", lambda: DN.limb_gemm_xf32(a, w)), ("pre-split", lambda: DN.limb_gemm(al, w))):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    t = buf.cpu().double()                              # [wg, wave, seg]
    nt = t[0, 0, 7].item()
    per = t[:, :, :7].mean(0) / max(nt - 1, 1)          # cycles (s_memtime ticks) per k-tile, per wave
    per[:, 0] *= 2; per[:, 6] *= 2                      # split work: per EVEN / per ODD k-tile
    print(name, "k-tiles", nt)
    print("  wave   split(even)  mfma-stretch  wait-dma  wait-lds  barrier  last-tile  split(odd)")
    for wv in range(8):
        r = per[wv].tolist()
        print("  %d   " % wv + "  ".join("%8.0f" % x for x in r))
