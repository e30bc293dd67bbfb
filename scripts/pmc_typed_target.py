#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes for the typed K = 128 products (C5: 737 k gathered rows, 23 edge types, 512-row tiles)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
L, tiles, V = 23, 1440, 120000
P = tiles * 512
tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
node = torch.randint(0, V, (P,), generator=g).to(torch.int32).to(dev)
H = (torch.rand((V, 128), generator=g) * 2 - 1).to(dev)
W = ((torch.rand((L, 128, 128), generator=g) * 2 - 1) * 0.1).to(dev)
for _ in range(5):
    DN.limb_dense_sel(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
    DN.panel_gemm(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
torch.cuda.synchronize()
