#!/usr/bin/env python
"""relgnn_panel_gemm_f32 against the library GEMMs (hipBLASLt through relgnn_blaslt_gemm_f32 / torch.bmm) on the shapes of the path, interleaved rounds in one process, random operands:
  dense   the C2 step's node-side products (aggregate-first order): [V, 768] @ [768, 256] forward and input gradient, the
          inter-layer Dense, the transform-first shapes, the [768, 256] weight gradient as K-split slabs
  typed   the C5 (GNN-FiLM, VarMisuse-shaped) per-(node, type) transforms: gathered rows x per-tile kernels, K = 128
One JSON line per shape."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch

from tf_gnn_samples_amd import dense as DN

NN, NT, TN = 0, 1, 2
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape):
    return torch.rand(shape, device=dev, generator=gen) * 2 - 1


def time_variants(variants, rounds=7, inner=5):
    """variants: {name: fn}.  Interleaved: every round runs every variant `inner` times between two events."""
    for fn in variants.values():
        fn(); fn()
    torch.cuda.synchronize()
    times = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                fn()
            b.record()
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / inner * 1e3)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in times.items()}


def report(what, flops, res, extra=None):
    row = {"what": what, "gflop": flops / 1e9}
    for k, (med, mn) in res.items():
        row[k + "_us"] = round(med, 1)
        row[k + "_TFLOPs"] = round(flops / (med * 1e-6) / 1e12, 1)
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)


def dense_shapes():
    for V in (32203, 36096, 40111):
        for (K, N, layout) in ((768, 256, NN), (256, 256, NN), (256, 768, NN), (768, 256, NT), (256, 768, NT)):
            a = rnd(V, K)
            b = rnd(K, N) * 0.1 if layout == NN else rnd(N, K) * 0.1
            v = {"panel": lambda: DN.panel_gemm(layout, a, b), "lib": lambda: DN.lib_gemm(layout, a, b)}
            res = time_variants(v)
            ref = (a.double() @ b.double()) if layout == NN else (a.double() @ b.double().t())
            err = float((DN.panel_gemm(layout, a, b).double() - ref).abs().max())
            report("%s [%d,%d]x[%d,%d]" % ("NN" if layout == NN else "NT", V, K, K, N), 2.0 * V * K * N, res, {"max_err": err})
        # weight gradient [768, 256] = agg^T @ gout, reduction over V
        a, g = rnd(V, 768), rnd(V, 256)
        for splits in (21, 42):
            chunk = ((V + splits - 1) // splits + 15) // 16 * 16
            nb = (V + chunk - 1) // chunk

            def panel_tn():
                slabs = DN.panel_gemm(TN, a, g, batch=nb, split_k_rows=chunk, dims=(768, 256, V))
                return slabs.sum(0) if nb > 1 else slabs
            res = time_variants({"panel_splitk_plus_sum": panel_tn, "lib_splitk": lambda: DN.matmul_tn_splitk(a, g)})
            err = float((panel_tn().double() - a.double().t() @ g.double()).abs().max())
            report("TN [%d,768]^T x [%d,256] %d slabs" % (V, V, nb), 2.0 * V * 768 * 256, res, {"max_err": err})


def typed_shapes():
    V, Din, L = 100000, 128, 23
    H = rnd(V, Din)
    for tiles in (1440,):
        P = tiles * 512
        node = torch.randint(0, V, (P,), device=dev, generator=gen, dtype=torch.int32)
        tile_type = torch.sort(torch.randint(0, L, (tiles,), device=dev, generator=gen, dtype=torch.int32)).values
        for Dout in (128, 256):
            W = rnd(L, Din, Dout) * 0.2

            def bmm_fwd():
                X = H.index_select(0, node.long())
                Wt = W.index_select(0, tile_type.long())
                return torch.bmm(X.view(-1, 512, Din), Wt).view(P, Dout)
            res = time_variants({"panel": lambda: DN.panel_gemm(NN, H, W, a_rows=node, num_rows=P, b_select=tile_type,
                                                                rows_per_select=512), "torch_bmm": bmm_fwd}, rounds=5, inner=3)
            report("typed fwd gather[%d rows] x W_type [%d,%d]" % (P, Din, Dout), 2.0 * P * Din * Dout, res)
            dY = rnd(P, Dout)
            Wt = W.index_select(0, tile_type.long())
            res = time_variants({"panel": lambda: DN.panel_gemm(NT, dY, W, b_select=tile_type, rows_per_select=512, dims=(P, Din, Dout)),
                                 "torch_bmm": lambda: torch.bmm(dY.view(-1, 512, Dout), Wt.transpose(1, 2))}, rounds=5, inner=3)
            report("typed dX [%d,%d] x W_type^T" % (P, Dout), 2.0 * P * Din * Dout, res)
            X = H.index_select(0, node.long())
            res = time_variants({"panel": lambda: DN.panel_gemm(TN, H, dY, a_rows=node, batch=tiles,
                                                                strides=(0, 512 * Dout, Din * Dout), dims=(Din, Dout, 512)),
                                 "torch_bmm": lambda: torch.bmm(X.view(-1, 512, Din).transpose(1, 2), dY.view(-1, 512, Dout))},
                                rounds=5, inner=3)
            report("typed dW partials %d x [%d,512]x[512,%d]" % (tiles, Din, Dout), 2.0 * P * Din * Dout, res)


def pmc_target():
    """A few launches of the panel kernel and of the library GEMM on the C2 layer shape (run under rocprofv3 --pmc)."""
    V, K, N = 36096, 768, 256
    a, b = rnd(V, K), rnd(K, N) * 0.1
    for _ in range(6):
        DN.panel_gemm(NN, a, b)
        DN.lib_gemm(NN, a, b)
    torch.cuda.synchronize()


if __name__ == "__main__":
    which = sys.argv[1:] or ["dense", "typed"]
    if "pmc" in which:
        pmc_target()
    if "dense" in which:
        dense_shapes()
    if "typed" in which:
        typed_shapes()
