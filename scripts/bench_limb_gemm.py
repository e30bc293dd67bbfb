#!/usr/bin/env python
"""relgnn_limb_gemm_f32 (fp32 operands as three bf16 limbs, six bf16 MFMA products) against the exact-fp32 library GEMM on the
C2 layer shapes: time of the product alone, of the limb split of the activation operand, and the error of both against float64."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import config, dense as DN
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def exact_lib_gemm(layout, a, b):
    """the exact-fp32 library GEMM whatever RELGNN_GEMM says"""
    with config.override(gemm="lib"):
        return DN.lib_gemm(layout, a, b)


def exact_tn(a, b):
    with config.override(gemm="lib"):
        return DN.matmul_tn_splitk(a, b)


def timed(fn, reps=7, inner=10):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


shapes = [("fwd  A[V,768] @ W[768,256]", 36096, 256, 768), ("dA   G[V,256] @ W^T -> [V,768]", 36096, 768, 256),
          ("fwd  V=40111", 40111, 256, 768), ("fwd  V=32203", 32203, 256, 768), ("units u=5 exact", 256 * 160, 256, 768),
          ("units u=4 exact", 256 * 128, 256, 768), ("K=256 N=256", 36096, 256, 256)]
for name, M, N, K in shapes:
    a = torch.rand((M, K), device=dev, generator=gen) * 2 - 1
    w = (torch.rand((N, K), device=dev, generator=gen) * 2 - 1) * 0.1          # B as [N, K]
    al, wl = DN.limb_split(a), DN.limb_split(w)
    assert torch.equal(al.to_float64(), a.double()), "limbs do not add up to the fp32 value"
    out = DN.limb_gemm(al, wl)
    ref32 = exact_lib_gemm(DN.GEMM_NT, a, w)
    rows = slice(0, 4096)
    truth = a[rows].double() @ w.double().t()
    e_limb = float((out[rows].double() - truth).abs().max())
    e_f32 = float((ref32[rows].double() - truth).abs().max())
    out_x = DN.limb_gemm_xf32(a, wl)
    assert torch.equal(out_x, out), "in-kernel split differs from the pre-split product"
    t_limb = timed(lambda: DN.limb_gemm(al, wl, out=out))
    t_xf32 = timed(lambda: DN.limb_gemm_xf32(a, wl, out=out))
    t_split = timed(lambda: DN.limb_split(a, out=al))
    t_lib = timed(lambda: exact_lib_gemm(DN.GEMM_NT, a, w))
    t_panel = timed(lambda: DN.panel_gemm(DN.GEMM_NT, a, w))
    fl = 2.0 * M * N * K
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "limb_xf32_us": round(t_xf32, 1), "limb_us": round(t_limb, 1), "split_a_us": round(t_split, 1),
                      "lib_f32_us": round(t_lib, 1), "panel_f32_us": round(t_panel, 1),
                      "xf32_TFLOPs_fp32_equiv": round(fl / t_xf32 / 1e6, 1), "limb_TFLOPs_fp32_equiv": round(fl / t_limb / 1e6, 1),
                      "lib_TFLOPs": round(fl / t_lib / 1e6, 1), "max_abs_out": round(float(truth.abs().max()), 3),
                      "err_limb_vs_f64": e_limb, "err_f32_vs_f64": e_f32,
                      "max_abs_limb_minus_f32": float((out - ref32).abs().max())}), flush=True)

for name, V, J, C in [("dW   A[V,768]^T @ G[V,256]", 36096, 768, 256), ("dW   V=40111", 40111, 768, 256), ("dW   [V,256]^T @ [V,256]", 36096, 256, 256)]:
    a = torch.rand((V, J), device=dev, generator=gen) * 2 - 1
    g = (torch.rand((V, C), device=dev, generator=gen) * 2 - 1) * 0.05
    out = DN.limb_gemm_tn(a, g)
    ref = exact_tn(a, g)
    truth = a.double().t() @ g.double()
    t_limb = timed(lambda: DN.limb_gemm_tn(a, g))
    t_lib = timed(lambda: exact_tn(a, g))
    # two fp16 limbs: one scale per column of each operand (two column-maximum passes in front) / one per operand (round 3's form)
    ca, cg = DN.col_absmax(a), DN.col_absmax(g)
    t_pair_cols = timed(lambda: DN.limb_gemm_tn(a, g, ca, cg))
    t_colmax = timed(lambda: (DN.col_absmax(a), DN.col_absmax(g)))
    ma, mg = DN.absmax(a), DN.absmax(g)
    t_pair_op = timed(lambda: DN.limb_gemm_tn(a, g, ma, mg))
    t_absmax = timed(lambda: (DN.absmax(a), DN.absmax(g)))
    pair = DN.limb_gemm_tn(a, g, ca, cg)
    print(json.dumps({"shape": name, "V": V, "J": J, "C": C, "limb_tn_us": round(t_limb, 1), "f32_route_us": round(t_lib, 1),
                      "pair_column_scales_us": round(t_pair_cols, 1), "col_absmax_both_us": round(t_colmax, 1),
                      "pair_operand_scale_us": round(t_pair_op, 1), "absmax_both_us": round(t_absmax, 1),
                      "err_pair_columns_vs_f64": float((pair.double() - truth).abs().max()),
                      "max_abs_out": round(float(truth.abs().max()), 3), "err_limb_vs_f64": float((out.double() - truth).abs().max()),
                      "err_f32_vs_f64": float((ref.double() - truth).abs().max())}), flush=True)
