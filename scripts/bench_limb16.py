#!/usr/bin/env python
"""The two-fp16-limb product (relgnn_limb16_gemm_xf32) next to the bf16 triple and the exact-fp32 library GEMM on the C2 layer
shapes: time and error against float64.  Row magnitudes from torch here (the path gets them from the gather's epilogue)."""
import ctypes, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import _lib, dense as DN
dev = torch.device("cuda:0")
lib = _lib.load_library()
gen = torch.Generator(device="cpu").manual_seed(0)


def image16(mats, transpose):
    """one image of the matrices laid side by side along k; returns (buf, wmax)"""
    n = len(mats)
    if transpose:      # NN: w_l [K_l, N] -> B [N, sum K_l]
        N, K = mats[0].shape[1], sum(m.shape[0] for m in mats)
    else:              # NT: w_l [N, K_l]
        N, K = mats[0].shape[0], sum(m.shape[1] for m in mats)
    buf = torch.empty(int(lib.relgnn_limb16_elements(N, K)), dtype=torch.float16, device=dev)
    wmax = torch.zeros(1, dtype=torch.float32, device=dev)
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    kt, offs = 0, []
    for m in mats:
        offs.append(kt); kt += (m.shape[0] if transpose else m.shape[1]) // 16
    _lib.check(lib.relgnn_limb16_split_multi_f32(n, vp(*[m.data_ptr() for m in mats]), i64(*[m.stride(0) for m in mats]),
                                                 i32(*[m.shape[0] for m in mats]), i32(*[m.shape[1] for m in mats]),
                                                 i32(*[1 if transpose else 0] * n), vp(*[buf.data_ptr()] * n), i32(*offs), i32(*[kt] * n),
                                                 wmax.data_ptr(), _lib.current_stream()), "limb16 split")
    return buf, wmax, N, K


def gemm16(x, xmax, groups, buf, wmax, N, K, act=0, bias=None):
    out = torch.empty((x.shape[0], N), dtype=torch.float32, device=dev)
    _lib.check(lib.relgnn_limb16_gemm_xf32(act, x.data_ptr(), x.stride(0), xmax.data_ptr(), groups, buf.data_ptr(), wmax.data_ptr(),
                                           _lib.ptr(bias), _lib.ptr(DN._zeros(dev)), out.data_ptr(), out.stride(0), x.shape[0], N, K,
                                           _lib.current_stream()), "limb16 gemm")
    return out


def timed(fn, reps=9, inner=10):
    for _ in range(200):
        fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]


def case(name, x, mats, transpose, groups):
    buf, wmax, N, K = image16(mats, transpose)
    xmax = x.abs().view(x.shape[0], groups, -1).amax(2).contiguous()
    out = gemm16(x, xmax, groups, buf, wmax, N, K)
    Wfull = torch.cat(mats, 0) if transpose else torch.cat([m.t() for m in mats], 0)
    truth = x.double() @ Wfull.double()
    ref3 = DN.limb_gemm_weight(x, mats, DN.WEIGHT_NN if transpose else DN.WEIGHT_NT)
    f32 = x @ Wfull
    row = truth.abs().amax(1, keepdim=True).clamp(min=1e-300)
    rec = {"case": name, "M": x.shape[0], "N": N, "K": K}
    for tag, v in (("fp16x2", out), ("bf16x3", ref3), ("fp32_lib", f32)):
        e = (v.double() - truth).abs()
        rec[tag + "_max_abs_err"] = float(e.max()); rec[tag + "_worst_row_rel"] = float((e.amax(1, keepdim=True) / row).max())
    rec["max_abs_out"] = float(truth.abs().max())
    rec["fp16x2_us"] = round(timed(lambda: gemm16(x, xmax, groups, buf, wmax, N, K)), 1)
    rec["bf16x3_us"] = round(timed(lambda: DN.limb_gemm_weight(x, mats, DN.WEIGHT_NN if transpose else DN.WEIGHT_NT)), 1)
    print(json.dumps(rec), flush=True)


V = 36096
Ws = [((torch.rand((256, 256), generator=gen) * 2 - 1) * 0.108).to(dev) for _ in range(3)]
act = (torch.relu(torch.randn((V, 768), generator=gen)) * torch.distributions.Gamma(2.0, 0.125).sample((V, 1))).to(dev)
case("fwd: aggregated post-ReLU activations", act, Ws, True, 3)
g = (torch.distributions.StudentT(3.0).sample((V, 768)) * 1e-5 * torch.exp(torch.empty(V, 1).uniform_(-4.6, 4.6))).to(dev)
case("dA: gradients, rows over four decades", g, Ws, False, 3)
mix = act.clone(); mix[:, ::2] *= 1e-6; mix[::7] = 0
case("fwd: rows mixing O(1) and O(1e-6), zero rows", mix, Ws, True, 3)
