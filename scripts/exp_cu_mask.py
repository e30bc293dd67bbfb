#!/usr/bin/env python
"""Do the kernels of the C2 backward need every CU?  (a fixed CU partition for the weight-gradient stream.)

Part 1: each kernel of one aggregate-first RGCN layer (C2 batch, V = 32 k, 1.85 M messages) ALONE on a stream created with
hipExtStreamCreateWithCUMask over the first n CU bits (the mask is interleaved over the 8 XCDs: n / 8 CUs per XCD), n from 256 down.
Part 2: one layer's backward as the step runs it — by-source gather + input-gradient product on the main stream, weight gradient
(three-limb TN + slab sum) on the side stream — with the main stream on 256 - S CUs and the side stream on S, for several S, against
both streams unmasked (today).  HIP events on the streams; one JSON line per measurement."""
import ctypes, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tf_gnn_samples_amd import _lib, config, dense as DN, ops
from tf_gnn_samples_amd.graph import RelGraph
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(first, count):
    """Stream over `count` CUs starting at CU `first` OF EVERY XCD: measured on MI355X (this script's first version), the mask is
    XCD-major — word x of the 8 x 32-bit mask is XCD x's 32 CUs (dropping the last word costs the gather 8 % and a persistent
    one-workgroup-per-CU product 50 %: a whole XCD gone) — so a partition that leaves every XCD's L2 to both sides takes the same
    bit range in each word.  first / count are per-XCD here: 0 <= first, first + count <= 32."""
    words = (ctypes.c_uint32 * 8)()
    for x in range(8):
        for i in range(first, first + count):
            words[x] |= 1 << i
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    got = (ctypes.c_uint32 * 8)()
    hip.hipExtStreamGetCUMask(st, 8, got)
    return torch.cuda.ExternalStream(st.value, device=dev), hex(got[0])


def timed(fn, stream, reps=5, inner=10):
    with torch.cuda.stream(stream):
        for _ in range(5):
            fn()
        stream.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(inner):
                fn()
            e1.record(stream); stream.synchronize()
            ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(16, 1, seed=0)
mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
batch = DeviceBatch(mb, dev)
g = RelGraph(batch.adjacency_lists, mb.num_nodes)
V, L, M, D = g.V, g.L, g.M, 256
w = g.degree_scale(batch.type_to_num_incoming_edges)
gen = torch.Generator(device=dev).manual_seed(0)
H = torch.rand((V, D), device=dev, generator=gen) * 2 - 1
kernels = [(torch.rand((D, D), device=dev, generator=gen) * 2 - 1) * 0.08 for _ in range(L)]
plan = g.plan_transformed(w)
gout = (torch.rand((V, D), device=dev, generator=gen) * 2 - 1) * 0.05
agg = torch.rand((V, L * D), device=dev, generator=gen) * 2 - 1
gT = torch.rand((V, L * D), device=dev, generator=gen) * 0.05
ops.handover_word(dev)


def gather_fwd():          # rows of the [V, 256] state table into the V*L (target, type) buckets
    return ops._seg_reduce_raw(_lib.AGG_SUM, H, g.rowptr_t, 1, g.src_t, w, V * L)


def gather_bwd():          # by source: rows of the [V, 256] output gradient into the V*L (source, type) buckets
    return ops._seg_reduce_raw(_lib.AGG_SUM, gout, plan.rowptr_b, plan.stride_b, plan.col_b, plan.w_bwd(_lib.AGG_SUM), plan.num_rows_x)


def product_fwd():         # [V, 768] @ [768, 256], ReLU: the wave-role kernel
    return DN.grouped_nn_gemm(agg, kernels, relu=True)


def product_nt():          # input gradient [V, 768] @ [768, 256] (W_l^T): limb_gemm_kernel
    with config.override(limb_pc="fwd"):
        return DN.grouped_nt_gemm(gT, kernels)


def product_nt_pc():       # the same on the wave-role kernel
    with config.override(limb_pc="1"):
        return DN.grouped_nt_gemm(gT, kernels)


def weight_grad():         # [V, 768]^T @ [V, 256]: three-limb TN + slab sum
    return DN.matmul_tn_splitk(agg, gout)


cands = {"gather_fwd": gather_fwd, "gather_bwd": gather_bwd, "product_fwd_pc": product_fwd, "product_nt": product_nt,
         "product_nt_pc": product_nt_pc, "weight_grad_tn": weight_grad}

null = torch.cuda.current_stream(dev)
row = {"part": 1, "cus": "unmasked (torch stream)"}
for name, fn in cands.items():
    row[name + "_us"] = timed(fn, torch.cuda.Stream(device=dev))
print(json.dumps(row), flush=True)
for n in (32, 30, 28, 26, 24, 22, 20, 16, 12, 8, 6, 4):
    st, got = masked_stream(0, n)
    row = {"part": 1, "cus_per_xcd": n, "cus": 8 * n, "mask_per_xcd": got}
    for name, fn in cands.items():
        row[name + "_us"] = timed(fn, st)
    print(json.dumps(row), flush=True)


# ---- part 2: the layer's backward on two streams ---------------------------------------------------------------------------
def layer_backward(main, side, reps=5, inner=8):
    def once():
        # what ops._AggregateThenTransform.backward enqueues: weight gradient aside, input-gradient product + by-source gather here
        side.wait_stream(main)
        with torch.cuda.stream(side):
            weight_grad()
        with torch.cuda.stream(main):
            gather_bwd()
            product_nt()
        main.wait_stream(side)

    for _ in range(4):
        once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(inner):
            once()
        e1.record(main); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


print(json.dumps({"part": 2, "split": "both unmasked", "layer_backward_us": layer_backward(torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))}), flush=True)
one = torch.cuda.Stream(device=dev)
print(json.dumps({"part": 2, "split": "one stream (no overlap)", "layer_backward_us": layer_backward(one, one)}), flush=True)
for S in (4, 6, 8, 10, 12, 14, 16):
    side, _ = masked_stream(0, S)
    main, _ = masked_stream(S, 32 - S)
    print(json.dumps({"part": 2, "split": "side %d / main %d" % (8 * S, 256 - 8 * S), "layer_backward_us": layer_backward(main, side)}), flush=True)
    side2, _ = masked_stream(0, S)
    print(json.dumps({"part": 2, "split": "side %d / main unmasked" % (8 * S),
                      "layer_backward_us": layer_backward(torch.cuda.Stream(device=dev), side2)}), flush=True)
