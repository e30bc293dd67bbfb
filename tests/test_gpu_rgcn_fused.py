"""relgnn_rgcn_fused_fwd (csrc/rgcn_fused.hip): the aggregate-first RGCN layer (gnns/rgcn.py:84-114) with gather waves feeding the
matrix waves of the same workgroup through LDS.

The bar is BIT identity with the two-kernel route it replaces — relgnn_seg_reduce_fwd (sequential fp32 fold per (target, type)
bucket) followed by relgnn_limb_gemm_xf32 (three bf16 limbs, six products) — for the output AND for the bucket sums it stores for
the weight gradient, on every panel geometry (1 .. many 32-row units per workgroup, odd unit counts, rows % 32 != 0), with empty
buckets, empty edge types, long buckets, unit weights, and values the limb split has to saturate.  The two-kernel route is tied to
the oracle and the reference-run fixtures by tests/test_gpu_reference_run.py and tests/test_gpu_baseline_size.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D = 256


def _graph(dev, V, edge_counts, seed, hub=None):
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(seed)
    adj = []
    for l, e in enumerate(edge_counts):
        src = rng.integers(0, V, e)
        tgt = rng.integers(0, V, e)
        if hub is not None and l == hub[0] and e:
            tgt[: min(hub[2], e)] = hub[1]                 # one long bucket
        adj.append(torch.as_tensor(np.stack([src, tgt], axis=1).astype(np.int32), device=dev))
    return RelGraph(adj, V)


def _weights(dev, L, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [((torch.rand((D, D), generator=g) * 2 - 1) * 0.08).to(dev) for _ in range(L)]


def _two_kernels(H, graph, w, kernels, relu):
    from tf_gnn_samples_amd import _lib, dense, ops
    L = len(kernels)
    agg = ops._seg_reduce_raw(_lib.AGG_SUM, H, graph.rowptr_t, 1, graph.src_t, w, graph.V * L).view(graph.V, L * D)
    out = dense.limb_gemm_weight(agg, kernels, dense.WEIGHT_NN, None, _lib.ACT_RELU if relu else _lib.ACT_LINEAR)
    return agg, out


def _status():
    from tf_gnn_samples_amd import ops
    return ops.handover_status()


CASES = [
    # V, edges per type, weights?, hub (type, node, length)
    (1, [3], True, None),
    (31, [200, 31, 150], True, None),
    (32, [100, 0, 40], True, None),                         # an empty edge type
    (33, [500, 33, 500], False, None),                      # unit weights (w = NULL)
    (64, [64], True, None),
    (65, [900, 65], True, None),
    (97, [10, 0, 0, 5, 1], True, None),                     # mostly empty buckets, L = 5
    (500, [9000, 500, 9000], True, (0, 77, 700)),           # a 700-message bucket
    (2250, [60000, 2250, 60000], True, (2, 5, 3000)),
    (8229, [200000, 8229, 200000], True, None),             # 258 units: one workgroup per CU plus a remainder
]


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("V,edges,has_w,hub", CASES)
def test_fused_layer_is_bit_identical_to_gather_then_product(gpu_device, V, edges, has_w, hub, relu):
    from tf_gnn_samples_amd import ops
    dev = gpu_device
    graph = _graph(dev, V, edges, seed=V + len(edges), hub=hub)
    g = torch.Generator(device="cpu").manual_seed(V)
    H = torch.randn((V, D), generator=g).to(dev)
    w = (torch.rand(graph.M, generator=g) + 0.01).to(dev) if has_w else None
    kernels = _weights(dev, len(edges), seed=7)
    agg_ref, out_ref = _two_kernels(H, graph, w, kernels, relu)
    agg, out = ops._rgcn_fused(H, graph, w, kernels, relu, True)
    torch.cuda.synchronize()
    assert _status() == 0
    assert torch.equal(agg, agg_ref)
    assert torch.equal(out, out_ref)
    _, out2 = ops._rgcn_fused(H, graph, w, kernels, relu, False)       # inference form: no bucket sums written
    assert torch.equal(out2, out_ref)
    assert _status() == 0


def test_fused_layer_reads_strided_states_and_survives_extreme_values(gpu_device):
    """Rows of a wider table (ldh > 256); float32 lowest / largest (what unsorted_segment_max leaves in an empty segment,
    utils/utils.py:23-33: the split's saturating branch), an inf and a NaN row: the same bits as the two-kernel route, NaNs in the
    same places."""
    from tf_gnn_samples_amd import ops
    dev = gpu_device
    V = 300
    graph = _graph(dev, V, [3000, 300, 2000], seed=5)
    g = torch.Generator(device="cpu").manual_seed(11)
    wide = torch.randn((V, 320), generator=g).to(dev)
    H = wide[:, 32:288]                                               # 16-byte aligned view, row stride 320
    fmax = torch.finfo(torch.float32).max
    H[3, :] = -fmax
    H[4, 10] = fmax
    H[5, 0] = float("inf")
    H[6, 1] = float("nan")
    H[7, :] = 1e-41                                                   # denormals
    w = torch.ones(graph.M, device=dev)
    kernels = _weights(dev, 3, seed=3)
    agg_ref, out_ref = _two_kernels(H, graph, w, kernels, False)
    agg, out = ops._rgcn_fused(H, graph, w, kernels, False, True)
    assert _status() == 0
    assert torch.equal(torch.isnan(agg), torch.isnan(agg_ref)) and torch.equal(torch.isnan(out), torch.isnan(out_ref))
    assert torch.equal(torch.nan_to_num(agg, nan=0.0), torch.nan_to_num(agg_ref, nan=0.0))
    assert torch.equal(torch.nan_to_num(out, nan=0.0), torch.nan_to_num(out_ref, nan=0.0))
    assert bool(torch.isfinite(out_ref).any()) and bool(torch.isnan(out_ref).any())


@pytest.mark.parametrize("aggregation", ["sum", "mean"])
def test_layer_with_the_switch_on_gives_the_same_bits_and_gradients(gpu_device, aggregation):
    """ops.aggregate_then_transform under config.rgcn_fused = 1 vs 0: output, input gradient and every weight gradient identical
    (the backward reads the bucket sums the fused kernel stored)."""
    from tf_gnn_samples_amd import config, ops
    dev = gpu_device
    V = 5000
    graph = _graph(dev, V, [90000, 5000, 90000], seed=1)
    g = torch.Generator(device="cpu").manual_seed(2)
    H0 = torch.randn((V, D), generator=g).to(dev)
    w = (torch.rand(graph.M, generator=g) * 0.1).to(dev)
    k0 = _weights(dev, 3, seed=9)
    gout = torch.randn((V, D), generator=g).to(dev)
    results = {}
    for switch in ("0", "1"):
        with config.override(rgcn_fused=switch):
            H = H0.clone().requires_grad_(True)
            ks = [k.clone().requires_grad_(True) for k in k0]
            out = ops.aggregate_then_transform(H, ks, graph, w, aggregation, "relu")
            out.backward(gout)
            torch.cuda.synchronize()
            results[switch] = [out.detach(), H.grad] + [k.grad for k in ks]
    assert _status() == 0
    for a, b in zip(results["0"], results["1"]):
        assert torch.equal(a, b)


def test_unsupported_shapes_are_refused_not_computed(gpu_device):
    import ctypes
    from tf_gnn_samples_amd import _lib
    lib = _lib.load_library()
    dev = gpu_device
    graph = _graph(dev, 40, [100], seed=0)
    H = torch.zeros((40, 128), device=dev)
    out = torch.zeros((40, 128), device=dev)
    buf = torch.zeros(1 << 16, dtype=torch.bfloat16, device=dev)
    rc = lib.relgnn_rgcn_fused_fwd(H.data_ptr(), 40, 128, graph.rowptr_t.data_ptr(), 40, 1, graph.src_t.data_ptr(), None,
                                   buf.data_ptr(), None, 0, None, 0, out.data_ptr(), 128, 128, 128, None, None)
    assert rc == _lib.EUNSUPPORTED
    rc = lib.relgnn_rgcn_fused_fwd(None, 40, 256, graph.rowptr_t.data_ptr(), 40, 1, graph.src_t.data_ptr(), None,
                                   buf.data_ptr(), None, 0, None, 0, out.data_ptr(), 256, 256, 256, None, None)
    assert rc == _lib.EINVAL


@pytest.mark.parametrize("V,edges", [(4800, [60000, 4800, 60000]), (8229, [200000, 8229, 200000])])
def test_fused_layer_against_the_oracle_directly(gpu_device, V, edges):
    """The fused kernel against the NumPy oracle's sparse_rgcn_layer (gnns/rgcn.py:60-117 in the reference's op order: per-type
    Dense on gathered rows, 1/in-degree scale, concat, unsorted_segment_sum, ReLU) — not through the two-kernel HIP route: the
    north star's 1e-5 absolute on bounded states."""
    from oracle import bookkeeping, gnns as OG
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.graph import RelGraph
    dev = gpu_device
    rng = np.random.default_rng(V)
    adj = [np.stack([rng.integers(0, V, e), rng.integers(0, V, e)], axis=1).astype(np.int32) for e in edges]
    deg = bookkeeping.in_degree_table(adj, V).astype(np.float32)
    H = rng.uniform(-1, 1, size=(V, D)).astype(np.float32)
    Ws = {"Edge_%i_Weight/kernel" % l: (rng.uniform(-1, 1, size=(D, D)) * (6.0 / (2 * D)) ** 0.5).astype(np.float32)
          for l in range(len(edges))}
    want = OG.sparse_rgcn_layer(H, adj, deg, D, 1, "ReLU", "sum", weights=Ws)
    graph = RelGraph([torch.as_tensor(a, device=dev) for a in adj], V)
    w = graph.degree_scale(torch.as_tensor(deg, device=dev))
    kernels = [torch.as_tensor(Ws["Edge_%i_Weight/kernel" % l], device=dev) for l in range(len(edges))]
    _, out = ops._rgcn_fused(torch.as_tensor(H, device=dev), graph, w, kernels, True, False)
    assert _status() == 0
    err = float(np.abs(out.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max())
    assert err <= 1e-5, (err, float(np.abs(want).max()))
