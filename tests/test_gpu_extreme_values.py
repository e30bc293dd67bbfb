"""Extreme and non-finite values through every Dense route.

tf.unsorted_segment_max writes float32 LOWEST (-3.4028235e38) for an empty segment (utils/utils.py:23-33), and the next
op of the reference multiplies that row by a Dense kernel (gnns/ggnn.py:86-92, gnns/rgcn.py:109-114 in the next timestep,
models/sparse_graph_model.py:194-200): fp32 arithmetic yields a finite (huge) number as long as no partial sum leaves the range.
Until round 4 the bf16-limb split turned |x| >= 3.3962e38 into hi = inf, mid = -inf, lo = NaN and the product row into NaN.

What the routes must do, and what is asserted here for each of them (limb triple: relgnn_limb_dense_f32 / relgnn_limb_gemm_xf32 /
relgnn_limb_dense_sel_f32 / relgnn_limb_gemm_tn_f32; limb pair: relgnn_limb16_gemm_xf32 / relgnn_limb16_gemm_tn_f32; lib; panel):
  * finite inputs of ANY magnitude (+-FLT_MAX, the bf16 rounding boundary 0x7F7F8000, 2^127, denormals) with weights small enough
    that no evaluation order overflows: finite outputs, within 4e-6 of the row's largest |output| of the float64 product;
  * a +-inf / NaN input element makes exactly the outputs it takes part in non-finite (its row of a forward product; its row or
    column of a weight gradient) and changes no other output by a single bit.
"""
import numpy as np
import pytest
import torch

from oracle import gnns as G, model as OM
from helpers import glorot, random_relational_graph, rgcn_weights, degree_table

pytestmark = pytest.mark.gpu

FLT_MAX = float(np.finfo(np.float32).max)


def _f(bits):
    return float(np.array([bits], dtype=np.uint32).view(np.float32)[0])


SPECIALS = [FLT_MAX, -FLT_MAX, _f(0x7F7F7FFF), _f(0x7F7F8000), _f(0x7F7F8001), _f(0xFF7F8000), _f(0xFF7FFFFE), 2.0 ** 127,
            -(2.0 ** 127) * (1 + 127 / 128), 3.3e38, 1e38, _f(0x00000001), _f(0x007FFFFF), -_f(0x00400000), _f(0x00800000),
            -_f(0x00800000), 1.0, -1.0, 0.0, -0.0, 65504.0, 65520.0, 1e-30, -7e22]


@pytest.mark.parametrize("transpose", [False, True])
def test_limbs_stay_exact_and_finite_at_both_ends_of_the_range(gpu_device, transpose):
    from tf_gnn_samples_amd import dense as DN
    R, C = 70, 48
    vals = np.array(SPECIALS, dtype=np.float32)
    x = torch.as_tensor(np.resize(vals, (R, C)).copy(), device=gpu_device)
    src = x.t().contiguous() if transpose else x
    limbs = DN.limb_split(src, transpose=transpose)
    assert bool(torch.isfinite(limbs.data.float()).all())
    got, want = limbs.to_float64(), x.double()
    # three bf16 limbs reach down to bf16's smallest denormal, 2^-133: hi + mid + lo == x EXACTLY for every float32 whose lowest set
    # bit is at or above it — every |x| >= 2^-110, and zero — and up to the top of the range (hi saturates at the largest finite
    # bf16 instead of rounding to infinity); below 2^-110 the low bits of x fall under 2^-133 (and the matrix pipe may flush
    # denormal limbs): an absolute error of at most 2^-126 = 1.2e-38, eleven orders of magnitude under any tolerance of the path
    big = want.abs() >= 2.0 ** -110
    assert torch.equal(got[big], want[big])
    assert float((got - want).abs().max()) <= 2.0 ** -126
    assert bool((got[want == 0] == 0).all())


def _extreme_left_operand(M, K, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.rand((M, K), generator=g) * 2 - 1)
    a[3] = -FLT_MAX                                               # an empty unsorted_segment_max row
    a[7, 5] = FLT_MAX
    a[11] = 1e-40                                                 # denormals
    a[13] = torch.as_tensor(np.resize(np.array(SPECIALS, dtype=np.float32), K))
    a[13].clamp_(-3.4e38, 3.4e38)
    a[17] = _f(0x7F7F8000)                                        # the smallest magnitude that rounds to a bf16 infinity
    a[M - 1] = -FLT_MAX                                           # last row of the last (ragged) panel
    a[M // 2, K - 1] = _f(0xFF7F8001)
    return a.to(dev)


def _small_weights(K, N, dev, seed):
    """sum_k |w[k, n]| <= 0.5: FLT_MAX * that stays in range whatever the order of the additions."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = (torch.rand((K, N), generator=g) * 2 - 1)
    w[w.abs() < 1e-3] = 0.5                                       # no zero weights (inf * 0 would be NaN where inf * w is inf)
    return (w * (0.5 / K)).to(dev)


def _forward_product(route, a, w, settings):
    """a [M, K] @ w [K, N] through one route."""
    from tf_gnn_samples_amd import config, dense as DN
    M, K = a.shape
    N = w.shape[1]
    if route == "limb_dense":                                     # relgnn_limb_dense_f32: weights split per call, a in flight
        return DN.limb_dense(DN.GEMM_NN, a, w)
    if route == "limb_dense_nt":
        return DN.limb_dense(DN.GEMM_NT, a, w.t().contiguous())
    if route == "limb_weight_image":                              # relgnn_limb_gemm_xf32 on a cached weight image
        return DN.limb_gemm_weight(a, w, DN.WEIGHT_NN)
    if route == "limb_sel":                                       # relgnn_limb_dense_sel_f32, 128-column panels (also N = 121: cut)
        return DN.limb_dense_sel(DN.GEMM_NN, a, w)
    if route == "limb_sel_gathered":
        rows = torch.arange(M, dtype=torch.int32, device=a.device)
        return DN.limb_dense_sel(DN.GEMM_NN, a, w, a_rows=rows, num_rows=M)
    if route == "limb16_pair":                                    # relgnn_limb16_gemm_xf32: row magnitudes as the gather writes them
        xmax = a.abs().amax(1).contiguous()
        return DN.limb_gemm_weight(a, w, DN.WEIGHT_NN, xmax=xmax, xgroups=1)
    if route == "limb16_pair_groups":
        xmax = a.abs().view(M, 2, K // 2).amax(2).contiguous().view(-1)
        return DN.limb_gemm_weight(a, w, DN.WEIGHT_NN, xmax=xmax, xgroups=2)
    with config.override(gemm=route):                             # lib / panel / torch through the public entry
        return DN.lib_gemm(DN.GEMM_NN, a, w)


FWD_ROUTES = [("limb_dense", 256), ("limb_dense_nt", 256), ("limb_weight_image", 256), ("limb_weight_image", 768), ("limb_sel", 128),
              ("limb_sel", 121), ("limb_sel_gathered", 384), ("limb16_pair", 256), ("limb16_pair_groups", 256), ("lib", 256),
              ("panel", 256), ("torch", 256)]


@pytest.mark.parametrize("route,N", FWD_ROUTES)
@pytest.mark.parametrize("M,K", [(5000, 256), (4133, 48), (36096, 128)])
def test_forward_products_of_huge_and_tiny_finite_rows(gpu_device, route, N, M, K):
    from tf_gnn_samples_amd import config
    if M > 6000 and route not in ("limb_dense", "limb_sel", "limb16_pair"):
        pytest.skip("the large height once per kernel family")
    if route == "limb16_pair_groups" and K % 32:
        pytest.skip("two k-groups per row need K % 32 == 0")
    a = _extreme_left_operand(M, K, gpu_device, M + K)
    w = _small_weights(K, N, gpu_device, N + K)
    out = _forward_product(route, a, w, config.settings)
    truth = a.double() @ w.double()
    assert out.shape == (M, N)
    assert bool(torch.isfinite(out).all()), "non-finite outputs from finite inputs: rows %s" % (
        (~torch.isfinite(out)).any(1).nonzero().flatten()[:8].tolist())
    err = (out.double() - truth).abs().amax(1)
    bound = 4e-6 * truth.abs().amax(1) + 1e-37
    bad = (err > bound).nonzero().flatten()
    assert bad.numel() == 0, (route, bad[:8].tolist(), err[bad[:8]].tolist(), bound[bad[:8]].tolist())


@pytest.mark.parametrize("route,N", FWD_ROUTES)
def test_forward_products_confine_inf_and_nan_to_their_rows(gpu_device, route, N):
    from tf_gnn_samples_amd import config
    M, K = 4500, 256
    a = _extreme_left_operand(M, K, gpu_device, 5)
    w = _small_weights(K, N, gpu_device, 6)
    clean = _forward_product(route, a, w, config.settings)
    b = a.clone()
    b[19, 100] = float("inf")
    b[23, 7] = float("nan")
    b[29] = -float("inf")
    b[31, 0] = float("inf"); b[31, 255] = -float("inf")
    b[M - 2, 33] = float("nan")
    got = _forward_product(route, b, w, config.settings)
    touched = torch.zeros(M, dtype=torch.bool, device=gpu_device)
    touched[[19, 23, 29, 31, M - 2]] = True
    assert not bool(torch.isfinite(got[touched]).any()), "a row with inf / NaN inputs has finite outputs"
    assert torch.equal(got[~touched], clean[~touched]), "an inf / NaN input changed another row"


@pytest.mark.parametrize("form", ["triple", "pair_columns", "pair_operand", "lib"])
@pytest.mark.parametrize("V,J,C", [(5000, 256, 256), (4099, 768, 256), (9001, 128, 512)])
def test_weight_gradient_of_huge_and_tiny_finite_entries(gpu_device, form, V, J, C):
    """dW = A^T G with a float32-lowest row of A (the Dense behind an empty max segment: its weight gradient reduces over that
    row), denormal columns, and a column of G at the top of the range."""
    from tf_gnn_samples_amd import config, dense as DN
    g = torch.Generator(device="cpu").manual_seed(V + J)
    a = (torch.rand((V, J), generator=g) * 2 - 1)
    b = (torch.rand((V, C), generator=g) * 2 - 1) * (0.25 / V)     # |sum_v a[v, j] b[v, c]| <= 0.25 * max|a|
    a[5] = -FLT_MAX
    a[V - 1, 3] = FLT_MAX
    a[:, 9] = 1e-40
    a[:, 10] *= 1e-30
    b[:, 7] *= 1e-25
    b[:, 11] = 1e-41
    a, b = a.to(gpu_device), b.to(gpu_device)
    if form == "triple":
        out = DN.limb_gemm_tn(a, b)
    elif form == "pair_columns":
        out = DN.limb_gemm_tn(a, b, DN.col_absmax(a), DN.col_absmax(b))
    elif form == "pair_operand":
        out = DN.limb_gemm_tn(a, b, DN.absmax(a), DN.absmax(b))
    else:
        with config.override(gemm="lib"):
            out = DN.matmul_tn_splitk(a, b)
    truth = a.double().t() @ b.double()
    assert bool(torch.isfinite(out).all())
    err = (out.double() - truth).abs()
    if form == "pair_operand":          # one scale per operand: normwise only (what round 3 shipped; kept as an ABI form)
        assert float(err.max()) <= 4e-6 * float(truth.abs().max())
        return
    # per output ROW j (one column of A) and per output COLUMN c (one column of G): relative to that row's / column's largest entry
    row_bound = 8e-6 * truth.abs().amax(1, keepdim=True) + 1e-37
    assert bool((err <= row_bound).all()), (form, float((err / row_bound).max()))
    if form != "lib":
        # (the float32-lowest row of A enters EVERY output column and dominates it: the per-column criterion is checked on the same
        #  operands without that row)
        a2 = a.clone(); a2[5] = 0.5; a2[V - 1, 3] = -0.25
        t2 = a2.double().t() @ b.double()
        o2 = DN.limb_gemm_tn(a2, b) if form == "triple" else DN.limb_gemm_tn(a2, b, DN.col_absmax(a2), DN.col_absmax(b))
        e2 = (o2.double() - t2).abs()
        assert bool((e2 <= 8e-6 * t2.abs().amax(0, keepdim=True) + 1e-37).all()), (form, "columns")
        assert bool((e2 <= 8e-6 * t2.abs().amax(1, keepdim=True) + 1e-37).all()), (form, "rows")


@pytest.mark.parametrize("form", ["triple", "pair_columns", "pair_operand"])
def test_weight_gradient_confines_inf_and_nan(gpu_device, form):
    from tf_gnn_samples_amd import dense as DN
    V, J, C = 4500, 768, 256
    g = torch.Generator(device="cpu").manual_seed(1)
    a0 = (torch.rand((V, J), generator=g) * 2 - 1).to(gpu_device)
    b0 = ((torch.rand((V, C), generator=g) * 2 - 1) * 1e-3).to(gpu_device)

    def run(a, b):
        if form == "triple":
            return DN.limb_gemm_tn(a, b)
        if form == "pair_columns":
            return DN.limb_gemm_tn(a, b, DN.col_absmax(a), DN.col_absmax(b))
        return DN.limb_gemm_tn(a, b, DN.absmax(a), DN.absmax(b))

    clean = run(a0, b0)
    a, b = a0.clone(), b0.clone()
    a[7, 20] = float("inf")
    a[V - 3, 700] = float("nan")           # (inside the last V % 32 rows: the exact-fp32 tail of relgnn_sum_slabs_tail_f32)
    b[9, 30] = float("nan")
    b[4000, 255] = -float("inf")
    got = run(a, b)
    bad = torch.zeros((J, C), dtype=torch.bool, device=gpu_device)
    bad[[20, 700], :] = True
    bad[:, [30, 255]] = True
    assert not bool(torch.isfinite(got[bad]).any())
    assert torch.equal(got[~bad], clean[~bad]), "a non-finite element changed a sum it does not take part in"


def test_magnitude_reductions_skip_non_finite_elements(gpu_device):
    from tf_gnn_samples_amd import dense as DN
    x = torch.randn((4097, 260), device=gpu_device)
    x[5, 3] = 77.0
    x[9, 8] = float("inf"); x[100, 8] = -9.0; x[11, 12] = float("nan"); x[4096, 259] = -float("inf")
    want = torch.where(torch.isfinite(x), x.abs(), torch.zeros_like(x))
    assert torch.equal(DN.col_absmax(x), want.amax(0))
    assert float(DN.absmax(x)) == float(want.max())
    assert torch.equal(DN.col_absmax(torch.zeros((33, 8), device=gpu_device)), torch.zeros(8, device=gpu_device))
    v = x[:, :256]                                                   # strided rows
    assert torch.equal(DN.col_absmax(v), want[:, :256].amax(0))
    for cols in (17, 18, 23, 254):                                   # wider than 16 and not a multiple of 4 (the bucket magnitudes
        assert torch.equal(DN.col_absmax(x[:, :cols]), want[:, :cols].amax(0)), cols   # of an 18- or 23-type layer): zero-padded


def _isolating_graph(rng, V, L, edges):
    """Heavy-tailed random edges; every 13th node RECEIVES nothing (but sends): an empty max segment in every timestep."""
    adj = random_relational_graph(rng, V, L, edges)
    return [a[(a[:, 1] % 13) != 0] for a in adj]


def _dev(x, dev):
    if isinstance(x, dict):
        return {k: _dev(v, dev) for k, v in x.items()}
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    return torch.as_tensor(x, device=dev)


ROUTES = [dict(gemm="limb", limb="pair"), dict(gemm="limb", limb="triple"), dict(gemm="lib"), dict(gemm="panel")]


@pytest.mark.parametrize("route", ROUTES, ids=lambda r: "-".join(r.values()))
@pytest.mark.parametrize("D,cell", [(256, "GRU"), (128, "GRU"), (128, "RNN")])
def test_ggnn_max_with_nodes_that_receive_nothing(gpu_device, route, D, cell):
    """gnns/ggnn.py:86-92 with message_aggregation_function='max' on a batch tall enough for the limb route (V >= 8192): the cell's
    input rows of the nodes without incoming edges are float32 lowest in BOTH timesteps; the HIP path must give what the oracle
    gives (the gates saturate), on every route."""
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.gnns import sparse_ggnn_layer
    from tf_gnn_samples_amd.graph import clear_graph_cache
    rng = np.random.default_rng(D)
    V, L = 9000, 4
    adj = _isolating_graph(rng, V, L, [30000, 9000, 0, 20000])
    scope = {"GRU": "gru_cell", "RNN": "simple_rnn_cell"}[cell]
    g = 3 if cell == "GRU" else 1
    w = rgcn_weights(rng, L, D, D)
    # cell kernels small enough that FLT_MAX * sum_k |w_k| stays in range in any summation order (no inf - inf in any evaluation)
    w.update({scope + "/kernel": glorot(rng, (D, g * D)) * np.float32(0.03), scope + "/recurrent_kernel": glorot(rng, (D, g * D)),
              scope + "/bias": (rng.standard_normal(g * D) * 0.1).astype(np.float32)})
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ref = G.sparse_ggnn_layer(h, adj, D, 2, cell, "tanh", "max", weights=w)
    assert np.isfinite(ref).all()
    clear_graph_cache()
    with config.override(**route):
        out = sparse_ggnn_layer(_dev(h, gpu_device), _dev(adj, gpu_device), D, 2, cell, "tanh", "max", weights=_dev(w, gpu_device))
    out = out.cpu().numpy()
    assert np.isfinite(out).all(), "NaN / inf rows: %s" % np.nonzero(~np.isfinite(out).all(1))[0][:8]
    assert np.abs(out - ref).max() < 1e-5


@pytest.mark.parametrize("route", ROUTES, ids=lambda r: "-".join(r.values()))
def test_rgcn_max_linear_through_the_driver_loop(gpu_device, route):
    """models/sparse_graph_model.py:176-200 around gnns/rgcn.py:81-115 with max aggregation and a linear layer activation, two
    timesteps: the float32-lowest rows of timestep 1 go through the per-type Dense of timestep 2 (their huge finite products win or
    lose the next max) and through the driver's Dense(h, tanh) behind layer 0.  HIP path against the oracle, relative to each row's
    magnitude (the states are not O(1) any more)."""
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.graph import clear_graph_cache
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import PPI_Task
    rng = np.random.default_rng(9)
    V, L = 8200, 3
    adj = _isolating_graph(rng, V, L, [40000, 8200, 40000])
    deg = degree_table(adj, V)
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(1, 1, seed=1, mean_nodes=100, std_nodes=10, min_nodes=50, max_nodes=150, fwd_edges_per_node=3.0)
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=2, graph_num_timesteps_per_layer=2, message_aggregation_function="max",
             graph_activation_function="linear", graph_layer_input_dropout_keep_prob=1.0)
    model = RGCN_Model(p, task, device=str(gpu_device))
    with torch.no_grad():
        for n in model.variables.names():                 # kernels small enough that nothing overflows in any summation order
            if n.endswith("/kernel") and "gnn_layer" in n:
                model.variables[n].mul_(0.03)
    feats = rng.standard_normal((V, 50)).astype(np.float32)
    W = {k[len("graph_model/"):]: v.detach().cpu().numpy() for k, v in
         ((n, model.variables[n]) for n in model.variables.names()) if k.startswith("graph_model/")}
    ref = OM.graph_propagation(feats, adj, deg.astype(np.float32), p, W, OM.rgcn_apply(p))
    assert np.isfinite(ref).all()
    clear_graph_cache()
    with config.override(**route), torch.no_grad():
        out = model.compute_final_node_representations(_dev(feats, gpu_device), _dev(adj, gpu_device),
                                                       _dev(deg.astype(np.float32), gpu_device)).cpu().numpy()
    assert np.isfinite(out).all(), "NaN / inf rows: %s" % np.nonzero(~np.isfinite(out).all(1))[0][:8]
    scale = np.maximum(1.0, np.abs(ref).max(1, keepdims=True))
    assert (np.abs(out - ref) / scale).max() < 1e-5
