"""relgnn_limb_gemm_f32 / relgnn_limb_split_f32 (csrc/limb_gemm.hip): fp32 products from three bf16 limbs per operand.

The limbs must add up to the fp32 value EXACTLY; the product must be as close to the float64 result as the exact-fp32 matrix
pipe is (same error class: the dropped limb products are < 2^-23 of each term), on every panel geometry and with the epilogue."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return ((torch.rand(shape, generator=g) * 2 - 1) * scale).to(dev)


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("R,C", [(517, 272), (32, 16), (1, 48), (272, 517 + 11)])
def test_limbs_are_exact(gpu_device, transpose, R, C):
    from tf_gnn_samples_amd import dense as DN
    x = _rand((R, C), gpu_device, 1) * torch.logspace(-20, 20, C, device=gpu_device)     # wide exponent range
    x[0, 5] = 0.0
    x[0, 7] = -0.0
    if transpose:
        x = x.t().contiguous()                   # [C, R]: the limbs of its transpose are an [R, C] matrix again
    limbs = DN.limb_split(x, transpose=transpose)
    assert limbs.data.dtype == torch.bfloat16 and (limbs.rows, limbs.cols) == (R, C)
    want = x.t() if transpose else x
    assert torch.equal(limbs.to_float64(), want.double())
    # rows past R inside the last 32-row tile are zeros
    RB = (R + 31) // 32
    full = limbs.data.view(RB, C // 16, 3, 2, 32, 8)
    if R % 32:
        assert float(full[-1, :, :, :, R % 32:, :].float().abs().max()) == 0.0


@pytest.mark.parametrize("M", [1, 31, 32, 33, 160, 161, 1000, 5000, 36096])
@pytest.mark.parametrize("N,K", [(256, 768), (768, 256), (256, 16)])
def test_limb_gemm_matches_float64(gpu_device, M, N, K):
    from tf_gnn_samples_amd import dense as DN
    if M > 5000 and N == 768:
        pytest.skip("one large shape per column count is enough")
    a = _rand((M, K), gpu_device, M + K)
    w = _rand((N, K), gpu_device, N + K + 7, 0.1)
    wl = DN.limb_split(w)
    out = DN.limb_gemm(DN.limb_split(a), wl)
    out_x = DN.limb_gemm_xf32(a, wl)             # the left operand split inside the kernel: the same limbs, the same products
    truth = a.double() @ w.double().t()
    f32 = a @ w.t()                               # exact fp32 (library)
    e_limb = float((out.double() - truth).abs().max())
    e_f32 = float((f32.double() - truth).abs().max())
    scale = float(truth.abs().max())
    assert e_limb <= max(2.0 * e_f32, 4e-7 * scale), (e_limb, e_f32, scale)
    assert torch.equal(out_x, out)


@pytest.mark.parametrize("K", [16, 32, 48, 80, 1008])
@pytest.mark.parametrize("M", [5, 100, 160, 700])
def test_limb_gemm_xf32_k_tails_and_strided_rows(gpu_device, M, K):
    """Odd numbers of k-tiles (the last 32-k super-tile half empty), a row stride larger than K, a poisoned neighbourhood."""
    from tf_gnn_samples_amd import dense as DN
    big = torch.full((M, K + 24), float("nan"), device=gpu_device)
    a = big[:, 4:4 + K]
    a.copy_(_rand((M, K), gpu_device, M * 7 + K))
    w = _rand((256, K), gpu_device, K + 3, 0.2)
    out = DN.limb_gemm_xf32(a, DN.limb_split(w))
    truth = a.double() @ w.double().t()
    e_f32 = float(((a.contiguous() @ w.t()).double() - truth).abs().max())
    assert float((out.double() - truth).abs().max()) <= max(3.0 * e_f32, 8e-7 * max(1.0, float(truth.abs().max())))
    assert torch.isnan(big[:, :4]).all() and torch.isnan(big[:, 4 + K:]).all()


def test_limb_gemm_from_transposed_weights_bias_act(gpu_device):
    """The forward use: out = relu(bias + A @ W) with W [K, N] split transposed."""
    from tf_gnn_samples_amd import _lib, dense as DN
    a = _rand((3000, 768), gpu_device, 11)
    W = _rand((768, 256), gpu_device, 12, 0.08)
    b = _rand((256,), gpu_device, 13)
    wl = DN.limb_split(W, transpose=True)
    out = DN.limb_gemm(DN.limb_split(a), wl, bias=b, act=_lib.ACT_RELU)
    assert torch.equal(DN.limb_gemm_xf32(a, wl, bias=b, act=_lib.ACT_RELU), out)
    truth = torch.relu(a.double() @ W.double() + b.double())
    f32 = torch.relu(a @ W + b)
    assert float((out.double() - truth).abs().max()) <= 1.5 * float((f32.double() - truth).abs().max())


def test_limb_gemm_strided_output(gpu_device):
    from tf_gnn_samples_amd import dense as DN
    a = _rand((700, 256), gpu_device, 21)
    w = _rand((256, 256), gpu_device, 22, 0.1)
    big = torch.full((700, 1024), 7.0, device=gpu_device)
    out = big[:, 256:512]
    DN.limb_gemm(DN.limb_split(a), DN.limb_split(w), out=out)
    truth = a.double() @ w.double().t()
    assert float((out.double() - truth).abs().max()) <= 2e-6
    assert float(big[:, :256].min()) == 7.0 and float(big[:, 512:].min()) == 7.0            # nothing else written


def test_limb_rejects_unsupported(gpu_device):
    from tf_gnn_samples_amd import dense as DN
    with pytest.raises(ValueError):
        DN.limb_split(_rand((64, 24), gpu_device, 1))          # columns % 16 != 0
    a = DN.limb_split(_rand((64, 32), gpu_device, 1))
    w = DN.limb_split(_rand((128, 32), gpu_device, 2))
    with pytest.raises(ValueError):
        DN.limb_gemm(a, w)                                     # N % 256 != 0
    with pytest.raises(ValueError):
        DN.limb_gemm(a, DN.limb_split(_rand((256, 48), gpu_device, 3)))      # reduction lengths differ


@pytest.mark.parametrize("layout", ["NN", "NT"])
def test_limb_dense_is_the_product_of_the_path(gpu_device, layout):
    """relgnn_limb_dense_f32: fp32 operands in, the weights split on the fly; equals split + product bit for bit."""
    from tf_gnn_samples_amd import _lib, dense as DN
    a = _rand((5000, 768), gpu_device, 31)
    if layout == "NN":
        W = _rand((768, 256), gpu_device, 32, 0.08)
        b = _rand((256,), gpu_device, 33)
        out = DN.limb_dense(DN.GEMM_NN, a, W, b, _lib.ACT_RELU)
        want = DN.limb_gemm_xf32(a, DN.limb_split(W, transpose=True), bias=b, act=_lib.ACT_RELU)
        truth = torch.relu(a.double() @ W.double() + b.double())
    else:
        W = _rand((256, 768), gpu_device, 34, 0.08)
        out = DN.limb_dense(DN.GEMM_NT, a, W)
        want = DN.limb_gemm_xf32(a, DN.limb_split(W))
        truth = a.double() @ W.double().t()
    assert torch.equal(out, want)
    assert float((out.double() - truth).abs().max()) <= 6e-6


@pytest.mark.parametrize("V", [32, 33, 95, 257, 4099, 36096])
@pytest.mark.parametrize("J,C", [(768, 256), (256, 256), (32, 256), (64, 512), (96, 256)])
def test_limb_gemm_tn_matches_float64(gpu_device, V, J, C):
    """dW = A^T @ G with both operands split and transposed in flight; every panel geometry (T32 = 4, 2, 1), ragged chunks."""
    from tf_gnn_samples_amd import dense as DN
    if V > 5000 and (J, C) not in ((768, 256), (256, 256)):
        pytest.skip("two large shapes are enough")
    big = torch.full((V, J + 8), float("nan"), device=gpu_device)
    a = big[:, 4:4 + J]
    a.copy_(_rand((V, J), gpu_device, V + J))
    g = _rand((V, C), gpu_device, V + C + 5, 0.05)
    out = DN.limb_gemm_tn(a, g)
    truth = a.double().t() @ g.double()
    f32 = a.contiguous().t() @ g
    e_limb = float((out.double() - truth).abs().max())
    e_f32 = float((f32.double() - truth).abs().max())
    assert e_limb <= max(3.0 * e_f32, 8e-7 * max(1.0, float(truth.abs().max()))), (e_limb, e_f32)


@pytest.mark.parametrize("M", [1, 127, 128, 129, 300, 5000, 70000])
@pytest.mark.parametrize("layout,N,K", [("NN", 128, 128), ("NT", 128, 128), ("NN", 640, 128), ("NN", 384, 16), ("NN", 128, 32),
                                        ("NT", 128, 272), ("NN", 256, 48)])
def test_limb_dense_sel_plain(gpu_device, M, layout, N, K):
    """relgnn_limb_dense_sel_f32 without gather / selection: 128 x 128 panels, every k-tile count parity, bias + activation."""
    from tf_gnn_samples_amd import _lib, dense as DN
    if M > 5000 and (N, K) != (128, 128):
        pytest.skip("one large case per layout")
    a = _rand((M, K), gpu_device, M + K)
    if layout == "NN":
        W = _rand((K, N), gpu_device, N + 1, 0.1)
        b = _rand((N,), gpu_device, 3)
        out = DN.limb_dense_sel(DN.GEMM_NN, a, W, b, _lib.ACT_TANH)
        truth = torch.tanh(a.double() @ W.double() + b.double())
        f32 = torch.tanh(a @ W + b)
    else:
        W = _rand((N, K), gpu_device, N + 2, 0.1)
        out = DN.limb_dense_sel(DN.GEMM_NT, a, W)
        truth = a.double() @ W.double().t()
        f32 = a @ W.t()
    e, e32 = float((out.double() - truth).abs().max()), float((f32.double() - truth).abs().max())
    assert e <= max(3.0 * e32, 8e-7 * max(1.0, float(truth.abs().max()))), (e, e32)


@pytest.mark.parametrize("layout,Din,Dout", [("NN", 128, 128), ("NN", 128, 256), ("NT", 128, 128), ("NT", 256, 128)])
def test_limb_dense_sel_gathered_rows_and_per_tile_weights(gpu_device, layout, Din, Dout):
    """The typed transform Y[r] = H[node[r]] @ W_type(tile(r)) and its input gradient dX[r] = dY[r] @ W_type^T: 512-row tiles, 23
    kernels, padding rows (-1), against float64 and against relgnn_panel_gemm_f32 (exact fp32) on the same operands."""
    from tf_gnn_samples_amd import dense as DN
    L, tiles, V = 23, 71, 5000
    g = torch.Generator(device="cpu").manual_seed(Din + Dout)
    tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(gpu_device)
    P = tiles * 512
    W = _rand((L, Din, Dout), gpu_device, 5, 0.1)
    if layout == "NN":
        H = _rand((V, Din), gpu_device, 6)
        node = torch.randint(0, V, (P,), generator=g).to(torch.int32)
        node[torch.rand(P, generator=g) < 0.1] = -1
        node = node.to(gpu_device)
        out = DN.limb_dense_sel(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
        ref = DN.panel_gemm(DN.GEMM_NN, H, W, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
        Hg = torch.where((node >= 0).unsqueeze(1), H[node.clamp(min=0).long()], torch.zeros((), device=gpu_device)).double()
        truth = torch.bmm(Hg.view(tiles, 512, Din), W.double()[tile_type.long()]).view(P, Dout)
    else:
        gY = _rand((P, Dout), gpu_device, 7)
        out = DN.limb_dense_sel(DN.GEMM_NT, gY, W, b_select=tile_type, rows_per_select=512)
        ref = DN.panel_gemm(DN.GEMM_NT, gY, W, b_select=tile_type, rows_per_select=512, dims=(P, Din, Dout))
        truth = torch.bmm(gY.double().view(tiles, 512, Dout), W.double()[tile_type.long()].transpose(1, 2)).view(P, Din)
    e, e32 = float((out.double() - truth).abs().max()), float((ref.double() - truth).abs().max())
    assert e <= max(3.0 * e32, 8e-7 * max(1.0, float(truth.abs().max()))), (e, e32)


def test_weight_limb_images_equal_single_splits(gpu_device):
    """relgnn_limb_split_multi_f32 (one launch for the weights of a step) writes the images relgnn_limb_split_f32 writes, for the
    three operand kinds, incl. the stacked [W_0 | W_1 | ..] operand that is never formed in fp32, and for more matrices than one
    launch takes."""
    from tf_gnn_samples_amd import dense as DN
    DN._WEIGHT_LIMBS.clear()
    W = _rand((3, 256, 256), gpu_device, 1, 0.1)
    Wd = _rand((256, 512), gpu_device, 2, 0.1)
    a = DN.weight_limbs(W.view(768, 256), DN.WEIGHT_NN)
    assert torch.equal(a, DN.limb_split(W.view(768, 256), transpose=True).data)
    b = DN.weight_limbs(Wd, DN.WEIGHT_NT)
    assert torch.equal(b, DN.limb_split(Wd).data)
    c = DN.weight_limbs(W, DN.WEIGHT_NT)                               # [W_0 | W_1 | W_2]: B [Din, L*Dout]
    stacked = W.permute(0, 2, 1).reshape(768, 256)                    # B^T = [K = L*Dout, N = Din]
    assert torch.equal(c, DN.limb_split(stacked, transpose=True).data)
    separate = [_rand((256, 256), gpu_device, 40 + l, 0.1) for l in range(3)]      # three variables, not one stack
    d = DN.weight_limbs(separate, DN.WEIGHT_NN)
    assert torch.equal(d, DN.limb_split(torch.cat(separate, 0), transpose=True).data)
    e = DN.weight_limbs(separate, DN.WEIGHT_NT)
    assert torch.equal(e, DN.limb_split(torch.cat(separate, 1)).data)
    many = [_rand((48, 32), gpu_device, 10 + i) for i in range(30)]    # 30 items: two launches
    DN.weights_changed()
    for m in many:
        DN.weight_limbs(m, DN.WEIGHT_NT)
    DN.weights_changed()
    got = [DN.weight_limbs(m, DN.WEIGHT_NT) for m in many]            # the first request re-splits all 30 + the 3 above
    for m, g in zip(many, got):
        assert torch.equal(g, DN.limb_split(m).data)
    DN._WEIGHT_LIMBS.clear()


def test_weight_limb_cache_follows_the_weights(gpu_device):
    """A cached image is replaced after an in-place write (version counter), after weights_changed() (writes through raw
    pointers, as the fused optimizer launch does), and when another tensor takes the address."""
    from tf_gnn_samples_amd import config, dense as DN
    if not config.settings.limb_gemm:
        pytest.skip("the public Dense entry takes another route in this run (RELGNN_GEMM)")
    DN._WEIGHT_LIMBS.clear()
    x = _rand((4608, 256), gpu_device, 3)
    w = _rand((256, 256), gpu_device, 4, 0.1)
    y0 = DN.lib_gemm(DN.GEMM_NN, x, w, weight=True)
    assert torch.equal(y0, DN.lib_gemm(DN.GEMM_NN, x, w))
    with torch.no_grad():
        w.mul_(2.0)
    assert torch.equal(DN.lib_gemm(DN.GEMM_NN, x, w, weight=True), DN.lib_gemm(DN.GEMM_NN, x, w))
    w.data.mul_(0.5)                                                  # (a write the version counter of w does not see)
    stale = DN.lib_gemm(DN.GEMM_NN, x, w, weight=True)
    DN.weights_changed()
    fresh = DN.lib_gemm(DN.GEMM_NN, x, w, weight=True)
    assert torch.equal(fresh, y0) and torch.equal(fresh, DN.lib_gemm(DN.GEMM_NN, x, w))
    assert not torch.equal(stale, fresh)                              # documents why weights_changed() exists
    ptr = w.data_ptr()
    del w
    w2 = _rand((256, 256), gpu_device, 5, 0.1)
    if w2.data_ptr() == ptr:
        assert torch.equal(DN.lib_gemm(DN.GEMM_NN, x, w2, weight=True), DN.lib_gemm(DN.GEMM_NN, x, w2))
    DN._WEIGHT_LIMBS.clear()


def test_grouped_gemms_equal_the_stacked_products(gpu_device):
    """The aggregate-first layer's two products over the per-edge-type kernels as they are (no stacked operand in fp32): same bits
    as the products with the stacked operands."""
    from tf_gnn_samples_amd import dense as DN
    Ws = [_rand((256, 256), gpu_device, 6 + l, 0.1) for l in range(3)]
    x = _rand((9000, 768), gpu_device, 7)
    for relu in (False, True):
        assert torch.equal(DN.grouped_nn_gemm(x, Ws, relu=relu), DN.lib_gemm(DN.GEMM_NN, x, torch.cat(Ws, 0), relu=relu))
    stackedT = torch.cat([w.t() for w in Ws], 0)
    ref = DN.lib_gemm(DN.GEMM_NN, x, stackedT)
    assert torch.equal(DN.grouped_nt_gemm(x, Ws), ref)
    truth = x.double() @ stackedT.double()
    assert float((ref.double() - truth).abs().max()) < 4e-6 * max(1.0, float(truth.abs().max()))
    W5 = [_rand((128, 64), gpu_device, 20 + l, 0.1) for l in range(5)]           # not a limb shape: the library route
    g5 = _rand((5000, 320), gpu_device, 9)
    t5 = g5.double() @ torch.cat([w.t() for w in W5], 0).double()
    assert float((DN.grouped_nt_gemm(g5, W5).double() - t5).abs().max()) < 1e-5
    x5 = _rand((5000, 640), gpu_device, 10)
    assert float((DN.grouped_nn_gemm(x5, W5).double() - x5.double() @ torch.cat(W5, 0).double()).abs().max()) < 1e-5


def test_training_steps_with_and_without_the_weight_limb_cache(gpu_device):
    """Three fused clip + Adam steps of the C2-shaped model: the parameters are the same bits whether the weights' limbs are
    split per product or once per step."""
    from tf_gnn_samples_amd import dense as DN
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(3, 1, seed=3)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    results = []
    for cached in (True, False):
        DN._WEIGHT_LIMBS.clear()
        config.settings.weight_limb_cache = "1" if cached else "0"
        try:
            torch.manual_seed(0)
            p = RGCN_Model.default_params()
            p.update(hidden_size=256, graph_num_layers=3, graph_layer_input_dropout_keep_prob=1.0, random_seed=0)
            model = RGCN_Model(p, task, device=str(gpu_device))
            batch = DeviceBatch(mb, gpu_device)
            for _ in range(3):
                model.train_step(batch)
            with torch.no_grad():
                model.forward_batch(batch, training=False)            # (an evaluation pass right behind an update)
            results.append([q.detach().clone() for q in model.optimizer.params])
        finally:
            config.settings.weight_limb_cache = "1"
    assert mb.num_nodes >= 4096                                       # (the limb route is what ran)
    for a, b in zip(*results):
        assert torch.equal(a, b)
    DN._WEIGHT_LIMBS.clear()


@pytest.mark.parametrize("rps,tiles,last", [(128, 300, 128), (256, 140, 200), (384, 90, 1), (512, 70, 300), (512, 65, 512)])
def test_limb_dense_sel_tile_sizes_and_ragged_last_tile(gpu_device, rps, tiles, last):
    """K = 128 and >= 32 k rows take the persistent weights-resident form (runs of 128-row panels per workgroup, the weights
    replaced where the select tile's type changes): every tile height, unsorted types, a last tile that is cut short, rows of
    zeros; against float64 and against relgnn_panel_gemm_f32 (exact fp32) where that kernel takes the tile height."""
    from tf_gnn_samples_amd import dense as DN
    L, V = 7, 3000
    g = torch.Generator(device="cpu").manual_seed(rps)
    M = (tiles - 1) * rps + last
    tile_type = torch.randint(0, L, (tiles,), generator=g).to(torch.int32).to(gpu_device)
    W = _rand((L, 128, 256), gpu_device, 5, 0.1)
    H = _rand((V, 128), gpu_device, 6)
    node = torch.randint(0, V, (M,), generator=g).to(torch.int32)
    node[torch.rand(M, generator=g) < 0.1] = -1
    node = node.to(gpu_device)
    out = DN.limb_dense_sel(DN.GEMM_NN, H, W, a_rows=node, num_rows=M, b_select=tile_type, rows_per_select=rps)
    Hg = torch.where((node >= 0).unsqueeze(1), H[node.clamp(min=0).long()], torch.zeros((), device=gpu_device)).double()
    sel = tile_type.long().repeat_interleave(rps)[:M]
    truth = torch.einsum("rk,rkn->rn", Hg, W.double()[sel])
    assert float((out.double() - truth).abs().max()) <= 2e-6 * max(1.0, float(truth.abs().max()))
    if rps == 512 and last == 512:
        ref = DN.panel_gemm(DN.GEMM_NN, H, W, a_rows=node, num_rows=M, b_select=tile_type, rows_per_select=rps)
        assert float((out - ref).abs().max()) <= 4e-6 * max(1.0, float(truth.abs().max()))


@pytest.mark.parametrize("M,N,K,bias,act", [(9000, 121, 256, True, "linear"), (5000, 250, 64, False, "tanh"), (4100, 97, 128, True, "relu"),
                                            (40000, 121, 128, True, "linear")])
def test_limb_dense_sel_cut_last_chunk(gpu_device, M, N, K, bias, act):
    """N that is not a multiple of 128 (the 121 labels of the PPI head): the last chunk is cut at N, rows of C are not 16-byte
    aligned; nothing is written past a row's N columns."""
    from tf_gnn_samples_amd import _lib, dense as DN
    a = _rand((M, K), gpu_device, M + N)
    W = _rand((K, N), gpu_device, N + 1, 0.1)
    b = _rand((N,), gpu_device, 3) if bias else None
    code = {"linear": _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH, "relu": _lib.ACT_RELU}[act]
    out = DN.limb_dense_sel(DN.GEMM_NN, a, W, b, code)
    assert out.shape == (M, N) and out.is_contiguous()
    z = a.double() @ W.double() + (b.double() if bias else 0.0)
    truth = {"linear": z, "tanh": torch.tanh(z), "relu": torch.relu(z)}[act]
    z32 = a @ W + (b if bias else 0.0)
    f32 = {"linear": z32, "tanh": torch.tanh(z32), "relu": torch.relu(z32)}[act]
    e, e32 = float((out.double() - truth).abs().max()), float((f32.double() - truth).abs().max())
    assert e <= max(3.0 * e32, 8e-7 * max(1.0, float(truth.abs().max()))), (e, e32)
    # the route the Dense layers take, and its gradients through the library routes
    from tf_gnn_samples_amd import config
    if act == "linear" and N == 121 and config.settings.limb_gemm:
        x = a.clone().requires_grad_(True)
        k = W.clone().requires_grad_(True)
        bb = b.clone().requires_grad_(True)
        y = DN.dense(x, k, bb)
        assert torch.equal(y, out)
        y.backward(torch.ones_like(y))
        assert float((x.grad.double() - W.double().sum(1)).abs().max()) < 1e-4


def test_limb_dense_sel_cut_last_chunk_nt_layout(gpu_device):
    """The same cut for the NT layout (B given as [N, K] with N = 121 rows)."""
    from tf_gnn_samples_amd import dense as DN
    a = _rand((6000, 256), gpu_device, 1)
    W = _rand((121, 256), gpu_device, 2, 0.1)
    out = DN.limb_dense_sel(DN.GEMM_NT, a, W)
    truth = a.double() @ W.double().t()
    assert out.shape == (6000, 121)
    e, e32 = float((out.double() - truth).abs().max()), float(((a @ W.t()).double() - truth).abs().max())
    assert e <= max(3.0 * e32, 8e-7 * max(1.0, float(truth.abs().max()))), (e, e32)


@pytest.mark.parametrize("agg,scale", [("sum", 1.0), ("mean", 1.0), ("sum", 1e-4)])
def test_two_fp16_limb_products_of_the_aggregate_first_layer(gpu_device, monkeypatch, agg, scale):
    """RELGNN_LIMB=pair: the gather writes every bucket's largest magnitude, the products of the aggregate-first layer (forward and
    input gradient) run from two fp16 limbs per value behind power-of-two row scales.  Output and gradients against float64,
    next to the bf16-triple route's errors; rows of zeros (isolated nodes) and tiny inputs included."""
    from helpers import random_relational_graph
    from tf_gnn_samples_amd import config, dense as DN, ops
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(3)
    V, L, D = 9000, 3, 256
    adj = random_relational_graph(rng, V, L, [70000, 9000, 70000])
    adj = [a[(a[:, 1] % 17) != 0] for a in adj]                      # every 17th node receives nothing: rows of zeros
    g = RelGraph([torch.as_tensor(a, device=gpu_device) for a in adj], V)
    w = g.degree_scale(torch.as_tensor(np.stack([np.bincount(a[:, 1], minlength=V) for a in adj]).astype(np.float32),
                                       device=gpu_device)) if agg == "mean" else None
    H0 = torch.as_tensor((np.maximum(rng.standard_normal((V, D)), 0) * scale).astype(np.float32), device=gpu_device)
    W0 = [torch.as_tensor((rng.standard_normal((D, D)) * 0.06).astype(np.float32), device=gpu_device) for _ in range(L)]
    gout = torch.as_tensor((rng.standard_normal((V, D)) * np.exp(rng.uniform(-6, 2, (V, 1)))).astype(np.float32), device=gpu_device)

    def run(pair):
        monkeypatch.setattr(config.settings, "limb", "pair" if pair else "triple")
        H = H0.clone().requires_grad_(True)
        Ws = [x.clone().requires_grad_(True) for x in W0]
        out = ops.aggregate_then_transform(H, Ws, g, w, "sum", "relu")
        out.backward(gout)
        return [out.detach(), H.grad] + [x.grad for x in Ws]

    triple, pair = run(False), run(True)
    if w is None:                                                    # float64 truth through the same function in plain torch
        Hd = H0.double().requires_grad_(True)
        Wd = [x.double().requires_grad_(True) for x in W0]
        acc = 0
        for l in range(L):
            src, tgt = (torch.as_tensor(adj[l][:, i].astype(np.int64), device=gpu_device) for i in (0, 1))
            acc = acc + torch.zeros((V, D), dtype=torch.float64, device=gpu_device).index_add_(0, tgt, Hd[src]) @ Wd[l]
        truth_out = torch.relu(acc)
        truth_out.backward(gout.double())
        truth = [truth_out.detach(), Hd.grad] + [x.grad for x in Wd]
        for name, t, a, b in zip(["out", "dH", "dW0", "dW1", "dW2"], truth, triple, pair):
            et, ep = float((a.double() - t).abs().max()), float((b.double() - t).abs().max())
            assert ep <= max(2.0 * et, 4e-6 * max(float(t.abs().max()), 1e-30)), (name, ep, et)
    for name, a, b in zip(["out", "dH", "dW0", "dW1", "dW2"], triple, pair):
        assert float((a - b).abs().max()) <= 8e-6 * max(float(a.abs().max()), 1e-30), name


@pytest.mark.parametrize("V,J,C", [(36096, 768, 256), (5000, 128, 256), (9001, 256, 512)])
def test_limb16_gemm_tn_matches_float64(gpu_device, V, J, C):
    """The weight gradient from two fp16 limbs behind one power-of-two scale per column of each operand (and the per-operand form
    of the ABI), against float64 next to the bf16 triple; operands whose rows span four decades (what gradients look like)."""
    from tf_gnn_samples_amd import dense as DN
    g = torch.Generator(device="cpu").manual_seed(V + J)
    a = (torch.relu(torch.randn((V, J), generator=g)) * torch.distributions.Gamma(2.0, 0.125).sample((V, 1))).to(gpu_device)
    b = (torch.randn((V, C), generator=g) * 1e-4 * torch.exp(torch.empty(V, 1).uniform_(-4.6, 4.6))).to(gpu_device)
    truth = a.double().t() @ b.double()
    pair = DN.limb_gemm_tn(a, b, DN.col_absmax(a), DN.col_absmax(b))
    pair1 = DN.limb_gemm_tn(a, b, DN.absmax(a), DN.absmax(b))
    triple = DN.limb_gemm_tn(a, b)
    e2, e1, e3 = (float((x.double() - truth).abs().max()) for x in (pair, pair1, triple))
    assert e2 <= max(2.0 * e3, 2e-6 * float(truth.abs().max())), (e2, e3)
    assert e1 <= max(2.0 * e3, 2e-6 * float(truth.abs().max())), (e1, e3)
    assert float(DN.absmax(b)) == float(b.abs().max()) and float(DN.absmax(a[:7, :3].contiguous())) == float(a[:7, :3].abs().max())
    assert torch.equal(DN.col_absmax(b), b.abs().amax(0))


def test_limb16_gemm_tn_columns_over_ten_decades(gpu_device):
    """the columns of both operands scaled 1e0 .. 1e-10 (Adam divides every weight's gradient by ITS OWN running
    magnitude, so a small column's relative error reaches the update).  Worst relative error per output row (one column of A) and
    per output column (one column of G), next to the exact-fp32 product's: the per-column scales must stay within 4x of fp32;
    the per-operand scale of round 3 does not (recorded, not asserted).  Numbers -> gpurun_out/limb16_tn_column_range.json."""
    import json
    import os
    from tf_gnn_samples_amd import config, dense as DN
    V, J, C = 36096, 768, 256
    g = torch.Generator(device="cpu").manual_seed(11)
    sa = torch.logspace(0, -10, J)[torch.randperm(J, generator=g)]
    sb = torch.logspace(0, -10, C)[torch.randperm(C, generator=g)]
    a = (torch.relu(torch.randn((V, J), generator=g)) * sa).to(gpu_device)
    b = (torch.randn((V, C), generator=g) * 1e-3 * sb).to(gpu_device)
    truth = a.double().t() @ b.double()
    with config.override(gemm="lib"):
        fp32 = DN.matmul_tn_splitk(a, b)
    outs = {"fp32_library_split_k": fp32, "bf16_triple": DN.limb_gemm_tn(a, b),
            "fp16_pair_column_scales": DN.limb_gemm_tn(a, b, DN.col_absmax(a), DN.col_absmax(b)),
            "fp16_pair_operand_scale": DN.limb_gemm_tn(a, b, DN.absmax(a), DN.absmax(b)),
            # what the aggregate-first layer passes (ops._weight_gradient): one scale per 256-column group (edge type) of the
            # bucket sums, one per column of the gradient — on THESE operands (A's columns ten decades apart inside a group) recorded only
            "fp16_pair_layer_form_on_these_operands": DN.limb_gemm_tn(a, b, a.abs().view(V, 3, 256).amax(2).amax(0), DN.col_absmax(b))}
    rep = {}
    outer = (sa.double()[:, None] * sb.double()[None, :]).to(gpu_device)      # entry (j, c) is a sum of terms of size sa[j] * sb[c]
    for name, o in outs.items():
        err = (o.double() - truth).abs()
        rep[name] = {"worst_entry_over_its_column_scales": float((err / outer).max()),
                     "worst_row_rel": float((err.amax(1) / truth.abs().amax(1)).max()),
                     "worst_col_rel": float((err.amax(0) / truth.abs().amax(0)).max()),
                     "max_abs_over_max_abs": float(err.max() / truth.abs().max())}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/limb16_tn_column_range.json", "w") as f:
        json.dump({"shape": [V, J, C], "columns_scaled": "1e0 .. 1e-10 (log-spaced, shuffled), both operands", "errors": rep}, f, indent=1)
    print(rep)
    f32, pc = rep["fp32_library_split_k"], rep["fp16_pair_column_scales"]
    assert pc["worst_row_rel"] <= 4.0 * f32["worst_row_rel"] and pc["worst_col_rel"] <= 4.0 * f32["worst_col_rel"], rep
    assert pc["worst_entry_over_its_column_scales"] <= 4.0 * f32["worst_entry_over_its_column_scales"], rep
    assert rep["bf16_triple"]["worst_row_rel"] <= 4.0 * f32["worst_row_rel"], rep
    # the layer's form where it is used: gradient columns ten decades apart, the bucket sums' columns within two decades of each
    # other inside an edge type (sums of post-activation states), the three edge types three decades apart
    sa2 = (10.0 ** (-2 * torch.rand(J, generator=g))) * torch.tensor([1.0, 1e-3, 30.0]).repeat_interleave(256)
    a2 = (torch.relu(torch.randn((V, J), generator=g)) * sa2).to(gpu_device)
    truth2 = a2.double().t() @ b.double()
    with config.override(gemm="lib"):
        f2 = DN.matmul_tn_splitk(a2, b)
    l2 = DN.limb_gemm_tn(a2, b, a2.abs().view(V, 3, 256).amax(2).amax(0), DN.col_absmax(b))
    outer2 = (sa2.double()[:, None] * sb.double()[None, :]).to(gpu_device)
    e_f, e_l = ((x.double() - truth2).abs() / outer2 for x in (f2, l2))
    rep2 = {"fp32_library_split_k": float(e_f.max()), "fp16_pair_layer_form": float(e_l.max())}
    with open("gpurun_out/limb16_tn_column_range.json", "w") as f:
        json.dump({"shape": [V, J, C], "columns_scaled": "1e0 .. 1e-10 (log-spaced, shuffled), both operands", "errors": rep,
                   "layer_form_regime": {"what": "worst entry error over the scales of its row and column; A's columns within two "
                                                 "decades inside an edge type, edge types three decades apart, G's columns ten decades",
                                         "errors": rep2}}, f, indent=1)
    assert rep2["fp16_pair_layer_form"] <= 4.0 * rep2["fp32_library_split_k"], rep2


@pytest.mark.parametrize("J,C,tiles", [(128, 128, 37), (128, 256, 21), (64, 128, 9), (64, 256, 5), (256, 128, 4)])
def test_typed_weight_gradient_tiles_from_three_limbs(gpu_device, J, C, tiles):
    """relgnn_limb_gemm_tn_tiles_f32 (round 5): part[z] = A[a_rows[tile z]]^T @ G[tile z] over 512-row tiles of a compact pair table —
    gathered rows (padding = -1 = zeros, repeated nodes, whole tiles of padding), both operands split into three bf16 limbs in
    flight.  Against float64, next to the exact-fp32 panel kernel it replaces in ops.typed_linear's backward; magnitudes spread
    over six decades per column (gradient-like)."""
    from tf_gnn_samples_amd import config, dense as DN
    if not config.settings.limb_gemm:
        pytest.skip("the gathered three-limb TN belongs to RELGNN_GEMM=limb (limb_tn_tiles_supported says no under any other route)")
    torch.manual_seed(J * 7 + C + tiles)
    chunk = 512
    P, N = tiles * chunk, 3000
    A = torch.randn(N, J, device=gpu_device)
    G = torch.randn(P, C, device=gpu_device) * torch.exp(torch.empty(1, C, device=gpu_device).uniform_(-7, 7))
    rows = torch.randint(0, N, (P,), device=gpu_device, dtype=torch.int32)
    rows[torch.rand(P, device=gpu_device) < 0.2] = -1
    if tiles > 4:
        rows[3 * chunk:4 * chunk] = -1                                # a tile of padding only
        rows[chunk:chunk + 100] = 17                                  # one node many times
    assert DN.limb_tn_tiles_supported(A, G, rows, chunk)
    got = DN.limb_gemm_tn_tiles(A, G, rows, chunk)
    Ag = torch.where(rows.view(-1, 1) >= 0, A.double()[rows.clamp(min=0).long()], torch.zeros((), dtype=torch.float64, device=gpu_device))
    want = torch.einsum("tkj,tkc->tjc", Ag.view(tiles, chunk, J), G.double().view(tiles, chunk, C))
    panel = DN.panel_gemm(DN.GEMM_TN, A, G, a_rows=rows, batch=tiles, strides=(0, chunk * C, J * C), dims=(J, C, chunk))
    assert got.shape == want.shape == panel.shape
    scale = want.abs().amax(dim=(0, 1), keepdim=True).clamp(min=1e-30)            # per output column (its gradient column's magnitude)
    e_limb = float(((got.double() - want).abs() / scale).max())
    e_panel = float(((panel.double() - want).abs() / scale).max())
    assert e_limb <= max(2.0 * e_panel, 2e-6), (e_limb, e_panel)
    if tiles > 4:
        assert float(got[3].abs().max()) == 0.0
    # strided operands (a column block of a wider table) and a non-contiguous gradient fall back or are taken in place
    wide = torch.randn(N, J + 64, device=gpu_device)
    Aw = wide[:, :J]
    if DN.limb_tn_tiles_supported(Aw, G, rows, chunk):
        got_w = DN.limb_gemm_tn_tiles(Aw, G, rows, chunk)
        Awg = torch.where(rows.view(-1, 1) >= 0, Aw.double()[rows.clamp(min=0).long()], torch.zeros((), dtype=torch.float64, device=gpu_device))
        want_w = torch.einsum("tkj,tkc->tjc", Awg.view(tiles, chunk, J), G.double().view(tiles, chunk, C))
        assert float(((got_w.double() - want_w).abs() / want_w.abs().amax(dim=(0, 1), keepdim=True).clamp(min=1e-30)).max()) <= 2e-6


@pytest.mark.parametrize("act", ["tanh", "relu", "leaky_relu", "elu", "selu"])
def test_activation_gradient_in_the_input_gradient_epilogue(gpu_device, act):
    """relgnn_limb_gemm_xf32_dact / relgnn_limb16_gemm_xf32_dact (round 5): (g @ W^T) * act'(y) in the product's epilogue is the
    same bits as the product followed by relgnn_act_bwd_from_output — plain Dense shape (K = 256) and the aggregate-first layer's
    grouped K = 768 product, bf16 triples and fp16 pairs, rows that do not fill the last panel."""
    from tf_gnn_samples_amd import _lib, dense as DN, ops
    torch.manual_seed(5)
    a = ops.activation_id(act)
    V = 4096 + 77
    y = torch.tanh(torch.randn(V, 256, device=gpu_device)) if act == "tanh" else \
        torch.nn.functional.elu(torch.randn(V, 256, device=gpu_device)) if act in ("elu", "selu") else \
        torch.relu(torch.randn(V, 256, device=gpu_device)) if act == "relu" else \
        torch.nn.functional.leaky_relu(torch.randn(V, 256, device=gpu_device), 0.2)
    W = (torch.randn(256, 256, device=gpu_device) * 0.06).requires_grad_(True)          # a Dense kernel [in, out]
    g = torch.randn(V, 256, device=gpu_device)
    fused = DN.lib_gemm(DN.GEMM_NT, g, W, weight=True, premask=(a, y))
    plain = DN.lib_gemm(DN.GEMM_NT, g, W, weight=True)
    assert torch.equal(fused, DN.act_bwd_from_output(a, y, plain))
    # grouped: three per-type kernels, g [V, 768]
    Ws = [(torch.randn(256, 256, device=gpu_device) * 0.06).requires_grad_(True) for _ in range(3)]
    gT = torch.randn(V, 768, device=gpu_device)
    fused = DN.grouped_nt_gemm(gT, Ws, premask=(a, y))
    assert torch.equal(fused, DN.act_bwd_from_output(a, y, DN.grouped_nt_gemm(gT, Ws)))
    gmax = gT.abs().view(V, 3, 256).amax(2).contiguous().view(-1)
    fused = DN.grouped_nt_gemm(gT, Ws, xmax=gmax, xgroups=3, premask=(a, y))
    assert torch.equal(fused, DN.act_bwd_from_output(a, y, DN.grouped_nt_gemm(gT, Ws, xmax=gmax, xgroups=3)))
    # a route without the epilogue applies the factor as a pass: same function
    small = DN.lib_gemm(DN.GEMM_NT, g[:100], W, weight=True, premask=(a, y[:100]))
    want = (g[:100].double() @ W.detach().double().t()) * {
        "tanh": lambda t: 1 - t * t, "relu": lambda t: (t > 0).double(), "leaky_relu": lambda t: torch.where(t > 0, 1.0, 0.2),
        "elu": lambda t: torch.where(t > 0, torch.ones_like(t), t + 1), "selu": lambda t: torch.where(
            t > 0, torch.full_like(t, 1.0507009873554805), t + 1.7580993408473768)}[act](y[:100].double())
    assert float((small.double() - want).abs().max()) <= 2e-5


def test_folded_activation_gradients_leave_the_training_step_s_bits_alone(gpu_device, monkeypatch):
    """The premask protocol end to end (dense.py): a 3-layer RGCN + PPI head on a batch tall enough for the limb kernels — ReLU
    layers, tanh Dense between layers, tanh projection — with the activation gradients folded into the input-gradient products
    (two ReLU' and two tanh' passes disappear) against the same step with the folding refused (every producer runs its own pass):
    the SAME bits in the loss and in every gradient, and the fused step launches fewer relgnn_act_bwd_from_output passes."""
    from tf_gnn_samples_amd import dense as DN
    from tf_gnn_samples_amd.graph import clear_graph_cache
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(3, 1, seed=3)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes >= 4096
    calls = {"n": 0}
    real = DN.act_bwd_from_output

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(DN, "act_bwd_from_output", counting)

    def step(fold: bool):
        clear_graph_cache()
        if not fold:
            monkeypatch.setattr(DN, "fusable_activation_of", lambda x, sole_reader=False: 0)
        p = RGCN_Model.default_params()
        p.update(hidden_size=256, graph_num_layers=3, graph_layer_input_dropout_keep_prob=1.0, random_seed=0)
        model = RGCN_Model(p, task, device=str(gpu_device))
        batch = DeviceBatch(mb, gpu_device)
        model.optimizer.zero_grad()
        calls["n"] = 0
        m = model.forward_batch(batch, training=True)
        m['loss'].backward()
        torch.cuda.synchronize()
        return float(m['loss']), {n: model.variables[n].grad.detach().clone() for n in model.variables.names()}, calls["n"]

    loss_f, grads_f, passes_f = step(True)
    loss_u, grads_u, passes_u = step(False)
    assert loss_f == loss_u
    for n in grads_u:
        assert torch.equal(grads_f[n], grads_u[n]), n
    from tf_gnn_samples_amd import config
    if config.settings.limb_gemm:            # (the passes fold into the LIMB products' epilogues; other routes keep them, same bits)
        assert passes_u == 5 and passes_f <= 2, (passes_u, passes_f)     # 3 ReLU' + 2 tanh' -> the head's ReLU' (+ nothing else)


def test_a_tanh_output_with_two_readers_is_never_folded(gpu_device):
    """The failure the two-word protocol exists for (found by the reference-run RMSProp case of a GGNN model in round 5): y = tanh(.)
    handed to a layer with the hander's word "only you read it", and the layer reads it TWICE (messages and cell).  Folding tanh'
    into one of the two input-gradient products would leave the producer multiplying the sum again.  Neither reader passes
    sole_reader, so nothing is folded and the gradients equal the plain composition; with ONE reader that says so, it is folded —
    same bits."""
    from tf_gnn_samples_amd import _lib, dense as DN
    torch.manual_seed(11)
    x0 = torch.randn(5000, 256, device=gpu_device)
    W0 = torch.randn(256, 256, device=gpu_device) * 0.06
    U0 = torch.randn(256, 256, device=gpu_device) * 0.06
    R0 = torch.randn(256, 256, device=gpu_device) * 0.06

    def run(two_readers: bool, words: bool):
        x = x0.clone().requires_grad_(True)
        W, U, R = (t.clone().requires_grad_(True) for t in (W0, U0, R0))
        y = DN.dense_act(x, W, None, _lib.ACT_TANH, sole_consumer=True)
        if two_readers:
            out = DN.dense(y, U, sole_reader=words) + DN.dense(y, R, sole_reader=words)
        else:
            out = DN.dense(y, U, sole_reader=words)
        out.square().sum().backward()
        return [t.grad.clone() for t in (x, W, U)] + ([R.grad.clone()] if two_readers else [])

    one_folded, one_plain = run(False, True), run(False, False)
    for a, b in zip(one_folded, one_plain):
        assert torch.equal(a, b)
    two = run(True, False)
    xd = x0.double().requires_grad_(True)
    Wd, Ud, Rd = (t.double().requires_grad_(True) for t in (W0, U0, R0))
    yd = torch.tanh(xd @ Wd)
    (yd @ Ud + yd @ Rd).square().sum().backward()
    for got, want in zip(two, (xd.grad, Wd.grad, Ud.grad, Rd.grad)):
        assert float((got.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_zero_padded_gradient_route(gpu_device):
    """The PPI head's backward (tasks/ppi_task.py:183-191 behind models/sparse_graph_model.py's last layer): the loss gradient
    written into rows zero-padded from 121 to 128 columns (relgnn_sigmoid_ce_bwd_padded) meets the head's kernel [256, 121] as a
    limb product with K = 128 — the weight image's last k-tile filled with zeros by the split — and the ReLU' of the layer below in
    the epilogue.  Against float64, and against the route it replaces (library product + relgnn_act_bwd_from_output)."""
    from tf_gnn_samples_amd import _lib, config, dense as DN
    from tf_gnn_samples_amd.tasks.ppi_task import _SigmoidCEStats
    dev = gpu_device
    V, D, C = 6000, 256, 121
    x = _rand((V, D), dev, 1).relu_()                       # the last GNN layer's output
    kernel = _rand((D, C), dev, 2, 0.1)
    bias = _rand((C,), dev, 3, 0.1)
    labels = (_rand((V, C), dev, 4) > 0.4).float()
    res = {}
    for pad in ("0", "1"):
        with config.override(head_pad=pad):
            xx = DN.mark_activation_output(x.clone().requires_grad_(True), _lib.ACT_RELU)
            kk, bb = kernel.clone().requires_grad_(True), bias.clone().requires_grad_(True)
            logits = DN.dense(xx, kk, bb)
            mean, total, f1, counts = _SigmoidCEStats.apply(logits, labels, 1.0 / V)
            total.backward()
            res[pad] = (xx.grad.clone(), kk.grad.clone(), bb.grad.clone())
    # float64 truth: g = sigmoid(logits) - labels; gx = (g @ kernel^T) * relu'(x)
    lg = x.double() @ kernel.double() + bias.double()
    g = torch.sigmoid(lg) - labels.double()
    gx = (g @ kernel.double().t()) * (x > 0).double()
    gk = x.double().t() @ g
    for pad in ("0", "1"):
        gxx, gkk, gbb = res[pad]
        scale = float(gx.abs().max())
        assert float((gxx.double() - gx).abs().max()) <= 2e-6 * max(scale, 1.0), pad
        assert float((gkk.double() - gk).abs().max()) <= 4e-6 * float(gk.abs().max()), pad
        assert float((gbb.double() - g.sum(0)).abs().max()) <= 4e-6 * float(g.sum(0).abs().max()), pad
    assert torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][2], res["1"][2])     # weight / bias gradients: the same kernels
    # the padded loss gradient itself: the unpadded kernel's values, zeros behind them
    lib = _lib.load_library()
    logits = (x @ kernel + bias).contiguous()
    one = torch.ones(1, device=dev)
    plain = torch.empty_like(logits)
    _lib.check(lib.relgnn_sigmoid_ce_bwd(logits.data_ptr(), labels.data_ptr(), logits.numel(), None, 0.0, one.data_ptr(),
                                         plain.data_ptr(), None), "relgnn_sigmoid_ce_bwd")
    padded = torch.full((V, 128), 7.0, device=dev)
    _lib.check(lib.relgnn_sigmoid_ce_bwd_padded(logits.data_ptr(), labels.data_ptr(), V, C, None, 0.0, one.data_ptr(),
                                                padded.data_ptr(), 128, None), "relgnn_sigmoid_ce_bwd_padded")
    assert torch.equal(padded[:, :C], plain) and float(padded[:, C:].abs().max()) == 0.0


def test_weight_image_with_a_ragged_last_k_tile(gpu_device):
    """relgnn_limb_split_multi_f32 on a [256, 121] matrix with unaligned rows (ld = 121) and on its transpose-read twin [121, 256]:
    k-tiles 0 .. 7, the last one zero behind column 121; the limbs add up to the fp32 values exactly."""
    from tf_gnn_samples_amd import dense as DN
    dev = gpu_device
    w = _rand((256, 121), dev, 5) * torch.logspace(-10, 10, 121, device=dev)
    x = _rand((5000, 128), dev, 6)
    x[:, 121:] = 0.0
    out = DN.limb_gemm_weight(x, w, DN.WEIGHT_NT)                       # x @ w_padded^T
    truth = x[:, :121].double() @ w.double().t()
    assert float((out.double() - truth).abs().max()) <= 4e-7 * float(truth.abs().max())
    wn = w.t().contiguous()                                             # [121, 256]: the forward operand, read transposed
    out2 = DN.limb_gemm_weight(x, wn, DN.WEIGHT_NN)
    assert torch.equal(out2, out)
    x[:, 121:] = 3.0                                                    # the zero k-tile really is zero: junk there changes nothing
    assert torch.equal(DN.limb_gemm_weight(x, w, DN.WEIGHT_NT), out)


@pytest.mark.parametrize("M", [4096, 4097, 4160, 5000, 36096])
@pytest.mark.parametrize("K,N", [(768, 256), (256, 768), (256, 256), (512, 256), (256, 1024), (128, 256), (384, 256), (128, 512)])
def test_producer_consumer_product_is_bit_identical(gpu_device, M, K, N):
    """relgnn_limb_gemm_xf32_pc (wave roles, LDS hand-over) against relgnn_limb_gemm_xf32(_dact): forward with bias + ReLU, input
    gradient with the activation-gradient epilogue; odd unit counts, rows % 32 != 0, a strided left operand."""
    import ctypes
    from tf_gnn_samples_amd import _lib, config, dense as DN
    dev = gpu_device
    wide = _rand((M, K + 32), dev, M + K)
    a = wide[:, 16:16 + K]                                  # row stride K + 32, 16-byte aligned
    a[7, 3] = -torch.finfo(torch.float32).max               # the saturating split
    w = [_rand((N, 128), dev, 100 + i, 0.1) for i in range(K // 128)]       # NT operands side by side along k
    bias = _rand((N,), dev, 5, 0.1)
    y = _rand((M, N), dev, 6).relu_()
    outs = {}
    for pc in ("0", "1"):
        with config.override(limb_pc=pc):
            outs[pc] = (DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, bias, _lib.ACT_RELU),
                        DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, bias, _lib.ACT_TANH),
                        DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_LINEAR, dact=_lib.ACT_RELU, dy=y),
                        DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_LINEAR, dact=_lib.ACT_TANH, dy=y))
    from tf_gnn_samples_amd import ops
    assert ops.handover_status() == 0
    for p, q in zip(outs["0"], outs["1"]):
        assert torch.equal(p, q)


@pytest.mark.parametrize("K,N,act", [(768, 256, "relu"), (768, 256, "linear"), (256, 256, "tanh"), (128, 256, "tanh"), (512, 256, "tanh"),
                                     (256, 768, "relu"), (128, 1024, "linear"), (256, 512, "tanh"), (384, 256, "relu"),
                                     (1024, 256, "linear")])
def test_every_wave_role_variant_against_float64_at_the_baseline_height(gpu_device, K, N, act):
    """relgnn_limb_gemm_xf32_pc — the DEFAULT forward product since round 5 — against float64 directly (not through the older HIP
    kernel), on the C2 batch's height (32 203 rows: 1007 units, an odd count, the last one ragged) for every instantiated variant:
    ReLU / linear at K = 768 (the layer's product), tanh at K in {128, 256, 512} (the Dense layers), several column chunks at
    K <= 256, and the activation-gradient epilogues.  Bar: 1e-5 absolute (north_star) with outputs up to ~|2|."""
    from tf_gnn_samples_amd import _lib, config, dense as DN, ops
    dev, M = gpu_device, 32203
    a = _rand((M, K), dev, 11 + K)
    w = [_rand((N, 128), dev, 300 + i, 0.08) for i in range(K // 128)]
    bias = _rand((N,), dev, 7, 0.1)
    act_id = {"relu": _lib.ACT_RELU, "linear": _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH}[act]
    assert _lib.load_library().relgnn_limb_gemm_xf32_pc_supported(act_id, M, N, K) == 1
    with config.override(limb_pc="1"):
        assert DN._limb_pc_ok(a, N, K, bias, act_id, None, torch.empty((M, N), device=dev), DN.WEIGHT_NT)
        got = DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, bias, act_id)
    W = torch.cat([x.double() for x in w], dim=1)                       # [N, K]
    z = a.double() @ W.t() + bias.double()
    want = {"relu": torch.relu, "linear": lambda t: t, "tanh": torch.tanh}[act](z)
    err = float((got.double() - want).abs().max())
    assert err <= 1e-5, (K, N, act, err, float(want.abs().max()))
    if act == "linear":                                                  # the input-gradient form: times act'(y) in the epilogue
        y = _rand((M, N), dev, 9)
        for dact, factor in ((_lib.ACT_RELU, (y > 0).double()), (_lib.ACT_TANH, 1.0 - y.double() ** 2)):
            with config.override(limb_pc="1"):
                g = DN.limb_gemm_weight(a, w, DN.WEIGHT_NT, None, _lib.ACT_LINEAR, dact=dact, dy=y)
            e = float((g.double() - (a.double() @ W.t()) * factor).abs().max())
            assert e <= 1e-5, (K, N, dact, e)
    assert ops.handover_status() == 0


def test_a_product_that_gives_up_on_a_hand_over_fails_the_next_metrics_fetch(gpu_device):
    """The wave-role kernel bounds every poll of its LDS counters; a wave whose poll runs out finishes with wrong numbers and ORs a
    bit into the CALLER's status block (include/relgnn.h: relgnn_limb_gemm_xf32_pc, `status`).  The model owns that block and reads
    it with every step's metrics copy (MetricsReadback): with the poll bound set to 1 — word 1 of the block, the debug knob — a
    C2-size forward product gives up and the step's metrics fetch raises instead of training on."""
    from tf_gnn_samples_amd import config as _config
    if not (_config.settings.limb_gemm and _config.settings.limb == "triple" and _config.settings.limb_pc != "0"):
        pytest.skip("the wave-role product kernel runs on the exact-split limb route only")
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.models.sparse_graph_model import MetricsReadback
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(4, 1, seed=3)
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3, graph_layer_input_dropout_keep_prob=1.0)
    model = RGCN_Model(p, task, device=str(gpu_device))
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes >= 4096                                          # tall enough for the limb kernels
    batch = DeviceBatch(mb, gpu_device)
    word = ops.handover_word(gpu_device)
    assert word is model.handover_word and word.dtype == torch.int32 and word.numel() == 2
    assert ops.handover_status() == 0
    good = MetricsReadback(model.train_step(batch)).get()               # the default bound: clean
    assert np.isfinite(good['loss'])
    word[1] = 1
    try:
        with torch.no_grad():                                            # (forward only: the garbage must not reach the weights)
            rb = MetricsReadback(model.forward_batch(batch, training=False))
        with pytest.raises(ops.HandoverError, match="gave up on an LDS hand-over"):
            rb.get()
        assert int(word[0].item()) == 0                                  # reported once, then cleared
        # the epoch loop of the model goes through the same fetch
        with pytest.raises(ops.HandoverError):
            model.test(task._loaded_data[DataFold.TRAIN], quiet=True)
    finally:
        word[1] = 0
        word[0] = 0
    again = MetricsReadback(model.train_step(batch)).get()
    assert np.isfinite(again['loss']) and ops.handover_status() == 0


def test_the_status_block_survives_a_captured_step(gpu_device):
    """A hipGraph replays the pointers it was captured with: the status block exists before the capture (the model allocates it
    when it is built), so a captured step reports like an eager one."""
    from tf_gnn_samples_amd import config as _config
    if not (_config.settings.limb_gemm and _config.settings.limb == "triple" and _config.settings.limb_pc != "0"):
        pytest.skip("the wave-role product kernel runs on the exact-split limb route only")
    from tf_gnn_samples_amd import ops
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(4, 1, seed=4)
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=2, graph_layer_input_dropout_keep_prob=1.0)
    model = RGCN_Model(p, task, device=str(gpu_device))
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, gpu_device)
    cap = model.capture_train_step(batch)
    cap.replay()
    assert cap.handover_status() == 0
    word = ops.handover_word(gpu_device)
    word[1] = 1
    try:
        cap.replay()
        torch.cuda.synchronize()
        assert cap.handover_status() & 12                               # bits 2 / 3: the product kernel's waves
    finally:
        word[1] = 0
        word[0] = 0


@pytest.mark.parametrize("model_name", ["RGCN", "GGNN", "RGAT", "RGIN", "GNN-FiLM", "GNN-Edge-MLP0"])
@pytest.mark.parametrize("res_every,dense_every", [(1, 1), (2, 1), (2, 2), (10000, 1), (10000, 2)])
def test_folding_changes_no_bit_of_any_model_s_step(gpu_device, monkeypatch, model_name, res_every, dense_every):
    """The folding of activation gradients rides on tags and on two 'I am the only reader' words (dense.py) — nothing at run time
    cross-checks them.  This does, for every model class x residual cadence x Dense cadence with tanh Dense layers (the
    non-idempotent case): one training step's loss and EVERY gradient with the folding allowed against the same step with every
    producer running its own pass, bit for bit.  A layer that reads its input twice and says sole_reader, or a driver loop that
    vouches for a tensor it keeps, shows up here as a differing gradient."""
    from tf_gnn_samples_amd import dense as DN
    from tf_gnn_samples_amd.graph import clear_graph_cache
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(2, 1, seed=5, mean_nodes=2400, std_nodes=100, min_nodes=2200, max_nodes=2600, fwd_edges_per_node=6.0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes >= 4096                                         # tall enough for the limb kernels' epilogues
    cls, extra = name_to_model_class(model_name)

    def step(fold: bool):
        clear_graph_cache()
        with monkeypatch.context() as mp:
            if not fold:
                mp.setattr(DN, "fusable_activation_of", lambda x, sole_reader=False: 0)
            p = cls.default_params()
            p.update(extra)
            p.update(hidden_size=256, graph_num_layers=4, graph_layer_input_dropout_keep_prob=1.0, random_seed=0,
                     graph_residual_connection_every_num_layers=res_every, graph_dense_between_every_num_gnn_layers=dense_every,
                     graph_model_activation_function="tanh")
            model = cls(p, task, device=str(gpu_device))
            batch = DeviceBatch(mb, gpu_device)
            model.optimizer.zero_grad()
            m = model.forward_batch(batch, training=True)
            m['loss'].backward()
            torch.cuda.synchronize()
            return (float(m['loss'].detach()),
                    {n: model.variables[n].grad.detach().clone() for n in model.variables.names() if model.variables[n].grad is not None})

    loss_f, grads_f = step(True)
    loss_u, grads_u = step(False)
    assert loss_f == loss_u and sorted(grads_f) == sorted(grads_u)
    for n in grads_u:
        assert torch.equal(grads_f[n], grads_u[n]), n


def test_panel_products_take_cached_weight_images_and_give_the_same_bits(gpu_device):
    """relgnn_limb_gemm_sel_xf32 (round 6: the 128-column panel products read their weights' limb images from the step's cache,
    dense.weight_image(separate=True)) against relgnn_limb_dense_sel_f32 (which splits a stacked copy of the weights in front of
    every product): typed forward with gathered rows + per-tile kernels, typed input gradient, a plain Dense with a tanh epilogue —
    the same bits; and the cache follows the weights (in-place write + weights_changed(): the next product uses the new values)."""
    from tf_gnn_samples_amd import config as _config
    if not _config.settings.limb_gemm:
        pytest.skip("the cached panel images belong to the limb route (RELGNN_GEMM is set to another route in this run)")
    from tf_gnn_samples_amd import _lib, dense as DN
    dev = gpu_device
    L, tiles, V = 7, 40, 9000
    P = tiles * 512
    g = torch.Generator(device="cpu").manual_seed(0)
    tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
    node = torch.randint(-1, V, (P,), generator=g).to(torch.int32).to(dev)              # (-1: a padding row -> zeros)
    H = _rand((V, 128), dev, 1)
    for Dout in (128, 256):
        Ws = [_rand((128, Dout), dev, 10 + l, 0.1) for l in range(L)]
        assert DN.sel_weights_cacheable(Ws, DN.GEMM_NN) and DN.sel_weights_cacheable(Ws, DN.GEMM_NT)
        assert DN.sel_image(Ws, DN.GEMM_NN) is DN.sel_image(Ws, DN.GEMM_NN)           # the second lookup: by identity
        stacked = torch.stack(Ws)
        want = DN.limb_dense_sel(DN.GEMM_NN, H, stacked, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
        got = DN.limb_dense_sel(DN.GEMM_NN, H, Ws, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512, cached=True)
        assert torch.equal(got, want)
        gY = _rand((P, Dout), dev, 3)
        want = DN.limb_dense_sel(DN.GEMM_NT, gY, stacked, b_select=tile_type, rows_per_select=512)
        got = DN.limb_dense_sel(DN.GEMM_NT, gY, Ws, b_select=tile_type, rows_per_select=512, cached=True)
        assert torch.equal(got, want)
        # the cache follows the weights
        with torch.no_grad():
            Ws[2].mul_(-0.5)                                                          # moves the tensor's version
        stacked = torch.stack(Ws)
        assert torch.equal(DN.limb_dense_sel(DN.GEMM_NT, gY, Ws, b_select=tile_type, rows_per_select=512, cached=True),
                           DN.limb_dense_sel(DN.GEMM_NT, gY, stacked, b_select=tile_type, rows_per_select=512))
        Ws[3].data.copy_(Ws[3].data * 2.0)                                            # behind torch's back: the caller says so
        DN.weights_changed()
        stacked = torch.stack(Ws)
        assert torch.equal(DN.limb_dense_sel(DN.GEMM_NN, H, Ws, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512,
                                             cached=True),
                           DN.limb_dense_sel(DN.GEMM_NN, H, stacked, a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512))
    # a plain D = 128 Dense with tanh in the epilogue: cached image vs split-per-call, and vs float64
    x, k, b = _rand((9000, 128), dev, 5), _rand((128, 128), dev, 6, 0.1), _rand((128,), dev, 7, 0.1)
    a1 = DN.limb_dense_sel(DN.GEMM_NN, x, k, b, _lib.ACT_TANH, cached=True)
    a0 = DN.limb_dense_sel(DN.GEMM_NN, x, k, b, _lib.ACT_TANH)
    assert torch.equal(a1, a0)
    assert float((a1.double() - torch.tanh(x.double() @ k.double() + b.double())).abs().max()) <= 2e-6
    assert torch.equal(DN.lib_gemm(DN.GEMM_NN, x, k, b, weight=True, act=_lib.ACT_TANH), a0)      # the route the Dense layers take


def test_dense_multi_is_the_product_with_the_concatenated_kernels(gpu_device):
    """dense.dense_multi(x, [k_0 .. k_4]) = x @ [k_0 | .. | k_4] (gnns/ggnn.py:60-64,81 for every node) without the concatenated
    operand: forward and input gradient bit for bit against dense(x, torch.cat(...)), the five weight gradients against float64,
    each a dense tensor of its own."""
    from tf_gnn_samples_amd import config as _config
    if not _config.settings.limb_gemm:
        pytest.skip("the cached panel images belong to the limb route (RELGNN_GEMM is set to another route in this run)")
    from tf_gnn_samples_amd import dense as DN
    dev = gpu_device
    V, K, N, L = 20000, 128, 128, 5
    x = _rand((V, K), dev, 1).requires_grad_(True)
    ks = [_rand((K, N), dev, 20 + l, 0.1).requires_grad_(True) for l in range(L)]
    gy = _rand((V, L * N), dev, 2, 0.01)
    y = DN.dense_multi(x, ks)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith("_DenseMultiFn")
    y.backward(gy)
    got = (y.detach().clone(), x.grad.clone(), [k.grad.clone() for k in ks])
    assert all(k.grad.is_contiguous() for k in ks)
    x.grad = None
    for k in ks:
        k.grad = None
    y2 = DN.dense(x, torch.cat(ks, dim=1))
    y2.backward(gy)
    assert torch.equal(got[0], y2.detach()) and torch.equal(got[1], x.grad)
    for l in range(L):
        want = x.detach().double().t() @ gy[:, l * N:(l + 1) * N].double()
        assert float((got[2][l].double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), l
        assert float((ks[l].grad.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), l


@pytest.mark.parametrize("Dout,kind", [(128, "nn"), (256, "nn"), (128, "nt"), (256, "nt")])
@pytest.mark.parametrize("tiles", [1, 7, 40, 300])
def test_typed_products_on_the_wave_role_kernel_give_the_same_bits(gpu_device, tiles, Dout, kind):
    """relgnn_limb_gemm_sel_pc_xf32 (csrc/limb_gemm_pc_typed.hip: gathering producer waves + barrier-free matrix waves) against the
    panel kernels behind relgnn_limb_gemm_sel_xf32 on the per-(node, type) products of a many-type graph: forward (gathered rows,
    padding rows, per-tile kernels; N = 128 and 256) and input gradient (K = 128 and 256) — bit for bit, and the hand-over status
    stays clean; 1 .. 300 tiles of 512 rows (fewer pairs than workgroups, ragged ranges, several pairs per workgroup)."""
    from tf_gnn_samples_amd import config, dense as DN, ops
    dev = gpu_device
    L, V = 5, 3000
    P = tiles * 512
    g = torch.Generator(device="cpu").manual_seed(tiles + Dout)
    tile_type = torch.sort(torch.randint(0, L, (tiles,), generator=g)).values.to(torch.int32).to(dev)
    node = torch.randint(-1, V, (P,), generator=g).to(torch.int32).to(dev)
    Ws = [_rand((128, Dout), dev, 10 + l, 0.1) for l in range(L)]
    if kind == "nn":
        H = _rand((V, 128), dev, 1)
        H[5, 3] = -torch.finfo(torch.float32).max                                       # the saturating split
        args = dict(a_rows=node, num_rows=P, b_select=tile_type, rows_per_select=512)
        a, layout = H, DN.GEMM_NN
    else:
        a, layout = _rand((P, Dout), dev, 3), DN.GEMM_NT
        args = dict(b_select=tile_type, rows_per_select=512)
    outs = {}
    for pc in ("0", "1"):
        with config.override(typed_pc=pc):
            outs[pc] = DN.limb_dense_sel(layout, a, Ws, image=DN.sel_image(Ws, layout), **args)
    assert ops.handover_status() == 0
    assert torch.equal(outs["1"], outs["0"])
    if tiles == 7:                                                                      # and against float64
        W = torch.stack(Ws).double()
        t = tile_type.long().repeat_interleave(512)
        if kind == "nn":
            rows = torch.where(node.long().unsqueeze(1) >= 0, a.double()[node.long().clamp(min=0)], torch.zeros(1, dtype=torch.float64, device=dev))
            want = torch.einsum("pk,pkn->pn", rows, W[t])
        else:
            want = torch.einsum("pn,pkn->pk", a.double(), W[t])
        scale = float(want.abs().max())
        assert float((outs["1"].double() - want).abs().max()) <= 4e-6 * max(scale, 1.0)
