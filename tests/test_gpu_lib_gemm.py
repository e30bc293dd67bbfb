"""relgnn_blaslt_gemm_f32 (csrc/blaslt_gemm.hip): the library GEMM with a cached solution must compute what torch.mm computes
for every layout, for node counts that share a cache bucket, with bias, accumulate and the strided-batched split-K form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(gen, *shape):
    return torch.rand(shape, device=gen.device, generator=gen) * 2 - 1


def _close(got, want, K):
    # fp32 products summed in a library-chosen order: compare against float64, tolerance ~ sqrt(K) ulps of the magnitude
    err = (got.double() - want).abs().max().item()
    assert err <= 2e-6 * max(1.0, float(np.sqrt(K))) * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_layouts_and_bucket_reuse(gpu_device, layout):
    from tf_gnn_samples_amd import dense as D
    gen = torch.Generator(device=gpu_device).manual_seed(layout)
    for V in (4099, 4100, 4611, 7000, 37, 1):                 # several node counts per 4096-row bucket, tiny ones
        for (n_in, n_out) in ((256, 768), (50, 256), (121, 256), (768, 256)):
            if layout == D.GEMM_NN:
                a, b = _rand(gen, V, n_in), _rand(gen, n_in, n_out)
                want = a.double() @ b.double()
            elif layout == D.GEMM_NT:
                a, b = _rand(gen, V, n_in), _rand(gen, n_out, n_in)
                want = a.double() @ b.double().t()
            else:
                a, b = _rand(gen, V, n_in), _rand(gen, V, n_out)
                want = a.double().t() @ b.double()
            got = D.lib_gemm(layout, a, b)
            assert got.shape == want.shape
            _close(got, want, a.shape[1] if layout != D.GEMM_TN else V)


def test_bias_accumulate_and_row_strided_operands(gpu_device):
    from tf_gnn_samples_amd import dense as D
    gen = torch.Generator(device=gpu_device).manual_seed(7)
    a, b, bias = _rand(gen, 5000, 256), _rand(gen, 256, 121), _rand(gen, 121)
    _close(D.lib_gemm(D.GEMM_NN, a, b, bias), a.double() @ b.double() + bias.double(), 256)
    _close(D.lib_gemm(D.GEMM_NN, a, b, relu=True), (a.double() @ b.double()).clamp_(min=0), 256)
    _close(D.lib_gemm(D.GEMM_NN, a, b, bias, relu=True), (a.double() @ b.double() + bias.double()).clamp_(min=0), 256)
    out = _rand(gen, 5000, 121)
    want = out.double() + a.double() @ b.double()
    _close(D.lib_gemm(D.GEMM_NN, a, b, out=out, accumulate=True), want, 256)
    wide = _rand(gen, 5000, 768)
    view = wide[:, 256:512]                                     # rows dense, row stride 768
    _close(D.lib_gemm(D.GEMM_NN, view, b), view.double() @ b.double(), 256)
    _close(D.lib_gemm(D.GEMM_TN, view, a), view.double().t() @ a.double(), 5000)


@pytest.mark.parametrize("V", [36411, 30011, 5000, 700])
def test_split_k_weight_gradient(gpu_device, V):
    from tf_gnn_samples_amd import dense as D
    gen = torch.Generator(device=gpu_device).manual_seed(V)
    for M, N in ((768, 256), (256, 256), (256, 121), (50, 256)):
        a, b = _rand(gen, V, M), _rand(gen, V, N)
        _close(D.matmul_tn_splitk(a, b), a.double().t() @ b.double(), V)


def test_dense_autograd_matches_torch(gpu_device):
    from tf_gnn_samples_amd.dense import dense
    gen = torch.Generator(device=gpu_device).manual_seed(3)
    x, k, bias = _rand(gen, 9001, 256), _rand(gen, 256, 121), _rand(gen, 121)
    xs = [t.clone().requires_grad_(True) for t in (x, k, bias)]
    ys = [t.clone().requires_grad_(True) for t in (x, k, bias)]
    out = dense(*xs)
    ref = torch.addmm(ys[2], ys[0], ys[1])
    g = _rand(gen, *out.shape)
    out.backward(g)
    ref.backward(g)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4)
    for got, want in zip(xs, ys):
        assert torch.allclose(got.grad, want.grad, rtol=1e-4, atol=2e-3), (got.grad - want.grad).abs().max()


@pytest.mark.parametrize("V", [36411, 4099, 130, 17, 2, 1])
def test_streaming_weight_gradient_kernel(gpu_device, V):
    """relgnn_gemm_tn_stream_f32: every output shape class (full tiles, ragged M / N, odd row strides -> 4-byte loads,
    row-strided views), chunk tails (odd V, V smaller than one unrolled step), bit-reproducible."""
    from tf_gnn_samples_amd import dense as D
    gen = torch.Generator(device=gpu_device).manual_seed(V)
    for M, N in ((256, 256), (256, 121), (50, 256), (121, 50), (128, 128), (1, 1), (65, 63), (768, 256)):
        a, b = _rand(gen, V, M), _rand(gen, V, N)
        got = D.tn_stream_gemm(a, b)
        _close(got, a.double().t() @ b.double(), V)
        assert torch.equal(got, D.tn_stream_gemm(a, b))
    wide = _rand(gen, V, 512)
    a, b = wide[:, 3:131], wide[:, 256:512]            # odd base offset -> scalar loads; even -> 8-byte loads
    _close(D.tn_stream_gemm(a, b), a.double().t() @ b.double(), V)
    acc = _rand(gen, 128, 256)
    want = acc.double() + a.double().t() @ b.double()
    _close(D.tn_stream_gemm(a, b, out=acc), want, V)    # accumulate into an existing product


@pytest.mark.parametrize("V", [51037, 4099, 130, 3])
@pytest.mark.parametrize("M,bc,L", [(128, 128, 5), (128, 128, 1), (64, 50, 3), (50, 121, 2), (256, 64, 4)])
def test_streaming_weight_gradient_into_blocks(gpu_device, V, M, bc, L):
    """relgnn_gemm_tn_stream_blocks_f32: the L column blocks of ONE a^T @ b as L dense matrices (the per-edge-type kernel gradients
    of gnns/ggnn.py:63-67,80-81) — the same numbers, bit for bit, as the one-matrix entry gives for the whole product (same
    chunking and summation order), each block contiguous, and against float64."""
    from tf_gnn_samples_amd import dense as D
    gen = torch.Generator(device=gpu_device).manual_seed(V + M + L)
    a, b = _rand(gen, V, M), _rand(gen, V, L * bc)
    assert D.tn_stream_blocks_ok(a, b)
    got = D.tn_stream_blocks(a, b, L)
    assert got.shape == (L, M, bc) and got.is_contiguous()
    whole = D.tn_stream_gemm(a, b)
    for l, blk in enumerate(got.unbind(0)):
        assert blk.is_contiguous()
        assert torch.equal(blk, whole[:, l * bc:(l + 1) * bc]), l
        _close(blk, a.double().t() @ b[:, l * bc:(l + 1) * bc].double(), V)
    assert torch.equal(got, D.tn_stream_blocks(a, b, L))


@pytest.mark.parametrize("V", [49986, 4099, 130, 3])
def test_streaming_weight_gradients_as_one_group(gpu_device, V):
    """relgnn_gemm_tn_stream_group_f32 on a GRU cell's three weight gradients (gnns/ggnn.py:92: x^T gxk, h^T gxk[:, :2u],
    (r*h)^T gxk[:, 2u:], the last two written as column blocks of ONE [u, 3u] gradient) and its bias gradient (the column sums of
    gxk): against float64, reproducible, and nothing written outside the blocks."""
    from tf_gnn_samples_amd import dense as D
    u = 128
    gen = torch.Generator(device=gpu_device).manual_seed(V)
    x, h, rh, gxk = _rand(gen, V, u), _rand(gen, V, u), _rand(gen, V, u), _rand(gen, V, 3 * u)
    gq = gxk[:, 2 * u:].contiguous()
    gK = torch.full((u, 3 * u), 9.0, device=gpu_device)
    gU = torch.full((u, 3 * u), 9.0, device=gpu_device)
    gb = torch.full((3 * u,), 9.0, device=gpu_device)
    products = [(x, gxk, gK), (h, gxk[:, :2 * u], gU[:, :2 * u]), (rh, gq, gU[:, 2 * u:])]
    assert D.tn_stream_group_ok(products)
    D.tn_stream_group(products, colsum=gb)
    _close(gK, x.double().t() @ gxk.double(), V)
    _close(gU[:, :2 * u], h.double().t() @ gxk[:, :2 * u].double(), V)
    _close(gU[:, 2 * u:], rh.double().t() @ gq.double(), V)
    _close(gb, gxk.double().sum(0), V)
    first = (gK.clone(), gU.clone(), gb.clone())
    D.tn_stream_group(products, colsum=gb)
    assert all(torch.equal(a, b) for a, b in zip(first, (gK, gU, gb)))
    # a group of one without the column sums, a group of four
    one = torch.empty((u, 3 * u), device=gpu_device)
    D.tn_stream_group([(x, gxk, one)])
    _close(one, x.double().t() @ gxk.double(), V)
    outs = [torch.empty((u, 64), device=gpu_device) for _ in range(4)]
    D.tn_stream_group([(x, gxk[:, 64 * i:64 * (i + 1)], outs[i]) for i in range(4)], colsum=gb[:64].clone())
    for i in range(4):
        _close(outs[i], x.double().t() @ gxk[:, 64 * i:64 * (i + 1)].double(), V)


def test_streaming_group_refuses_what_it_cannot_take(gpu_device):
    import ctypes
    from tf_gnn_samples_amd import _lib, dense as D
    lib = _lib.load_library()
    a = torch.zeros((64, 50), device=gpu_device)
    b = torch.zeros((64, 128), device=gpu_device)
    out = torch.zeros((50, 128), device=gpu_device)
    assert not D.tn_stream_group_ok([(a, b, out)])                # 50 columns: not whole tiles
    vp, i64, i32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int32 * 1
    ws = torch.zeros(1 << 20, device=gpu_device)
    rc = lib.relgnn_gemm_tn_stream_group_f32(1, vp(a.data_ptr()), i64(50), vp(b.data_ptr()), i64(128), vp(out.data_ptr()), i64(128),
                                             i32(50), i32(128), 64, None, ws.data_ptr(), ws.numel() * 4, None)
    assert rc == _lib.EUNSUPPORTED
    rc = lib.relgnn_gemm_tn_stream_group_f32(5, vp(a.data_ptr()), i64(50), vp(b.data_ptr()), i64(128), vp(out.data_ptr()), i64(128),
                                             i32(50), i32(128), 64, None, ws.data_ptr(), ws.numel() * 4, None)
    assert rc == _lib.EINVAL                                      # more than four products
    a64 = torch.zeros((64, 64), device=gpu_device)
    o64 = torch.ones((64, 128), device=gpu_device)
    cs = torch.ones(128, device=gpu_device)
    rc = lib.relgnn_gemm_tn_stream_group_f32(1, vp(a64.data_ptr()), i64(64), vp(b.data_ptr()), i64(128), vp(o64.data_ptr()), i64(128),
                                             i32(64), i32(128), 0, cs.data_ptr(), None, 0, None)
    assert rc == _lib.OK                                          # no rows: zeros
    torch.cuda.synchronize()
    assert float(o64.abs().sum()) == 0.0 and float(cs.abs().sum()) == 0.0


def test_streaming_weight_gradient_into_blocks_refuses_ragged_blocks(gpu_device):
    import ctypes
    from tf_gnn_samples_amd import _lib
    lib = _lib.load_library()
    a = torch.zeros((64, 32), device=gpu_device)
    b = torch.zeros((64, 100), device=gpu_device)
    out = torch.ones(4096, device=gpu_device)
    ws = torch.zeros(1 << 20, device=gpu_device)
    rc = lib.relgnn_gemm_tn_stream_blocks_f32(a.data_ptr(), 32, b.data_ptr(), 100, out.data_ptr(), 33, 32 * 33, 32, 100, 33, 64, 0,
                                              ws.data_ptr(), ws.numel() * 4, None)
    assert rc == _lib.EINVAL                      # 100 columns are not a whole number of 33-column blocks
    rc = lib.relgnn_gemm_tn_stream_blocks_f32(a.data_ptr(), 32, b.data_ptr(), 100, out.data_ptr(), 50, 32 * 50, 32, 100, 50, 0, 0,
                                              ws.data_ptr(), ws.numel() * 4, None)
    assert rc == _lib.OK                          # no rows: the two [32, 50] blocks are zeroed, nothing behind them
    torch.cuda.synchronize()
    assert float(out[:3200].abs().sum()) == 0.0 and float(out[3200:].sum()) == 896.0


def test_streaming_kernel_random_shapes(gpu_device):
    """Seeded sweep over node counts, output shapes and row strides (aligned / odd, views into wider storage): the kernel's
    load ring is hand-scheduled assembly, so every combination of its four load variants, of lean and masked passes and of
    chunk tails is compared with float64."""
    from tf_gnn_samples_amd import dense as D
    rng = np.random.default_rng(1234)
    gen = torch.Generator(device=gpu_device).manual_seed(99)
    for case in range(48):
        V = int(rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 500, 1023, 1024, 1025, 4097, 20011, 65537]))
        M = int(rng.choice([1, 2, 31, 32, 50, 63, 64, 65, 121, 128, 200, 256]))
        N = int(rng.choice([1, 3, 32, 50, 64, 100, 121, 128, 129, 255, 256]))
        pad_a, pad_b = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        off_a, off_b = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        wa = _rand(gen, V, off_a + M + pad_a)
        wb = _rand(gen, V, off_b + N + pad_b)
        a, b = wa[:, off_a:off_a + M], wb[:, off_b:off_b + N]
        got = D.tn_stream_gemm(a, b)
        want = a.double().t() @ b.double()
        err = (got.double() - want).abs().max().item()
        tol = 2e-6 * max(1.0, float(np.sqrt(V))) * max(1.0, want.abs().max().item())
        assert got.shape == (M, N) and err <= tol, (case, V, M, N, off_a, pad_a, off_b, pad_b, err, tol)


@pytest.mark.parametrize("bias", [False, True])
def test_dense_relu_matches_two_step(gpu_device, bias):
    """dense_relu = one GEMM with a ReLU epilogue; value and all three gradients against relu(x @ k + b) in torch."""
    from tf_gnn_samples_amd.dense import dense_relu
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(1000, 96, generator=g).to(gpu_device).requires_grad_(True)
    k = (torch.randn(96, 132, generator=g) * 0.1).to(gpu_device).requires_grad_(True)
    b = torch.randn(132, generator=g).to(gpu_device).requires_grad_(True) if bias else None
    go = torch.randn(1000, 132, generator=g).to(gpu_device)
    y = dense_relu(x, k, b)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), k.grad.clone()] + ([b.grad.clone()] if bias else [])
    x.grad = k.grad = None
    if bias:
        b.grad = None
    ref = torch.relu(x.double() @ k.double() + (b.double() if bias else 0.0))
    ref.backward(go.double())
    want = [ref.detach(), x.grad, k.grad] + ([b.grad] if bias else [])
    for a, w in zip(got, want):
        assert float((a.double() - w.double()).abs().max()) <= 2e-5 * max(1.0, float(w.abs().max()))
