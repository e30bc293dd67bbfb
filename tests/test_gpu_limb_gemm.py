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
    L, tiles, V = 23, 37, 5000
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
