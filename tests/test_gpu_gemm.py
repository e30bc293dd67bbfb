"""relgnn_gemm_f32 (csrc/gemm_f32.hip): the exact-fp32 MFMA GEMM behind the node-side Dense layers, against fp64
matmuls of the same operands.  Covers the three operand layouts (forward NN, input gradient NT, weight gradient TN
with split-K), every tile configuration the dispatcher can pick, ragged M / K tails, strided rows, the bias +
activation epilogue, and an ASYMMETRIC operand so that a transposed output tile cannot pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _route_gemms_through_the_mfma_kernel(monkeypatch):
    """The package's default for the node-side GEMMs is the library (measured faster at K = 256); these tests exercise
    the hand-written kernel, i.e. what RELGNN_GEMM=mfma selects."""
    from tf_gnn_samples_amd import dense
    monkeypatch.setattr(dense, "_OWN_GEMM", True)


def _err(out, ref):
    return float((out.double().cpu() - ref).abs().max())


def _budget(K, a, b):
    # fp32 roundoff of a K-term dot product of O(1) operands: ~1e-7 * sqrt(K) * |a||b|; generous factor
    return 2e-6 * max(1.0, K ** 0.5) * float(a.abs().max()) * float(b.abs().max())


@pytest.mark.parametrize("M,K,N", [(36411, 256, 768), (36411, 256, 256), (1000, 768, 256), (150, 64, 64), (333, 32, 128),
                                   (257, 260, 132), (64, 4, 4), (5, 256, 768)])
def test_gemm_nn_nt_against_fp64(gpu_device, M, K, N):
    from tf_gnn_samples_amd import dense as D
    g = torch.Generator(device=gpu_device).manual_seed(M + K + N)
    a = torch.rand((M, K), device=gpu_device, generator=g) * 2 - 1
    b = torch.rand((K, N), device=gpu_device, generator=g) * 2 - 1
    b[:, 0] += 3.0          # asymmetric: column 0 stands out
    assert D.own_gemm_supported(D.GEMM_NN, a, b)
    out = D.own_gemm(D.GEMM_NN, a, b)
    ref = a.double().cpu() @ b.double().cpu()
    assert out.shape == (M, N) and _err(out, ref) <= _budget(K, a, b)
    bt = b.t().contiguous()                                   # [N, K]: C = A @ Bt^T
    assert D.own_gemm_supported(D.GEMM_NT, a, bt)
    out = D.own_gemm(D.GEMM_NT, a, bt)
    assert _err(out, ref) <= _budget(K, a, b)


@pytest.mark.parametrize("V,Kin,N", [(36411, 256, 768), (36411, 256, 256), (5000, 128, 64), (777, 64, 192), (31, 256, 128),
                                     (3000, 52, 36)])
def test_gemm_tn_split_k_against_fp64(gpu_device, V, Kin, N):
    from tf_gnn_samples_amd import dense as D
    g = torch.Generator(device=gpu_device).manual_seed(V + Kin + N)
    x = torch.rand((V, Kin), device=gpu_device, generator=g) * 2 - 1
    gr = torch.rand((V, N), device=gpu_device, generator=g) * 2 - 1
    gr[:, 1] *= 4.0
    assert D.own_gemm_supported(D.GEMM_TN, x, gr)
    out = D.matmul_tn_splitk(x, gr)
    ref = x.double().cpu().t() @ gr.double().cpu()
    assert out.shape == (Kin, N) and _err(out, ref) <= _budget(V, x, gr)
    out2 = D.matmul_tn_splitk(x, gr)
    assert torch.equal(out, out2)                              # fixed-order split-K sum: bit-deterministic


def test_gemm_epilogue_strided_rows_and_unsupported_shapes(gpu_device):
    from tf_gnn_samples_amd import _lib, dense as D
    g = torch.Generator(device=gpu_device).manual_seed(5)
    big = torch.rand((500, 512), device=gpu_device, generator=g) - 0.5
    a = big[:, 128:384]                                        # row stride 512, 16-byte aligned start
    w = torch.rand((256, 128), device=gpu_device, generator=g) - 0.5
    bias = torch.rand(128, device=gpu_device, generator=g)
    assert D.own_gemm_supported(D.GEMM_NN, a, w)
    for act, fn in ((_lib.ACT_LINEAR, lambda t: t), (_lib.ACT_TANH, torch.tanh), (_lib.ACT_RELU, torch.relu)):
        out = D.own_gemm(D.GEMM_NN, a, w, bias, act)
        ref = fn(a.double().cpu() @ w.double().cpu() + bias.double().cpu())
        assert _err(out, ref) <= 1e-5
    # shapes outside the kernel's contract go to the library (K = 50: rows are not 16-byte multiples; N = 121)
    assert not D.own_gemm_supported(D.GEMM_NN, torch.zeros((10, 50), device=gpu_device), torch.zeros((50, 256), device=gpu_device))
    assert not D.own_gemm_supported(D.GEMM_NN, torch.zeros((10, 256), device=gpu_device), torch.zeros((256, 121), device=gpu_device))
    assert not D.own_gemm_supported(D.GEMM_NN, a[:, 1:], w[1:])  # misaligned row start


def test_dense_autograd_through_the_mfma_gemm(gpu_device):
    """dense() forward / input gradient / weight gradient / bias gradient against fp64 autograd."""
    from tf_gnn_samples_amd.dense import dense
    g = torch.Generator(device=gpu_device).manual_seed(9)
    x = (torch.rand((2111, 256), device=gpu_device, generator=g) - 0.5).requires_grad_(True)
    w = (torch.rand((256, 384), device=gpu_device, generator=g) - 0.5).requires_grad_(True)
    b = torch.rand(384, device=gpu_device, generator=g).requires_grad_(True)
    go = torch.rand((2111, 384), device=gpu_device, generator=g) - 0.5
    dense(x, w, b).backward(go)
    xr, wr, br = (t.detach().double().cpu().requires_grad_(True) for t in (x, w, b))
    (xr @ wr + br).backward(go.double().cpu())
    for got, ref in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        assert _err(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
