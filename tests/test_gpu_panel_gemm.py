"""relgnn_panel_gemm_f32 (csrc/panel_gemm.hip: row-panel exact-fp32 MFMA GEMM with direct-to-LDS staging) against float64
products: the three layouts, row counts that do not fill panels, K tails, bias + activation epilogues, gathered rows with
padding (-1), per-tile weight selection, independent batches, split-K slabs, gathered reduction rows.  Asymmetric random
operands (a transposed fragment or output tile cannot pass)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NN, NT, TN = 0, 1, 2


def _rand(gen, *shape):
    return torch.rand(shape, device=gen.device, generator=gen) * 2 - 1


def _close(got, want, K):
    want = want.to(torch.float64)
    err = float((got.to(torch.float64) - want).abs().max())
    scale = float(want.abs().max()) + 1e-30
    assert err <= 4e-7 * np.sqrt(K) * scale + 1e-6, (err, scale)


@pytest.mark.parametrize("M,N,K", [(36096, 256, 768), (32203, 256, 256), (1000, 128, 128), (17, 256, 64), (4097, 384, 132),
                                   (300, 768, 256), (50000, 128, 640), (129, 128, 20), (5000, 64, 128), (40000, 192, 64),
                                   (28001, 256, 768), (30017, 256, 256), (33333, 256, 768), (38911, 256, 768), (41000, 512, 128),
                                   (256 * 16 * 3, 256, 64), (256 * 16 * 5 + 1, 256, 64), (256 * 16 * 7 - 3, 256, 64)])
def test_nn_shapes(gpu_device, M, N, K):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(M + N + K)
    a, b = _rand(gen, M, K), _rand(gen, K, N)
    out = panel_gemm(NN, a, b)
    _close(out, a.double() @ b.double(), K)


@pytest.mark.parametrize("act", [0, 1, 2, 6])
def test_nn_bias_activation_epilogue(gpu_device, act):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(7 + act)
    a, b, bias = _rand(gen, 5000, 256), _rand(gen, 256, 256) * 0.1, _rand(gen, 256)
    out = panel_gemm(NN, a, b, bias, act)
    pre = a.double() @ b.double() + bias.double()
    want = {0: pre, 1: torch.tanh(pre), 2: torch.relu(pre), 6: torch.nn.functional.gelu(pre)}[act]
    _close(out, want, 256)


@pytest.mark.parametrize("M,N,K", [(36096, 256, 768), (999, 128, 256), (20000, 768, 256), (64, 128, 36), (3000, 64, 128),
                                   (33333, 256, 768), (7777, 128, 128)])
def test_nt_shapes(gpu_device, M, N, K):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(M + 3 * N + K)
    a, bt = _rand(gen, M, K), _rand(gen, N, K)
    out = panel_gemm(NT, a, bt)
    _close(out, a.double() @ bt.double().t(), K)


@pytest.mark.parametrize("K,M,N", [(36096, 768, 256), (32203, 768, 256), (5000, 128, 128), (1000, 256, 256), (999 * 4, 128, 384), (700, 64, 128),
                                   (2049, 128, 64), (8000, 320, 256)])
def test_tn_split_k(gpu_device, K, M, N):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(K + M + N)
    a, b = _rand(gen, K, M), _rand(gen, K, N)
    splits = 6
    chunk = ((K + splits - 1) // splits + 15) // 16 * 16
    splits = (K + chunk - 1) // chunk
    slabs = panel_gemm(TN, a, b, batch=splits, split_k_rows=chunk, dims=(M, N, K))
    assert slabs.shape == (splits, M, N) or splits == 1
    _close(slabs.double().reshape(-1, M, N).sum(0), a.double().t() @ b.double(), K)
    whole = panel_gemm(TN, a, b)
    _close(whole, a.double().t() @ b.double(), K)


def test_strided_rows_and_output(gpu_device):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(11)
    big_a, big_b = _rand(gen, 3000, 300), _rand(gen, 256, 400)
    a, b = big_a[:, 4:260], big_b[:, 8:264]               # 16-byte aligned column offsets, row strides 300 / 400
    big_out = torch.full((3000, 512), 7.0, device=gpu_device)
    out = big_out[:, 128:384]
    panel_gemm(NN, a, b, out=out)
    _close(out, a.double() @ b.double(), 256)
    assert float(big_out[:, :128].min()) == 7.0 and float(big_out[:, 384:].max()) == 7.0     # nothing outside the block


def test_gathered_rows_with_padding_and_typed_weights(gpu_device):
    """Y[r] = H[node[r]] @ W_type(r) over 512-row tiles of one type each, padding rows (-1) -> zeros: the compact
    (node, type) table transform of graph.SidePairs (gnns/gnn_film.py:92-106 on many-type graphs)."""
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(12)
    V, Din, L, tiles = 5000, 128, 7, 23
    H = _rand(gen, V, Din)
    for Dout in (64, 128, 256, 384):
        W = _rand(gen, L, Din, Dout) * 0.2
        tile_type = torch.randint(0, L, (tiles,), device=gpu_device, generator=gen, dtype=torch.int32)
        node = torch.randint(0, V, (tiles * 512,), device=gpu_device, generator=gen, dtype=torch.int32)
        pad = torch.rand(tiles * 512, device=gpu_device, generator=gen) < 0.1
        node = torch.where(pad, torch.full_like(node, -1), node)
        Y = panel_gemm(NN, H, W, a_rows=node, num_rows=tiles * 512, b_select=tile_type, rows_per_select=512)
        X = torch.where(pad.unsqueeze(1), torch.zeros((), device=gpu_device), H[node.clamp(min=0).long()])
        want = torch.bmm(X.double().view(tiles, 512, Din), W.double()[tile_type.long()]).view(-1, Dout)
        _close(Y, want, Din)
        assert float(Y[pad].abs().max()) == 0.0
        # input gradient of the same transform: dX[r] = dY[r] @ W_type(r)^T  (NT with the kernels as given)
        dY = _rand(gen, tiles * 512, Dout)
        dX = panel_gemm(NT, dY, W, b_select=tile_type, rows_per_select=512, dims=(tiles * 512, Din, Dout))
        want = torch.bmm(dY.double().view(tiles, 512, Dout), W.double()[tile_type.long()].transpose(1, 2)).view(-1, Din)
        _close(dX, want, Dout)
        # per-tile weight-gradient partials: part[z] = X_z^T @ dY_z with X_z = gathered rows of tile z
        part = panel_gemm(TN, H, dY, a_rows=node, batch=tiles, strides=(0, 512 * Dout, Din * Dout), dims=(Din, Dout, 512))
        want = torch.bmm(X.double().view(tiles, 512, Din).transpose(1, 2), dY.double().view(tiles, 512, Dout))
        _close(part, want, 512)


def test_independent_batches(gpu_device):
    from tf_gnn_samples_amd.dense import panel_gemm
    gen = torch.Generator(device=gpu_device).manual_seed(13)
    B_, M, K, N = 9, 512, 128, 256
    a, b = _rand(gen, B_, M, K), _rand(gen, B_, K, N)
    out = panel_gemm(NN, a, b, batch=B_, strides=(M * K, K * N, M * N), dims=(M, N, K))
    _close(out, torch.bmm(a.double(), b.double()), K)
    xt = _rand(gen, B_, M, K)          # TN per batch: [K_red = M rows, K cols]^T @ [M rows, N]
    g = _rand(gen, B_, M, N)
    out = panel_gemm(TN, xt, g, batch=B_, strides=(M * K, M * N, K * N), dims=(K, N, M))
    _close(out, torch.bmm(xt.double().transpose(1, 2), g.double()), M)


def test_unsupported_shapes_are_refused(gpu_device):
    from tf_gnn_samples_amd.dense import panel_gemm, panel_gemm_supported
    a, b = torch.zeros(10, 50, device=gpu_device), torch.zeros(50, 256, device=gpu_device)
    assert not panel_gemm_supported(NN, a, b)
    with pytest.raises(ValueError):
        panel_gemm(NN, a, b)
    a, b = torch.zeros(10, 64, device=gpu_device), torch.zeros(64, 121, device=gpu_device)
    assert not panel_gemm_supported(NN, a, b)
    a, b = torch.zeros(10, 64, device=gpu_device), torch.zeros(64, 96, device=gpu_device)
    assert not panel_gemm_supported(NN, a, b)
