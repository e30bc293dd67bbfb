"""Host logic of the round-5 product routes in dense.py — no kernel runs: the zero-padded tag of a gradient / feature tensor
(mark_zero_padded / zero_padded_operand), the k extent of a weight image whose matrix has a ragged last k-tile, and which products
take the wave-role kernel (relgnn_limb_gemm_xf32_pc) under each value of config.limb_pc."""
import pytest
import torch


def test_zero_padded_tag_lives_and_dies_with_the_tensor_object():
    from tf_gnn_samples_amd import dense as DN
    buf = torch.zeros((10, 128))
    buf[:, :121] = torch.randn(10, 121)
    g = DN.mark_zero_padded(buf[:, :121], 128)
    ap = DN.zero_padded_operand(g)
    assert ap is not None and ap.shape == (10, 128) and ap.data_ptr() == buf.data_ptr() and torch.equal(ap, buf)
    # whatever autograd (or anybody) derives from it is a new object without the tag: the plain route
    assert DN.zero_padded_operand(g.contiguous()) is None
    assert DN.zero_padded_operand(g * 2.0) is None
    assert DN.zero_padded_operand(g[1:]) is None
    assert DN.zero_padded_operand(torch.randn(10, 121)) is None
    # a tag that no longer describes its tensor is ignored: another view carrying a copied tag, a row stride that is not the padded one
    other = buf[:, :100]
    other._relgnn_zero_pad = g._relgnn_zero_pad
    assert DN.zero_padded_operand(other) is None
    wide = torch.zeros((10, 130))[:, :121]
    wide._relgnn_zero_pad = (wide.data_ptr(), tuple(wide.shape), wide.stride(0), 128)
    assert DN.zero_padded_operand(wide) is None
    # a padded width that is not a multiple of 16 is not a padded operand
    odd = DN.mark_zero_padded(torch.zeros((4, 120))[:, :100], 120)
    assert DN.zero_padded_operand(odd) is None


def test_weight_image_extent_of_a_single_matrix_rounds_its_k_up_to_a_k_tile():
    from tf_gnn_samples_amd import dense as DN
    head = torch.zeros((256, 121))                                  # the PPI head's kernel^T view: NT, K = 121
    assert DN._weight_image_shape([head], DN.WEIGHT_NT) == (256, 128)
    proj = torch.zeros((50, 256))                                   # the input projection: NN, K = 50
    assert DN._weight_image_shape([proj], DN.WEIGHT_NN) == (256, 64)
    layer = [torch.zeros((256, 256)) for _ in range(3)]
    assert DN._weight_image_shape(layer, DN.WEIGHT_NN) == (256, 768)
    # matrices side by side along k must each fill whole k-tiles: only a single matrix may be ragged
    items = DN._weight_image_items([head], DN.WEIGHT_NT, torch.zeros(1))
    assert items[0][6:] == (0, 8)                                   # k-tiles 0 .. 7 of 8


@pytest.mark.parametrize("mode,kind,n,k,act,expect", [
    ("fwd", "nn", 256, 768, 2, True),        # the layer's forward product, ReLU
    ("fwd", "nt", 256, 768, 0, False),       # its input gradient: next to the side stream's weight gradient
    ("1", "nt", 256, 768, 0, True),
    ("0", "nn", 256, 768, 2, False),
    ("fwd", "nn", 256, 256, 1, True),        # a tanh Dense layer
    ("fwd", "nn", 256, 768, 1, False),       # tanh: K in {128, 256, 512} only
    ("fwd", "nn", 768, 256, 0, True),        # several column chunks: K <= 256
    ("fwd", "nn", 768, 512, 0, False),
    ("fwd", "nn", 256, 640, 0, False),       # no variant for five half slabs
    ("fwd", "nn", 128, 256, 0, False),       # N % 256
    ("fwd", "nn", 256, 768, 5, False),       # selu: the old kernel's epilogue
])
def test_which_products_take_the_wave_role_kernel(mode, kind, n, k, act, expect):
    from tf_gnn_samples_amd import config, dense as DN
    a = torch.zeros((8192, k))
    out = torch.zeros((8192, n))
    with config.override(limb_pc=mode):
        assert DN._limb_pc_ok(a, n, k, None, act, None, out, kind) is expect
        assert DN._limb_pc_ok(a[:100], n, k, None, act, None, out[:100], kind) is False      # below the limb routes' minimum height


def test_a_backward_that_raises_leaves_no_deferred_join_state_behind():
    """ops.deferred_weight_gradient_join: when the backward inside it raises, nobody calls join_deferred() for that pass — the
    context forgets the pass itself, so that the next step's join does not verify (or keep alive) this step's parameters."""
    from tf_gnn_samples_amd import ops
    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(RuntimeError, match="boom"):
        with ops.deferred_weight_gradient_join():
            assert ops._DEFER["on"]
            ops._DEFER["handed"].append((p, 1234, 0))
            ops._DEFER["targets"].add(id(p))
            raise RuntimeError("boom")
    assert not ops._DEFER["on"] and ops._DEFER["handed"] == [] and not ops._DEFER["targets"] and ops._DEFER["pending"] == []
    ops.join_deferred()                                   # nothing to verify: does not raise
    # a clean exit keeps the pass for join_deferred()
    with ops.deferred_weight_gradient_join():
        ops._DEFER["targets"].add(id(p))
    assert ops._DEFER["targets"]
    ops.join_deferred()
    assert not ops._DEFER["targets"]


def test_combined_node_csr_of_both_pair_tables_sums_what_the_two_reductions_sum():
    """graph.PairTables.node_csr_both: one CSR node -> (its by-source table rows, then its by-target table rows + P_s) — the
    single reduction of ops._TypedLinearPair — gives every node the sum of the two separate reductions, the by-source rows first."""
    import types
    from tf_gnn_samples_amd.graph import PairTables
    g = torch.Generator().manual_seed(0)
    V = 50

    def side(P):
        cnt = torch.randint(0, 4, (V,), dtype=torch.int32, generator=g)
        rp = torch.zeros(V + 1, dtype=torch.int32)
        rp[1:] = torch.cumsum(cnt, 0)
        n = int(rp[-1])
        return types.SimpleNamespace(V=V, node_rowptr=rp, node_col=torch.randperm(P, generator=g)[:n].to(torch.int32), num_pairs=n, P=P)

    pt = PairTables.__new__(PairTables)
    pt.src, pt.tgt = side(200), side(300)
    rowptr, col = pt.node_csr_both()
    assert pt.node_csr_both()[1] is col                                 # built once per graph
    assert int(rowptr[-1]) == pt.src.num_pairs + pt.tgt.num_pairs and rowptr.dtype == col.dtype == torch.int32
    Xa, Xb = torch.randn(200, 3, dtype=torch.float64, generator=g), torch.randn(300, 3, dtype=torch.float64, generator=g)
    X = torch.cat([Xa, Xb])
    for v in range(V):
        rows_a = pt.src.node_col[pt.src.node_rowptr[v]:pt.src.node_rowptr[v + 1]].long()
        rows_b = pt.tgt.node_col[pt.tgt.node_rowptr[v]:pt.tgt.node_rowptr[v + 1]].long()
        mine = col[rowptr[v]:rowptr[v + 1]].long()
        assert mine.tolist() == rows_a.tolist() + (rows_b + 200).tolist()
        assert torch.allclose(X[mine].sum(0), Xa[rows_a].sum(0) + Xb[rows_b].sum(0))


def test_the_grouped_weight_gradient_and_the_cell_kernels_are_gpu_routes_only():
    """Host tensors never reach relgnn_gemm_tn_stream_group_f32 / _blocks_f32 or the GRU cell kernels: the route predicates say no
    and the cell falls back to the composition, whose numbers are the oracle's (oracle/tf_ops.py: gru_cell)."""
    import numpy as np
    import torch
    from oracle import tf_ops
    from tf_gnn_samples_amd import dense as DN, utils
    u = 128
    g = torch.Generator().manual_seed(0)
    x, h = torch.randn((40, u), generator=g), torch.rand((40, u), generator=g) * 2 - 1
    K, R = (torch.rand((u, 3 * u), generator=g) - 0.5) * 0.2, (torch.rand((u, 3 * u), generator=g) - 0.5) * 0.2
    b = torch.rand((3 * u,), generator=g) - 0.5
    gxk = torch.randn((40, 3 * u), generator=g)
    assert not DN.tn_stream_blocks_ok(x, gxk)
    assert not DN.tn_stream_group_ok([(x, gxk, torch.empty((u, 3 * u)))])
    assert not DN.tn_stream_group_ok([])
    assert not utils._gru_cell_kernel_ok(x, h, K, R, b, 1)
    cell = utils.get_gated_unit(u, "gru", "tanh", {"kernel": K, "recurrent_kernel": R, "bias": b})
    out = cell(x, [h])[0]
    want = tf_ops.gru_cell(x.numpy(), h.numpy(), K.numpy(), R.numpy(), b.numpy(), np.tanh)
    assert float(np.abs(out.numpy() - want).max()) <= 2e-6
