"""relgnn_gru_cell_fwd_xf32 (csrc/gru_cell.hip): the Keras GRUCell of sparse_ggnn_layer (gnns/ggnn.py:92 through utils/utils.py:15-16;
reset_after=False, hard_sigmoid gates, order z, r, h) as one wave-role kernel — both products, the gates, r * h handed to the
candidate's k-loop through LDS, the blend.

Checked against float64 (every output: z, r, r * h, the candidate, the new state), against the oracle's cell, against the composition
it replaces (three limb products + gru.hip's two elementwise kernels: same numbers to the last bits, gradients included), on panel
geometries from one row to C3's 49 986 (odd unit counts, rows % 32 != 0, fewer panels than CUs), row-strided operands, every
activation it takes, with and without the tensors the backward reads, and with rows that hold inf / NaN / float32 lowest."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

U = 128


def _weights(dev, seed, scale=0.15):
    g = torch.Generator(device="cpu").manual_seed(seed)
    K = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * scale).to(dev)
    R = ((torch.rand((U, 3 * U), generator=g) * 2 - 1) * scale).to(dev)
    b = ((torch.rand((3 * U,), generator=g) * 2 - 1) * 0.3).to(dev)
    return K, R, b


def _states(dev, V, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn((V, U), generator=g) * 1.5).to(dev), (torch.rand((V, U), generator=g) * 2 - 1).to(dev)


def _cell64(x, h, K, R, b, act):
    x, h, K, R, b = (t if t.dtype == torch.float64 else t.double() for t in (x, h, K, R, b))
    hs = lambda t: torch.clamp(0.2 * t + 0.5, 0.0, 1.0)
    xk = x @ K + b
    z = hs(xk[:, :U] + h @ R[:, :U])
    r = hs(xk[:, U:2 * U] + h @ R[:, U:2 * U])
    pre = xk[:, 2 * U:] + (r * h) @ R[:, 2 * U:]
    hh = {"tanh": torch.tanh, "relu": torch.relu, "linear": lambda t: t,
          "leaky_relu": lambda t: torch.where(t > 0, t, 0.2 * t)}[act](pre)
    return z, r, r * h, hh, z * h + (1.0 - z) * hh


def _launch(x, h, K, R, b, act, train=True):
    from tf_gnn_samples_amd import _lib, dense as DN, ops
    lib = _lib.load_library()
    V = h.shape[0]
    im_zr = DN.weight_image([K[:, :2 * U], R[:, :2 * U]], DN.WEIGHT_NN)
    im_h = DN.weight_image([K[:, 2 * U:], R[:, 2 * U:]], DN.WEIGHT_NN)
    outs = [torch.full((V, U), 7.0, device=h.device) for _ in range(5 if train else 1)]
    z, r, rh, hh = (outs[:4] if train else (None,) * 4)
    out = outs[-1]
    act_id = {"linear": _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH, "relu": _lib.ACT_RELU, "leaky_relu": _lib.ACT_LEAKY_RELU}[act]
    _lib.check(lib.relgnn_gru_cell_fwd_xf32(_lib.ptr(x, rows_strided=True), x.stride(0), _lib.ptr(h, rows_strided=True), h.stride(0),
                                            im_zr.buf.data_ptr(), im_h.buf.data_ptr(), _lib.ptr(b), act_id, _lib.ptr(z), _lib.ptr(r),
                                            _lib.ptr(rh), _lib.ptr(hh), _lib.ptr(out), V, U, U,
                                            ops.handover_word(h.device).data_ptr(), _lib.current_stream()),
               "relgnn_gru_cell_fwd_xf32")
    torch.cuda.synchronize()
    assert ops.handover_status() == 0
    return z, r, rh, hh, out


@pytest.mark.parametrize("act", ["tanh", "relu", "linear", "leaky_relu"])
@pytest.mark.parametrize("V", [1, 31, 32, 33, 64, 65, 97, 2250, 8229, 16416, 49986])
def test_cell_kernel_against_float64(gpu_device, V, act):
    dev = gpu_device
    K, R, b = _weights(dev, 3)
    x, h = _states(dev, V, V)
    got = _launch(x, h, K, R, b, act)
    want = _cell64(x, h, K, R, b, act)
    for name, g, w in zip(("z", "r", "rh", "hh", "out"), got, want):
        err = float((g.double() - w).abs().max())
        assert err <= 3e-6 * max(1.0, float(w.abs().max())), (name, V, act, err)
    # an inference pass (nothing kept for a backward) gives the same new states, bit for bit
    only = _launch(x, h, K, R, b, act, train=False)[-1]
    assert torch.equal(only, got[-1])


def test_cell_kernel_reads_row_strided_operands_and_is_reproducible(gpu_device):
    dev = gpu_device
    V = 5000
    K, R, b = _weights(dev, 5)
    g = torch.Generator(device="cpu").manual_seed(1)
    wide = torch.randn((V, 400), generator=g).to(dev)
    x, h = wide[:, 4:132], wide[:, 256:384]                    # 16-byte aligned views, row stride 400
    a = _launch(x, h, K, R, b, "tanh")
    bb = _launch(x.contiguous(), h.contiguous(), K, R, b, "tanh")
    for s, t in zip(a, bb):
        assert torch.equal(s, t)
    again = _launch(x, h, K, R, b, "tanh")
    for s, t in zip(a, again):
        assert torch.equal(s, t)


def test_cell_kernel_against_the_oracle_cell(gpu_device):
    """The oracle's GRU cell (oracle/tf_ops.py: the Keras expressions in float32 NumPy) on C3-like magnitudes: 1e-5 absolute."""
    from oracle import tf_ops
    dev = gpu_device
    V = 4100
    K, R, b = _weights(dev, 11)
    x, h = _states(dev, V, 12)
    got = _launch(x, h, K, R, b, "tanh")[-1]
    want = tf_ops.gru_cell(x.cpu().numpy(), h.cpu().numpy(), K.cpu().numpy(), R.cpu().numpy(), b.cpu().numpy(), np.tanh)
    assert float(np.abs(got.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max()) <= 1e-5


def test_rows_with_non_finite_values_spoil_their_own_rows_only(gpu_device):
    dev = gpu_device
    V = 300
    K, R, b = _weights(dev, 7)
    x, h = _states(dev, V, 8)
    fmax = torch.finfo(torch.float32).max
    x[3, 5] = float("nan")
    x[40, :] = -fmax                                           # what unsorted_segment_max leaves for a node without messages
    h[77, 9] = float("inf")
    x[150, 0] = fmax
    got = _launch(x, h, K, R, b, "tanh")[-1]
    bad = torch.zeros(V, dtype=torch.bool, device=dev)
    bad[[3, 40, 77, 150]] = True
    want = _cell64(x, h, K, R, b, "tanh")[-1]
    assert bool(torch.isfinite(got[~bad]).all())
    assert float((got[~bad].double() - want[~bad]).abs().max()) <= 3e-6
    assert bool(torch.isnan(got[3]).any()) and not bool(torch.isfinite(got[77, 9]))
    # float32 lowest in every input column: the limbs saturate instead of overflowing (limb_split.h) and the row stays what fp32
    # arithmetic makes of it — gates clipped to {0, 1}, a finite candidate
    ref40 = _cell64(x[40:41], h[40:41], K, R, b, "tanh")[-1]
    assert torch.equal(torch.isfinite(got[40]), torch.isfinite(ref40[0]))


def _cell_backward64(x, h, K, R, z, r, hh, g, act):
    """The cell's gradients in float64 from the float32 forward's own z, r and candidate — the derivative of hard_sigmoid / ReLU
    is a step, and a gate that float32 put ON the step (z == 1.0f where float64 has 1 - 1e-9) would make an autograd reference
    differ by a whole term."""
    x, h, K, R, z, r, hh, g = (t.double() for t in (x, h, K, R, z, r, hh, g))
    dact = {"tanh": lambda y: 1 - y * y, "relu": lambda y: (y > 0).double(), "linear": lambda y: torch.ones_like(y),
            "leaky_relu": lambda y: torch.where(y > 0, 1.0, 0.2)}[act]
    hsg = lambda y: ((y > 0) & (y < 1)).double() * 0.2
    gpre = g * (1 - z) * dact(hh)
    gzp = g * (h - hh) * hsg(z)
    grh = gpre @ R[:, 2 * U:].t()
    grp = grh * h * hsg(r)
    gxk = torch.cat([gzp, grp, gpre], 1)
    gx = gxk @ K.t()
    gh = g * z + grh * r + gxk[:, :2 * U] @ R[:, :2 * U].t()
    gR = torch.cat([h.t() @ gxk[:, :2 * U], (r * h).t() @ gpre], 1)
    return gx, gh, x.t() @ gxk, gR, gxk.sum(0)


@pytest.mark.parametrize("act", ["tanh", "relu", "linear", "leaky_relu"])
@pytest.mark.parametrize("V", [1, 33, 64, 97, 2250, 16416, 49986])
def test_cell_backward_kernel_against_float64(gpu_device, V, act):
    """relgnn_gru_cell_bwd_xf32 through utils._GRUCellFn: the gradients of the inputs, the states and the three variables against
    the float64 formulas (gnns/ggnn.py:92's cell differentiated by hand, on the forward's own gate values); the hand-over status
    stays clean."""
    from tf_gnn_samples_amd import _lib, config, ops, utils
    if not config.settings.limb_gemm:
        pytest.skip("the cell kernels belong to the limb route (RELGNN_GEMM is set to another route in this run)")
    dev = gpu_device
    K, R, b = _weights(dev, 31)
    x, h = _states(dev, V, V + 5)
    g = torch.Generator(device="cpu").manual_seed(V)
    gout = torch.randn((V, U), generator=g).to(dev)
    z, r, rh, hh, _ = _launch(x, h, K, R, b, act)
    leaves = [t.clone().requires_grad_(True) for t in (x, h, K, R, b)]
    act_id = {"linear": _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH, "relu": _lib.ACT_RELU, "leaky_relu": _lib.ACT_LEAKY_RELU}[act]
    out = utils._GRUCellFn.apply(*leaves, act_id)
    out.backward(gout)
    torch.cuda.synchronize()
    assert ops.handover_status() == 0
    want = _cell_backward64(x, h, K, R, z, r, hh, gout, act)
    for name, got, w in zip(("x", "h", "kernel", "recurrent_kernel", "bias"), leaves, want):
        scale = max(1.0, float(w.abs().max()))
        err = float((got.grad.double() - w).abs().max())
        assert err <= 4e-6 * scale * max(1.0, float(np.sqrt(V)) / 30), (name, V, act, err, scale)


def test_cell_backward_kernel_is_reproducible_and_reads_strided_gradients(gpu_device):
    from tf_gnn_samples_amd import _lib, utils
    dev = gpu_device
    V = 4097
    K, R, b = _weights(dev, 41)
    x, h = _states(dev, V, 42)
    g = torch.Generator(device="cpu").manual_seed(2)
    wide = torch.randn((V, 260), generator=g).to(dev)
    results = []
    for gout in (wide[:, 4:132], wide[:, 4:132].contiguous(), wide[:, 4:132]):
        leaves = [t.clone().requires_grad_(True) for t in (x, h, K, R, b)]
        utils._GRUCellFn.apply(*leaves, _lib.ACT_TANH).backward(gout)
        torch.cuda.synchronize()
        results.append([t.grad.clone() for t in leaves])
    for other in results[1:]:
        assert all(torch.equal(a, c) for a, c in zip(results[0], other))


@pytest.mark.parametrize("aggregation", ["sum", "max"])
def test_ggnn_layer_with_and_without_the_cell_kernel(gpu_device, aggregation):
    """sparse_ggnn_layer under config.gru_cell = 1 vs 0: the new states agree to 2e-6 and so do the gradients of the states and of
    every variable (the backward is the same chain of kernels on both sides, fed with z, r, r * h and the candidate either
    route saved)."""
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.gnns import ggnn
    dev = gpu_device
    V, L = 6000, 4
    rng = np.random.default_rng(5)
    adj = [torch.as_tensor(np.stack([rng.integers(0, V, 9000), rng.integers(0, V, 9000)], 1).astype(np.int32), device=dev)
           for _ in range(L)]
    g = torch.Generator(device="cpu").manual_seed(3)
    H0 = (torch.rand((V, U), generator=g) * 2 - 1).to(dev)
    K, R, b = _weights(dev, 21, 0.1)
    W0 = {"Edge_%d_Weight/kernel" % l: ((torch.rand((U, U), generator=g) * 2 - 1) * 0.1).to(dev) for l in range(L)}
    W0.update({"gru_cell/kernel": K, "gru_cell/recurrent_kernel": R, "gru_cell/bias": b})
    gout = torch.randn((V, U), generator=g).to(dev)
    res = {}
    for sw in ("0", "1"):
        with config.override(gru_cell=sw):
            H = H0.clone().requires_grad_(True)
            W = {k: v.clone().requires_grad_(True) for k, v in W0.items()}
            out = ggnn.sparse_ggnn_layer(H, adj, U, 2, "gru", "tanh", aggregation, weights=W)
            out.backward(gout)
            torch.cuda.synchronize()
            res[sw] = [out.detach(), H.grad] + [W[k].grad for k in sorted(W)]
    for a, c in zip(res["0"], res["1"]):
        scale = max(1.0, float(a.abs().max()))
        assert float((a.double() - c.double()).abs().max()) <= 2e-5 * scale


def test_unsupported_cells_are_refused(gpu_device):
    from tf_gnn_samples_amd import _lib
    lib = _lib.load_library()
    assert lib.relgnn_gru_cell_fwd_supported(_lib.ACT_TANH, 128, 128) == 1
    assert lib.relgnn_gru_cell_fwd_supported(_lib.ACT_TANH, 256, 128) == 0
    assert lib.relgnn_gru_cell_fwd_supported(_lib.ACT_ELU, 128, 128) == 0
    t = torch.zeros((64, 128), device=gpu_device)
    w = torch.zeros(1 << 18, dtype=torch.bfloat16, device=gpu_device)
    b = torch.zeros(384, device=gpu_device)
    rc = lib.relgnn_gru_cell_fwd_xf32(t.data_ptr(), 128, t.data_ptr(), 128, w.data_ptr(), w.data_ptr(), b.data_ptr(), _lib.ACT_TANH,
                                      None, None, None, None, t.data_ptr(), 64, 64, 128, None, None)
    assert rc == _lib.EUNSUPPORTED
    rc = lib.relgnn_gru_cell_fwd_xf32(t.data_ptr(), 128, t.data_ptr(), 128, w.data_ptr(), w.data_ptr(), b.data_ptr(), _lib.ACT_TANH,
                                      t.data_ptr(), None, None, None, t.data_ptr(), 64, 128, 128, None, None)
    assert rc == _lib.EINVAL                                   # the backward's tensors: all four or none


def test_cell_kernels_that_give_up_on_a_hand_over_say_so_and_end(gpu_device):
    """Every poll of the cell kernels' LDS counters is bounded (csrc/handover.h): with the poll bound at 1 (word 1 of the caller's
    status block, the debug knob) both kernels run to their end — with wrong numbers — and OR the give-up bits into word 0, which
    the model reads with every step's metrics (tests/test_gpu_limb_gemm.py covers that fetch)."""
    from tf_gnn_samples_amd import _lib, config, ops, utils
    if not config.settings.limb_gemm:
        pytest.skip("the cell kernels belong to the limb route (RELGNN_GEMM is set to another route in this run)")
    dev = gpu_device
    V = 20000
    K, R, b = _weights(dev, 51)
    x, h = _states(dev, V, 52)
    gout = torch.ones((V, U), device=dev)
    word = ops.handover_word(dev)
    assert ops.handover_status() == 0
    leaves = [t.clone().requires_grad_(True) for t in (x, h, K, R, b)]
    word[1] = 1
    try:
        out = utils._GRUCellFn.apply(*leaves, _lib.ACT_TANH)
        torch.cuda.synchronize()
        assert int(word[0].item()) & (4 | 8)                  # RELGNN_HANDOVER_PC_MATRIX | RELGNN_HANDOVER_PC_PRODUCER (include/relgnn.h)
        word[0] = 0
        out.backward(gout)
        torch.cuda.synchronize()
        assert int(word[0].item()) & (4 | 8)
    finally:
        word[1] = 0
        word[0] = 0
    clean = [t.clone().requires_grad_(True) for t in (x, h, K, R, b)]
    utils._GRUCellFn.apply(*clean, _lib.ACT_TANH).backward(gout)
    torch.cuda.synchronize()
    assert ops.handover_status() == 0 and all(bool(torch.isfinite(t.grad).all()) for t in clean)
