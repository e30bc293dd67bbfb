"""The per-message activations of the edge kernels (csrc/common.h: act_fwd_fast / act_grad_fast — v_exp_f32, v_rcp_f32 and
the fitted two-piece erf of scripts/fit_fast_erf.py) against float64 restatements of utils/utils.py:36-58, through the one
C-ABI entry that exposes them element by element (relgnn_pair_materialize: hidden = act(P[row]), gpre = g * act'(P[row]))."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SELU_SCALE, SELU_ALPHA = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717


def _erf(x):
    return np.vectorize(math.erf)(x)


def _reference(name, x):
    """(act(x), act'(x)) in float64."""
    if name == "tanh":
        t = np.tanh(x)
        return t, 1 - t * t
    if name == "relu":
        return np.maximum(x, 0), (x > 0).astype(np.float64)
    if name == "leaky_relu":
        return np.where(x > 0, x, 0.2 * x), np.where(x > 0, 1.0, 0.2)
    if name == "elu":
        return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1), np.where(x > 0, 1.0, np.exp(np.minimum(x, 0)))
    if name == "selu":
        return (SELU_SCALE * np.where(x > 0, x, SELU_ALPHA * (np.exp(np.minimum(x, 0)) - 1)),
                SELU_SCALE * np.where(x > 0, 1.0, SELU_ALPHA * np.exp(np.minimum(x, 0))))
    if name == "gelu":
        cdf = 0.5 * (1 + _erf(x / math.sqrt(2)))
        return x * cdf, cdf + x * np.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["tanh", "relu", "leaky_relu", "elu", "selu", "gelu"])
def test_per_message_activation_and_derivative(gpu_device, name):
    from tf_gnn_samples_amd import _lib, ops
    lib = _lib.load_library()
    act = ops.activation_id(name)
    D = 64
    grid = np.concatenate([np.linspace(-30, 30, 4001), np.linspace(-1.5, 1.5, 4001), np.linspace(-6, 6, 4001),
                           [0.0, -0.0, 1e-30, -1e-30, 1e-8, -1e-8, 1.0, -1.0, 4.2, -4.2, 88.0, -88.0, 1e4, -1e4]])
    M = (len(grid) + D - 1) // D
    x = np.zeros(M * D, dtype=np.float32)
    x[:len(grid)] = grid.astype(np.float32)
    P = torch.as_tensor(x.reshape(M, D), device=gpu_device)
    rows = torch.arange(M, dtype=torch.int32, device=gpu_device)
    out = torch.empty_like(P)
    st = _lib.current_stream()
    _lib.check(lib.relgnn_pair_materialize(act, _lib.ptr(P), D, None, D, D, _lib.ptr(rows), None, M, None, _lib.ptr(out), D, st),
               "relgnn_pair_materialize")
    g = torch.full_like(P, 1.0)
    gpre = torch.empty_like(P)
    _lib.check(lib.relgnn_pair_materialize(act, _lib.ptr(P), D, None, D, D, _lib.ptr(rows), None, M, _lib.ptr(g), _lib.ptr(gpre), D, st),
               "relgnn_pair_materialize")
    ref, dref = _reference(name, x.astype(np.float64))
    got, dgot = out.cpu().numpy().reshape(-1).astype(np.float64), gpre.cpu().numpy().reshape(-1).astype(np.float64)
    assert np.isfinite(got).all() and np.isfinite(dgot).all()
    # 3e-7 absolute where the result is O(1), 3e-7 relative beyond: three fp32 ulps at 1.0
    assert np.all(np.abs(got - ref) <= 3e-7 * np.maximum(1.0, np.abs(ref)))
    assert np.all(np.abs(dgot - dref) <= 3e-7 * np.maximum(1.0, np.abs(dref)))
    # signs and exact points the reference's activations have
    i0 = len(grid) - 14
    assert got[i0] == 0.0 and got[i0 + 1] == 0.0                     # act(+-0) = 0 for all six
    if name in ("relu", "leaky_relu", "elu", "selu", "gelu"):
        assert got[-2] == pytest.approx(ref[-2], rel=3e-7)            # act(1e4): the identity branch
