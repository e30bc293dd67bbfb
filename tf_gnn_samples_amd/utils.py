"""Host-side mirror of the reference's utils/utils.py (same names, argument meaning and
error behaviour) on PyTorch-ROCm tensors.

  get_aggregation_function  utils/utils.py:23-33   -> HIP segment kernels (ops.py)
  get_activation            utils/utils.py:36-58
  get_gated_unit            utils/utils.py:10-20   (Keras SimpleRNNCell / GRUCell semantics, TF 1.13)
  MLP                       utils/utils.py:77-126
  micro_f1                  utils/utils.py:61-74
  SMALL_NUMBER, BIG_NUMBER  utils/utils.py:6-7

Node-wise GEMMs (Dense, GRU) are library GEMMs (hipBLASLt, dense.lib_gemm); the elementwise halves of the GRU cell and the layer
normalisation are HIP kernels of librelgnn (csrc/gru.hip, csrc/layer_norm.hip).
"""
import math
from typing import Callable, List, Mapping, Optional, Union

import torch

from . import ops
from .dense import dense, dense_relu

BIG_NUMBER = 1e7
SMALL_NUMBER = 1e-7


def get_aggregation_function(aggregation_fun: Optional[str]):
    if aggregation_fun in ['sum', 'unsorted_segment_sum']:
        return ops.unsorted_segment_sum
    if aggregation_fun in ['max', 'unsorted_segment_max']:
        return ops.unsorted_segment_max
    if aggregation_fun in ['mean', 'unsorted_segment_mean']:
        return ops.unsorted_segment_mean
    if aggregation_fun in ['sqrt_n', 'unsorted_segment_sqrt_n']:
        return ops.unsorted_segment_sqrt_n
    else:
        raise ValueError("Unknown aggregation function '%s'!" % aggregation_fun)


def _gelu(x):
    # erf form of the reference (utils/utils.py:53-55)
    cdf = 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    return x * cdf


def _leaky_relu(x):
    return torch.nn.functional.leaky_relu(x, 0.2)  # tf.nn.leaky_relu default alpha


def get_activation(activation_fun: Optional[str]):
    if activation_fun is None:
        return None
    activation_fun = activation_fun.lower()
    if activation_fun == 'linear':
        return None
    if activation_fun == 'tanh':
        return torch.tanh
    if activation_fun == 'relu':
        return torch.relu
    if activation_fun == 'leaky_relu':
        return _leaky_relu
    if activation_fun == 'elu':
        return torch.nn.functional.elu
    if activation_fun == 'selu':
        return torch.selu
    if activation_fun == 'gelu':
        return _gelu
    else:
        raise ValueError("Unknown activation function '%s'!" % activation_fun)


def apply_activation(fn, x):
    return x if fn is None else fn(x)


def hard_sigmoid(x):
    """Keras hard_sigmoid (GRUCell's default recurrent_activation in TF 1.13): clip(0.2x+0.5, 0, 1)."""
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


class _LayerNormFn(torch.autograd.Function):
    """csrc/layer_norm.hip: one read + one write forward, d gamma / d beta without atomics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        from . import _lib
        lib = _lib.load_library()
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        V, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(V, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _lib.check(lib.relgnn_layer_norm_fwd(_lib.ptr(x), D, V, D, _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.ptr(y), D,
                                             _lib.ptr(mean), _lib.ptr(rstd), _lib.current_stream()), "relgnn_layer_norm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        from .dense import column_sum
        lib = _lib.load_library()
        x, gamma, mean, rstd = ctx.saved_tensors
        V, D = x.shape
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        if V == 0:
            return dx, torch.zeros_like(gamma), torch.zeros_like(gamma), None
        groups = int(lib.relgnn_layer_norm_groups(V, D))
        partial = torch.empty((groups, 2 * D), dtype=torch.float32, device=x.device)
        _lib.check(lib.relgnn_layer_norm_bwd(_lib.ptr(x), D, _lib.ptr(gy), D, V, D, _lib.ptr(gamma), _lib.ptr(mean),
                                             _lib.ptr(rstd), _lib.ptr(dx), D, _lib.ptr(partial), groups,
                                             _lib.current_stream()), "relgnn_layer_norm_bwd")
        gsum = column_sum(partial)
        return dx, gsum[:D], gsum[D:], None


def layer_norm_scope(i: int) -> str:
    """Variable scope of the i-th tf.contrib.layers.layer_norm call inside one variable_scope (TF uniquifies the default
    scope: LayerNorm, LayerNorm_1, ...).  The reference's layers call it once per TIMESTEP (gnns/gnn_film.py:120,
    rgin.py:139, gnn_edge_mlp.py:119): every timestep owns its gamma/beta; the driver's inter-layer norm
    (models/sparse_graph_model.py:192-193) is the next scope after them."""
    return "LayerNorm" if i == 0 else "LayerNorm_%d" % i


def layer_norm_variables(state_dim: int, count: int):
    specs = {}
    for i in range(count):
        specs[layer_norm_scope(i) + "/beta"] = ((state_dim,), "zeros")
        specs[layer_norm_scope(i) + "/gamma"] = ((state_dim,), "ones")
    return specs


def layer_norm(x, gamma, beta, eps: float = 1e-12):
    """tf.contrib.layers.layer_norm on a [V, D] tensor: moments over the last axis (biased
    variance), variance_epsilon 1e-12, learnable gamma/beta over the last axis."""
    D = x.shape[-1]
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and D % 4 == 0 and D <= 1024:
        return _LayerNormFn.apply(x, gamma, beta, float(eps))
    return torch.nn.functional.layer_norm(x, (D,), gamma, beta, eps)


class _FusedGRU(torch.autograd.Function):
    """Keras GRUCell given xk = x@K+b and rec = h@U[:, :2u]: two fused elementwise kernels around the inner GEMM
    (r*h) @ U_h forward, two backward (csrc/gru.hip) instead of ~40 elementwise launches."""

    @staticmethod
    def forward(ctx, xk, rec, h, u_h, act: int):
        from . import _lib
        lib = _lib.load_library()
        st = _lib.current_stream()
        xk, rec, h = xk.contiguous(), rec.contiguous(), h.contiguous()
        # (u_h = recurrent_kernel[:, 2u:] stays the strided view of the parameter it is: the product reads it with its leading
        #  dimension and keeps its limb image with the step's other weights — round 6: no copy, no split launch per cell)
        if not (u_h.dim() == 2 and u_h.stride(1) == 1 and u_h.stride(0) % 4 == 0 and u_h.data_ptr() % 16 == 0):
            u_h = u_h.contiguous()
        V, u = h.shape
        z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        _lib.check(lib.relgnn_gru_gates_fwd(_lib.ptr(xk), _lib.ptr(rec), _lib.ptr(h), V, u, _lib.ptr(z), _lib.ptr(r),
                                            _lib.ptr(rh), st), "relgnn_gru_gates_fwd")
        from .dense import GEMM_NN, lib_gemm
        q = lib_gemm(GEMM_NN, rh, u_h, weight=True)
        hh, out = torch.empty_like(h), torch.empty_like(h)
        _lib.check(lib.relgnn_gru_out_fwd(_lib.ptr(xk), _lib.ptr(q), _lib.ptr(z), _lib.ptr(h), V, u, act, _lib.ptr(hh),
                                          _lib.ptr(out), st), "relgnn_gru_out_fwd")
        ctx.act = act
        ctx.save_for_backward(z, r, rh, h, hh, u_h)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import _lib
        from .dense import matmul_tn_splitk
        lib = _lib.load_library()
        st = _lib.current_stream()
        z, r, rh, h, hh, u_h = ctx.saved_tensors
        V, u = h.shape
        gout = gout.contiguous()
        gxk = torch.empty((V, 3 * u), dtype=torch.float32, device=h.device)
        gq, gz, gh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        _lib.check(lib.relgnn_gru_out_bwd(_lib.ptr(gout), _lib.ptr(z), _lib.ptr(h), _lib.ptr(hh), V, u, ctx.act,
                                          _lib.ptr(gxk), _lib.ptr(gq), _lib.ptr(gz), _lib.ptr(gh), st), "relgnn_gru_out_bwd")
        from .dense import GEMM_NT, lib_gemm
        grh = lib_gemm(GEMM_NT, gq, u_h, weight=True)
        gu_h = matmul_tn_splitk(rh, gq) if ctx.needs_input_grad[3] else None
        _lib.check(lib.relgnn_gru_gates_bwd(_lib.ptr(grh), _lib.ptr(gz), _lib.ptr(z), _lib.ptr(r), _lib.ptr(h), V, u,
                                            _lib.ptr(gxk), _lib.ptr(gh), st), "relgnn_gru_gates_bwd")
        return gxk, gxk[:, :2 * u], gh, gu_h, None


class _GRUCellFn(torch.autograd.Function):
    """The whole Keras GRUCell (utils/utils.py:15-16 through gnns/ggnn.py:92; reset_after=False, gate order z, r, h) as ONE autograd
    node over the cell's three variables as they are stored — kernel [D, 3u], recurrent_kernel [u, 3u], bias [3u]:
        xk = x @ K + b;  rec = h @ U[:, :2u];  z, r = hs(xk_zr + rec);  q = (r * h) @ U[:, 2u:];  out = z h + (1 - z) act(xk_h + q)
    Same kernels and the same arithmetic as the composition of dense() / _FusedGRU it replaces (round 6); what goes away is what
    autograd wrapped around the two column views of the recurrent kernel — per cell and step two zero-fills, two copies and an add
    for the views' gradients, a copy of U[:, 2u:] — because the recurrent kernel's gradient is written block by block into ONE
    [u, 3u] tensor (dense.tn_stream_into).  The weight gradients go to the side stream like a Dense layer's (dense._on_side_stream)."""

    @staticmethod
    def forward(ctx, x, h, K, U, b, act: int):
        from . import _lib
        from .dense import GEMM_NN, lib_gemm
        lib = _lib.load_library()
        st = _lib.current_stream()
        x, h = x.contiguous(), h.contiguous()
        V, u = h.shape
        if _gru_cell_kernel_ok(x, h, K, U, b, act):
            # one launch (csrc/gru_cell.hip): both products, the gates, the candidate and the blend; z, r, r * h and the candidate
            # are written only when a backward will read them
            from . import ops
            from .dense import WEIGHT_NN, weight_image
            train = any(ctx.needs_input_grad[:5])
            im_zr = weight_image([K[:, :2 * u], U[:, :2 * u]], WEIGHT_NN)  # B [2u, D + u]: k = [x | h]
            im_h = weight_image([K[:, 2 * u:], U[:, 2 * u:]], WEIGHT_NN)   # B [u, D + u]:  k = [x | r * h]
            z, r, rh, hh = (torch.empty_like(h) for _ in range(4)) if train else (None,) * 4
            out = torch.empty_like(h)
            _lib.check(lib.relgnn_gru_cell_fwd_xf32(_lib.ptr(x), x.stride(0), _lib.ptr(h), h.stride(0), im_zr.buf.data_ptr(),
                                                    im_h.buf.data_ptr(), _lib.ptr(b), act, _lib.ptr(z), _lib.ptr(r), _lib.ptr(rh),
                                                    _lib.ptr(hh), _lib.ptr(out), V, u, x.shape[1],
                                                    ops.handover_word(h.device).data_ptr(), st), "relgnn_gru_cell_fwd_xf32")
            ctx.act, ctx.cell_kernel = act, True
            if train:
                ctx.save_for_backward(x, h, K, U, z, r, rh, hh)
            ctx.leaf_params = (K, U, b) if all(p.is_leaf and p.requires_grad for p in (K, U, b)) else None
            return out
        ctx.cell_kernel = False
        xk = lib_gemm(GEMM_NN, x, K, b, weight=True)                       # [V, 3u]
        rec = lib_gemm(GEMM_NN, h, U[:, :2 * u], weight=True)              # [V, 2u] (the view read with its leading dimension)
        z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        _lib.check(lib.relgnn_gru_gates_fwd(_lib.ptr(xk), _lib.ptr(rec), _lib.ptr(h), V, u, _lib.ptr(z), _lib.ptr(r),
                                            _lib.ptr(rh), st), "relgnn_gru_gates_fwd")
        q = lib_gemm(GEMM_NN, rh, U[:, 2 * u:], weight=True)
        hh, out = torch.empty_like(h), torch.empty_like(h)
        _lib.check(lib.relgnn_gru_out_fwd(_lib.ptr(xk), _lib.ptr(q), _lib.ptr(z), _lib.ptr(h), V, u, act, _lib.ptr(hh),
                                          _lib.ptr(out), st), "relgnn_gru_out_fwd")
        ctx.act = act
        ctx.save_for_backward(x, h, K, U, z, r, rh, hh)
        ctx.leaf_params = (K, U, b) if all(p.is_leaf and p.requires_grad for p in (K, U, b)) else None
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import _lib
        from .dense import GEMM_NT, _on_side_stream, column_sum, lib_gemm, matmul_tn_splitk, tn_stream_into
        lib = _lib.load_library()
        st = _lib.current_stream()
        x, h, K, U, z, r, rh, hh = ctx.saved_tensors
        V, u = h.shape
        gout = gout.contiguous()
        gxk = torch.empty((V, 3 * u), dtype=torch.float32, device=h.device)
        fused = ctx.cell_kernel and _gru_cell_kernel_ok(x, h, K, U, None, ctx.act)
        if fused:
            # one launch (csrc/gru_cell.hip): the gate / candidate gradients and the three input-gradient products; gxk for the
            # weight gradients below
            from . import ops
            from .dense import WEIGHT_NT, weight_image
            im_h = weight_image([K[:, 2 * u:], U[:, 2 * u:]], WEIGHT_NT, separate=True)
            im_zr = weight_image([K[:, :2 * u], U[:, :2 * u]], WEIGHT_NT, separate=True)
            gx, gh = torch.empty_like(x), torch.empty_like(h)
            _lib.check(lib.relgnn_gru_cell_bwd_xf32(_lib.ptr(gout), gout.stride(0), _lib.ptr(z), _lib.ptr(r), _lib.ptr(h), h.stride(0),
                                                    _lib.ptr(hh), im_h.buf.data_ptr(), im_zr.buf.data_ptr(), ctx.act, _lib.ptr(gxk),
                                                    _lib.ptr(gx), _lib.ptr(gh), V, u, x.shape[1],
                                                    ops.handover_word(h.device).data_ptr(), st), "relgnn_gru_cell_bwd_xf32")
            gq = gxk[:, 2 * u:]
        else:
            gq, gz, gh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
            _lib.check(lib.relgnn_gru_out_bwd(_lib.ptr(gout), _lib.ptr(z), _lib.ptr(h), _lib.ptr(hh), V, u, ctx.act,
                                              _lib.ptr(gxk), _lib.ptr(gq), _lib.ptr(gz), _lib.ptr(gh), st), "relgnn_gru_out_bwd")
            grh = lib_gemm(GEMM_NT, gq, U[:, 2 * u:], weight=True)
            _lib.check(lib.relgnn_gru_gates_bwd(_lib.ptr(grh), _lib.ptr(gz), _lib.ptr(z), _lib.ptr(r), _lib.ptr(h), V, u,
                                                _lib.ptr(gxk), _lib.ptr(gh), st), "relgnn_gru_gates_bwd")
        grec = gxk[:, :2 * u]                                              # d loss / d rec = the z, r columns of d loss / d xk

        def weight_side():
            if all(ctx.needs_input_grad[2:5]):
                # the three products over the same rows and the bias gradient (the column sums of gxk, which the first product
                # reads anyway): one launch pair instead of four products, four reductions and two column-sum launches
                from .dense import tn_stream_group, tn_stream_group_ok
                gK = torch.empty((x.shape[1], 3 * u), dtype=torch.float32, device=h.device)
                gU = torch.empty((u, 3 * u), dtype=torch.float32, device=h.device)
                products = [(x, gxk, gK), (h, grec, gU[:, :2 * u]), (rh, gq, gU[:, 2 * u:])]
                if tn_stream_group_ok(products):
                    gb = torch.empty(3 * u, dtype=torch.float32, device=h.device)
                    tn_stream_group(products, colsum=gb)
                    return gK, gU, gb
            gK = matmul_tn_splitk(x, gxk) if ctx.needs_input_grad[2] else None
            gU = None
            if ctx.needs_input_grad[3]:
                gU = torch.empty((u, 3 * u), dtype=torch.float32, device=h.device)
                tn_stream_into(h, grec, gU[:, :2 * u])
                tn_stream_into(rh, gq, gU[:, 2 * u:])
            gb = column_sum(gxk) if ctx.needs_input_grad[4] else None
            return gK, gU, gb

        aside = _on_side_stream(weight_side, (x, h, rh, gxk, gq), ctx.leaf_params,
                                want=(ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and any(ctx.needs_input_grad[2:5]))
        if not fused:
            gx = lib_gemm(GEMM_NT, gxk, K, weight=True) if ctx.needs_input_grad[0] else None
            if ctx.needs_input_grad[1]:
                gh = gh.add_(lib_gemm(GEMM_NT, grec, U[:, :2 * u], weight=True))
            else:
                gh = None
        gK, gU, gb = aside if aside is not None else weight_side()
        return gx, gh, gK, gU, gb, None


def _gru_cell_kernel_ok(x, h, K, U, b, act: int) -> bool:
    """The one-kernel forward (relgnn_gru_cell_fwd_xf32): the limb route, 128 units over 128-wide inputs, rows it can read in place."""
    from . import _lib
    from .config import settings as cfg
    from .dense import WEIGHT_NN, weight_image_ok
    u = h.shape[1]
    if not (cfg.limb_gemm and cfg.gru_cell == "1" and h.is_cuda and 0 < h.shape[0] <= (1 << 21) and h.shape[0] * h.stride(0) < (1 << 30) and K.shape[0] == x.shape[1]
            and (b is None or (b.is_contiguous() and b.data_ptr() % 16 == 0))
            and all(t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in (x, h))):
        return False
    if not _lib.load_library().relgnn_gru_cell_fwd_supported(act, u, x.shape[1]):
        return False
    return weight_image_ok([K[:, :2 * u], U[:, :2 * u]], WEIGHT_NN) and weight_image_ok([K[:, 2 * u:], U[:, 2 * u:]], WEIGHT_NN)


def _gru_cell_fused_ok(x, h, K, U, b, u: int) -> bool:
    return (h.is_cuda and x.is_cuda and x.dim() == 2 and h.dim() == 2 and h.shape[1] == u and u % 4 == 0 and U.shape == (u, 3 * u)
            and K.shape[1] == 3 * u and b is not None and b.shape == (3 * u,) and U.stride(1) == 1 and U.is_contiguous()
            and x.dtype == h.dtype == torch.float32
            # (the recurrent kernel's gradient blocks are written by the streaming weight-gradient kernel: its size class)
            and 2 * u * u <= 256 * 256 and 0 < h.shape[0] <= (1 << 18))


_GRU_FUSABLE = {None: 0, "linear": 0, "tanh": 1, "relu": 2, "leaky_relu": 3, "elu": 4, "selu": 5}


class _GatedUnit:
    """Callable cell(inputs, [state]) -> (output, [new_state]) like a Keras RNN cell."""

    def __init__(self, units: int, kind: str, activation_fn, weights: Mapping[str, torch.Tensor], activation_name=None):
        self.units, self.kind, self.activation_fn, self.w = units, kind, activation_fn, weights
        self.activation_name = None if activation_name is None else activation_name.lower()

    def __call__(self, inputs, states):
        h = states[0]
        u = self.units
        act = self.activation_fn
        K, U, b = self.w["kernel"], self.w["recurrent_kernel"], self.w["bias"]
        if self.kind == 'rnn':
            out = dense(inputs, K, b) + dense(h, U)
            out = apply_activation(act, out)
            return out, [out]
        # GRU, reset_after=False, gate order z, r, h (Keras GRUCell, TF 1.13)
        if self.activation_name in _GRU_FUSABLE and _gru_cell_fused_ok(inputs, h, K, U, b, u):
            out = _GRUCellFn.apply(inputs, h, K, U, b, _GRU_FUSABLE[self.activation_name])
            return out, [out]
        xk = dense(inputs, K, b)                             # [V, 3u]
        rec = dense(h, U[:, :2 * u])                         # [V, 2u]
        if h.is_cuda and self.activation_name in _GRU_FUSABLE:
            out = _FusedGRU.apply(xk, rec, h, U[:, 2 * u:], _GRU_FUSABLE[self.activation_name])
            return out, [out]
        z = hard_sigmoid(xk[:, :u] + rec[:, :u])
        r = hard_sigmoid(xk[:, u:2 * u] + rec[:, u:])
        hh = apply_activation(act, xk[:, 2 * u:] + dense(r * h, U[:, 2 * u:]))
        out = z * h + (1.0 - z) * hh
        return out, [out]


GATED_UNIT_SCOPES = {'rnn': 'simple_rnn_cell', 'gru': 'gru_cell', 'lstm': 'lstm_cell'}


def gated_unit_variable_shapes(units: int, input_dim: int, gated_unit: str):
    """TF variable names (relative to the layer scope) and shapes of the cell's weights."""
    name = gated_unit.lower()
    if name == 'rnn':
        g = 1
    elif name == 'gru':
        g = 3
    elif name == 'lstm':
        g = 4
    else:
        raise Exception("Unknown RNN cell type '%s'." % gated_unit)
    scope = GATED_UNIT_SCOPES[name]
    return {
        scope + "/kernel": ((input_dim, g * units), "glorot_uniform"),
        scope + "/recurrent_kernel": ((units, g * units), "orthogonal"),
        scope + "/bias": ((g * units,), "zeros"),
    }


def get_gated_unit(units: int, gated_unit: str, activation_function: str, weights: Mapping[str, torch.Tensor]):
    """Mirror of get_gated_unit(units, gated_unit, activation_function); `weights` carries the
    cell variables (kernel [D_in, g*units], recurrent_kernel [units, g*units], bias [g*units])."""
    activation_fn = get_activation(activation_function)
    gated_unit_name = gated_unit.lower()
    if gated_unit_name == 'rnn':
        return _GatedUnit(units, 'rnn', activation_fn, weights, activation_function)
    if gated_unit_name == 'gru':
        return _GatedUnit(units, 'gru', activation_fn, weights, activation_function)
    if gated_unit_name == 'lstm':
        # The reference passes a single state to LSTMCell (gnns/ggnn.py:92), which fails inside
        # Keras (LSTM needs [h, c]); there is no behaviour to reproduce.
        raise NotImplementedError("LSTM cells cannot work in the reference's GGNN layer (single state)")
    else:
        raise Exception("Unknown RNN cell type '%s'." % gated_unit)


def micro_f1_label_masks(labels):
    """(int(label) != 0, int(label) != 1): the two label predicates micro_f1 needs; constant per batch."""
    li = labels.to(torch.int32)
    return li != 0, li != 1


def micro_f1(logits, labels, label_masks=None):
    """utils/utils.py:61-74 on integers, restated with boolean masks (identical counts):
      predicted = round(sigmoid(logits))  ==  sigmoid(logits) > 0.5   (round-half-even sends exactly 0.5 to 0)
      true_pos  = #(predicted * labels != 0);  false_pos = #(predicted * (labels - 1) != 0)
      false_neg = #((predicted - 1) * labels != 0)"""
    predicted = torch.sigmoid(logits) > 0.5
    lab_nz, lab_n1 = label_masks if label_masks is not None else micro_f1_label_masks(labels)
    true_pos = (predicted & lab_nz).sum()
    false_pos = (predicted & lab_n1).sum()
    false_neg = (~predicted & lab_nz).sum()
    precision = true_pos / (true_pos + false_pos)
    recall = true_pos / (true_pos + false_neg)
    fmeasure = (2 * precision * recall) / (precision + recall)
    return fmeasure.to(torch.float32)


class MLP(object):
    """utils/utils.py:77-126.  `weights` maps "dense/kernel", "dense_1/kernel", ... (and
    ".../bias" when use_biases) to tensors in TF layout [in, out]."""

    def __init__(self,
                 out_size: int,
                 hidden_layers: Union[List[int], int] = 1,
                 use_biases: bool = False,
                 activation_fun: Optional[Callable[[torch.Tensor], torch.Tensor]] = torch.relu,
                 dropout_rate: float = 0.0,
                 name: Optional[str] = "MLP",
                 weights: Optional[Mapping[str, torch.Tensor]] = None,
                 training: bool = False,
                 ):
        if isinstance(hidden_layers, int):
            hidden_layer_sizes = [out_size] * hidden_layers
        else:
            hidden_layer_sizes = hidden_layers
        if len(hidden_layer_sizes) > 1:
            assert activation_fun is not None, "Multiple linear layers without an activation"
        self.hidden_layer_sizes = list(hidden_layer_sizes)
        self.out_size = out_size
        self.use_biases = use_biases
        self.activation_fun = activation_fun
        self.dropout_rate = dropout_rate
        self.name = name
        self.weights = weights
        self.training = training

    @staticmethod
    def layer_names(num_layers: int):
        return ["dense" if i == 0 else "dense_%i" % i for i in range(num_layers)]

    @classmethod
    def variable_shapes(cls, in_size: int, out_size: int, hidden_layers: Union[List[int], int] = 1,
                        use_biases: bool = False, name: str = "MLP"):
        sizes = [out_size] * hidden_layers if isinstance(hidden_layers, int) else list(hidden_layers)
        dims = [in_size] + sizes + [out_size]
        res = {}
        for i, lname in enumerate(cls.layer_names(len(dims) - 1)):
            res["%s/%s/kernel" % (name, lname)] = ((dims[i], dims[i + 1]), "glorot_uniform")
            if use_biases:
                res["%s/%s/bias" % (name, lname)] = ((dims[i + 1],), "zeros")
        return res

    def _dense(self, i, x, fused_relu: bool = False):
        lname = self.layer_names(len(self.hidden_layer_sizes) + 1)[i]
        k = self.weights["%s/%s/kernel" % (self.name, lname)]
        b = self.weights["%s/%s/bias" % (self.name, lname)] if self.use_biases else None
        if fused_relu:
            return dense_relu(x, k, b)
        return dense(x, k, b)

    def __call__(self, input: torch.Tensor) -> torch.Tensor:
        activations = input
        n_hidden = len(self.hidden_layer_sizes)
        for i in range(n_hidden):
            if self.dropout_rate > 0.0 and self.training:
                activations = torch.nn.functional.dropout(activations, self.dropout_rate, True)
            if self.activation_fun is torch.relu:        # the default: ReLU in the GEMM's epilogue
                activations = self._dense(i, activations, fused_relu=True)
            else:
                activations = apply_activation(self.activation_fun, self._dense(i, activations))
        return self._dense(n_hidden, activations)
