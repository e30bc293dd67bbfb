"""torch.autograd wrappers over the librelgnn C ABI (include/relgnn.h).

`unsorted_segment_{sum,mean,max,sqrt_n}` mirror the TF ops returned by the reference's
get_aggregation_function (utils/utils.py:23-33), same argument names and meaning.
`seg_gather_reduce` is the fused form the layer functions use: the gather
(tf.nn.embedding_lookup), the per-message scale and the segment reduction in ONE kernel.
"""
from typing import Optional

import torch

from . import _lib
from .graph import GatherReducePlan, build_segment_plan

_MODE_IDS = {
    "sum": _lib.AGG_SUM, "unsorted_segment_sum": _lib.AGG_SUM,
    "max": _lib.AGG_MAX, "unsorted_segment_max": _lib.AGG_MAX,
    "mean": _lib.AGG_MEAN, "unsorted_segment_mean": _lib.AGG_MEAN,
    "sqrt_n": _lib.AGG_SQRT_N, "unsorted_segment_sqrt_n": _lib.AGG_SQRT_N,
}

_ACT_IDS = {
    None: _lib.ACT_LINEAR, "linear": _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH, "relu": _lib.ACT_RELU,
    "leaky_relu": _lib.ACT_LEAKY_RELU, "elu": _lib.ACT_ELU, "selu": _lib.ACT_SELU, "gelu": _lib.ACT_GELU,
}
# activations whose derivative can be evaluated from the OUTPUT (safe to fuse as an epilogue)
_FUSABLE_ACTS = {_lib.ACT_LINEAR, _lib.ACT_TANH, _lib.ACT_RELU, _lib.ACT_LEAKY_RELU, _lib.ACT_ELU, _lib.ACT_SELU}


def aggregation_mode_id(aggregation_fun: Optional[str]) -> int:
    """Same accepted strings and error as get_aggregation_function (utils/utils.py:23-33)."""
    if aggregation_fun not in _MODE_IDS:
        raise ValueError("Unknown aggregation function '%s'!" % aggregation_fun)
    return _MODE_IDS[aggregation_fun]


def activation_id(activation_fun: Optional[str]) -> int:
    """Same accepted strings and error as get_activation (utils/utils.py:36-58)."""
    if activation_fun is None:
        return _lib.ACT_LINEAR
    name = activation_fun.lower()
    if name not in _ACT_IDS:
        raise ValueError("Unknown activation function '%s'!" % activation_fun)
    return _ACT_IDS[name]


def _check_f32(x: torch.Tensor, what: str):
    if x.dtype != torch.float32:
        raise ValueError("%s must be float32, got %s" % (what, x.dtype))


def _seg_reduce_raw(mode, X, rowptr, stride, col, w, num_out, act=_lib.ACT_LINEAR):
    lib = _lib.load_library()
    D = X.shape[1]
    out = torch.empty((num_out, D), dtype=torch.float32, device=X.device)
    _lib.check(lib.relgnn_seg_reduce_fwd(
        mode, _lib.ptr(X), X.shape[0], X.stride(0), D, _lib.ptr(rowptr), num_out, stride,
        _lib.ptr(col), _lib.ptr(w), act, _lib.ptr(out), D, _lib.current_stream()),
        "relgnn_seg_reduce_fwd")
    return out


class _SegGatherReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, plan: GatherReducePlan, mode: int, act: int):
        _check_f32(X, "X")
        if X.dim() != 2 or X.shape[0] != plan.num_rows_x:
            raise ValueError("X must be [%d, D], got %s" % (plan.num_rows_x, tuple(X.shape)))
        if X.stride(1) != 1:
            X = X.contiguous()
        out = _seg_reduce_raw(mode, X, plan.rowptr, plan.stride, plan.col, plan.w, plan.num_out, act)
        ctx.plan, ctx.mode, ctx.act = plan, mode, act
        need_x = mode == _lib.AGG_MAX
        need_out = need_x or act != _lib.ACT_LINEAR
        ctx.save_for_backward(X if need_x else None, out if need_out else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load_library()
        st = _lib.current_stream()
        plan, mode, act = ctx.plan, ctx.mode, ctx.act
        X, out = ctx.saved_tensors
        gout = gout.contiguous()
        D = gout.shape[1]
        if act != _lib.ACT_LINEAR:
            g = torch.empty_like(gout)
            _lib.check(lib.relgnn_act_bwd_from_output(act, _lib.ptr(out), _lib.ptr(gout), gout.numel(),
                                                      _lib.ptr(g), st), "relgnn_act_bwd_from_output")
            gout = g
        if mode == _lib.AGG_MAX:
            if act != _lib.ACT_LINEAR:
                raise RuntimeError("max aggregation is never fused with an activation epilogue")
            gsel = torch.empty_like(gout)
            _lib.check(lib.relgnn_seg_max_count(
                _lib.ptr(X), X.stride(0), D, _lib.ptr(plan.rowptr), plan.num_out, plan.stride,
                _lib.ptr(plan.col), _lib.ptr(plan.w), _lib.ptr(out), _lib.ptr(gout), D, _lib.ptr(gsel), st),
                "relgnn_seg_max_count")
            gX = torch.empty((plan.num_rows_x, D), dtype=torch.float32, device=gout.device)
            _lib.check(lib.relgnn_seg_max_bwd(
                _lib.ptr(X), X.stride(0), D, _lib.ptr(plan.rowptr_b), plan.num_rows_x, plan.stride_b,
                _lib.ptr(plan.col_b), _lib.ptr(plan.w_bwd(_lib.AGG_SUM)), _lib.ptr(out), _lib.ptr(gsel), D,
                _lib.ptr(gX), D, st), "relgnn_seg_max_bwd")
            return gX, None, None, None
        # sum / mean / sqrt_n: the gradient is the same gather-reduce over the transposed buckets
        gX = _seg_reduce_raw(_lib.AGG_SUM, gout, plan.rowptr_b, plan.stride_b, plan.col_b,
                             plan.w_bwd(mode), plan.num_rows_x)
        return gX, None, None, None


def seg_gather_reduce(X: torch.Tensor, plan: GatherReducePlan, aggregation: str = "sum",
                      activation: Optional[str] = None) -> torch.Tensor:
    """out[s] = act( AGG_{p in segment s} w[p] * X[col[p]] )  — one fused HIP kernel.

    `activation` is fused as an epilogue when its derivative is recoverable from the output
    (everything in get_activation except gelu); gelu is applied by the caller."""
    mode = aggregation_mode_id(aggregation)
    act = activation_id(activation)
    if act not in _FUSABLE_ACTS or mode == _lib.AGG_MAX and act != _lib.ACT_LINEAR:
        raise ValueError("activation %r cannot be fused into the reduce epilogue" % activation)
    return _SegGatherReduce.apply(X, plan, mode, act)


# ---- drop-in tf.unsorted_segment_* --------------------------------------------------------
class SegmentPlanCache:
    """Plans for raw (segment_ids, num_segments) pairs, keyed by tensor identity."""

    def __init__(self, size=8):
        self._d = {}
        self._order = []
        self._size = size

    def get(self, segment_ids: torch.Tensor, num_segments: int) -> GatherReducePlan:
        key = (segment_ids.data_ptr(), segment_ids._version, segment_ids.numel(), int(num_segments))
        hit = self._d.get(key)
        if hit is not None:
            return hit[1]
        ids = segment_ids.to(torch.int32).contiguous()
        M = ids.numel()
        if M > 0:
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= num_segments:  # TF-CPU: InvalidArgumentError
                raise ValueError("segment id out of range [0, %d)" % num_segments)
        rowptr, perm, _ = build_segment_plan(ids, num_segments)
        lib = _lib.load_library()
        inv = torch.empty_like(perm)
        _lib.check(lib.relgnn_invert_perm(_lib.ptr(perm), M, _lib.ptr(inv), _lib.current_stream()),
                   "relgnn_invert_perm")
        plan = GatherReducePlan(
            rowptr=rowptr, stride=1, col=perm, w=None, num_out=num_segments, num_rows_x=M,
            rowptr_b=torch.arange(M + 1, dtype=torch.int32, device=ids.device), stride_b=1, col_b=ids,
            pos_b=inv, num_messages=M)
        self._d[key] = (segment_ids, plan)
        self._order.append(key)
        while len(self._order) > self._size:
            self._d.pop(self._order.pop(0), None)
        return plan


_SEGMENT_PLANS = SegmentPlanCache()


def _unsorted_segment(mode_name, data, segment_ids, num_segments):
    if data.dim() == 1:
        return _unsorted_segment(mode_name, data.unsqueeze(1), segment_ids, num_segments).squeeze(1)
    lead = data.shape[0]
    flat = data.reshape(lead, -1)
    plan = _SEGMENT_PLANS.get(segment_ids, int(num_segments))
    out = _SegGatherReduce.apply(flat, plan, aggregation_mode_id(mode_name), _lib.ACT_LINEAR)
    return out.reshape((int(num_segments),) + tuple(data.shape[1:]))


def unsorted_segment_sum(data, segment_ids, num_segments):
    """tf.unsorted_segment_sum(data, segment_ids, num_segments) on the HIP path."""
    return _unsorted_segment("sum", data, segment_ids, num_segments)


def unsorted_segment_mean(data, segment_ids, num_segments):
    return _unsorted_segment("mean", data, segment_ids, num_segments)


def unsorted_segment_sqrt_n(data, segment_ids, num_segments):
    return _unsorted_segment("sqrt_n", data, segment_ids, num_segments)


def unsorted_segment_max(data, segment_ids, num_segments):
    """Empty segments yield float32 lowest (-3.4028235e38), as TF does."""
    return _unsorted_segment("max", data, segment_ids, num_segments)
