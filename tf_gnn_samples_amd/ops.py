"""torch.autograd wrappers over the librelgnn C ABI (include/relgnn.h).

`unsorted_segment_{sum,mean,max,sqrt_n}` mirror the TF ops returned by the reference's
get_aggregation_function (utils/utils.py:23-33), same argument names and meaning.
`seg_gather_reduce` is the fused form the layer functions use: the gather
(tf.nn.embedding_lookup), the per-message scale and the segment reduction in ONE kernel.
"""
from typing import Optional

import torch

from . import _lib
from .config import settings as _cfg
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


# ---- long buckets -----------------------------------------------------------------------------------------------------
# The gather kernel gives ONE wave to a bucket and folds its messages in order: a hub with 1e5 incoming edges keeps one
# wave busy for milliseconds while the rest of the chip has long finished.  A graph that knows it has such buckets
# (RelGraph.split_long_segments: the lengths are read back once, where the graph is validated anyway) attaches a
# SplitPlan to the row-pointer tensor; _seg_reduce_raw then runs two launches of the SAME kernel: chunks of at most
# LONG_SEGMENT messages as virtual rows, then the virtual rows of a bucket combined in chunk order (deterministic; the
# fp32 summation order of a hub differs from the sequential fold by the chunking only).
LONG_SEGMENT = 4096


class SplitPlan:
    def __init__(self, rowptr: torch.Tensor, stride: int, num_out: int, chunk: int = LONG_SEGMENT):
        dev = rowptr.device
        bounds = rowptr[0:num_out * stride + 1:stride].long()                     # [num_out + 1]
        starts, ends = bounds[:-1], bounds[1:]
        self.lengths = ends - starts
        chunks = torch.clamp((self.lengths + chunk - 1) // chunk, min=1)
        off = torch.zeros(num_out + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(chunks, 0)
        self.num_virtual = int(off[-1])                                            # host sync, once per plan
        seg_of = torch.repeat_interleave(torch.arange(num_out, device=dev), chunks)
        first = starts[seg_of] + (torch.arange(self.num_virtual, device=dev) - off[seg_of]) * chunk
        self.virtual_rowptr = torch.cat([first, bounds[-1:]]).to(torch.int32).contiguous()
        self.combine_rowptr = off.to(torch.int32).contiguous()
        self.iota = torch.arange(self.num_virtual, dtype=torch.int32, device=dev)


_ACT_BY_ID = {_lib.ACT_TANH: torch.tanh, _lib.ACT_RELU: torch.relu,
              _lib.ACT_LEAKY_RELU: lambda x: torch.nn.functional.leaky_relu(x, 0.2),
              _lib.ACT_ELU: torch.nn.functional.elu, _lib.ACT_SELU: torch.selu,
              _lib.ACT_GELU: lambda x: torch.nn.functional.gelu(x, approximate="none")}


def _seg_reduce_split(mode, X, sp: SplitPlan, col, w, num_out, act):
    first = _lib.AGG_MAX if mode == _lib.AGG_MAX else _lib.AGG_SUM
    parts = _seg_reduce_raw(first, X, sp.virtual_rowptr, 1, col, w, sp.num_virtual)
    out = _seg_reduce_raw(first, parts, sp.combine_rowptr, 1, sp.iota, None, num_out)
    if mode in (_lib.AGG_MEAN, _lib.AGG_SQRT_N):
        n = sp.lengths.clamp(min=1).to(torch.float32).unsqueeze(1)
        out = out / (n if mode == _lib.AGG_MEAN else torch.sqrt(n))
    if act != _lib.ACT_LINEAR:
        out = _ACT_BY_ID[act](out)
    return out


def acc64_supported(D: int) -> bool:
    """Row widths relgnn_seg_reduce_acc64_fwd takes (the one-wave-per-row kernels)."""
    return D % 4 == 0 and 128 < D <= 1024


def rowmax_supported(X, rowptr, stride, acc64: bool = False) -> bool:
    """relgnn_seg_reduce_fwd_rowmax takes this gather (one wave holds a whole output row; not the hub-split route)."""
    split = getattr(rowptr, "_relgnn_split", None)
    D = X.shape[1]
    return (not acc64 and not (split is not None and stride in split) and X.is_cuda and D % 4 == 0 and 128 < D <= 1024
            and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0)


def _seg_reduce_raw(mode, X, rowptr, stride, col, w, num_out, act=_lib.ACT_LINEAR, acc64: bool = False, rowmax=None):
    """acc64: float64 bucket accumulators (relgnn_seg_reduce_acc64_fwd) — for sums that feed a GEMM, never for values that
    stand for the reference's own segment sums.  Split (hub) plans keep the float32 two-pass route.
    rowmax: a [num_out] float32 tensor that receives the largest magnitude of every output row (rowmax_supported)."""
    split = getattr(rowptr, "_relgnn_split", None)
    if split is not None and stride in split:
        return _seg_reduce_split(mode, X, split[stride], col, w, num_out, act)
    lib = _lib.load_library()
    D = X.shape[1]
    out = torch.empty((num_out, D), dtype=torch.float32, device=X.device)
    if rowmax is not None:
        _lib.check(lib.relgnn_seg_reduce_fwd_rowmax(
            mode, _lib.ptr(X, rows_strided=True), X.shape[0], X.stride(0), D, _lib.ptr(rowptr), num_out, stride,
            _lib.ptr(col), _lib.ptr(w), act, _lib.ptr(out), D, _lib.ptr(rowmax), _lib.current_stream()),
            "relgnn_seg_reduce_fwd_rowmax")
        return out
    if acc64 and mode != _lib.AGG_MAX and acc64_supported(D) and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0:
        _lib.check(lib.relgnn_seg_reduce_acc64_fwd(
            mode, _lib.ptr(X, rows_strided=True), X.shape[0], X.stride(0), D, _lib.ptr(rowptr), num_out, stride,
            _lib.ptr(col), _lib.ptr(w), act, _lib.ptr(out), D, _lib.current_stream()), "relgnn_seg_reduce_acc64_fwd")
        return out
    _lib.check(lib.relgnn_seg_reduce_fwd(
        mode, _lib.ptr(X, rows_strided=True), X.shape[0], X.stride(0), D, _lib.ptr(rowptr), num_out, stride,
        _lib.ptr(col), _lib.ptr(w), act, _lib.ptr(out), D, _lib.current_stream()),
        "relgnn_seg_reduce_fwd")
    return out


class _SegGatherReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, plan: GatherReducePlan, mode: int, act: int):
        _check_f32(X, "X")
        if X.dim() != 2 or X.shape[0] != plan.num_rows_x:
            raise ValueError("X must be [%d, D], got %s" % (plan.num_rows_x, tuple(X.shape)))
        if X.stride(1) != 1 or (mode == _lib.AGG_MAX and not X.is_contiguous()):
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
        segments = int(num_segments)
        if M > 0:
            lo, hi = int(ids.min()), int(ids.max())
            if hi >= num_segments:  # TF-CPU: InvalidArgumentError
                raise ValueError("segment id out of range [0, %d)" % num_segments)
            if lo < 0:
                # tf.unsorted_segment_*: "if the given segment ID is negative, the value is dropped".  Dropped rows go
                # to one extra bucket behind the real ones; the caller slices it off (and autograd's slice backward
                # hands those rows a zero gradient).
                ids = torch.where(ids < 0, torch.full_like(ids, segments), ids)
                segments += 1
        num_segments = segments
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
    if plan.num_out > int(num_segments):          # the bucket of rows with negative ids (dropped, like TF)
        out = out[:int(num_segments)]
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


# ---- fused edge-wise message kernels (csrc/edge_fused.hip) ----------------------------------
def _mode_factor(graph, mode: int):
    """d(finalised aggregate)/d(raw sum) per target node: 1, 1/max(n,1) or 1/sqrt(max(n,1))."""
    if mode == _lib.AGG_SUM:
        return None
    n = graph.messages_per_target()
    return (1.0 / n) if mode == _lib.AGG_MEAN else (1.0 / torch.sqrt(n))


class _FusedEdgeMessages(torch.autograd.Function):
    """kind 'film': out[v] = AGG act(gamma[v,l] * (w * T[src,l]) + beta[v,l]),  A = film [V*L, 2D]
       kind 'pair': out[v] = AGG act(w * (T[src,l] + A[v,l])),                  A = Q    [V*L, D]"""

    @staticmethod
    def forward(ctx, T, A, graph, w, mode: int, act: int, kind: str, pairs=None):
        """pairs (graph.PairTables, FiLM only): T / A hold rows for the non-empty (node,type) buckets only."""
        lib = _lib.load_library()
        _check_f32(T, "T"); _check_f32(A, "A")
        T, A = T.contiguous(), A.contiguous()
        V, L = graph.V, graph.L
        D = T.shape[1]
        rows_t, rows_a = (V * L, V * L) if pairs is None else (pairs.P_s, pairs.P_t)
        if pairs is not None and kind != "film":
            raise ValueError("compact pair tables are only wired into the FiLM kernels")
        if T.shape[0] != rows_t or A.shape[0] != rows_a or A.shape[1] != (2 * D if kind == "film" else D):
            raise ValueError("bad operand shapes for fused %s messages" % kind)
        out = torch.empty((V, D), dtype=torch.float32, device=T.device)
        if kind == "film":
            col = graph.col_t if pairs is None else pairs.col_t
            brow = None if pairs is None else pairs.tgt.bucket_row
            _lib.check(lib.relgnn_film_fwd(mode, act, _lib.ptr(T), D, _lib.ptr(A), A.shape[1], D, _lib.ptr(graph.rowptr_t),
                                           V, L, _lib.ptr(col), _lib.ptr(w), _lib.ptr(out), D, _lib.ptr(brow),
                                           _lib.current_stream()), "relgnn_film_fwd")
        else:
            _lib.check(lib.relgnn_pair_fwd(mode, act, _lib.ptr(T), D, _lib.ptr(A), A.shape[1], D, _lib.ptr(graph.rowptr_t),
                                           V, L, _lib.ptr(graph.col_t), _lib.ptr(w), _lib.ptr(out), D,
                                           _lib.current_stream()), "relgnn_pair_fwd")
        ctx.graph, ctx.w, ctx.mode, ctx.act, ctx.kind, ctx.pairs = graph, w, mode, act, kind, pairs
        ctx.save_for_backward(T, A)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load_library()
        st = _lib.current_stream()
        graph, w, mode, act, kind, pairs = ctx.graph, ctx.w, ctx.mode, ctx.act, ctx.kind, ctx.pairs
        if mode == _lib.AGG_MAX:
            raise RuntimeError("fused %s messages have no max-aggregation backward; the layer uses the unfused path" % kind)
        T, A = ctx.saved_tensors
        V, L = graph.V, graph.L
        D = T.shape[1]
        f = _mode_factor(graph, mode)
        gagg = (gout * f.unsqueeze(1)).contiguous() if f is not None else gout.contiguous()
        gA = torch.empty_like(A)
        if pairs is not None:
            # compact tables: every real row is written by the kernels; the few padding rows are not and feed the
            # batched weight-gradient GEMM (against all-zero inputs, but 0 * NaN garbage would still poison it)
            _fill_rows(gA, pairs.tgt.pad_rows, 0.0)
        # Two ways to the gradient of the gathered rows (RELGNN_EDGE_BWD=emit|regather overrides the choice):
        #  emit:     pass A (by target) also writes every message's gradient w.r.t. its gathered row ([M, D]); gT is then
        #            one plain gather-reduce of those rows over the by-source buckets;
        #  regather: pass B walks the by-source buckets and re-gathers the per-bucket row (8D bytes for FiLM) and the
        #            target's gradient row once per MESSAGE, recomputing the pre-activation: nothing [M, D] is written.
        # Measured on MI355X (scripts/bench_configs.py): the wave kernels (D > 128) on dense tables win with regather
        # (FiLM on the C2 batch: 833 + 426 us -> 330 + 536 us per layer, step 9.06 -> 7.66 ms); the lane-group kernels on
        # compact pair tables (C5: D = 128, 23 types) with emit (47.4 vs 52.9 ms).
        choice = _cfg.edge_bwd if _cfg.edge_bwd != "auto" else ("emit" if (pairs is not None or D <= 128) else "regather")
        emit = choice == "emit"
        dmsg = torch.empty((graph.M, D), dtype=torch.float32, device=T.device) if emit else None
        # RELGNN_EDGE_SIGN_MASK=1 (opt-in; regather + piecewise-linear activation + wave kernels with one float4 per lane):
        # pass A leaves one sign bit per feature and message, pass B then gathers gamma and the target's gradient row only.
        # Measured on the C2 batch: pass B 536 -> 419 us, but the four ballots + the mask store cost pass A 330 -> 402 us and
        # the step does not get faster (7.31 / 7.27 ms without, 7.43 / 7.41 ms with, one A/B call) -> off by default.
        smask = None
        if (not emit and kind == "film" and pairs is None and act in (_lib.ACT_LINEAR, _lib.ACT_RELU, _lib.ACT_LEAKY_RELU)
                and 128 < D <= 256 and L <= 63 and V * L * (D // 4) < 2 ** 32 and graph.M > 0
                and _cfg.edge_sign_mask == "1"):
            smask = torch.empty((graph.M, 4), dtype=torch.int64, device=T.device)
        if kind == "film":
            col = graph.col_t if pairs is None else pairs.col_t
            brow_t = None if pairs is None else pairs.tgt.bucket_row
            _lib.check(lib.relgnn_film_bwd_film(act, _lib.ptr(T), D, _lib.ptr(A), A.shape[1], D, _lib.ptr(graph.rowptr_t),
                                                V, L, _lib.ptr(col), _lib.ptr(w), _lib.ptr(gagg), D,
                                                _lib.ptr(gA), A.shape[1], _lib.ptr(brow_t), _lib.ptr(dmsg), _lib.ptr(smask), st),
                       "relgnn_film_bwd_film")
        else:
            _lib.check(lib.relgnn_pair_bwd_q(act, _lib.ptr(T), D, _lib.ptr(A), D, D, _lib.ptr(graph.rowptr_t), V, L,
                                             _lib.ptr(graph.col_t), _lib.ptr(w), _lib.ptr(gagg), D, _lib.ptr(gA), D,
                                             _lib.ptr(dmsg), st), "relgnn_pair_bwd_q")
        if emit:
            if pairs is None:
                gT = _seg_reduce_raw(_lib.AGG_SUM, dmsg, graph.rowptr_s, 1, graph.pos_t_of_s, None, V * L)
            else:
                rowptr_c, _, pos_c = pairs._messages_by_source_row()
                gT = _seg_reduce_raw(_lib.AGG_SUM, dmsg, rowptr_c, 1, pos_c, None, pairs.P_s)   # padding rows: zero
        elif smask is not None:
            gT = torch.empty_like(T)
            _lib.check(lib.relgnn_film_bwd_msg_masked(act, _lib.ptr(A), A.shape[1], D, _lib.ptr(graph.rowptr_s), V * L,
                                                      _lib.ptr(graph.tgt_s), _lib.ptr(graph.frow_s),
                                                      _lib.ptr(graph.w_by_source(w)), _lib.ptr(graph.pos_t_of_s),
                                                      _lib.ptr(smask), _lib.ptr(gagg), D, _lib.ptr(gT), D, st),
                       "relgnn_film_bwd_msg_masked")
        elif kind == "film":
            gT = torch.empty_like(T)
            if pairs is not None:
                _fill_rows(gT, pairs.src.pad_rows, 0.0)
            frow = graph.frow_s if pairs is None else pairs.frow_s
            brow_s = None if pairs is None else pairs.src.bucket_row
            _lib.check(lib.relgnn_film_bwd_msg(act, _lib.ptr(T), D, _lib.ptr(A), A.shape[1], D, _lib.ptr(graph.rowptr_s),
                                               V * L, _lib.ptr(graph.tgt_s), _lib.ptr(frow),
                                               _lib.ptr(graph.w_by_source(w)), _lib.ptr(gagg), D, _lib.ptr(gT), D,
                                               _lib.ptr(brow_s), st), "relgnn_film_bwd_msg")
        else:
            gT = torch.empty_like(T)
            _lib.check(lib.relgnn_pair_bwd_p(act, _lib.ptr(T), D, _lib.ptr(A), D, D, _lib.ptr(graph.rowptr_s), V * L,
                                             _lib.ptr(graph.tgt_s), _lib.ptr(graph.frow_s), _lib.ptr(graph.w_by_source(w)),
                                             _lib.ptr(gagg), D, _lib.ptr(gT), D, st), "relgnn_pair_bwd_p")
        return gT, gA, None, None, None, None, None, None


def film_messages_reduce(T, film, graph, w, aggregation: str, activation: Optional[str], pairs=None):
    """gnns/gnn_film.py:92-116 in one kernel (sum / mean / sqrt_n; max forward only).  With `pairs`
    (graph.PairTables) T is [P_s, D] and film [P_t, 2D]: rows for the non-empty (node,type) buckets only."""
    D = T.shape[1]
    pad = (-D) % 4
    if pad:                                       # [gamma | beta] halves padded separately (see _pad_columns)
        film = torch.nn.functional.pad(film.reshape(film.shape[0], 2, D), (0, pad)).reshape(film.shape[0], 2 * (D + pad))
        return _FusedEdgeMessages.apply(_pad_columns(T, pad), film, graph, w, aggregation_mode_id(aggregation),
                                        activation_id(activation), "film", pairs)[:, :D]
    return _FusedEdgeMessages.apply(T, film, graph, w, aggregation_mode_id(aggregation), activation_id(activation),
                                    "film", pairs)


class _TypedLinear(torch.autograd.Function):
    """Y[r] = H[node[r]] @ W_{type(r)} for the compact rows of one graph.SidePairs.

    The table is a batch of [chunk, Din] tiles, each of ONE edge type (type blocks are padded to the tile size), so the
    L per-type transforms are ONE batched GEMM with the weight picked per tile; the dense path computes
    H @ [W_0|..|W_{L-1}] for ALL V*L (node,type) rows instead.  Backward: dX = dY W^T (batched), per-tile partial
    weight gradients X^T dY (batched) summed per type by the gather-reduce kernel, and the per-row dX summed back into
    their nodes over the node -> rows CSR (ascending type order); no atomics anywhere."""

    @staticmethod
    def forward(ctx, H, side, *weights):
        V, Din = H.shape
        Dout = weights[0].shape[1]
        c = side.chunk
        Hz = torch.cat([H, H.new_zeros((1, Din))])                            # row V = the padding rows' input
        X = Hz.index_select(0, side.node)                                     # [P, Din]
        Wt = torch.stack(weights).index_select(0, side.chunk_type)            # [tiles, Din, Dout]
        Y = torch.bmm(X.view(-1, c, Din), Wt).view(side.P, Dout)
        ctx.side, ctx.num_nodes, ctx.shape = side, V, (len(weights), Din, Dout)
        ctx.save_for_backward(X, Wt)
        return Y

    @staticmethod
    def backward(ctx, gY):
        X, Wt = ctx.saved_tensors
        side, c = ctx.side, ctx.side.chunk
        L, Din, Dout = ctx.shape
        gY = gY.contiguous()
        gH = None
        if ctx.needs_input_grad[0]:
            gX = torch.bmm(gY.view(-1, c, Dout), Wt.transpose(1, 2)).view(side.P, Din)
            gH = _seg_reduce_raw(_lib.AGG_SUM, gX, side.node_rowptr, 1, side.node_col, None, ctx.num_nodes)
        gW = [None] * L
        if any(ctx.needs_input_grad[2:]):
            part = torch.bmm(X.view(-1, c, Din).transpose(1, 2), gY.view(-1, c, Dout))   # [tiles, Din, Dout]
            n = Din * Dout
            sub = 1024 if n % 1024 == 0 else (n if n <= 1024 and n % 4 == 0 else 0)
            if sub:
                K = n // sub
                rowptr, col = side.weight_grad_plan(K)
                g = _seg_reduce_raw(_lib.AGG_SUM, part.view(-1, sub), rowptr, 1, col, None, L * K).view(L, Din, Dout)
            else:                                                                 # odd shapes: per-type sums
                g = torch.stack([part[a:a + k].sum(0) for a, k in zip(_tile_starts(side), side.chunk_counts)])
            gW = [g[l] if ctx.needs_input_grad[2 + l] else None for l in range(L)]
        return (gH, None, *gW)


def _tile_starts(side):
    out, at = [], 0
    for k in side.chunk_counts:
        out.append(at)
        at += k
    return out


class _BlockedLinear(torch.autograd.Function):
    """Y[a_l:b_l] = X[a_l:b_l] @ W_l for consecutive row blocks (the per-edge-type Dense layers of an edge MLP on the
    type-major message list, utils/utils.py:120-126 via gnns/gnn_edge_mlp.py:102).  One autograd node instead of L
    slices + L GEMM nodes + cat: outputs and input gradients are written straight into their row blocks (no concat,
    no zero-filled accumulation buffer), weight gradients use the split-K reduction over the block's rows."""

    @staticmethod
    def forward(ctx, X, offsets, *weights):
        from .dense import GEMM_NN, mm_into
        X = X.contiguous()
        Y = torch.empty((X.shape[0], weights[0].shape[1]), dtype=X.dtype, device=X.device)
        for l, W in enumerate(weights):
            a, b = offsets[l], offsets[l + 1]
            if b > a:
                mm_into(GEMM_NN, X[a:b], W, Y[a:b])
        ctx.offsets = offsets
        ctx.save_for_backward(X, *weights)
        return Y

    @staticmethod
    def backward(ctx, gY):
        from .dense import GEMM_NT, matmul_tn_splitk, mm_into
        X, *weights = ctx.saved_tensors
        offsets = ctx.offsets
        gY = gY.contiguous()
        gX = torch.empty_like(X) if ctx.needs_input_grad[0] else None
        gW = []
        for l, W in enumerate(weights):
            a, b = offsets[l], offsets[l + 1]
            if b > a:
                if gX is not None:
                    mm_into(GEMM_NT, gY[a:b], W, gX[a:b])
                gW.append(matmul_tn_splitk(X[a:b], gY[a:b]) if ctx.needs_input_grad[2 + l] else None)
            else:
                gW.append(torch.zeros_like(W) if ctx.needs_input_grad[2 + l] else None)
        return (gX, None, *gW)


def blocked_linear(X, offsets, weights):
    """X [M, Din] in consecutive row blocks offsets[l]:offsets[l+1]; weights: one [Din, Dout] kernel per block."""
    _check_f32(X, "X")
    return _BlockedLinear.apply(X, [int(o) for o in offsets], *weights)


def _fill_rows(X: torch.Tensor, rows: torch.Tensor, value: float) -> None:
    """X[rows] = value (rows: int64 ids on the device) — relgnn_fill_rows_f32; torch's index_fill_ took 30 us for a few thousand rows."""
    if rows.numel() == 0:
        return
    if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1 and rows.dtype == torch.int64 and rows.is_contiguous()):
        X.index_fill_(0, rows, value)
        return
    _lib.check(_lib.load_library().relgnn_fill_rows_f32(X.data_ptr(), X.stride(0), X.shape[1], rows.data_ptr(), rows.numel(), float(value),
                                                        _lib.current_stream()), "relgnn_fill_rows_f32")


def _typed_weight_gradient(H, gY, side, L: int, Din: int, Dout: int):
    """[L, Din, Dout]: per-tile partials gather(H)^T @ gY (one launch), summed per edge type in tile order."""
    from .dense import GEMM_TN, limb_gemm_tn_tiles, limb_tn_tiles_supported, panel_gemm
    node32, _ = side.panel_indices()
    tiles = side.P // side.chunk
    if ((_cfg.typed_tn == "limb" or (_cfg.typed_tn == "auto" and Dout % 256 == 0))
            and limb_tn_tiles_supported(H, gY, node32, side.chunk)):
        # (round 5, isolated at a C5-sized table of 1400 tiles: three-limb TN 405 us vs exact-fp32 panel TN 484 us for
        #  [128, 256] partials, 325 vs 262 us for [128, 128] ones — the panel kernel runs near the fp32 matrix pipe's rate;
        #  the 0.6 / 0.96 ms per launch of the round-4 traces were contention on the side stream, not the kernel)
        part = limb_gemm_tn_tiles(H, gY, node32, side.chunk)                 # [tiles, Din, Dout]
    else:
        part = panel_gemm(GEMM_TN, H, gY, a_rows=node32, batch=tiles, strides=(0, side.chunk * Dout, Din * Dout),
                          dims=(Din, Dout, side.chunk))                      # [tiles, Din, Dout]
    n = Din * Dout
    sub = 1024 if n % 1024 == 0 else n
    K = n // sub
    rowptr, col = side.weight_grad_plan(K)
    return _seg_reduce_raw(_lib.AGG_SUM, part.view(-1, sub), rowptr, 1, col, None, L * K).view(L, Din, Dout)


class _TypedLinearPair(torch.autograd.Function):
    """The two per-(node, type) transforms of a GNN-FiLM layer (gnns/gnn_film.py:92-106: messages h_u W_l over the by-source pair
    table, FiLM weights h_v F_l over the by-target one) as ONE autograd node: the forward is the two products of
    _TypedLinearPanel; the backward sums BOTH tables' per-row input gradients into the nodes with one gather-reduce over a
    combined node -> rows CSR (graph.PairTables.node_csr_both) instead of two reductions and an addition of their [V, D] results
    (round 6: one launch and ~0.1 ms less per layer of the C5 step).  Cached limb images only (dense.sel_image)."""

    @staticmethod
    def forward(ctx, H, pairs, leaves, La: int, *weights):
        from .dense import GEMM_NN, limb_dense_sel, sel_image
        wa, wb = weights[:La], weights[La:]
        ctx.leaf_params, ctx.pairs, ctx.La = leaves, pairs, La
        outs = []
        for side, ws in ((pairs.src, wa), (pairs.tgt, wb)):
            node32, tile_type = side.panel_indices()
            outs.append(limb_dense_sel(GEMM_NN, H, ws, a_rows=node32, num_rows=side.P, b_select=tile_type,
                                       rows_per_select=side.chunk, image=sel_image(ws, GEMM_NN)))
        ctx.save_for_backward(H, *weights)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, gYa, gYb):
        from .dense import GEMM_NT, limb_dense_sel, sel_image
        H, *weights = ctx.saved_tensors
        pairs, La = ctx.pairs, ctx.La
        wa, wb = weights[:La], weights[La:]
        sides = ((pairs.src, wa, gYa.contiguous()), (pairs.tgt, wb, gYb.contiguous()))
        Din = wa[0].shape[0]
        gH, gWs = None, None
        want_w = any(ctx.needs_input_grad[4:])

        def weight_gradients():
            return tuple(_typed_weight_gradient(H, g, side, len(ws), Din, ws[0].shape[1]) for side, ws, g in sides)

        side_stream = None
        if (want_w and ctx.needs_input_grad[0] and _cfg.bwd_overlap_on and H.is_cuda
                and (not _DEFER["on"] or deferred_targets_ok(ctx.leaf_params, H.device))):
            side_stream = _side_stream(H.device)
            cur = torch.cuda.current_stream(H.device)
            side_stream.wait_stream(cur)
            with torch.cuda.stream(side_stream):
                gWs = weight_gradients()
            for t in (H, sides[0][2], sides[1][2]):
                t.record_stream(side_stream)
        if ctx.needs_input_grad[0]:
            Pa, Pb = pairs.src.P, pairs.tgt.P
            gX = torch.empty((Pa + Pb, Din), dtype=torch.float32, device=H.device)
            at = 0
            images = [sel_image(ws, GEMM_NT) for _, ws, _ in sides]
            for (side, ws, g), im in zip(sides, images):
                _, tile_type = side.panel_indices()
                if im is not None:
                    limb_dense_sel(GEMM_NT, g, ws, b_select=tile_type, rows_per_select=side.chunk, image=im, out=gX[at:at + side.P])
                else:                                                   # (switched off between forward and backward)
                    gX[at:at + side.P] = limb_dense_sel(GEMM_NT, g, torch.stack(ws), b_select=tile_type, rows_per_select=side.chunk)
                at += side.P
            rowptr, col = pairs.node_csr_both()
            gH = _seg_reduce_raw(_lib.AGG_SUM, gX, rowptr, 1, col, None, H.shape[0])
        if side_stream is not None:
            flat = gWs[0].unbind(0) + gWs[1].unbind(0)
            if _DEFER["on"]:
                hand_over_deferred(H.device, side_stream, ctx.leaf_params, flat)
            else:
                torch.cuda.current_stream(H.device).wait_stream(side_stream)
            for g in gWs:
                g.record_stream(torch.cuda.current_stream(H.device))
        elif want_w:
            wait_if_in_flight(ctx.leaf_params, H.device)
            gWs = weight_gradients()
        need = ctx.needs_input_grad[4:]
        grads = (gWs[0].unbind(0) + gWs[1].unbind(0)) if gWs is not None else (None,) * len(weights)
        return (gH, None, None, None) + tuple(g if n else None for g, n in zip(grads, need))


def typed_linear_pair(H, pairs, weights_src, weights_tgt):
    """(typed_linear(H, pairs.src, weights_src), typed_linear(H, pairs.tgt, weights_tgt)) with ONE reduction of both input
    gradients into the nodes (_TypedLinearPair) where the cached limb route applies; else the two separate calls."""
    from .dense import GEMM_NN, GEMM_NT, sel_image
    weights_src, weights_tgt = list(weights_src), list(weights_tgt)
    _check_f32(H, "H")
    H = H.contiguous()
    ok = (_typed_panel_ok(H, pairs.src, weights_src) and _typed_panel_ok(H, pairs.tgt, weights_tgt)
          and weights_src[0].shape[0] == weights_tgt[0].shape[0])
    if ok:
        for ws in (weights_src, weights_tgt):
            Din, Dout = ws[0].shape
            ok = ok and _typed_limb_ok(Din, Dout) and _typed_limb_ok(Dout, Din) and sel_image(ws, GEMM_NN) is not None \
                and sel_image(ws, GEMM_NT) is not None
    if not ok:
        return typed_linear(H, pairs.src, weights_src), typed_linear(H, pairs.tgt, weights_tgt)
    allw = weights_src + weights_tgt
    leaves = tuple(allw) if all(w.is_leaf and w.requires_grad for w in allw) else None
    return _TypedLinearPair.apply(H, pairs, leaves, len(weights_src), *allw)


class _TypedLinearPanel(torch.autograd.Function):
    """_TypedLinear on the row-panel MFMA kernel (csrc/panel_gemm.hip): the gather H[node[r]] and the per-tile kernel
    selection happen in the kernel's load addresses.  Nothing [P, Din] (the gathered rows) and nothing [tiles, Din, Dout] (a
    per-tile copy of the kernels) is materialised, forward or backward:
      forward   Y  = gather(H) @ W_type            one launch (NN, gathered rows, per-tile B)
      backward  dX = dY @ W_type^T                 one launch (NT with the kernels as stored), summed into nodes over the
                                                   node -> rows CSR by the gather-reduce kernel (ascending type order)
                dW = per-tile gather(H)^T @ dY     one launch (TN, gathered reduction rows, one product per tile), the per-tile
                                                   partials summed per type by the gather-reduce kernel (tile order)"""

    @staticmethod
    def forward(ctx, H, side, leaves, *weights):
        from .dense import GEMM_NN, GEMM_NT, limb_dense_sel, panel_gemm, sel_image
        ctx.leaf_params = leaves            # the per-type weights themselves, when every one is a leaf parameter
        L, (Din, Dout) = len(weights), weights[0].shape
        node32, tile_type = side.panel_indices()
        # round 6: the limb images of the per-type weights come from the step's cache (dense.weight_image(separate=True)): neither a
        # stacked [L, Din, Dout] copy nor a split launch in front of the product
        im = sel_image(weights, GEMM_NN) if (_typed_limb_ok(Din, Dout) and _typed_limb_ok(Dout, Din)) else None
        cached = im is not None
        W = None if cached else torch.stack(weights)
        if cached:
            Y = limb_dense_sel(GEMM_NN, H, weights, a_rows=node32, num_rows=side.P, b_select=tile_type,
                               rows_per_select=side.chunk, image=im)
        elif _typed_limb_ok(Din, Dout):
            Y = limb_dense_sel(GEMM_NN, H, W, a_rows=node32, num_rows=side.P, b_select=tile_type, rows_per_select=side.chunk)
        else:
            Y = panel_gemm(GEMM_NN, H, W, a_rows=node32, num_rows=side.P, b_select=tile_type, rows_per_select=side.chunk)
        ctx.side, ctx.shape, ctx.cached = side, (L, Din, Dout), cached
        if cached:
            ctx.save_for_backward(H, *weights)
        else:
            ctx.save_for_backward(H, W)
        return Y

    @staticmethod
    def backward(ctx, gY):
        from .dense import GEMM_NT, GEMM_TN, limb_dense_sel, panel_gemm
        H, *saved = ctx.saved_tensors
        W = None if ctx.cached else saved[0]
        side = ctx.side
        L, Din, Dout = ctx.shape
        gY = gY.contiguous()
        node32, tile_type = side.panel_indices()
        gH = gW = None

        def weight_gradient():
            return _typed_weight_gradient(H, gY, side, L, Din, Dout)

        # The weight gradient (exact-fp32 matrix pipe, ~0.25-0.5 ms per product on a 23-type batch) depends on nothing the input
        # gradient computes: it runs on the side stream of the aggregate-first layer's weight gradient, under the memory-bound
        # kernels that follow on the main stream (the other typed product's row sums, the next layer's fused edge backward); the
        # join is deferred behind the whole backward inside train_step (deferred_weight_gradient_join above).
        side_stream = None
        want_w = any(ctx.needs_input_grad[3:])
        if (want_w and ctx.needs_input_grad[0] and _cfg.bwd_overlap_on and gY.is_cuda
                and (not _DEFER["on"] or deferred_targets_ok(ctx.leaf_params, gY.device))):
            side_stream = _side_stream(gY.device)
            cur = torch.cuda.current_stream(gY.device)
            side_stream.wait_stream(cur)
            with torch.cuda.stream(side_stream):
                gW = weight_gradient()
            for t in (H, gY):
                t.record_stream(side_stream)
        if ctx.needs_input_grad[0]:
            im = None
            if ctx.cached:
                from .dense import sel_image
                im = sel_image(saved, GEMM_NT)
                if im is None:                       # (switched off between forward and backward)
                    W = torch.stack(saved)
            if im is not None:
                gX = limb_dense_sel(GEMM_NT, gY, saved, b_select=tile_type, rows_per_select=side.chunk, image=im)
            elif _typed_limb_ok(Dout, Din):
                gX = limb_dense_sel(GEMM_NT, gY, W, b_select=tile_type, rows_per_select=side.chunk)
            else:
                gX = panel_gemm(GEMM_NT, gY, W, b_select=tile_type, rows_per_select=side.chunk, dims=(side.P, Din, Dout))
            gH = _seg_reduce_raw(_lib.AGG_SUM, gX, side.node_rowptr, 1, side.node_col, None, H.shape[0])
        if side_stream is not None:
            if _DEFER["on"]:
                hand_over_deferred(gY.device, side_stream, ctx.leaf_params, gW.unbind(0))
            else:
                torch.cuda.current_stream(gY.device).wait_stream(side_stream)
            gW.record_stream(torch.cuda.current_stream(gY.device))
        elif want_w:
            wait_if_in_flight(ctx.leaf_params, gY.device)     # (no-op unless an earlier use of these weights went aside)
            gW = weight_gradient()
        need = ctx.needs_input_grad[3:]
        return (gH, None, None) + (tuple(g if n else None for g, n in zip(gW.unbind(0), need)) if gW is not None else (None,) * L)


def _typed_limb_ok(k: int, n: int) -> bool:
    """The limb route (relgnn_limb_dense_sel_f32) for a typed product with reduction length k and n output columns."""
    from . import dense
    return _cfg.limb_gemm and n % 128 == 0 and k % 16 == 0 and 16 <= k <= dense._LIMB_MAX_K


def _typed_panel_ok(H, side, weights) -> bool:
    Din, Dout = weights[0].shape
    # (forward: N = Dout; input gradient: N = Din; both must be panel widths)
    return (_cfg.typed == "panel" and H.is_cuda and side.chunk == 512 and Din % 64 == 0
            and Dout % 64 == 0 and side.P > 0 and H.data_ptr() % 16 == 0)


def typed_linear(H, side, weights):
    """[P, Dout] table over the non-empty (node,type) buckets of `side` (graph.SidePairs); weights: L x [Din, Dout]."""
    _check_f32(H, "H")
    H = H.contiguous()
    if _typed_panel_ok(H, side, weights):
        weights = list(weights)
        leaves = tuple(weights) if all(w.is_leaf and w.requires_grad for w in weights) else None
        return _TypedLinearPanel.apply(H, side, leaves, *weights)
    return _TypedLinear.apply(H, side, *weights)


def pair_messages_reduce_fused(P, Q, graph, w, aggregation: str, activation: Optional[str]):
    D = P.shape[1]
    pad = (-D) % 4
    if pad:
        return _FusedEdgeMessages.apply(_pad_columns(P, pad), _pad_columns(Q, pad), graph, w, aggregation_mode_id(aggregation),
                                        activation_id(activation), "pair")[:, :D]
    return _FusedEdgeMessages.apply(P, Q, graph, w, aggregation_mode_id(aggregation), activation_id(activation), "pair")


class _PairMaterialize(torch.autograd.Function):
    """hidden[m] = act(P[src_m*L+l_m] + Q[tgt_m*L+l_m]) in the reference's type-major message order."""

    @staticmethod
    def forward(ctx, P, Q, graph, act: int):
        lib = _lib.load_library()
        P = P.contiguous()
        Q = Q.contiguous() if Q is not None else None
        M, D = graph.M, P.shape[1]
        out = torch.empty((M, D), dtype=torch.float32, device=P.device)
        _lib.check(lib.relgnn_pair_materialize(act, _lib.ptr(P), D, _lib.ptr(Q), D, D, _lib.ptr(graph.key_by_source),
                                               _lib.ptr(graph.key_by_target), M, None, _lib.ptr(out), D,
                                               _lib.current_stream()), "relgnn_pair_materialize")
        ctx.graph, ctx.act, ctx.has_q = graph, act, Q is not None
        ctx.save_for_backward(P, Q)
        return out

    @staticmethod
    def backward(ctx, ghidden):
        lib = _lib.load_library()
        graph, act = ctx.graph, ctx.act
        P, Q = ctx.saved_tensors
        M, D = graph.M, P.shape[1]
        ghidden = ghidden.contiguous()
        S = graph.V * graph.L
        if (ctx.has_q and _cfg.edge_bwd != "emit" and D % 4 == 0 and 128 < D <= 1024
                and M * (D // 4) < 2 ** 32 and S > 0):
            # No [M, D] gradient of the pre-activation is written and re-read twice: the by-(source,type) pass of the pair
            # kernels sums g_m * act'(P[r] + Q[f_m]) per bucket r straight from ghidden's rows (its "target gradient row" is
            # the message's own row here), once over the by-source buckets for gP and once over the by-target buckets with
            # the roles of P and Q swapped for gQ.  C2 shape, elu: 959 + 2 x 423 us -> 2 x ~500 us per layer.
            gP, gQ = torch.empty_like(P), torch.empty_like(Q)
            st = _lib.current_stream()
            _lib.check(lib.relgnn_pair_bwd_p(act, _lib.ptr(P), D, _lib.ptr(Q), D, D, _lib.ptr(graph.rowptr_s), S,
                                             _lib.ptr(graph.perm_s), _lib.ptr(graph.frow_s), None, _lib.ptr(ghidden), D,
                                             _lib.ptr(gP), D, st), "relgnn_pair_bwd_p")
            _lib.check(lib.relgnn_pair_bwd_p(act, _lib.ptr(Q), D, _lib.ptr(P), D, D, _lib.ptr(graph.rowptr_t), S,
                                             _lib.ptr(graph.perm_t), _lib.ptr(graph.col_t), None, _lib.ptr(ghidden), D,
                                             _lib.ptr(gQ), D, st), "relgnn_pair_bwd_p")
            return gP, gQ, None, None
        gpre = torch.empty_like(ghidden)
        _lib.check(lib.relgnn_pair_materialize(act, _lib.ptr(P), D, _lib.ptr(Q), D, D, _lib.ptr(graph.key_by_source),
                                               _lib.ptr(graph.key_by_target), M, _lib.ptr(ghidden), _lib.ptr(gpre), D,
                                               _lib.current_stream()), "relgnn_pair_materialize")
        # gP[r] = sum of gpre over the messages whose source row is r; gQ[f] likewise by target row
        gP = _seg_reduce_raw(_lib.AGG_SUM, gpre, graph.rowptr_s, 1, graph.perm_s, None, S)
        gQ = _seg_reduce_raw(_lib.AGG_SUM, gpre, graph.rowptr_t, 1, graph.perm_t, None, S) if ctx.has_q else None
        return gP, gQ, None, None


def _pad_columns(X, pad: int):
    """[rows, D] -> [rows, D + pad] with zero columns: the edge kernels move 16-byte pieces of a row (D % 4 == 0).  A width the
    reference accepts and they do not (hidden_size 15) runs on padded tables: every message activation maps 0 to 0, so the padded
    columns stay 0 through product, scale, activation and every aggregation, and are cut off again (differentiable: F.pad / slice)."""
    return torch.nn.functional.pad(X, (0, pad)) if X is not None else None


def pair_materialize(P, Q, graph, activation: Optional[str]):
    D = P.shape[1]
    pad = (-D) % 4
    if pad:
        return _PairMaterialize.apply(_pad_columns(P, pad), _pad_columns(Q, pad), graph, activation_id(activation))[:, :D]
    return _PairMaterialize.apply(P, Q, graph, activation_id(activation))


# ---- RGAT (csrc/rgat_fast.hip, generic fallback csrc/rgat.hip) -------------------------------------
def _rgat_fast_ok(D: int, K: int) -> bool:
    return K in (1, 2, 4, 8) and D % K == 0 and (D // K) % 4 == 0 and D <= 1024


def _rgat_dz_fast_ok(D: int, K: int) -> bool:
    dh4 = D // K // 4 if K and D % (4 * K) == 0 else 0
    return _rgat_fast_ok(D, K) and D <= 256 and dh4 > 0 and (dh4 & (dh4 - 1)) == 0


class _RgatAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T, s_src, s_tgt, graph, num_heads: int, slope: float):
        lib = _lib.load_library()
        st = _lib.current_stream()
        T, s_src, s_tgt = T.contiguous(), s_src.contiguous(), s_tgt.contiguous()
        V, L, M = graph.V, graph.L, graph.M
        D = T.shape[1]
        out = torch.empty((V, D), dtype=torch.float32, device=T.device)
        alpha = torch.empty((M, num_heads), dtype=torch.float32, device=T.device)
        fast = _rgat_fast_ok(D, num_heads)
        if fast:
            _lib.check(lib.relgnn_rgat_alpha(_lib.ptr(s_src), _lib.ptr(s_tgt), num_heads, _lib.ptr(graph.rowptr_t), V, L,
                                             _lib.ptr(graph.col_t), slope, _lib.ptr(alpha), st), "relgnn_rgat_alpha")
            _lib.check(lib.relgnn_headw_reduce(_lib.ptr(T), V * L, D, D, num_heads, _lib.ptr(graph.rowptr_t), V, L,
                                               _lib.ptr(graph.col_t), _lib.ptr(alpha), None, _lib.ptr(out), D, None, None, st),
                       "relgnn_headw_reduce")
        else:
            _lib.check(lib.relgnn_rgat_fwd(_lib.ptr(T), D, D, num_heads, _lib.ptr(s_src), _lib.ptr(s_tgt),
                                           _lib.ptr(graph.rowptr_t), V, L, _lib.ptr(graph.col_t), slope, _lib.ptr(out), D,
                                           _lib.ptr(alpha), st), "relgnn_rgat_fwd")
        ctx.graph, ctx.K, ctx.slope, ctx.fast = graph, num_heads, slope, fast
        ctx.save_for_backward(T, s_src, s_tgt, alpha, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load_library()
        st = _lib.current_stream()
        graph, K, slope = ctx.graph, ctx.K, ctx.slope
        T, s_src, s_tgt, alpha, out = ctx.saved_tensors
        V, L, M = graph.V, graph.L, graph.M
        D = T.shape[1]
        gout = gout.contiguous()
        dz = torch.empty((M, K), dtype=torch.float32, device=T.device)
        # the two score-table gradients are sums of dz over the (target, type) and the (source, type) buckets: the dz
        # pass keeps the first in registers and the by-source gather of the gT pass carries the second along
        # (RELGNN_RGAT_FUSED_SUMS=0: two separate gather-reduces over dz, 70 + 60 us per layer at the C2 shape)
        fuse = _cfg.rgat_fused_sums != "0"
        if ctx.fast and _rgat_dz_fast_ok(D, K):
            fuse_t = fuse and L * K <= 64
            gs_tgt = torch.empty((V * L, K), dtype=torch.float32, device=T.device) if fuse_t else None
            _lib.check(lib.relgnn_rgat_dz(_lib.ptr(T), V * L, D, D, K, _lib.ptr(s_src), _lib.ptr(s_tgt),
                                          _lib.ptr(graph.rowptr_t), V, L, _lib.ptr(graph.col_t), slope, _lib.ptr(alpha),
                                          _lib.ptr(out), _lib.ptr(gout), D, _lib.ptr(dz), _lib.ptr(gs_tgt), st), "relgnn_rgat_dz")
            if not fuse_t:
                gs_tgt = _seg_reduce_raw(_lib.AGG_SUM, dz, graph.rowptr_t, 1, graph.iota, None, V * L)
        else:
            gs_tgt = torch.empty((V * L, K), dtype=torch.float32, device=T.device)
            _lib.check(lib.relgnn_rgat_bwd_logits(_lib.ptr(T), D, D, K, _lib.ptr(s_src), _lib.ptr(s_tgt),
                                                  _lib.ptr(graph.rowptr_t), V, L, _lib.ptr(graph.col_t), slope,
                                                  _lib.ptr(alpha), _lib.ptr(out), _lib.ptr(gout), D, _lib.ptr(dz),
                                                  _lib.ptr(gs_tgt), st), "relgnn_rgat_bwd_logits")
        if ctx.fast:
            gT = torch.empty_like(T)
            gs_src = torch.empty((V * L, K), dtype=torch.float32, device=T.device) if fuse else None
            _lib.check(lib.relgnn_headw_reduce(_lib.ptr(gout), V, D, D, K, _lib.ptr(graph.rowptr_s), V * L, 1,
                                               _lib.ptr(graph.tgt_s), _lib.ptr(alpha), _lib.ptr(graph.pos_t_of_s),
                                               _lib.ptr(gT), D, _lib.ptr(dz) if fuse else None, _lib.ptr(gs_src), st),
                       "relgnn_headw_reduce")
            if not fuse:
                gs_src = _seg_reduce_raw(_lib.AGG_SUM, dz, graph.rowptr_s, 1, graph.pos_t_of_s, None, V * L)
        else:
            gT = torch.empty_like(T)
            gs_src = torch.empty((V * L, K), dtype=torch.float32, device=T.device)
            _lib.check(lib.relgnn_rgat_bwd_msg(D, K, _lib.ptr(graph.rowptr_s), V * L, _lib.ptr(graph.tgt_s),
                                               _lib.ptr(graph.pos_t_of_s), _lib.ptr(alpha), _lib.ptr(dz), _lib.ptr(gout), D,
                                               _lib.ptr(gT), D, _lib.ptr(gs_src), st), "relgnn_rgat_bwd_msg")
        return gT, gs_src, gs_tgt, None, None, None


def rgat_scores_supported(D: int, K: int) -> bool:
    g = D // 4
    return D % 4 == 0 and g in (8, 16, 32, 64) and K > 0 and g % K == 0 and ((g // K) & (g // K - 1)) == 0


class _RgatLayerAttention(torch.autograd.Function):
    """rgat.py:103-136 from the transformed rows T [V*L, D] and the stacked attention parameters att [L, 2D]:
    score tables (csrc/rgat_scores.hip) -> segmented softmax -> attention-weighted sum, one autograd node; the
    backward folds the score-table gradients into gT in place and reduces d att without atomics."""

    @staticmethod
    def forward(ctx, T, att, graph, num_heads: int, slope: float):
        lib = _lib.load_library()
        st = _lib.current_stream()
        T, att = T.contiguous(), att.contiguous()
        V, L = graph.V, graph.L
        D = T.shape[1]
        s_src = torch.empty((V * L, num_heads), dtype=torch.float32, device=T.device)
        s_tgt = torch.empty_like(s_src)
        _lib.check(lib.relgnn_rgat_scores_fwd(_lib.ptr(T), D, D, num_heads, _lib.ptr(att), L, V, _lib.ptr(s_src),
                                              _lib.ptr(s_tgt), st), "relgnn_rgat_scores_fwd")
        out = _RgatAttention.forward(ctx, T, s_src, s_tgt, graph, num_heads, slope)   # saves T, s_*, alpha, out
        ctx.att = att
        return out

    @staticmethod
    def backward(ctx, gout):
        from .dense import column_sum
        lib = _lib.load_library()
        T = ctx.saved_tensors[0]
        att, graph, K = ctx.att, ctx.graph, ctx.K
        V, L = graph.V, graph.L
        D = T.shape[1]
        gT, gs_src, gs_tgt = _RgatAttention.backward(ctx, gout)[:3]
        groups = int(lib.relgnn_rgat_scores_groups(V))
        partial = torch.empty((groups, L * 2 * D), dtype=torch.float32, device=T.device)
        _lib.check(lib.relgnn_rgat_scores_bwd(_lib.ptr(T), D, D, K, _lib.ptr(att), L, V, _lib.ptr(gs_src), _lib.ptr(gs_tgt),
                                              _lib.ptr(gT), D, _lib.ptr(partial), groups, _lib.current_stream()),
                   "relgnn_rgat_scores_bwd")
        gatt = column_sum(partial).view(L, 2 * D)
        return gT, gatt, None, None, None


def rgat_layer_attention(T, att, graph, num_heads: int, slope: float = 0.2):
    """T [V*L, D] (row v*L+l = h_v W_l), att [L, 2D] = the stacked Edge_%i_Attention_Parameters."""
    return _RgatLayerAttention.apply(T, att, graph, int(num_heads), float(slope))


def rgat_attention(T, s_src, s_tgt, graph, num_heads: int, slope: float = 0.2):
    """gnns/rgat.py:98-136: segmented softmax over all incoming messages + attention-weighted sum."""
    return _RgatAttention.apply(T, s_src, s_tgt, graph, int(num_heads), float(slope))


# ---- aggregate, then transform (two kernels, no fusion): gather from the SMALL table -------------------------------------
def aggregate_acc64() -> bool:
    """RELGNN_AGG_ACC=f32|f64: accumulator width of the bucket sums that feed the K = L*D GEMM of the aggregate-first order.
    Measured at the full C2 batch (profiles/r03_parity_margin.json): float64 accumulators change the distance to the oracle from
    6.2e-6 to 6.0e-6 and the distance to the float64 truth from 4.8e-6 to 5.1e-6 — the bucket sums are NOT where the
    aggregate-first order spends its error budget (the K = 768 dot products are), so the default stays the float32 kernel."""
    return _cfg.agg_acc == "f64"


# config.settings.bwd_overlap (RELGNN_BWD_OVERLAP; auto = on with the limb route): the weight gradient of the aggregate-first RGCN layer on a side stream next to the input gradient's
# gather.  Measured on the C2 step, alternated twice in one process group: 2.006 / 2.014 ms without, 1.955 / 1.955 ms with (round 2
# measured the opposite, 3.08 vs 2.94 ms, with the library's split-K GEMM in that place: it wanted the same CUs and the same L2 as the
# gather; the limb kernel is one 147 KB-LDS workgroup per CU that leaves registers and the L2 path to the gather's waves).
# With the exact-fp32 routes (RELGNN_GEMM=lib / panel) the default is off: 2.45 vs 2.22 ms.
_SIDE_STREAMS = {}


# The join of the side stream (main stream waits for the weight gradient) normally sits at the end of the layer's backward: whatever
# reads the gradient next finds it complete.  A training step that owns the whole backward can do better: nothing reads a weight
# gradient before the optimizer, and next to the gather the side stream's workgroups starve (the gather's 27 k four-wave workgroups
# hold every wave slot and register), so its kernels really start at the gather's tail and finish AFTER the input-gradient product
# — the main stream then idled at every layer's join.  deferred_weight_gradient_join() (models/sparse_graph_model.py: train_step)
# moves the joins to join_deferred(), called once behind the backward.
_DEFER = {"on": False, "pending": [], "targets": set(), "handed": []}


class deferred_weight_gradient_join:
    def __enter__(self):
        self._old = _DEFER["on"]
        _DEFER["on"] = True
        return self

    def __exit__(self, exc_type, exc, tb):
        _DEFER["on"] = self._old
        if exc_type is not None:
            # the backward raised (out of memory, a check inside a Function): nobody will call join_deferred() for this pass.  Wait
            # for the side streams and forget the pass — stale `handed` entries would hold the parameters alive and make the NEXT
            # step's join verify gradients that belong to this one
            pending, _DEFER["pending"] = _DEFER["pending"], []
            _DEFER["handed"] = []
            _DEFER["targets"].clear()
            for device, side in pending:
                try:
                    torch.cuda.current_stream(device).wait_stream(side)
                except Exception:
                    pass
        return False


def _accumulator_keeps_the_tensor(p) -> bool:
    """Will autograd's AccumulateGrad take the gradient tensor of leaf `p` as it is, launching nothing on the main stream?  It
    copies (reads the tensor at once) when a tensor hook sits on the parameter (the gradient passes through Python and gains a
    reference), when the layouts differ, and under create_graph (grad mode on inside the backward); anomaly mode inspects every
    gradient; a post-accumulate hook reads p.grad right behind the accumulation."""
    return (p.is_leaf and p.requires_grad and p.grad is None and p.is_contiguous()
            and not getattr(p, "_backward_hooks", None) and not getattr(p, "_post_accumulate_grad_hooks", None))


def deferred_targets_ok(params, device) -> bool:
    """May a weight gradient be left in flight on the side stream until join_deferred()?  Only if nothing on the main stream reads
    it before: every target must be a LEAF that has no gradient yet and has not been a target in this backward pass — autograd's
    accumulator then keeps the tensor itself and launches nothing (_accumulator_keeps_the_tensor lists what else makes it copy;
    join_deferred() verifies afterwards that it did keep it).  A parameter used twice in the graph (the timesteps of a GGNN
    layer share their weights) has its contributions SUMMED on the main stream (in the engine's input buffer, or `grad += new`),
    which would read tensors that are still being written: on the second sight of a parameter the main stream is made to wait for
    the side stream here and the caller computes on one stream."""
    seen = _DEFER["targets"]
    if (params is not None and not torch.is_grad_enabled() and not torch.is_anomaly_enabled()
            and all(_accumulator_keeps_the_tensor(p) and id(p) not in seen for p in params)):
        seen.update(id(p) for p in params)
        return True
    wait_if_in_flight(params, device)
    return False


def wait_if_in_flight(params, device) -> None:
    """A contribution to `params` is about to be produced on the main stream.  If an earlier one of this backward went aside,
    autograd will sum the two on the main stream: it waits for the side stream first.  The parameters are marked as seen either
    way — a LATER contribution must not go aside either (the engine would add it, still in flight, to the one buffered here).
    Called on every sight of a parameter that does not go aside itself, whatever else the caller computes.  (A view's or a
    non-leaf's gradient never goes aside and is consumed by the view's backward at once: params is None for those.)"""
    if params is None:
        return
    seen = _DEFER["targets"]
    if any(id(p) in seen for p in params):
        side = _SIDE_STREAMS.get(device)
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)
        # their first contributions are complete as far as the main stream is concerned from here on: summing into them is safe and
        # join_deferred() has nothing left to verify for these parameters
        mine = {id(p) for p in params}
        _DEFER["handed"] = [h for h in _DEFER["handed"] if id(h[0]) not in mine]
    if _DEFER["on"]:
        seen.update(id(p) for p in params)


def hand_over_deferred(device, side, params, grads) -> None:
    """Record that `grads` (in flight on `side`) are being returned to autograd as the gradients of the leaves `params`."""
    _DEFER["pending"].append((device, side))
    for p, g in zip(params, grads):
        if g is not None:
            _DEFER["handed"].append((p, g.data_ptr(), g._version))


def join_deferred() -> None:
    """Make the current stream wait for every side stream whose join was deferred (no host synchronisation), then check that
    autograd did what the deferral relies on: every parameter's .grad IS the tensor that was handed over (same storage, never
    written in place since).  Anything else means the main stream read or wrote a gradient that was still being produced — raised
    here rather than left as a silently wrong update."""
    pending, _DEFER["pending"] = _DEFER["pending"], []
    handed, _DEFER["handed"] = _DEFER["handed"], []
    _DEFER["targets"].clear()
    done = set()
    for device, side in pending:
        if id(side) not in done:
            torch.cuda.current_stream(device).wait_stream(side)
            done.add(id(side))
    for p, ptr, version in handed:
        g = p.grad
        if g is None or g.data_ptr() != ptr or g._version != version:
            raise RuntimeError(
                "deferred weight-gradient join: the gradient of a %s parameter was %s on the main stream while its producer was "
                "still in flight on the side stream (a second use of the parameter outside this package's layers, a hook, or a "
                "copying accumulator); run the backward without ops.deferred_weight_gradient_join() or set bwd_overlap=0"
                % (tuple(p.shape), "dropped" if g is None else "copied" if g.data_ptr() != ptr else "accumulated into in place"))


def _side_stream(device):
    st = _SIDE_STREAMS.get(device)
    if st is None:
        # (a high-priority queue changes nothing here: measured 1.810 / 1.813 ms on C2, 32.5 / 33.0 ms on C5 — the side stream's
        #  large workgroups still become resident only where the main stream's small ones leave room)
        st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


def _weight_gradient(agg, gsc, amax, L: int):
    """agg^T @ gsc (agg [V, L*Din]: the bucket sums, gsc [V, Dout]).  On the two-fp16-limb route (RELGNN_LIMB=pair: `amax`, the
    forward gather's per-bucket magnitudes [V*L], says the layer took it) the operands go behind exact power-of-two scales that
    factor out of the product: gsc one per COLUMN (gradient columns differ by decades and Adam normalises every weight by its own
    history: relgnn_col_absmax_f32, one streaming pass over [V, Dout]), agg one per EDGE TYPE — the largest of that type's bucket
    magnitudes, a reduction over the [V, L] table the gather already wrote.  (One magnitude per column of agg as well would cost a
    pass over the [V, L*Din] sums — 110 MB at C2, measured +50 us per layer even on a side stream, more than the two-limb
    arithmetic gains; within an edge type an element keeps all 22 bits down to 4e-6 of the type's largest bucket entry, and the
    columns of one type's bucket sums are sums of the same post-activation states.  The kernel and the C ABI take any grouping:
    tests/test_gpu_limb_gemm.py measures per-column scales on both sides.)"""
    from . import dense as DN
    if (amax is not None and _cfg.limb_pair and _cfg.pair_part("tn") and DN.limb_tn_supported(agg, gsc)
            and agg.shape[1] * gsc.shape[1] > 256 * 256):
        return DN.limb_gemm_tn(agg, gsc, DN.col_absmax(amax.view(-1, L)), DN.col_absmax(gsc))
    return DN.matmul_tn_splitk(agg, gsc)


def _pair_products(X, rowptr, stride, V, k, n, kernels, kind) -> bool:
    """The two-fp16-limb form for this gather + product pair (dense.RELGNN_LIMB=pair): the gather can write the per-bucket
    magnitudes and the product takes the limb route."""
    from . import dense as DN
    if not (_cfg.limb_pair and X.is_cuda and V >= DN._LIMB_MIN_ROWS and n % 256 == 0 and k % 16 == 0
            and 16 <= k <= DN._LIMB_MAX_K):
        return False
    if not _cfg.pair_part(kind):                         # (diagnostics: which products take the form)
        return False
    return (rowmax_supported(X, rowptr, stride, aggregate_acc64())
            and DN.weight_image_ok(list(kernels), DN.WEIGHT_NN if kind == "nn" else DN.WEIGHT_NT))


_HANDOVER_WORDS = {}


def handover_word(device=None) -> torch.Tensor:
    """The hand-over status block of `device` (int32[2] in HBM, include/relgnn.h RELGNN_HANDOVER_*): the kernels whose wave roles hand
    data over through LDS counters (relgnn_limb_gemm_xf32_pc: the default forward products; relgnn_rgcn_fused_fwd) OR a bit into
    word 0 when one of their bounded polls runs out — such a launch has written wrong numbers.  The library neither allocates nor
    remembers anything: the block is the caller's, like `err_flag`.  One per device, allocated by the first caller — a model does it
    when it is built, so that a step captured into a hipGraph carries a live pointer (an allocation cannot be captured)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    w = _HANDOVER_WORDS.get(dev.index)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the hand-over status block of %s does not exist yet and cannot be allocated inside a stream "
                               "capture: call ops.handover_word(device) (or build the model) before capturing" % dev)
        w = torch.zeros(2, dtype=torch.int32, device=dev)
        _HANDOVER_WORDS[dev.index] = w
    return w


def handover_status(reset: bool = True, device=None) -> int:
    """Word 0 of handover_word(device): 0 = every poll of every launch so far completed.  Synchronises: a training / evaluation
    loop does not call this — it reads the word with every step's metrics copy (models.sparse_graph_model.MetricsReadback) and
    raises there; this is for code that launches the kernels directly (tests, scripts, smoke())."""
    w = handover_word(device)
    v = int(w[0].item())
    if reset and v:
        w[0].zero_()
    return v


class HandoverError(RuntimeError):
    pass


def raise_on_handover(value: int):
    if value:
        raise HandoverError(
            "a wave-role kernel gave up on an LDS hand-over (status 0x%x:%s%s%s%s): the results of that launch are wrong — every "
            "step since the last clean check is suspect" % (
                value, " fused-layer matrix wave" if value & 1 else "", " fused-layer gather wave" if value & 2 else "",
                " product matrix wave" if value & 4 else "", " product producer wave" if value & 8 else ""))


def _fused_layer_ok(H, graph, w, kernels) -> bool:
    """relgnn_rgcn_fused_fwd applies: switch on, exact-split limb route, 256 -> 256, no hub plan on the by-target buckets."""
    from .dense import WEIGHT_NN, _rows_ok, weight_image_ok
    if _cfg.rgcn_fused != "1" or not _cfg.limb_gemm or aggregate_acc64():
        return False
    if not (H.is_cuda and _rows_ok(H) and kernels[0].shape == (256, 256) and H.shape[1] == 256 and weight_image_ok(kernels, WEIGHT_NN)):
        return False
    split = getattr(graph.rowptr_t, "_relgnn_split", None)
    if (split is not None and 1 in split) or graph.V <= 0 or len(kernels) > 64:
        return False
    return w is None or (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous())


def _rgcn_fused(H, graph, w, kernels, relu: bool, want_sums: bool):
    from .dense import WEIGHT_NN, weight_limbs
    lib = _lib.load_library()
    V, L = graph.V, len(kernels)
    out = torch.empty((V, 256), dtype=torch.float32, device=H.device)
    agg = torch.empty((V, L * 256), dtype=torch.float32, device=H.device) if want_sums else None
    buf = weight_limbs(list(kernels), WEIGHT_NN)
    _lib.check(lib.relgnn_rgcn_fused_fwd(_lib.ptr(H, rows_strided=True), H.shape[0], H.stride(0), _lib.ptr(graph.rowptr_t), V, L,
                                         _lib.ptr(graph.src_t), _lib.ptr(w), buf.data_ptr(), None,
                                         _lib.ACT_RELU if relu else _lib.ACT_LINEAR, _lib.ptr(agg), L * 256, _lib.ptr(out), 256,
                                         256, 256, handover_word(H.device).data_ptr(), _lib.current_stream()), "relgnn_rgcn_fused_fwd")
    return agg, out


class _AggregateThenTransform(torch.autograd.Function):
    """out = act(f_mode(sum_l A_l @ W_l)),  A_l[v] = sum_{p in (v,l)} w_p H[src_p]   (W: [L, Din, Dout]).

    Same function as transform-then-aggregate (gnns/rgcn.py:84-114: sum / mean / sqrt_n commute with the per-type linear
    map), different memory behaviour: the gather reads the [V, Din] state table (C2: 33 MB, one graph's slab fits the 4 MiB
    L2 of its XCD) instead of the L-times larger table of transformed states, and the GEMM gets K = L*Din.  Measured on
    MI355X, C2, one layer forward: 86.6 + 107.5 us vs 120.8 + 107.4 us (scripts/exp_agg_first.py).
    Backward: dH through the by-source buckets exactly as before (gather dOut rows, GEMM with the stacked W_l^T),
    dW_l = A_l^T @ dOut from the saved aggregated rows."""

    @staticmethod
    def forward(ctx, H, graph, w, mode: int, act: int, act_name, h_act: int, *kernels):
        from .dense import grouped_nn_gemm
        H = H.contiguous()
        L = len(kernels)                                # kernels[l]: [Din, Dout], the per-edge-type variables themselves
        d_in, d_out = kernels[0].shape
        V = graph.V
        # RELGNN_LIMB=pair: the gather also writes every bucket's largest magnitude — the row scales of the two-fp16-limb product
        amax = None
        if _pair_products(H, graph.rowptr_t, 1, V, L * d_in, d_out, kernels, "nn"):
            amax = torch.empty(V * L, dtype=torch.float32, device=H.device)
        want_w = any(ctx.needs_input_grad[7:])
        f = _mode_factor(graph, mode)
        fused_relu = act == _lib.ACT_RELU and f is None       # sum aggregation: ReLU rides in the product's epilogue
        if amax is None and _fused_layer_ok(H, graph, w, kernels):
            # one kernel: the gather waves hand the bucket sums to the matrix waves through LDS (csrc/rgcn_fused.hip); the sums are
            # also stored as fp32 when the weight gradient will read them.  Same bits as the two launches below.
            agg, out = _rgcn_fused(H, graph, w, kernels, fused_relu, want_w)
        else:
            agg = _seg_reduce_raw(_lib.AGG_SUM, H, graph.rowptr_t, 1, graph.src_t, w, V * L,
                                  acc64=aggregate_acc64(), rowmax=amax).view(V, L * d_in)
            out = grouped_nn_gemm(agg, kernels, relu=fused_relu, xmax=amax, xgroups=L)
        if f is not None:
            out.mul_(f.unsqueeze(1))
        if act == _lib.ACT_RELU:
            if not fused_relu:
                out.relu_()
        elif act == _lib.ACT_TANH:
            out.tanh_()
        elif act != _lib.ACT_LINEAR:
            from .utils import apply_activation, get_activation
            out = apply_activation(get_activation(act_name), out)
        ctx.graph, ctx.w, ctx.mode, ctx.act, ctx.L = graph, w, mode, act, L
        # h_act: H is itself the output of that activation and its gradient factor may ride in the input-gradient product's
        # epilogue (dense.fusable_activation_of; needs d_in == the product's N, i.e. H's own rows as the epilogue operand)
        ctx.h_act = h_act if ctx.needs_input_grad[0] else 0
        ctx.save_for_backward(agg if want_w else None, out if act != _lib.ACT_LINEAR else None,
                              amax if want_w else None, H if ctx.h_act else None, *kernels)
        ctx.leaf_params = tuple(kernels) if all(k.is_leaf for k in kernels) else None
        return out

    @staticmethod
    def backward(ctx, gout):
        from .dense import act_bwd_from_output, grouped_nt_gemm, is_premasked, mark_premasked, matmul_tn_splitk
        graph, w, mode, act, L = ctx.graph, ctx.w, ctx.mode, ctx.act, ctx.L
        agg, out, amax, H_in, *kernels = ctx.saved_tensors
        d_in, d_out = kernels[0].shape
        V = graph.V
        # g * act'(out) — unless the consumer of `out` folded that factor into the product that made this gradient (the tag is on
        # the tensor object: look before anything copies it)
        premasked = act != _lib.ACT_LINEAR and is_premasked(gout, out, act)
        gout = gout.contiguous()
        if act != _lib.ACT_LINEAR and not premasked:
            gout = act_bwd_from_output(act, out, gout)
        gH = gW = None
        want_w = any(ctx.needs_input_grad[7:])
        # The weight gradient (matrix-pipe bound, one workgroup per CU, 147 KB of LDS, no L2 pressure) does not depend on the input
        # gradient's gather (L2-latency bound, no LDS, few registers): it runs on a side stream next to it (fork / join by events,
        # capturable in a hipGraph; the result is the same bits, the kernels are the same).
        side = None
        if (want_w and ctx.needs_input_grad[0] and _cfg.bwd_overlap_on and gout.is_cuda
                and (not _DEFER["on"] or deferred_targets_ok(ctx.leaf_params, gout.device))):
            side = _side_stream(gout.device)
            cur = torch.cuda.current_stream(gout.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                f = _mode_factor(graph, mode)
                gsc = gout if f is None else gout * f.unsqueeze(1)
                gW = _weight_gradient(agg, gsc, amax, L)
            for t in (agg, gout, gsc) + ((amax,) if amax is not None else ()):
                t.record_stream(side)
        if ctx.needs_input_grad[0]:
            plan = graph.plan_transformed(w)            # by-source buckets; weights carry the mean / sqrt_n factor
            gmax = None
            if (plan.num_rows_x == V * L and
                    _pair_products(gout, plan.rowptr_b, plan.stride_b, V, L * d_out, d_in, kernels, "nt")):
                gmax = torch.empty(V * L, dtype=torch.float32, device=gout.device)
            gT = _seg_reduce_raw(_lib.AGG_SUM, gout, plan.rowptr_b, plan.stride_b, plan.col_b, plan.w_bwd(mode),
                                 plan.num_rows_x, acc64=aggregate_acc64(), rowmax=gmax).view(V, L * d_out)   # row u: [dT_0 | .. | dT_{L-1}]
            if ctx.h_act and H_in is not None:                               # dH = sum_l dT_l @ W_l^T (* act'(H): H's producer skips its pass)
                gH = mark_premasked(grouped_nt_gemm(gT, kernels, xmax=gmax, xgroups=L, premask=(ctx.h_act, H_in)), H_in, ctx.h_act)
            else:
                gH = grouped_nt_gemm(gT, kernels, xmax=gmax, xgroups=L)
        if side is not None:
            if not _DEFER["on"]:
                torch.cuda.current_stream(gout.device).wait_stream(side)
            gW.record_stream(torch.cuda.current_stream(gout.device))
        elif want_w:
            wait_if_in_flight(ctx.leaf_params, gout.device)     # (no-op unless an earlier use of these kernels went aside)
            f = _mode_factor(graph, mode)               # agg holds the raw sums: the factor multiplies dOut
            gsc = gout if f is None else gout * f.unsqueeze(1)
            gW = _weight_gradient(agg, gsc, amax, L)
        gWs = tuple(gW[l * d_in:(l + 1) * d_in] if ctx.needs_input_grad[7 + l] else None for l in range(L)) \
            if gW is not None else (None,) * L           # dW_l = A_l^T @ dOut: row block l of [L*Din, Dout]
        if side is not None and _DEFER["on"]:
            hand_over_deferred(gout.device, side, kernels, gWs)
        return (gH, None, None, None, None, None, None) + gWs


def aggregate_then_transform(H, W, graph, w, aggregation: str, activation: Optional[str], sole_reader: bool = False):
    """sole_reader: this call is the only reader of H (dense.py, "activation gradients folded into the product that feeds them")."""
    mode, act = aggregation_mode_id(aggregation), activation_id(activation)
    if mode == _lib.AGG_MAX or act not in _FUSABLE_ACTS:
        raise ValueError("aggregate_then_transform: max aggregation / %r do not apply" % activation)
    # W: the per-edge-type kernels [Din, Dout] (a sequence: the variables themselves — nothing is stacked) or one [L, Din, Dout] tensor
    kernels = W.unbind(0) if torch.is_tensor(W) else tuple(W)
    from .dense import fusable_activation_of, mark_activation_output
    h_act = fusable_activation_of(H, sole_reader) if (H.requires_grad and H.is_contiguous() and H.shape[1] == kernels[0].shape[0]) else 0
    out = _AggregateThenTransform.apply(H, graph, w, mode, act, activation, h_act, *kernels)
    return mark_activation_output(out, act)     # (ReLU: any consumer may fold its gradient; the others need a caller's word that it is the only one)


# ---- RGDCN dynamic kernels applied node-side (csrc/rgdcn.hip) --------------------------------------------------------
class _RgdcnApply(torch.autograd.Function):
    """out[v,c,:] = out_act(f_mode(sum_l A[v,l,c,:] @ weight_act(P[v,l,c]))); A [V*L, C*K] (bucket-aggregated source
    states), P pre-activation dynamic weights in either GEMM layout ([V, L, C, K*K] or [C, V, L, K*K])."""

    @staticmethod
    def forward(ctx, A, P, graph, C: int, K: int, mode: int, weight_act: int, out_act: int, channel_major: bool):
        lib = _lib.load_library()
        A, P = A.contiguous(), P.contiguous()
        V, L = graph.V, graph.L
        KK = K * K
        strides = (L * KK, KK, V * L * KK) if channel_major else (L * C * KK, C * KK, KK)
        out = torch.empty((V, C * K), dtype=torch.float32, device=A.device)
        _lib.check(lib.relgnn_rgdcn_apply_fwd(mode, weight_act, out_act, _lib.ptr(A), _lib.ptr(P), *strides, V, L, C, K,
                                              _lib.ptr(graph.rowptr_t), _lib.ptr(out), _lib.current_stream()),
                   "relgnn_rgdcn_apply_fwd")
        ctx.graph, ctx.geom, ctx.strides = graph, (C, K, mode, weight_act, out_act), strides
        ctx.save_for_backward(A, P, out if out_act != _lib.ACT_LINEAR else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load_library()
        st = _lib.current_stream()
        A, P, out = ctx.saved_tensors
        graph = ctx.graph
        C, K, mode, weight_act, out_act = ctx.geom
        V, L = graph.V, graph.L
        g = gout.contiguous()
        if out_act != _lib.ACT_LINEAR:
            g2 = torch.empty_like(g)
            _lib.check(lib.relgnn_act_bwd_from_output(out_act, _lib.ptr(out), _lib.ptr(g), g.numel(), _lib.ptr(g2), st),
                       "relgnn_act_bwd_from_output")
            g = g2
        f = _mode_factor(graph, mode)
        if f is not None:
            g = (g * f.unsqueeze(1)).contiguous()
        gA, gP = torch.empty_like(A), torch.empty_like(P)
        _lib.check(lib.relgnn_rgdcn_apply_bwd(weight_act, _lib.ptr(A), _lib.ptr(P), *ctx.strides, V, L, C, K, _lib.ptr(g),
                                              _lib.ptr(gA), _lib.ptr(gP), st), "relgnn_rgdcn_apply_bwd")
        return gA, gP, None, None, None, None, None, None, None


def rgdcn_apply_supported(K: int, out_act: int) -> bool:
    return 0 < K <= 64 and (K & (K - 1)) == 0 and out_act in _FUSABLE_ACTS


def rgdcn_apply(A, P, graph, num_channels: int, channel_dim: int, aggregation: str, weight_activation: Optional[str],
                output_activation: Optional[str], channel_major: bool):
    return _RgdcnApply.apply(A, P, graph, int(num_channels), int(channel_dim), aggregation_mode_id(aggregation),
                             activation_id(weight_activation), activation_id(output_activation), bool(channel_major))


# ---- materialised messages: scale + activation fused into the segment reduce ------------------------------------
class _MessageActReduce(torch.autograd.Function):
    """out[v] = f_mode( sum_{m -> v} act( w_m * msgs[m] ) ) for a message tensor [M, D] in the reference's type-major
    order (gnns/gnn_edge_mlp.py:104-116): no elementwise pass over [M, D] forward, one fused pass backward."""

    @staticmethod
    def forward(ctx, msgs, graph, w, mode: int, act: int):
        lib = _lib.load_library()
        msgs = msgs.contiguous()
        M, D = msgs.shape
        plan = graph.plan_messages()
        out = torch.empty((graph.V, D), dtype=torch.float32, device=msgs.device)
        _lib.check(lib.relgnn_seg_reduce_msgact_fwd(mode, act, _lib.ptr(msgs), M, D, D, _lib.ptr(plan.rowptr), graph.V,
                                                    plan.stride, _lib.ptr(plan.col), _lib.ptr(w), _lib.ptr(out), D,
                                                    _lib.current_stream()), "relgnn_seg_reduce_msgact_fwd")
        ctx.graph, ctx.w, ctx.mode, ctx.act = graph, w, mode, act
        ctx.save_for_backward(msgs)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load_library()
        graph, w, mode, act = ctx.graph, ctx.w, ctx.mode, ctx.act
        (msgs,) = ctx.saved_tensors
        M, D = msgs.shape
        f = _mode_factor(graph, mode)
        gagg = (gout * f.unsqueeze(1)).contiguous() if f is not None else gout.contiguous()
        plan = graph.plan_messages()
        w_orig = graph.w_original_order(w) if w is not None else None
        gX = torch.empty_like(msgs)
        _lib.check(lib.relgnn_msg_act_bwd(act, _lib.ptr(msgs), D, _lib.ptr(w_orig), _lib.ptr(plan.col_b), _lib.ptr(gagg), M,
                                          _lib.ptr(gX), _lib.current_stream()), "relgnn_msg_act_bwd")
        return gX, None, None, None, None


def message_act_reduce(msgs, graph, w, aggregation: str, activation: Optional[str]):
    """Sum-like aggregations only (max: apply the activation, then seg_gather_reduce over plan_messages()); graphs with hub
    buckets take that route too (its gather-reduce splits long buckets into chunked virtual rows)."""
    mode = aggregation_mode_id(aggregation)
    if mode == _lib.AGG_MAX or msgs.shape[1] % 4 != 0 or graph.has_long_buckets:
        from .utils import apply_activation, get_activation
        if w is not None:
            msgs = graph.w_original_order(w).unsqueeze(1) * msgs
        msgs = apply_activation(get_activation(activation), msgs)
        return seg_gather_reduce(msgs, graph.plan_messages(), aggregation, None)
    return _MessageActReduce.apply(msgs, graph, w, mode, activation_id(activation))
