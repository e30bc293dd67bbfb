"""Node-side dense layers (Keras `Dense` of the reference: y = x @ kernel (+ bias), kernel [in, out]).

Forward and input-gradient products are plain library GEMMs (hipBLASLt, called through relgnn_blaslt_gemm_f32 so that the
library's solution is cached per shape class instead of being looked up for every new node count: lib_gemm).  The
WEIGHT-GRADIENT GEMM  dW = x^T @ g  reduces over the node dimension (K = V ~ 3e4..1e6, output only in x out <= 256 x 768):
a single library call leaves most of the 256 CUs idle (12 output tiles) and measured 400 us for [256 x 32k] @ [32k x 768] on
MI355X.  Outputs up to 256 x 256 go through the streaming MFMA kernel (relgnn_gemm_tn_stream_f32: 59 us at 256 x 256);
the [768 x 256] ones split the node dimension into S chunks, run one strided-batched library GEMM and sum the S partial
products (split-K, ~130 us = 13.6 GFLOP at ~105 TFLOP/s fp32).  The bias gradient (column sums over V rows) is a two-stage
HIP reduction.
"""
import os

import torch

# The route of the node-side Dense products (forward / input gradient / weight gradient) is config.settings.gemm (RELGNN_GEMM):
#   limb  (default) tall operands (>= _LIMB_MIN_ROWS rows; N % 256 == 0, K % 16 == 0, K <= _LIMB_MAX_K; weight gradients with
#         J % 32 == 0, C % 256 == 0 and more than 256 x 256 outputs) through csrc/limb_gemm.hip: every fp32 value as three bf16
#         limbs (exact), six bf16 MFMA products per fp32 product, fp32 accumulation.  Against float64 at [36 k, 768] x [768, 256]:
#         4.0e-6 (exact-fp32 library GEMM: 5.3e-6); the C2 layer against the oracle: 3.8e-6 abs (library: 6.2e-6,
#         profiles/r03_parity_margin*.json).  The error grows faster with K than an fmaf chain's (2.2x the fp32 product's at
#         K = 1040 .. 4096), hence the K limit.  Everything else falls through to `lib`.
#   lib   exact fp32 through relgnn_blaslt_gemm_f32 (hipBLASLt, solution cached per (layout, N, K, V / 4096)), small weight
#         gradients through relgnn_gemm_tn_stream_f32
#   panel forward / input-gradient products through the exact-fp32 row-panel MFMA kernel (csrc/panel_gemm.hip) wherever its shape
#         constraints hold (N % 64 == 0, K % 4 == 0); the weight gradients keep their `lib` routes
#   torch library GEMMs through torch.mm (a hipBLASLt solution lookup per call: ~70 us of host time for every node count not
#         seen before, i.e. for every batch of a shuffled epoch)
# config.settings.limb (RELGNN_LIMB) = pair (default): where the producer of the left operand supplies per-row magnitudes (the
# gather in front of the aggregate-first layer's products), the product is evaluated from TWO fp16 limbs per value behind exact
# power-of-two scales — three MFMA products instead of the six of the bf16 triple (csrc/limb_gemm.hip, NL = 2; its weight gradient:
# one scale per column of each operand).  Per product its operands carry 22 instead of 24 significant bits; measured against
# float64 on the C2 shapes neither arithmetic is systematically closer end to end (profiles/r04_*);
# `triple` keeps the exact split everywhere.
from .config import settings as _cfg



def _ops():
    from . import ops            # (ops imports this module lazily too)
    return ops


_LIMB_MIN_ROWS, _LIMB_MAX_K = 4096, 1024
GEMM_NN, GEMM_NT, GEMM_TN = 0, 1, 2
_WARNED_UNSUPPORTED = False


# ---- activation gradients folded into the product that feeds them ------------------------------------------------------------------
# y = act(z) is differentiated from its OUTPUT (relgnn_act_bwd_from_output: tanh, relu, leaky_relu, elu, selu) by the function that
# produced it: g_z = g_y * act'(y) — one pass over [V, D] per activation and step (three ReLU' and two tanh' passes per C2 step,
# 16-34 us each).  The function that CONSUMES y computes g_y as an input-gradient product and can apply act'(y) in that product's
# epilogue (relgnn_limb_gemm_xf32_dact: same bits, no pass).  Protocol, all on Python attributes of the tensors involved:
#   * a producer tags its output:            mark_activation_output(y, act, sole_consumer=...)
#   * a consumer that sees a tagged input x and whose input-gradient route can fuse returns g_x already multiplied by act'(x) and
#     tags it:                               mark_premasked(g_x, x, act)
#   * the producer's backward skips its own pass iff the gradient it receives IS that tagged tensor:  is_premasked(g, y, act)
# A gradient that autograd had to sum with other contributions, copy or pass through a hook arrives as another tensor object
# without the tag, and the producer multiplies as always.  That is exact for ReLU whatever happened in between (its factor is 0 or
# 1: applying it twice, or to a sum whose first term already carries it, changes nothing).  The other activations' factors are not
# idempotent: folding one into ONE of several contributions would leave the producer multiplying the sum again.  They are only
# folded when y provably has exactly one reader in the autograd graph, which takes two words: whoever hands y over vouches that
# only the function it is handed to will read it (the tag's flag: the driver loop of models/sparse_graph_model.py knows its own
# dataflow), and that function says that it reads it exactly once (sole_reader=True: the aggregate-first RGCN layer's first
# timestep, the driver's Dense between layers; a GGNN layer, which feeds its input to the messages AND to the cell, does not).
_FROM_OUTPUT_ACTS = (1, 2, 3, 4, 5)          # _lib.ACT_TANH .. ACT_SELU (GELU needs the pre-activation)
_IDEMPOTENT_ACTS = (2,)                      # _lib.ACT_RELU


def mark_activation_output(y: torch.Tensor, act: int, sole_consumer: bool = False) -> torch.Tensor:
    if act in _FROM_OUTPUT_ACTS and y.is_cuda and y.dtype == torch.float32 and y.dim() == 2:
        y._relgnn_act = (int(act), y._version, bool(sole_consumer))
    return y


def vouch_sole_consumer(y: torch.Tensor, sole: bool) -> torch.Tensor:
    """The caller knows how many functions will read y (a tagged activation output): set / clear the tag's sole-consumer word."""
    tag = getattr(y, "_relgnn_act", None)
    if tag is not None:
        y._relgnn_act = (tag[0], tag[1], bool(sole))
    return y


def fusable_activation_of(x: torch.Tensor, sole_reader: bool = False) -> int:
    """The activation whose gradient a consumer of x may apply in its input-gradient product (0 = none).  sole_reader: the caller
    reads x exactly once (needed, together with the hander's word in the tag, for every activation but ReLU)."""
    tag = getattr(x, "_relgnn_act", None)
    if tag is None or tag[1] != x._version or getattr(x, "_backward_hooks", None) or _cfg.act_fusion != "1":
        return 0
    act, _, only_this_callee = tag
    return act if (act in _IDEMPOTENT_ACTS or (only_this_callee and sole_reader)) else 0


def mark_premasked(g: torch.Tensor, y: torch.Tensor, act: int) -> torch.Tensor:
    g._relgnn_premasked = (y.data_ptr(), y._version, int(act), tuple(y.shape))
    return g


def is_premasked(g: torch.Tensor, y: torch.Tensor, act: int) -> bool:
    return getattr(g, "_relgnn_premasked", None) == (y.data_ptr(), y._version, int(act), tuple(y.shape))


def mark_zero_padded(g: torch.Tensor, ld: int) -> torch.Tensor:
    """g [M, K] is a view of rows of ld >= K floats whose columns K .. ld-1 hold ZEROS (written by g's producer: the loss gradient
    of tasks/ppi_task.py through relgnn_sigmoid_ce_bwd_padded).  A consumer whose product reduces over K may then read [M, ld]
    and meet a reduction length that is a multiple of 16 — the limb route — without a padding copy.  The tag is on the tensor
    object: anything autograd copies, sums or passes through a hook arrives untagged and takes the plain route."""
    g._relgnn_zero_pad = (g.data_ptr(), tuple(g.shape), g.stride(0), int(ld))
    return g


def zero_padded_operand(g: torch.Tensor):
    """The [M, ld] view behind a tensor tagged by mark_zero_padded (None: not tagged, or no longer the tensor that was tagged)."""
    tag = getattr(g, "_relgnn_zero_pad", None)
    if (tag is None or g.dim() != 2 or tag != (g.data_ptr(), tuple(g.shape), g.stride(0), tag[3]) or g.stride(1) != 1
            or g.stride(0) != tag[3] or tag[3] < g.shape[1] or tag[3] % 16):
        return None
    return torch.as_strided(g, (g.shape[0], tag[3]), (tag[3], 1))


def act_bwd_from_output(act: int, y: torch.Tensor, g: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """g * act'(y) with the derivative taken from the activation's output (relgnn_act_bwd_from_output); out may be g itself."""
    from . import _lib
    g = g if g.is_contiguous() else g.contiguous()
    if out is None:
        out = torch.empty_like(g)
    _lib.check(_lib.load_library().relgnn_act_bwd_from_output(act, _lib.ptr(y), _lib.ptr(g), g.numel(), _lib.ptr(out),
                                                              _lib.current_stream()), "relgnn_act_bwd_from_output")
    return out


def _premask_operand_ok(y: torch.Tensor, rows: int, cols: int) -> bool:
    return (y is not None and y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and tuple(y.shape) == (rows, cols)
            and y.stride(1) == 1 and y.stride(0) % 4 == 0 and y.stride(0) >= cols and y.data_ptr() % 16 == 0)


class _PerStream(dict):
    """Scratch keyed by (device, raw stream handle): products issued on different streams may run concurrently and must not
    share it.  Bounded: at most `limit` streams per cache are remembered, the least recently used entry goes first (a process
    that keeps creating streams — graph captures, user streams — would otherwise pin 64 MB per stream for its lifetime, and a
    recycled stream handle would find another stream's entry).  Dropping an entry only returns its memory to torch's caching
    allocator, which hands it out again in stream order of the stream it was allocated on; clear_caches() empties all of them."""

    def __init__(self, limit: int = 4):
        super().__init__()
        self.limit = limit

    def lookup(self, key):
        v = self.get(key)
        if v is not None:                      # move to the back: most recently used
            del self[key]
            self[key] = v
        return v

    def store(self, key, value):
        self.pop(key, None)
        while len(self) >= self.limit:
            del self[next(iter(self))]
        self[key] = value
        return value


_LIMB_WS = _PerStream()
_WORKSPACE = _PerStream()


def clear_caches() -> None:
    """Drop every per-stream scratch buffer and every cached limb image of a weight (dense.weight_image re-splits on the next
    request).  For callers that retire streams or devices; never needed for correctness."""
    _LIMB_WS.clear()
    _WORKSPACE.clear()
    _WEIGHT_LIMBS.clear()
    _FAST_IMAGES.clear()
    _SPLIT_ARGS.clear()
    _ZEROS.clear()


def _lib_rows_ok(t: torch.Tensor) -> bool:
    # row-dense: unit column stride AND rows that do not overlap (an expand()-backed gradient, e.g. from dense(x, W).sum(0),
    # arrives with strides (0, 1): every C entry point below would read it with ld = 0)
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] > 0
            and t.shape[1] > 0 and (t.stride(0) >= t.shape[1] or t.shape[0] == 1))


def _workspace(device):
    """hipBLASLt scratch (split-K / stream-K solutions write partial products there), one buffer per (device, stream):
    GEMMs issued on different streams may run concurrently and must not share it.  The library checks the size it is handed
    against the solution's need on every call (a cached solution that wants more fails and is re-queried, blaslt_gemm.hip)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACE.lookup(key)
    if ws is None:
        ws = _WORKSPACE.store(key, torch.empty(64 << 20, dtype=torch.uint8, device=device))
    return ws


def lib_gemm(layout: int, a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor = None, out: torch.Tensor = None,
             accumulate: bool = False, relu: bool = False, weight: bool = False, act: int = None, premask=None) -> torch.Tensor:
    """Plain library GEMM with a cached solution (relgnn_blaslt_gemm_f32): NN a @ b (+ bias) | NT a @ b^T | TN a^T @ b.
    Falls back to torch for operands the C entry point does not take (not fp32 / not row-dense / CPU).
    weight=True: b is a parameter (or a view of one) — the limb route keeps its limb image across the step (weight_limbs).
    act: an activation id for the epilogue (overrides relu; routes without that epilogue apply it in a pass of their own).
    premask = (act id, y): the result times act'(y), y [M, N] the OUTPUT of that activation (an input-gradient product meeting the
    activation gradient of the layer below): in the limb kernel's epilogue, or as relgnn_act_bwd_from_output behind any other route."""
    if act is None:
        act = 2 if relu else 0                                     # _lib.ACT_RELU / ACT_LINEAR
    relu = act == 2
    if layout == GEMM_NT and weight and out is None and not accumulate and a.is_cuda and _cfg.limb_gemm:
        # a gradient whose rows are zero-padded to a multiple of 16 columns by its producer (mark_zero_padded): the limb route over
        # the padded reduction length, with the activation gradient of the layer below in its epilogue — for the 121-label PPI
        # head that is one launch instead of a library product (K = 121) and a ReLU' pass over [V, 256]
        ap = zero_padded_operand(a)
        if ap is not None and _limb_padded_ok(ap, a.shape[1], b, bias, WEIGHT_NT) and (premask is None or _premask_operand_ok(premask[1], a.shape[0], b.shape[0])):
            return limb_gemm_weight(ap, b, WEIGHT_NT, bias, act, dact=premask[0] if premask is not None else 0,
                                    dy=premask[1] if premask is not None else None)
    if premask is not None or act not in (0, 2):
        return _gemm_with_epilogues(layout, a, b, bias, act, weight, premask)
    if _cfg.limb_gemm and layout != GEMM_TN and out is None and not accumulate:
        from . import _lib
        if _limb_route_ok(layout, a, b, bias):
            return limb_dense(layout, a, b, bias, _lib.ACT_RELU if relu else _lib.ACT_LINEAR, weight=weight)
        if _limb_route_ok(layout, a, b, bias, columns=128):        # the D = 128 models: 128 x 128 panels, two workgroups per CU
            return limb_dense_sel(layout, a, b, bias, _lib.ACT_RELU if relu else _lib.ACT_LINEAR,
                                  image=sel_image(b, layout) if weight else None)
        if layout == GEMM_NN and _limb_cut_route_ok(a, b, bias):   # N just short of a multiple of 128 (the 121 labels of the PPI head)
            return limb_dense_sel(layout, a, b, bias, _lib.ACT_RELU if relu else _lib.ACT_LINEAR)
    if ((_cfg.gemm == "panel") and layout != GEMM_TN and out is None and not accumulate and panel_gemm_supported(layout, a, b)
            and (bias is None or (bias.is_cuda and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        from . import _lib
        return panel_gemm(layout, a, b, bias, _lib.ACT_RELU if relu else _lib.ACT_LINEAR)
    if not ((_cfg.gemm != "torch") and _lib_rows_ok(a) and _lib_rows_ok(b) and (bias is None or (bias.is_cuda and bias.is_contiguous()
                                                                                          and bias.dtype == torch.float32))):
        if layout == GEMM_NN:
            res = torch.addmm(bias, a, b) if bias is not None else a @ b
        elif layout == GEMM_NT:
            res = a @ b.t()
        else:
            res = a.t() @ b
        if relu:
            res = res.relu_()
        if out is None:
            return res
        return out.add_(res) if accumulate else out.copy_(res)
    from . import _lib
    lib = _lib.load_library()
    if layout == GEMM_NN:
        M, K, N = a.shape[0], a.shape[1], b.shape[1]
    elif layout == GEMM_NT:
        M, K, N = a.shape[0], a.shape[1], b.shape[0]
    else:
        K, M, N = a.shape[0], a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws = _workspace(a.device)
    code = lib.relgnn_blaslt_gemm_f32(layout, _lib.ACT_RELU if relu else _lib.ACT_LINEAR,
                                      _lib.ptr(a, rows_strided=True), a.stride(0), _lib.ptr(b, rows_strided=True),
                                      b.stride(0), _lib.ptr(bias), _lib.ptr(out, rows_strided=True), out.stride(0), M, N, K,
                                      1, 0, 0, 0, 1 if accumulate else 0, _lib.ptr(ws), ws.numel(), _lib.current_stream())
    if code == _lib.EUNSUPPORTED:
        # the library has no solution for this problem through the direct interface (never seen on the shapes of the
        # path): same library through torch, said once — still the GPU, still fp32
        global _WARNED_UNSUPPORTED
        if not _WARNED_UNSUPPORTED:
            _WARNED_UNSUPPORTED = True
            import warnings
            warnings.warn("relgnn_blaslt_gemm_f32: no hipBLASLt solution for layout %d, M=%d N=%d K=%d; using torch.mm "
                          "for such shapes" % (layout, M, N, K))
        res = (a @ b) if layout == GEMM_NN else (a @ b.t()) if layout == GEMM_NT else (a.t() @ b)
        if bias is not None:
            res = res + bias
        if relu:
            res = res.relu_()
        return out.add_(res) if accumulate else out.copy_(res)
    _lib.check(code, "relgnn_blaslt_gemm_f32")
    return out


_TORCH_ACT_ = {1: torch.tanh_, 2: torch.relu_, 3: lambda t: torch.nn.functional.leaky_relu_(t, 0.2), 4: torch.nn.functional.elu_,
               5: torch.selu_}


def _gemm_with_epilogues(layout: int, a, b, bias, act: int, weight: bool, premask) -> torch.Tensor:
    """lib_gemm's products that carry an activation other than ReLU and / or an activation-gradient factor: both ride in the limb
    kernel's epilogue where that route applies (the tall products of the path); elsewhere they follow the plain product as passes."""
    from . import _lib
    M = a.shape[0]
    N = b.shape[1] if layout == GEMM_NN else b.shape[0]
    if premask is not None and not _premask_operand_ok(premask[1], M, N):
        raise ValueError("lib_gemm: premask operand must be a float32 device [%d, %d] matrix with 16-byte aligned rows" % (M, N))
    if (_cfg.limb_gemm and layout != GEMM_TN and weight and _limb_route_ok(layout, a, b, bias)
            and weight_image_ok(_weight_matrices(b), WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT)):
        return limb_gemm_weight(a, b, WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT, bias, act,
                                dact=premask[0] if premask is not None else 0, dy=premask[1] if premask is not None else None)
    if (premask is None and _cfg.limb_gemm and layout != GEMM_TN and act in _TORCH_ACT_
            and _limb_route_ok(layout, a, b, bias, columns=128)):
        # the D = 128 models' Dense layers (C3, C5: tanh between GNN layers): the 128-column panel kernels take any activation
        # of the path in their epilogue (act_rt) — round 6: no tanh pass behind the product
        return limb_dense_sel(layout, a, b, bias, act, image=sel_image(b, layout) if weight else None)
    res = lib_gemm(layout, a, b, bias, relu=(act == _lib.ACT_RELU), weight=weight)
    if act not in (_lib.ACT_LINEAR, _lib.ACT_RELU):
        fn = _TORCH_ACT_.get(act)
        if fn is None:
            raise ValueError("lib_gemm: no epilogue for activation id %d" % act)
        res = fn(res)
    if premask is not None:
        res = act_bwd_from_output(premask[0], premask[1], res, out=res)
    return res


def _rows_ok(t: torch.Tensor) -> bool:
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and (t.stride(0) >= t.shape[1] or t.shape[0] == 1)
            and t.data_ptr() % 16 == 0 and t.shape[0] < 2 ** 31 and t.shape[1] < 2 ** 31)


_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        from . import _lib
        z = _ZEROS[device] = torch.zeros(int(_lib.load_library().relgnn_panel_gemm_zeros_floats()), dtype=torch.float32,
                                         device=device)
    return z


def panel_gemm_supported(layout: int, a: torch.Tensor, b: torch.Tensor, n_out: int = None) -> bool:
    """Shapes relgnn_panel_gemm_f32 takes: fp32 device operands with 16-byte aligned dense rows, N % 64 == 0, K % 4 == 0
    (TN: M % 4 == 0 instead)."""
    if not (_rows_ok(a) and _rows_ok(b)):
        return False
    if layout == GEMM_NN:
        K, N = a.shape[1], b.shape[1]
    elif layout == GEMM_NT:
        K, N = a.shape[1], b.shape[0]
    else:
        K, N = a.shape[0], b.shape[1]
        if a.shape[1] % 4 != 0:
            return False
    if n_out is not None:
        N = n_out
    return N % 64 == 0 and (K % 4 == 0 or layout == GEMM_TN) and K > 0


def panel_gemm(layout: int, a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor = None, act: int = 0, *,
               a_rows: torch.Tensor = None, num_rows: int = None, b_select: torch.Tensor = None, rows_per_select: int = 0,
               batch: int = 1, strides=(0, 0, 0), split_k_rows: int = 0, dims=None, out: torch.Tensor = None) -> torch.Tensor:
    """relgnn_panel_gemm_f32 (csrc/panel_gemm.hip).  NN a @ b | NT a @ b^T | TN a^T @ b on the exact-fp32 matrix pipe.
      a_rows / num_rows : gathered left rows (NN / NT: output row r uses a[a_rows[r]], < 0 = zeros) or gathered reduction
                          rows of `a` (TN)
      b_select          : [num_rows / rows_per_select] int32, b is then [num_select, ...] and block p of rows_per_select output
                          rows multiplies b[b_select[p]]
      batch / strides   : independent products (element strides of a, b, out) or, with split_k_rows, K chunks -> out [batch, M, N]
      dims              : (M, N, K) when they do not follow from the operand shapes (batched / typed operands)"""
    from . import _lib
    lib = _lib.load_library()
    if dims is not None:
        M, N, K = dims
    elif layout == GEMM_NN:
        M, K, N = (num_rows if a_rows is not None else a.shape[0]), a.shape[1], b.shape[-1]
    elif layout == GEMM_NT:
        M, K, N = (num_rows if a_rows is not None else a.shape[0]), a.shape[1], b.shape[-2]
    else:
        K, M, N = (num_rows if a_rows is not None else a.shape[0]), a.shape[1], b.shape[-1]
    ldb = b.stride(-2)
    sel_stride = b.stride(0) if b_select is not None else 0
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=torch.float32, device=a.device)
    ldc = out.stride(-2)
    _lib.check(lib.relgnn_panel_gemm_f32(
        layout, act, a.data_ptr(), a.stride(-2), _lib.ptr(a_rows), b.data_ptr(), ldb, _lib.ptr(b_select), int(rows_per_select),
        sel_stride, _lib.ptr(bias), _lib.ptr(_zeros(a.device)), out.data_ptr(), ldc, M, N, K, batch, strides[0], strides[1],
        strides[2] if batch > 1 and strides[2] else (M * ldc if batch > 1 else 0), int(split_k_rows), _lib.current_stream()),
        "relgnn_panel_gemm_f32")
    return out


class Limbs:
    """An fp32 [rows, cols] matrix as three bf16 limbs per element in the tiled layout of csrc/limb_gemm.hip (include/relgnn.h:
    "limb tiles").  `data` is the flat bf16 buffer."""
    __slots__ = ("data", "rows", "cols")

    def __init__(self, data: torch.Tensor, rows: int, cols: int):
        self.data, self.rows, self.cols = data, int(rows), int(cols)

    def to_float64(self) -> torch.Tensor:
        """hi + mid + lo as float64 [rows, cols] (tests)."""
        RB, KT = (self.rows + 31) // 32, self.cols // 16
        t = self.data.view(RB, KT, 3, 2, 32, 8).double().sum(2)              # [RB, KT, h, i, 8]
        return t.permute(0, 3, 1, 2, 4).reshape(RB * 32, self.cols)[:self.rows]


def limb_split(x: torch.Tensor, transpose: bool = False, out: "Limbs" = None) -> "Limbs":
    """fp32 [R, C] -> the three bf16 limbs of x (or of x^T), x = hi + mid + lo exactly (relgnn_limb_split_f32)."""
    from . import _lib
    lib = _lib.load_library()
    if not _rows_ok(x):
        x = x.contiguous()
    R, C = x.shape
    rows, cols = (C, R) if transpose else (R, C)
    if out is None:
        out = Limbs(torch.empty(int(lib.relgnn_limb_elements(rows, cols)), dtype=torch.bfloat16, device=x.device), rows, cols)
    elif (out.rows, out.cols) != (rows, cols):
        raise ValueError("limb_split: out holds a [%d, %d] matrix, not [%d, %d]" % (out.rows, out.cols, rows, cols))
    _lib.check(lib.relgnn_limb_split_f32(x.data_ptr(), x.stride(0), R, C, 1 if transpose else 0, out.data.data_ptr(),
                                         _lib.current_stream()), "relgnn_limb_split_f32")
    return out


def limb_gemm(a: "Limbs", b: "Limbs", bias: torch.Tensor = None, act: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """act(bias + A @ B^T) in fp32 from the limbs of A [M, K] and B [N, K] (relgnn_limb_gemm_f32): six bf16 MFMA products per
    fp32 product, fp32 accumulation — fp32-class accuracy at up to 2.7x the fp32-input MFMA rate."""
    from . import _lib
    lib = _lib.load_library()
    if a.cols != b.cols:
        raise ValueError("limb_gemm: reduction lengths differ (%d, %d)" % (a.cols, b.cols))
    M, N, K = a.rows, b.rows, a.cols
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.data.device)
    _lib.check(lib.relgnn_limb_gemm_f32(act, a.data.data_ptr(), b.data.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.data.device)),
                                        out.data_ptr(), out.stride(0), M, N, K, _lib.current_stream()), "relgnn_limb_gemm_f32")
    return out


def limb_gemm_xf32(a: torch.Tensor, b: "Limbs", bias: torch.Tensor = None, act: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """act(bias + a @ B^T) with a fp32 [M, K] (dense rows) split inside the kernel and B [N, K] as limbs (relgnn_limb_gemm_xf32)."""
    from . import _lib
    lib = _lib.load_library()
    if not _rows_ok(a):
        a = a.contiguous()
    M, K = a.shape
    if K != b.cols:
        raise ValueError("limb_gemm_xf32: reduction lengths differ (%d, %d)" % (K, b.cols))
    if out is None:
        out = torch.empty((M, b.rows), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_limb_gemm_xf32(act, a.data_ptr(), a.stride(0), b.data.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.device)),
                                         out.data_ptr(), out.stride(0), M, b.rows, K, _lib.current_stream()), "relgnn_limb_gemm_xf32")
    return out


def _limb_padded_ok(ap: torch.Tensor, k: int, b: torch.Tensor, bias, kind: str) -> bool:
    """ap [M, ld] (zero_padded_operand) times a weight matrix b — [N, k] (WEIGHT_NT: ap @ b^T) or [k, N] (WEIGHT_NN: ap @ b) — with
    k <= ld = the next multiple of 16."""
    n, kb = (b.shape[0], b.shape[1]) if kind == WEIGHT_NT else (b.shape[1], b.shape[0])
    return (_rows_ok(ap) and ap.shape[0] >= _LIMB_MIN_ROWS and b.dim() == 2 and kb == k and ap.shape[1] == (k + 15) // 16 * 16
            and n % 256 == 0 and 16 <= ap.shape[1] <= _LIMB_MAX_K and weight_image_ok([b], kind)
            and (bias is None or (bias.is_cuda and bias.is_contiguous() and bias.dtype == torch.float32 and bias.data_ptr() % 16 == 0)))


def _limb_route_ok(layout: int, a: torch.Tensor, b: torch.Tensor, bias, columns: int = 256) -> bool:
    if not (_rows_ok(a) and _rows_ok(b)) or a.shape[0] < _LIMB_MIN_ROWS:
        return False
    K, N = (a.shape[1], b.shape[1]) if layout == GEMM_NN else (a.shape[1], b.shape[0])
    if (b.shape[0] if layout == GEMM_NN else b.shape[1]) != K:
        return False
    return (N % columns == 0 and K % 16 == 0 and 16 <= K <= _LIMB_MAX_K
            and (bias is None or (bias.is_cuda and bias.is_contiguous() and bias.dtype == torch.float32 and bias.data_ptr() % 16 == 0)))


# ---- the limbs of the step's WEIGHT operands: split once per optimizer step, all of them in one launch ----------------------------
# A weight is the right operand of two or three products per step (forward, input gradient) and changes once per step.  Its limb
# image is kept per (device, stream) until the weights change: the optimizer's fused update writes through raw pointers (tensor
# versions do not move) and says so through weights_changed(); every other in-place write moves the tensor's version, which is
# compared too.  The first request after a change re-splits every image that the previous step used (relgnn_limb_split_multi_f32);
# an image may be several matrices side by side along k (the per-edge-type kernels of a layer), so neither the stacked
# [L*Din, Dout] operand of the forward product nor the stacked W^T of the input gradient is ever formed in fp32.
# C2: 8 split launches + 3 stacks + 3 re-layouts per step -> 1 launch.  Under stream capture nothing is cached (a replay re-runs
# kernels, not this code): the image is split on every request.
_WEIGHT_LIMBS = _PerStream(limit=8)      # (device, stream) -> {operand key: _WeightImage}
_WEIGHT_GEN = [0]
WEIGHT_NN, WEIGHT_NT = "nn", "nt"


_CAPTURE_IMAGES = {"on": False, "images": {}}


class capture_image_cache:
    """Around the capture of ONE training step into a hipGraph: limb images split inside the capture are reused by later products of
    the same capture (forward -> backward) until weights_changed() — which the captured optimizer update calls — drops them."""

    def __enter__(self):
        _CAPTURE_IMAGES["on"], _CAPTURE_IMAGES["images"] = True, {}
        return self

    def __exit__(self, *exc):
        _CAPTURE_IMAGES["on"], _CAPTURE_IMAGES["images"] = False, {}
        return False


def weights_changed() -> None:
    """Tell the limb-image cache that parameters were rewritten in place by something torch's version counters do not see:
    a kernel that writes through raw pointers (models/sparse_graph_model.py: the fused clip + Adam launch; a hipGraph replay of
    it), `p.data.copy_()` / `p.data.mul_()`, a third-party optimizer that updates `.data`, a parameter broadcast.  Ordinary
    in-place tensor operations on the parameter itself (`p.add_()`, `p.copy_()` under no_grad) move its version and are noticed
    without this call.  PUBLIC CONTRACT of the default route (config gemm=limb, weight_limb_cache=1): whoever writes weights
    behind torch's back calls tf_gnn_samples_amd.dense.weights_changed() (cheap: a counter) — or runs with
    RELGNN_WEIGHT_LIMB_CACHE=0, which re-splits on every product."""
    _WEIGHT_GEN[0] += 1
    _CAPTURE_IMAGES["images"] = {}


class _WeightImage:
    # (no strong reference to the weights; pair: two fp16 limbs, `wmax` = the device float the image's scale comes from)
    __slots__ = ("refs", "items", "versions", "gen", "used_gen", "buf", "pair", "wmax")


def _weight_matrices(w):
    """A weight operand as a list of 2-D matrices laid side by side along k: a matrix, a [L, ., .] stack or a sequence."""
    if torch.is_tensor(w):
        return [w] if w.dim() == 2 else list(w.unbind(0))
    return list(w)


def _weight_image_shape(ws, kind: str):
    """(N, K) of B [N, K] = [w_0^T | w_1^T | ..] (WEIGHT_NN: w_l [K_l, N]) or [w_0 | w_1 | ..] (WEIGHT_NT: w_l [N, K_l])."""
    # (a single matrix whose k extent is not a multiple of 16 — the 121-label head — fills its last k-tile with zeros)
    k = sum(m.shape[0] if kind == WEIGHT_NN else m.shape[1] for m in ws)
    if len(ws) == 1:
        k = (k + 15) // 16 * 16
    return (ws[0].shape[1] if kind == WEIGHT_NN else ws[0].shape[0]), k


def weight_image_ok(ws, kind: str) -> bool:
    ws = _weight_matrices(ws)
    n = ws[0].shape[1] if kind == WEIGHT_NN else ws[0].shape[0]
    for m in ws:
        if not (m.is_cuda and m.dtype == torch.float32 and m.dim() == 2 and m.stride(1) == 1 and m.stride(0) >= m.shape[1]
                and (len(ws) == 1 or (m.stride(0) % 4 == 0 and m.data_ptr() % 16 == 0))):
            return False                   # (a single matrix may have rows of any alignment — [256, 121]: the split reads element-wise)
        if (m.shape[1] if kind == WEIGHT_NN else m.shape[0]) != n or ((m.shape[0] if kind == WEIGHT_NN else m.shape[1]) % 16 != 0
                                                                     and len(ws) > 1):
            return False
    return True


def _weight_image_items(ws, kind: str, buf: torch.Tensor):
    """(X, ldx, rows, cols, transpose, out, kt_offset, kt_total) per matrix of the image."""
    total = _weight_image_shape(ws, kind)[1] // 16
    items, kt = [], 0
    for m in ws:
        items.append((m.data_ptr(), m.stride(0), m.shape[0], m.shape[1], 1 if kind == WEIGHT_NN else 0, buf.data_ptr(), kt, total))
        kt += ((m.shape[0] if kind == WEIGHT_NN else m.shape[1]) + 15) // 16
    return items


_SPLIT_ARGS = {}


def _split_weight_images(images) -> None:
    import ctypes
    from . import _lib
    lib = _lib.load_library()
    triples = [im for im in images if not im.pair]
    pairs = [im for im in images if im.pair]
    if triples:
        # the argument arrays of a set of images are the same every step (an image's items never change): built once per set — a
        # 23-type, 10-layer model re-splits ~1400 matrices per step, and marshalling them anew cost milliseconds of host time
        key = tuple(map(id, triples))
        ent = _SPLIT_ARGS.get(key)
        if ent is None or len(ent[0]) != len(triples) or any(a is not b for a, b in zip(ent[0], triples)):
            items = [it for im in triples for it in im.items]
            n = len(items)
            cols = list(zip(*items))
            vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
            ent = (list(triples), n, (vp(*cols[0]), i64(*cols[1]), i32(*cols[2]), i32(*cols[3]), i32(*cols[4]), vp(*cols[5]),
                                      i32(*cols[6]), i32(*cols[7])))
            if len(_SPLIT_ARGS) > 64:
                _SPLIT_ARGS.clear()
            if len(triples) > 4:                       # (small sets are cheap to marshal and vary more)
                _SPLIT_ARGS[key] = ent
        _lib.check(lib.relgnn_limb_split_multi_f32(ent[1], *ent[2], _lib.current_stream()), "relgnn_limb_split_multi_f32")
    if pairs:           # two fp16 limbs: one magnitude per image first (its power-of-two scale), then the limbs — three launches
        wm = torch.empty(len(pairs), dtype=torch.float32, device=pairs[0].buf.device)
        items, image = [], []
        for i, im in enumerate(pairs):
            im.wmax = wm[i:i + 1]
            items += im.items
            image += [i] * len(im.items)
        n = len(items)
        cols = list(zip(*items))
        vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
        _lib.check(lib.relgnn_limb16_split_multi_f32(n, vp(*cols[0]), i64(*cols[1]), i32(*cols[2]), i32(*cols[3]), i32(*cols[4]),
                                                     vp(*cols[5]), i32(*cols[6]), i32(*cols[7]), i32(*image), len(pairs),
                                                     wm.data_ptr(), _lib.current_stream()), "relgnn_limb16_split_multi_f32")


def weight_limbs(w, kind: str) -> torch.Tensor:
    """The bf16-triple limb image (flat buffer) of a weight operand: weight_image(w, kind).buf."""
    return weight_image(w, kind).buf


def weight_image(w, kind: str, pair: bool = False, separate: bool = False) -> "_WeightImage":
    """The limb image of a weight operand as the right operand B [N, K] of relgnn_limb_gemm_xf32 (pair: of relgnn_limb16_gemm_xf32:
    two fp16 limbs, .wmax = the device float its scale comes from); .buf is the flat 16-bit buffer.  w: a matrix, a
    [L, ., .] stack or a sequence of matrices (laid side by side along k):
      WEIGHT_NN  w_l [K_l, N]:  [x_0 | x_1 | ..] @ [w_0; w_1; ..] = sum_l x_l @ w_l     (Dense forward; gnns/rgcn.py:96-98 summed over
                                                                                          the edge types in one product)
      WEIGHT_NT  w_l [N, K_l]:  [g_0 | g_1 | ..] @ [w_0 | w_1 | ..]^T = sum_l g_l @ w_l^T   (the input gradients of the same)
    separate: one image PER matrix, one behind the other in the buffer (matrix l at element l * relgnn_limb_elements(N, K); every
    matrix the same shape, N % 128 == 0) — the per-edge-type operands of relgnn_limb_gemm_sel_xf32 (round 6: the typed transforms
    and the D = 128 Dense layers no longer re-split their weights in front of every product).
    Valid until the next weights_changed() / in-place write to a matrix; on the current stream."""
    from . import _lib
    lib = _lib.load_library()
    ws = _weight_matrices(w)
    if separate:
        rows, cols = _weight_image_shape(ws[:1], kind)
        per = int(lib.relgnn_limb_elements(rows, cols))
        elements = per * len(ws)
    else:
        rows, cols = _weight_image_shape(ws, kind)
        elements = int(lib.relgnn_limb16_elements(rows, cols) if pair else lib.relgnn_limb_elements(rows, cols))
    dev = ws[0].device

    def make_items(buf):
        if not separate:
            return _weight_image_items(ws, kind, buf)
        return [(m.data_ptr(), m.stride(0), m.shape[0], m.shape[1], 1 if kind == WEIGHT_NN else 0, buf.data_ptr() + 2 * per * l, 0,
                 cols // 16) for l, m in enumerate(ws)]

    if torch.cuda.is_current_stream_capturing() or _cfg.weight_limb_cache != "1":
        # Under stream capture nothing outlives the capture — but WITHIN one captured training step the weights change once, at
        # its end (the optimizer's update calls weights_changed()): an image split for the forward serves the backward too
        # (capture_image_cache(): Sparse_Graph_Model.capture_train_step opens it around the capture).
        ckey = None
        if _CAPTURE_IMAGES["on"] and _cfg.weight_limb_cache == "1":
            ckey = (kind, pair, separate, torch.cuda.current_stream(dev).cuda_stream) + tuple(
                (m.data_ptr(), m.shape[0], m.shape[1], m.stride(0)) for m in ws)
            hit = _CAPTURE_IMAGES["images"].get(ckey)
            if hit is not None:
                return hit
        im = _WeightImage()
        im.pair, im.wmax = pair, None
        im.buf = torch.empty(elements, dtype=torch.bfloat16, device=dev)
        im.items = make_items(im.buf)
        _split_weight_images([im])
        if ckey is not None:
            im.refs = list(ws)                      # (keeps the addresses of the key alive for the duration of the capture)
            _CAPTURE_IMAGES["images"][ckey] = im
        return im
    import weakref
    skey = (dev, torch.cuda.current_stream(dev).cuda_stream)
    table = _WEIGHT_LIMBS.lookup(skey)
    if table is None:
        table = _WEIGHT_LIMBS.store(skey, {})
    key = (kind, pair, separate) + tuple((m.data_ptr(), m.shape[0], m.shape[1], m.stride(0)) for m in ws)
    gen = _WEIGHT_GEN[0]
    im = table.get(key)
    bases = [m._base if m._base is not None else m for m in ws]
    if im is not None and any(r() is not b for r, b in zip(im.refs, bases)):
        im = None                                        # another tensor lives at that address now
    if im is not None and im.gen == gen and im.versions == [m._version for m in ws]:
        im.used_gen = gen
        return im
    if im is None:
        im = table[key] = _WeightImage()
        im.refs, im.gen, im.versions, im.used_gen = [weakref.ref(b) for b in bases], -1, None, gen
        im.pair, im.wmax = pair, None
        im.buf = torch.empty(elements, dtype=torch.bfloat16, device=dev)
        im.items = make_items(im.buf)
    todo = [im]
    for k, other in list(table.items()):
        if other is im:
            continue
        alive = [r() for r in other.refs]
        if any(b is None for b in alive) or other.used_gen < gen - 1:       # gone, or not part of the last step: forget it
            del table[k]
        elif other.gen != gen or other.versions != [b._version for b in alive]:
            todo.append(other)
    _split_weight_images(todo)
    for t in todo:
        t.gen, t.versions = gen, [r()._version for r in t.refs]
    im.used_gen = gen
    return im


def limb_gemm_weight(a: torch.Tensor, w, kind: str, bias: torch.Tensor = None, act: int = 0,
                     out: torch.Tensor = None, xmax: torch.Tensor = None, xgroups: int = 0, dact: int = 0,
                     dy: torch.Tensor = None) -> torch.Tensor:
    """act(bias + a @ B^T) with B = weight_limbs(w, kind), a fp32 [M, K] split inside the kernel (relgnn_limb_gemm_xf32).
    xmax [M * xgroups] (per-row magnitudes of `a` from its producer, ops._seg_reduce_raw(rowmax=)): the two-fp16-limb form
    (relgnn_limb16_gemm_xf32)."""
    from . import _lib
    lib = _lib.load_library()
    n, k = _weight_image_shape(_weight_matrices(w), kind)
    if a.shape[1] != k:
        raise ValueError("limb_gemm_weight: a is [%d, %d], the weight operand has K = %d" % (a.shape[0], a.shape[1], k))
    if out is None:
        out = torch.empty((a.shape[0], n), dtype=torch.float32, device=a.device)
    if xmax is not None:
        if xmax.numel() != a.shape[0] * xgroups or xmax.dtype != torch.float32 or not xmax.is_contiguous():
            raise ValueError("limb_gemm_weight: xmax must be a contiguous float32 [%d * %d]" % (a.shape[0], xgroups))
        im = weight_image(w, kind, pair=True)
        if dy is not None:           # (dy [M, n]: the activation output whose gradient factor rides in the epilogue)
            _lib.check(lib.relgnn_limb16_gemm_xf32_dact(act, a.data_ptr(), a.stride(0), xmax.data_ptr(), int(xgroups), im.buf.data_ptr(),
                                                        im.wmax.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.device)), int(dact),
                                                        dy.data_ptr(), dy.stride(0), out.data_ptr(), out.stride(0), a.shape[0], n, k,
                                                        _lib.current_stream()), "relgnn_limb16_gemm_xf32_dact")
            return out
        _lib.check(lib.relgnn_limb16_gemm_xf32(act, a.data_ptr(), a.stride(0), xmax.data_ptr(), int(xgroups), im.buf.data_ptr(),
                                               im.wmax.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.device)), out.data_ptr(),
                                               out.stride(0), a.shape[0], n, k, _lib.current_stream()), "relgnn_limb16_gemm_xf32")
        return out
    buf = weight_limbs(w, kind)
    if _limb_pc_ok(a, n, k, bias, act, dy, out, kind):
        # wave roles instead of k-loop phases (csrc/limb_gemm_pc.hip): the same bits, the matrix waves at their MFMA-only time
        _lib.check(lib.relgnn_limb_gemm_xf32_pc(act, a.data_ptr(), a.stride(0), buf.data_ptr(), _lib.ptr(bias), int(dact),
                                                dy.data_ptr() if dy is not None else None, dy.stride(0) if dy is not None else 0,
                                                out.data_ptr(), out.stride(0), a.shape[0], n, k,
                                                _ops().handover_word(a.device).data_ptr(), _lib.current_stream()),
                   "relgnn_limb_gemm_xf32_pc")
        return out
    if dy is not None:
        _lib.check(lib.relgnn_limb_gemm_xf32_dact(act, a.data_ptr(), a.stride(0), buf.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.device)),
                                                  int(dact), dy.data_ptr(), dy.stride(0), out.data_ptr(), out.stride(0), a.shape[0], n, k,
                                                  _lib.current_stream()), "relgnn_limb_gemm_xf32_dact")
        return out
    _lib.check(lib.relgnn_limb_gemm_xf32(act, a.data_ptr(), a.stride(0), buf.data_ptr(), _lib.ptr(bias), _lib.ptr(_zeros(a.device)),
                                         out.data_ptr(), out.stride(0), a.shape[0], n, k, _lib.current_stream()),
               "relgnn_limb_gemm_xf32")
    return out


def _limb_pc_ok(a, n: int, k: int, bias, act: int, dy, out, kind: str) -> bool:
    """Shapes relgnn_limb_gemm_xf32_pc takes (config limb_pc): K % 128 == 0 (<= 1024), N % 256 == 0, N == 256 or K <= 256; ReLU / no
    activation.  limb_pc = fwd (the default): forward products only (WEIGHT_NN).  The kernel holds every CU for its whole run
    (one persistent 16-wave workgroup each); an input-gradient product runs next to the weight gradient on the side stream, whose
    workgroups then wait for CUs: measured in the C2 step, forward products 106 -> 87 us, input-gradient products 116 -> 128 us
    and the side stream's kernels twice as long (profiles/r05_g_limb_pc_step_timelines.txt)."""
    mode = _cfg.limb_pc
    if mode == "0" or (mode == "fwd" and kind != WEIGHT_NN):     # (the small input-gradient products, K <= 256 and N = 256, on it too: no
        return False                                              #  difference, 1.8115 vs 1.8101 ms over three alternations)
    from . import _lib
    if a.shape[0] < _LIMB_MIN_ROWS or not _lib.load_library().relgnn_limb_gemm_xf32_pc_supported(int(act), a.shape[0], n, k):
        return False                                              # (the shape list lives in the library: csrc/limb_gemm_pc.hip)
    return (a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
            and (bias is None or bias.data_ptr() % 16 == 0) and (dy is None or (dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0)))


def _limb_group_ok(a: torch.Tensor, ws, kind: str) -> bool:
    if not (_cfg.limb_gemm and _rows_ok(a) and a.shape[0] >= _LIMB_MIN_ROWS and weight_image_ok(ws, kind)):
        return False
    n, k = _weight_image_shape(ws, kind)
    return a.shape[1] == k and n % 256 == 0 and 16 <= k <= _LIMB_MAX_K


def grouped_nn_gemm(a: torch.Tensor, kernels, relu: bool = False, xmax: torch.Tensor = None, xgroups: int = 0) -> torch.Tensor:
    """(relu of) sum_l a[:, block l] @ kernels[l] for a [V, sum_l K_l], kernels[l] [K_l, N]: gnns/rgcn.py:96-98 summed over the
    edge types in one product (the aggregate-first layer's forward)."""
    from . import _lib
    kernels = list(kernels)
    if _limb_group_ok(a, kernels, WEIGHT_NN):
        return limb_gemm_weight(a, kernels, WEIGHT_NN, None, _lib.ACT_RELU if relu else _lib.ACT_LINEAR, xmax=xmax, xgroups=xgroups)
    return lib_gemm(GEMM_NN, a, torch.cat(kernels, dim=0) if len(kernels) > 1 else kernels[0], relu=relu)


def grouped_nt_gemm(g: torch.Tensor, kernels, xmax: torch.Tensor = None, xgroups: int = 0, premask=None) -> torch.Tensor:
    """sum_l g[:, block l] @ kernels[l]^T for g [V, sum_l K_l], kernels[l] [N, K_l]: the input gradient of grouped_nn_gemm's layer
    (dH = sum_l dT_l @ W_l^T).  premask = (act id, y): times act'(y), y [V, N] the layer's INPUT as the output of that activation
    (lib_gemm's premask)."""
    kernels = list(kernels)
    if _limb_group_ok(g, kernels, WEIGHT_NT) and (premask is None or _premask_operand_ok(premask[1], g.shape[0], kernels[0].shape[0])):
        return limb_gemm_weight(g, kernels, WEIGHT_NT, xmax=xmax, xgroups=xgroups,
                                dact=premask[0] if premask is not None else 0, dy=premask[1] if premask is not None else None)
    # (the stacked [sum K_l, N] right operand is W_l^T row blocks, 0.8 MB re-laid per call at C2)
    res = lib_gemm(GEMM_NN, g, torch.cat([k.t() for k in kernels], dim=0))
    return res if premask is None else act_bwd_from_output(premask[0], premask[1], res, out=res)


def _limb_cut_route_ok(a: torch.Tensor, b: torch.Tensor, bias) -> bool:
    """a @ b (+ bias) with b [K, N], N % 128 >= 96: the 128-column panels with the last one cut at N (library pick for
    [36 k, 256] @ [256, 121]: 59-65 us; this route: measured in profiles/)."""
    if not (_cfg.limb_cut == "1") or not _rows_ok(a) or a.shape[0] < _LIMB_MIN_ROWS:
        return False
    K, N = a.shape[1], b.shape[1]
    return (b.is_cuda and b.dtype == torch.float32 and b.dim() == 2 and b.is_contiguous() and b.shape[0] == K and b.data_ptr() % 16 == 0
            and N % 128 >= 96 and K % 16 == 0 and 16 <= K <= _LIMB_MAX_K
            and (bias is None or (bias.is_cuda and bias.is_contiguous() and bias.dtype == torch.float32)))


def limb_dense(layout: int, a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor = None, act: int = 0,
               out: torch.Tensor = None, weight: bool = False) -> torch.Tensor:
    """NN act(bias + a @ b) | NT a @ b^T through relgnn_limb_dense_f32: b (the weights) split into limbs in a per-(device, stream)
    scratch buffer, a split inside the product kernel."""
    from . import _lib
    if weight:           # b is a parameter (or a view of one): its limbs are kept across the products of a step
        return limb_gemm_weight(a, b, WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT, bias, act, out)
    lib = _lib.load_library()
    M, K = a.shape
    N = b.shape[1] if layout == GEMM_NN else b.shape[0]
    need = int(lib.relgnn_limb_elements(N, K))
    ws = _limb_ws(a.device, need)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_limb_dense_f32(layout, act, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), _lib.ptr(bias),
                                         _lib.ptr(_zeros(a.device)), ws.data_ptr(), ws.numel(), out.data_ptr(), out.stride(0), M, N, K,
                                         _lib.current_stream()), "relgnn_limb_dense_f32")
    return out


def mm_into(layout: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[:] = a @ b (GEMM_NN) | a @ b^T (GEMM_NT) into a preallocated row block: the limb route when the shapes allow
    (RELGNN_GEMM=limb), else the library through torch.mm.  For the per-edge-type row blocks of an edge MLP (ops._BlockedLinear)."""
    if _cfg.limb_gemm and _limb_route_ok(layout, a, b, None) and _rows_ok(out):
        return limb_dense(layout, a, b, out=out)
    return torch.mm(a, b if layout == GEMM_NN else b.t(), out=out)


def _limb_ws(device, need: int) -> torch.Tensor:
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _LIMB_WS.lookup(key)
    if ws is None or ws.numel() < need:
        ws = _LIMB_WS.store(key, torch.empty(max(need, 1 << 20), dtype=torch.bfloat16, device=device))
    return ws


_SEL_CACHE = True          # (scripts flip this for A/B runs of the cached panel-product images; not a route switch)


def sel_weights_cacheable(ws, layout: int) -> bool:
    """May the 128-column panel product take its weights from the step's limb-image cache (weight_image(separate=True))?  ws: the
    weight matrices as the caller holds them (parameters or views of parameters: something whose storage outlives the product and
    whose version moves when it is written) — all the same shape, N % 128 == 0, K % 16 == 0."""
    ws = _weight_matrices(ws)
    kind = WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT
    n, k = (ws[0].shape[1], ws[0].shape[0]) if layout == GEMM_NN else (ws[0].shape[0], ws[0].shape[1])
    return (_SEL_CACHE and _cfg.weight_limb_cache == "1" and n % 128 == 0 and k % 16 == 0 and all(m.shape == ws[0].shape for m in ws)
            and weight_image_ok(ws[:1], kind) and all(weight_image_ok([m], kind) for m in ws[1:]))


_FAST_IMAGES = {}


def sel_image(ws, layout: int):
    """The cached limb images of the weight matrices `ws` (one image per matrix, one behind the other) for the 128-column panel
    products, or None when they cannot come from the cache (shapes, switches).  Looked up by the IDENTITY of the weight tensors
    first — a training step asks for the same parameters' images five or six times per layer, and the general lookup
    (weight_image: addresses, shapes, strides, bases of every matrix) costs more host time than the split launch it saves when a
    layer has 23 of them (measured, round 6: C5 31.2 -> 32.7 ms with the general lookup alone, eager)."""
    if not (_SEL_CACHE and _cfg.weight_limb_cache == "1"):
        return None
    from . import _lib
    ws = _weight_matrices(ws)
    if not ws[0].is_cuda:
        return None
    kind = WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT
    capturing = torch.cuda.is_current_stream_capturing()
    key = None
    if not capturing:
        key = (kind, _lib.current_stream()) + tuple(map(id, ws))
        ent = _FAST_IMAGES.get(key)
        if ent is not None:
            im, refs = ent
            if im.gen == _WEIGHT_GEN[0]:
                for w, r, v in zip(ws, refs, im.versions):
                    if r() is not w or w._version != v:
                        break
                else:
                    im.used_gen = im.gen
                    return im
    if not sel_weights_cacheable(ws, layout):
        return None
    im = weight_image(ws, kind, separate=True)
    if key is not None and all(w._base is None for w in ws):
        import weakref
        if len(_FAST_IMAGES) > 512:
            _FAST_IMAGES.clear()
        _FAST_IMAGES[key] = (im, [weakref.ref(w) for w in ws])
    return im


def limb_dense_sel(layout: int, a: torch.Tensor, b, bias: torch.Tensor = None, act: int = 0, *,
                   a_rows: torch.Tensor = None, num_rows: int = None, b_select: torch.Tensor = None, rows_per_select: int = 0,
                   cached: bool = False, as_one: bool = False, image=None, out: torch.Tensor = None) -> torch.Tensor:
    """relgnn_limb_dense_sel_f32: the limb product in 128 x 128 panels.  b: [K, N] / [N, K] (NN / NT) or, with b_select,
    [num_b, K, N] / [num_b, N, K]; a_rows: int32 row ids of `a` per output row (< 0: zeros), num_rows output rows.
    cached=True (sel_weights_cacheable(b, layout)): b is the weight matrix / the LIST of per-type weight matrices themselves; their
    limbs come from the step's image cache (relgnn_limb_gemm_sel_xf32: no split launch, no stacked copy of the weights)."""
    from . import _lib
    lib = _lib.load_library()
    K = a.shape[1]
    if cached or image is not None:          # (image: sel_image(b, layout), looked up by the caller)
        ws = _weight_matrices(b)
        kind = WEIGHT_NN if layout == GEMM_NN else WEIGHT_NT
        N = ws[0].shape[1] if layout == GEMM_NN else ws[0].shape[0]
        M = int(num_rows) if a_rows is not None else a.shape[0]
        im = image if image is not None else weight_image(ws, kind, separate=True)
        if as_one:           # the images one behind the other = the image of [w_0 | w_1 | ..] stacked along N: ONE product, L*N columns
            return _sel_with_image(a, im, len(ws) * N, K, act, bias)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        elif out.shape != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError("limb_dense_sel: out must be a float32 [%d, %d] matrix with dense rows" % (M, N))
        if ((_cfg.typed_pc == "1" or (_cfg.typed_pc == "fwd" and a_rows is not None and N == 256)) and b_select is not None
                and bias is None and act == 0
                and lib.relgnn_limb_gemm_sel_pc_supported(M, N, K, int(rows_per_select))
                and (a_rows is None or a_rows.data_ptr() % 16 == 0)):
            # wave roles (csrc/limb_gemm_pc_typed.hip): the same bits, every gathered row read once
            _lib.check(lib.relgnn_limb_gemm_sel_pc_xf32(a.data_ptr(), a.stride(0), _lib.ptr(a_rows), im.buf.data_ptr(), len(ws),
                                                        _lib.ptr(b_select), int(rows_per_select), _lib.ptr(_zeros(a.device)),
                                                        out.data_ptr(), out.stride(0), M, N, K,
                                                        _ops().handover_word(a.device).data_ptr(), _lib.current_stream()),
                       "relgnn_limb_gemm_sel_pc_xf32")
            return out
        _lib.check(lib.relgnn_limb_gemm_sel_xf32(act, a.data_ptr(), a.stride(0), _lib.ptr(a_rows), im.buf.data_ptr(), len(ws),
                                                 _lib.ptr(b_select), int(rows_per_select), _lib.ptr(bias), _lib.ptr(_zeros(a.device)),
                                                 out.data_ptr(), out.stride(0), M, N, K, _lib.current_stream()),
                   "relgnn_limb_gemm_sel_xf32")
        return out
    num_b = b.shape[0] if b.dim() == 3 else 1
    N = b.shape[-1] if layout == GEMM_NN else b.shape[-2]
    M = int(num_rows) if a_rows is not None else a.shape[0]
    need = int(lib.relgnn_limb_elements((N + 127) // 128 * 128, K)) * num_b
    ws = _limb_ws(a.device, need)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_limb_dense_sel_f32(layout, act, a.data_ptr(), a.stride(0), _lib.ptr(a_rows), b.data_ptr(), b.stride(-2),
                                             num_b, b.stride(0) if b.dim() == 3 else 0, _lib.ptr(b_select), int(rows_per_select),
                                             _lib.ptr(bias), _lib.ptr(_zeros(a.device)), ws.data_ptr(), ws.numel(), out.data_ptr(),
                                             out.stride(0), M, N, K, _lib.current_stream()), "relgnn_limb_dense_sel_f32")
    return out


def limb_tn_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    return (_rows_ok(a) and _rows_ok(b) and a.shape[0] == b.shape[0] and a.shape[0] >= _LIMB_MIN_ROWS and a.shape[1] % 32 == 0
            and b.shape[1] % 256 == 0)


def limb_tn_tiles_supported(a: torch.Tensor, g: torch.Tensor, a_rows: torch.Tensor, rows_per_tile: int) -> bool:
    """Shapes relgnn_limb_gemm_tn_tiles_f32 takes (the typed weight-gradient partials of ops.typed_linear)."""
    return (_cfg.limb_gemm and _rows_ok(a) and _rows_ok(g) and a_rows.is_cuda and a_rows.dtype == torch.int32 and a_rows.is_contiguous()
            and a_rows.data_ptr() % 16 == 0 and a_rows.numel() == g.shape[0] and rows_per_tile % 32 == 0
            and g.shape[0] % rows_per_tile == 0 and a.shape[1] % 64 == 0 and g.shape[1] % 128 == 0)


def limb_gemm_tn_tiles(a: torch.Tensor, g: torch.Tensor, a_rows: torch.Tensor, rows_per_tile: int) -> torch.Tensor:
    """part[z] = a[a_rows[tile z]]^T @ g[tile z] for the P / rows_per_tile tiles of a compact pair table (a [*, J] node table, g [P, C]
    the table's gradient, a_rows [P] int32, < 0 = padding): [tiles, J, C], three bf16 limbs per value, gathered / transposed /
    split in flight (relgnn_limb_gemm_tn_tiles_f32)."""
    from . import _lib
    lib = _lib.load_library()
    P, C = g.shape
    J = a.shape[1]
    tiles = P // rows_per_tile
    part = torch.empty((tiles, J, C), dtype=torch.float32, device=g.device)
    _lib.check(lib.relgnn_limb_gemm_tn_tiles_f32(a.data_ptr(), a.stride(0), a_rows.data_ptr(), g.data_ptr(), g.stride(0),
                                                 _lib.ptr(_zeros(g.device)), part.data_ptr(), P, int(rows_per_tile), J, C,
                                                 _lib.current_stream()), "relgnn_limb_gemm_tn_tiles_f32")
    return part


def col_absmax(x: torch.Tensor) -> torch.Tensor:
    """[cols] float32 on the device: the largest finite magnitude of every column of x [rows, cols] (relgnn_col_absmax_f32)."""
    from . import _lib
    lib = _lib.load_library()
    cols = x.shape[1]
    if cols > 16 and cols % 4:
        # the kernel's wide form reads float4 columns: pad to the next multiple of 4 with zeros (a zero never is a column's largest
        # magnitude unless the column is zero) — e.g. the [V, L] bucket magnitudes of an 18-type aggregate-first layer
        return col_absmax(torch.nn.functional.pad(x, (0, (-cols) % 4)))[:cols]
    if x.shape[1] > 16 and not _rows_ok(x):
        x = x.contiguous()
    if x.shape[1] <= 16 and not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        x = x.contiguous()
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    nbytes = int(lib.relgnn_col_absmax_workspace_bytes(x.shape[0], x.shape[1]))
    ws = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device) if nbytes else None
    _lib.check(lib.relgnn_col_absmax_f32(_lib.ptr(x, rows_strided=True), x.stride(0) if x.shape[0] > 1 else x.shape[1], x.shape[0],
                                         x.shape[1], out.data_ptr(), _lib.ptr(ws), nbytes, _lib.current_stream()),
               "relgnn_col_absmax_f32")
    return out


def absmax(x: torch.Tensor) -> torch.Tensor:
    """[1] float32 on the device: max |x| over the finite elements (relgnn_absmax_f32; no host round trip)."""
    from . import _lib
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load_library().relgnn_absmax_f32(_lib.ptr(x), x.numel(), out.data_ptr(), _lib.current_stream()), "relgnn_absmax_f32")
    return out


def limb_gemm_tn(a: torch.Tensor, b: torch.Tensor, amax: torch.Tensor = None, bmax: torch.Tensor = None) -> torch.Tensor:
    """a^T @ b for a [V, J], b [V, C] (weight gradient) through relgnn_limb_gemm_tn_f32 + the in-order slab sum.
    amax, bmax (device floats): the two-fp16-limb form — [J] / [C] magnitudes per column (col_absmax(): one power-of-two scale
    per column of each operand), [1] / [1] (absmax(): one scale per operand), or any count that divides the operand's width (one
    per group of consecutive columns)."""
    from . import _lib
    lib = _lib.load_library()
    V, J = a.shape
    C = b.shape[1]
    Z = int(lib.relgnn_limb_gemm_tn_chunks(V, J, C))
    if Z <= 0:
        raise ValueError("limb_gemm_tn: unsupported shape [%d, %d]^T @ [%d, %d]" % (V, J, V, C))
    parts = torch.empty((Z, J, C), dtype=torch.float32, device=a.device)
    if amax is not None:
        na, nb = amax.numel(), bmax.numel()
        if na < 1 or nb < 1 or J % na or C % nb or amax.dtype != torch.float32 or bmax.dtype != torch.float32:
            raise ValueError("limb_gemm_tn: the magnitude counts (%d, %d) must divide the operand widths (%d, %d)" % (na, nb, J, C))
        _lib.check(lib.relgnn_limb16_gemm_tn_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), amax.data_ptr(), J // na,
                                                 bmax.data_ptr(), C // nb, parts.data_ptr(), V, J, C, _lib.current_stream()),
                   "relgnn_limb16_gemm_tn_f32")
    else:
        _lib.check(lib.relgnn_limb_gemm_tn_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), parts.data_ptr(), V, J, C,
                                               _lib.current_stream()), "relgnn_limb_gemm_tn_f32")
    # the slabs in chunk order + the last V % 32 rows (exact fp32), one pass
    R = V % 32
    out = torch.empty((J, C), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_sum_slabs_tail_f32(_lib.ptr(parts), Z, J, C, a[V - R:].data_ptr() if R else None, a.stride(0),
                                             b[V - R:].data_ptr() if R else None, b.stride(0), R, _lib.ptr(out),
                                             _lib.current_stream()), "relgnn_sum_slabs_tail_f32")
    return out


def enable_gemm_autotuning(max_tuning_ms_per_solution: int = 30, tune: bool = True) -> bool:
    """PyTorch TunableOp: for every GEMM shape the step uses, time the candidate rocBLAS / hipBLASLt solutions once
    (at first use) and keep the fastest.  Measured on MI355X, config C2: 2.72 -> 2.39 ms per training step (the fp32
    node-side GEMMs are ~half of the step).  The result table is written under the system temp directory, not into the
    working directory.  Returns False if this torch build has no TunableOp.  Call
    `enable_gemm_autotuning(tune=False)` after warm-up to freeze the choices."""
    import os
    import tempfile
    tun = getattr(torch.cuda, "tunable", None)
    if tun is None or not torch.cuda.is_available():
        return False
    try:
        tun.enable(True)
        tun.tuning_enable(bool(tune))
    except Exception:
        return False
    for fn, arg in (("set_filename", os.path.join(tempfile.gettempdir(), "relgnn_tunableop_%d.csv" % os.getpid())),
                    ("set_max_tuning_duration", int(max_tuning_ms_per_solution))):
        try:
            if tune:
                getattr(tun, fn)(arg)
        except Exception:
            pass
    return True


def _split_count(V: int, M: int, N: int) -> int:
    """Number of K-chunks: enough output tiles (~128x128) x chunks to fill 256 CUs, chunks >= 512 rows."""
    tiles = max(1, ((M + 127) // 128) * ((N + 127) // 128))
    want = max(1, 512 // tiles)
    return int(max(1, min(want, V // 512, 64)))


def tn_stream_into(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> None:
    """out[:] = a^T @ b through the streaming weight-gradient kernel, `out` a [M, N] block of a wider row-major matrix (dense rows,
    any row stride): the two column blocks of a GRU's recurrent-kernel gradient are written in place, no zeros + copies + add."""
    from . import _lib
    lib = _lib.load_library()
    V, M = a.shape
    N = b.shape[1]
    if out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise ValueError("tn_stream_into: out must be a float32 [%d, %d] block with dense rows" % (M, N))
    nbytes = lib.relgnn_gemm_tn_stream_workspace_bytes(M, N, V)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_gemm_tn_stream_f32(_lib.ptr(a, rows_strided=True), a.stride(0), _lib.ptr(b, rows_strided=True), b.stride(0),
                                             out.data_ptr(), out.stride(0), M, N, V, 0, _lib.ptr(ws), nbytes, _lib.current_stream()),
               "relgnn_gemm_tn_stream_f32")


_TN_BLOCKS_MAX_OUT = 128 * 1024      # outputs of the block form (measured up to [128, 640]; its partial sums are chunks * M * N floats)


def tn_stream_blocks_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    return (_cfg.tn == "stream" and a.is_cuda and a.shape[1] * b.shape[1] <= _TN_BLOCKS_MAX_OUT and 0 < a.shape[0] <= (1 << 18)
            and _lib_rows_ok(a) and _lib_rows_ok(b))


def tn_stream_blocks(a: torch.Tensor, b: torch.Tensor, L: int) -> torch.Tensor:
    """[L, M, N / L]: block l = a^T @ b[:, l * N / L : (l + 1) * N / L] for a [V, M], b [V, N] — ONE pass of the streaming
    weight-gradient kernel over a, every block a dense matrix of its own (relgnn_gemm_tn_stream_blocks_f32)."""
    from . import _lib
    lib = _lib.load_library()
    V, M = a.shape
    N = b.shape[1]
    bc = N // L
    out = torch.empty((L, M, bc), dtype=torch.float32, device=a.device)
    nbytes = lib.relgnn_gemm_tn_stream_workspace_bytes(M, N, V)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_gemm_tn_stream_blocks_f32(_lib.ptr(a, rows_strided=True), a.stride(0), _lib.ptr(b, rows_strided=True),
                                                    b.stride(0), out.data_ptr(), bc, M * bc, M, N, bc, V, 0, _lib.ptr(ws), nbytes,
                                                    _lib.current_stream()), "relgnn_gemm_tn_stream_blocks_f32")
    return out


def tn_stream_group_ok(products) -> bool:
    """May these (a [V, M], b [V, N], out [M, N] block) triples go through relgnn_gemm_tn_stream_group_f32?  At most four, the same
    V, whole 64 x 64 tiles, 8-byte aligned operands with even row strides."""
    if not (_cfg.tn == "stream" and 1 <= len(products) <= 4):
        return False
    V = products[0][0].shape[0]
    for a, b, out in products:
        if not (a.is_cuda and a.dtype == b.dtype == out.dtype == torch.float32 and a.shape[0] == b.shape[0] == V and 0 < V <= (1 << 18)
                and a.shape[1] % 64 == 0 and b.shape[1] % 64 == 0 and out.shape == (a.shape[1], b.shape[1])
                and all(t.stride(1) == 1 and t.stride(0) % 2 == 0 and t.data_ptr() % 8 == 0 for t in (a, b))
                and out.stride(1) == 1 and out.stride(0) >= out.shape[1]):
            return False
    return sum(a.shape[1] * b.shape[1] for a, b, _ in products) <= 4 * _TN_BLOCKS_MAX_OUT


def tn_stream_group(products, colsum: torch.Tensor = None) -> None:
    """out_i[:] = a_i^T @ b_i for every (a_i, b_i, out_i) — ONE pass of the streaming weight-gradient kernel and one reduction launch
    for all of them (relgnn_gemm_tn_stream_group_f32); colsum (contiguous [N_0]): also the column sums of b_0, the bias gradient of
    the layer whose kernel gradient product 0 is."""
    import ctypes
    from . import _lib
    lib = _lib.load_library()
    n = len(products)
    V = products[0][0].shape[0]
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    M = i32(*[a.shape[1] for a, _, _ in products])
    N = i32(*[b.shape[1] for _, b, _ in products])
    nbytes = lib.relgnn_gemm_tn_stream_group_workspace_bytes(n, M, N, V, 1 if colsum is not None else 0)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=products[0][0].device)
    _lib.check(lib.relgnn_gemm_tn_stream_group_f32(
        n, vp(*[a.data_ptr() for a, _, _ in products]), i64(*[a.stride(0) for a, _, _ in products]),
        vp(*[b.data_ptr() for _, b, _ in products]), i64(*[b.stride(0) for _, b, _ in products]),
        vp(*[o.data_ptr() for _, _, o in products]), i64(*[o.stride(0) for _, _, o in products]), M, N, V, _lib.ptr(colsum),
        _lib.ptr(ws), nbytes, _lib.current_stream()), "relgnn_gemm_tn_stream_group_f32")


def tn_stream_gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """a^T @ b for a [V, M], b [V, N] through the streaming weight-gradient kernel (csrc/gemm_tn_stream.hip);
    with `out` (contiguous [M, N]): out += a^T @ b."""
    from . import _lib
    lib = _lib.load_library()
    V, M = a.shape
    N = b.shape[1]
    accumulate = out is not None
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    nbytes = lib.relgnn_gemm_tn_stream_workspace_bytes(M, N, V)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_gemm_tn_stream_f32(_lib.ptr(a, rows_strided=True), a.stride(0), _lib.ptr(b, rows_strided=True),
                                             b.stride(0), _lib.ptr(out), N, M, N, V, 1 if accumulate else 0, _lib.ptr(ws),
                                             nbytes, _lib.current_stream()), "relgnn_gemm_tn_stream_f32")
    return out


def matmul_tn_splitk(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T @ b for a [V, M], b [V, N] (both row-major), reduction over V split into S chunks."""
    V, M = a.shape
    N = b.shape[1]
    # small outputs (every Dense of the path except the stacked per-type transforms): the streaming kernel — measured at
    # V = 36 k: [256 x 256] 59 us vs 171 us for the library's strided-batched split-K, [256 x 121] 43 vs 114, [50 x 256]
    # 33 vs 57, [128 x 128] 30 vs 56; the library wins for [768 x 256] (131 vs 223) and for V ~ 1e6 (scripts/exp_tn_stream.py)
    if (_cfg.tn == "stream") and M * N <= 256 * 256 and 0 < V <= (1 << 18) and _lib_rows_ok(a) and _lib_rows_ok(b):
        return tn_stream_gemm(a, b)
    if _cfg.limb_gemm and limb_tn_supported(a, b):
        return limb_gemm_tn(a, b)
    S = _split_count(V, M, N)
    if S <= 1:
        return lib_gemm(GEMM_TN, a, b)
    if not (a.is_contiguous() and b.is_contiguous()):      # the chunked forms below view the operands as [S, c, .]
        a, b = a.contiguous(), b.contiguous()
    c = V // S
    head = c * S
    # (outputs narrower than one 64-wide tile — the [50, 256] gradient of the input projection — keep torch.bmm: hipBLASLt's
    # strided-batched pick for them measured 148 us against 52 us, scripts/exp_cached_gemm_quality.py)
    if (_cfg.gemm != "torch") and min(M, N) >= 64 and _lib_rows_ok(a) and _lib_rows_ok(b) and a.is_contiguous() and b.is_contiguous():
        # one strided-batched library call: chunk z = rows [z*c, (z+1)*c) of both operands
        from . import _lib
        lib = _lib.load_library()
        parts = torch.empty((S, M, N), dtype=torch.float32, device=a.device)
        ws = _workspace(a.device)
        _lib.check(lib.relgnn_blaslt_gemm_f32(GEMM_TN, _lib.ACT_LINEAR, _lib.ptr(a), M, _lib.ptr(b), N, None, _lib.ptr(parts), N, M, N, c, S,
                                              c * M, c * N, M * N, 0, _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   "relgnn_blaslt_gemm_f32")
        # the partial products are summed in slab order and the < S leftover rows (V = S * c + R) are multiplied in by the
        # same pass (relgnn_sum_slabs_tail_f32): torch.sum + a second product + an accumulate were three launches, and the
        # library's pick for a [16, 128]^T @ [16, 640] leftover took 191 us (C3 timeline)
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        R = V - head
        _lib.check(lib.relgnn_sum_slabs_tail_f32(_lib.ptr(parts), S, M, N, _lib.ptr(a[head:]) if R else None, M,
                                                 _lib.ptr(b[head:]) if R else None, N, R, _lib.ptr(out),
                                                 _lib.current_stream()), "relgnn_sum_slabs_tail_f32")
        return out
    out = torch.bmm(a[:head].view(S, c, M).transpose(1, 2), b[:head].view(S, c, N)).sum(0)
    if head < V:
        out.addmm_(a[head:].t(), b[head:])       # the < c leftover rows, accumulated in place
    return out


def column_sum(g: torch.Tensor) -> torch.Tensor:
    """sum over rows of a [V, N] tensor (bias gradient).  torch's strided reduction took 330 us and rocBLAS gemv
    230 us for [32k, 121] on MI355X; the two-stage HIP kernel (csrc/dense_utils.hip) is bandwidth-bound."""
    if not g.is_cuda:
        return g.sum(0)
    from . import _lib
    lib = _lib.load_library()
    V, N = g.shape
    out = torch.empty(N, dtype=torch.float32, device=g.device)
    nbytes = lib.relgnn_column_sum_workspace_bytes(V, N)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device)
    _lib.check(lib.relgnn_column_sum(_lib.ptr(g, rows_strided=True), V, N, g.stride(0), _lib.ptr(out), _lib.ptr(ws), nbytes,
                                     _lib.current_stream()), "relgnn_column_sum")
    return out


def _leaf_params(kernel, bias):
    """(kernel, bias) when both are leaf parameters — their gradients go straight to the accumulator — else None: the gradient of a
    VIEW of a parameter (the GRU's recurrent_kernel[:, :2u]) is consumed by the view's backward on the main stream at once."""
    ok = kernel.is_leaf and kernel.requires_grad and (bias is None or (bias.is_leaf and bias.requires_grad))
    return ((kernel,) if bias is None else (kernel, bias)) if ok else None


def _on_side_stream(run, operands, params, want=True):
    """A Dense layer's weight and bias gradient on the weight-gradient side stream (ops._side_stream) — only while train_step defers
    the joins behind the whole backward (ops.deferred_weight_gradient_join): the gradients then leave the main stream's critical
    path and run under the next layer's gather (C2 step 1.826 -> 1.807 ms).  With the join inside backward() the same move was measured and lost (both
    products are matrix-pipe kernels: 2.02 vs 1.94 ms per C2 step), so outside train_step everything stays on one stream.
    `params`: the leaf parameters the gradients go to (ops.deferred_targets_ok: only a parameter that has no gradient yet takes its
    gradient tensor without launching anything on the main stream), or None.  Returns run()'s result, or None when not applicable."""
    from . import ops
    if not (want and ops._DEFER["on"] and _cfg.bwd_overlap_on and all(t.is_cuda for t in operands)):
        if operands[0].is_cuda:
            ops.wait_if_in_flight(params, operands[0].device)     # (no-op unless an earlier use of these parameters went aside)
        return None
    if not ops.deferred_targets_ok(params, operands[0].device):
        return None
    device = operands[0].device
    side = ops._side_stream(device)
    cur = torch.cuda.current_stream(device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = run()
    for t in operands:
        t.record_stream(side)
    for t in out:
        if t is not None:
            t.record_stream(cur)
    ops.hand_over_deferred(device, side, params, out)
    return out


class _DenseFn(torch.autograd.Function):
    """act(x @ kernel (+ bias)) for act in {linear, tanh, relu, leaky_relu, elu, selu} — the activation in the product's epilogue where
    the route has one, differentiated from the saved OUTPUT — with a split-K weight gradient.
    x_act: the activation x itself is the output of, when the caller may fold its gradient into this function's input-gradient
    product (fusable_activation_of(x)): g_x then leaves already multiplied by x_act'(x) and tagged (mark_premasked)."""

    @staticmethod
    def forward(ctx, x, kernel, bias, act: int, x_act: int):
        y = lib_gemm(GEMM_NN, x, kernel, bias, weight=True, act=act)
        ctx.save_for_backward(x, kernel, y if act else None)
        ctx.has_bias, ctx.act, ctx.x_act = bias is not None, act, x_act
        ctx.leaf_params = _leaf_params(kernel, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x, kernel, y = ctx.saved_tensors
        if ctx.act:
            # g * act'(y): nothing to do when the consumer of y already folded it into the product that made g
            if not is_premasked(g, y, ctx.act):
                g = act_bwd_from_output(ctx.act, y, g)
        if g.dim() != 2 or g.stride(1) != 1 or not g.is_cuda or (g.stride(0) < g.shape[1] and g.shape[0] != 1):
            g = g.contiguous()             # (a row-strided gradient, e.g. a column block of the GRU's gate gradients, is
        gx = None                          # read in place: every consumer below takes a leading dimension; an expanded one,
                                           # strides (0, 1), is materialised)
        def weight_side():
            gk = matmul_tn_splitk(x if (x.dim() == 2 and x.stride(1) == 1 and x.is_cuda) else x.contiguous(), g) \
                if ctx.needs_input_grad[1] else None
            gb = column_sum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
            return gk, gb

        aside = _on_side_stream(weight_side, (x, g), ctx.leaf_params, want=ctx.needs_input_grad[0] and ctx.needs_input_grad[1])
        if ctx.needs_input_grad[0]:
            if ctx.x_act and _premask_operand_ok(x, g.shape[0], kernel.shape[0]):
                gx = mark_premasked(lib_gemm(GEMM_NT, g, kernel, weight=True, premask=(ctx.x_act, x)), x, ctx.x_act)
            else:
                gx = lib_gemm(GEMM_NT, g, kernel, weight=True)
        gk, gb = aside if aside is not None else weight_side()
        return gx, gk, gb, None, None


class _DenseMultiFn(torch.autograd.Function):
    """x @ [k_0 | k_1 | ..] for L kernels [K, N] (the per-edge-type Dense kernels of a GGNN / FiLM layer applied to every node:
    gnns/ggnn.py:60-64,81) WITHOUT forming the column-concatenated [K, L*N] operand: the limb images of the kernels, one behind the
    other, ARE the image of the concatenation (row blocks of 32 output columns are contiguous), and they come from the step's cache.
    Round 6: per layer one torch.cat, its five slice copies in the backward and three split launches less."""

    @staticmethod
    def forward(ctx, x, *kernels):
        y = limb_dense_sel(GEMM_NN, x, list(kernels), image=sel_image(kernels, GEMM_NN), as_one=True)
        ctx.save_for_backward(x, *kernels)
        ctx.leaf_params = tuple(kernels) if all(k.is_leaf for k in kernels) else None
        return y

    @staticmethod
    def backward(ctx, g):
        x, *kernels = ctx.saved_tensors
        L, (K, N) = len(kernels), kernels[0].shape
        g = g.contiguous()
        gx = gks = None

        def weight_side():
            # dk_l = x^T @ g[:, block l].  Each gradient has to be a dense [K, N] tensor of its own — autograd's accumulator keeps
            # such a tensor as it is, a column block of a [K, L*N] product it would clone (five copies, and on the main stream
            # while the side stream still writes it) — so the ONE product x^T @ g writes its column blocks as L matrices
            # (tn_stream_blocks: x read once, two launches instead of 2 L; C3: 50 us instead of 5 x 50 per layer)
            if all(ctx.needs_input_grad[1:]) and tn_stream_blocks_ok(x, g):
                return tuple(tn_stream_blocks(x, g, L).unbind(0))
            return tuple(matmul_tn_splitk(x, g[:, l * N:(l + 1) * N]) if ctx.needs_input_grad[1 + l] else None for l in range(L))

        aside = _on_side_stream(weight_side, (x, g), ctx.leaf_params, want=ctx.needs_input_grad[0] and any(ctx.needs_input_grad[1:]))
        if ctx.needs_input_grad[0]:
            # gx = sum_l g[:, block l] @ k_l^T: the kernels side by side along the reduction (WEIGHT_NT image), 128 output columns
            im = weight_image(kernels, WEIGHT_NT)                        # B [K, L*N] = [k_0 | k_1 | ..] as stored
            gx = _sel_with_image(g, im, K, L * N)
        gks = aside if aside is not None else (weight_side() if any(ctx.needs_input_grad[1:]) else (None,) * L)
        return (gx,) + tuple(gks)


def _sel_with_image(a: torch.Tensor, im, n: int, k: int, act: int = 0, bias: torch.Tensor = None) -> torch.Tensor:
    """act(bias + a @ B^T) on the 128-column panels, B [n, k] = the limb image im (relgnn_limb_gemm_sel_xf32, one matrix)."""
    from . import _lib
    lib = _lib.load_library()
    out = torch.empty((a.shape[0], n), dtype=torch.float32, device=a.device)
    _lib.check(lib.relgnn_limb_gemm_sel_xf32(act, a.data_ptr(), a.stride(0), None, im.buf.data_ptr(), 1, None, 0, _lib.ptr(bias),
                                             _lib.ptr(_zeros(a.device)), out.data_ptr(), out.stride(0), a.shape[0], n, k,
                                             _lib.current_stream()), "relgnn_limb_gemm_sel_xf32")
    return out


def dense_multi(x: torch.Tensor, kernels) -> torch.Tensor:
    """x @ [k_0 | k_1 | ..] ([V, L*N]; row v viewed as [L, N] is (x_v k_0, .., x_v k_{L-1})) for L same-shaped kernels [K, N]."""
    kernels = list(kernels)
    K, N = kernels[0].shape
    if (_cfg.limb_gemm and _rows_ok(x) and x.shape[0] >= _LIMB_MIN_ROWS and x.shape[1] == K and N % 128 == 0 and K % 128 == 0
            and len(kernels) * N <= _LIMB_MAX_K and all(k.is_contiguous() and k.data_ptr() % 16 == 0 for k in kernels)
            and sel_image(kernels, GEMM_NN) is not None):
        return _DenseMultiFn.apply(x, *kernels)
    return dense(x, torch.cat(kernels, dim=1))


def dense(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor = None, sole_reader: bool = False) -> torch.Tensor:
    """x @ kernel (+ bias) with a split-K weight gradient.  sole_reader: this call is the only reader of x (see the protocol at the
    top of this file; it only matters when x is a tagged activation output)."""
    return _DenseFn.apply(x, kernel, bias, 0, fusable_activation_of(x, sole_reader) if x.requires_grad else 0)


def dense_act(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor = None, act: int = 0,
              sole_consumer: bool = False, sole_reader: bool = False) -> torch.Tensor:
    """act(dense(x, kernel, bias)) as ONE product with the activation in its epilogue (activation ids of _lib; gelu and anything the
    epilogue does not take: the two-step route).  The result is tagged as an activation output (mark_activation_output) so that the
    function that consumes it may fold act' into its input-gradient product; sole_consumer: the caller vouches that only the function
    it hands the result to will read it; sole_reader: this call is the only reader of x (both: the protocol at the top of this file)."""
    from . import _lib
    if act == _lib.ACT_LINEAR:
        return dense(x, kernel, bias, sole_reader=sole_reader)
    if not (act in _FROM_OUTPUT_ACTS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and _cfg.gemm != "torch"
            and (_cfg.act_fusion == "1" or act == _lib.ACT_RELU)):
        from .utils import apply_activation, get_activation
        return apply_activation(get_activation(_lib.ACT_NAMES[act]), dense(x, kernel, bias, sole_reader=sole_reader))
    y = _DenseFn.apply(x, kernel, bias, act, fusable_activation_of(x, sole_reader) if x.requires_grad else 0)
    return mark_activation_output(y, act, sole_consumer)


def dense_relu(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    """relu(dense(x, kernel, bias)) as one GEMM with a ReLU epilogue (CUDA fp32 operands; anything else takes the two-step
    route)."""
    from . import _lib
    return dense_act(x, kernel, bias, _lib.ACT_RELU)
