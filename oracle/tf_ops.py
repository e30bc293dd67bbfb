"""NumPy restatement of the TensorFlow-1 ops the reference's hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py); PARITY UNPINNED.  Each function names the
reference call sites it stands in for and the TF semantics it assumes ([TF-internal] = recalled
from TensorFlow 1.13 / dpu_utils 0.1.30 sources, not verifiable offline).

Every function computes in the dtype of its floating-point inputs: call with float32 arrays
for the "TF order, TF precision" oracle and with float64 arrays for the high-precision truth.
"""
import ctypes
import math
import os
from pathlib import Path

import numpy as np

SMALL_NUMBER = 1e-7  # utils/utils.py:7
BIG_NUMBER = 1e7     # utils/utils.py:6

_HERE = Path(__file__).resolve().parent
_CLIB = None


def _clib():
    """The C sequential segment kernels (oracle/segment_ops.c); None if not built."""
    global _CLIB
    if _CLIB is None:
        p = _HERE / "_build" / "liboracle_segment.so"
        if p.exists():
            lib = ctypes.CDLL(str(p))
            for n in ("oracle_seg_sum_f32", "oracle_seg_sum_f64", "oracle_seg_max_f32", "oracle_seg_max_f64"):
                getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_void_p]
                getattr(lib, n).restype = None
            lib.oracle_scaled_seg_sum_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
            lib.oracle_scaled_seg_sum_f32.restype = None
            lib.oracle_gather_scaled_seg_sum_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
            lib.oracle_gather_scaled_seg_sum_f32.restype = None
            _CLIB = lib
        else:
            _CLIB = False
    return _CLIB or None


# ---- gather / dense -----------------------------------------------------------------------
def embedding_lookup(params, ids):
    """tf.nn.embedding_lookup(params, ids) == params[ids] (gnns/rgcn.py:87-89 and the same
    idiom in every layer).  Out-of-range ids raise, as TF-CPU's Gather does [TF-internal]."""
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= params.shape[0]):
        raise IndexError("embedding_lookup: index out of range (TF: InvalidArgumentError)")
    return params[ids]


def dense(x, kernel, bias=None, activation=None):
    """tf.keras.layers.Dense / tf.layers.Dense: x @ kernel (+ bias), kernel laid out [in, out]
    (gnns/rgcn.py:70-74,98; utils/utils.py:111-118) [TF-internal layout]."""
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return y if activation is None else activation(y)


# ---- segment reductions (utils/utils.py:23-33) --------------------------------------------
def _seg_fold(data, segment_ids, num_segments, kind):
    data = np.ascontiguousarray(data)
    ids = np.ascontiguousarray(np.asarray(segment_ids, dtype=np.int32))
    if ids.size and ids.max() >= num_segments:
        raise IndexError("segment id out of range (TF-CPU: InvalidArgumentError)")
    M = data.shape[0]
    trailing = data.shape[1:]
    D = int(np.prod(trailing)) if trailing else 1
    out = np.empty((num_segments, D), dtype=data.dtype)
    lib = _clib()
    if lib is not None and data.dtype in (np.float32, np.float64):
        fn = getattr(lib, "oracle_seg_%s_%s" % (kind, "f32" if data.dtype == np.float32 else "f64"))
        fn(data.ctypes.data, ids.ctypes.data, M, D, num_segments, out.ctypes.data)
    else:  # pure NumPy: unbuffered in-order accumulation
        flat = data.reshape(M, D)
        keep = ids >= 0
        if kind == "sum":
            out[...] = 0
            np.add.at(out, ids[keep], flat[keep])
        else:
            out[...] = np.finfo(data.dtype).min  # lowest(), not -inf [TF-internal]
            np.maximum.at(out, ids[keep], flat[keep])
    return out.reshape((num_segments,) + trailing)


def unsorted_segment_sum(data, segment_ids, num_segments):
    """tf.unsorted_segment_sum: sequential in-order accumulation, zeros for empty segments."""
    return _seg_fold(data, segment_ids, num_segments, "sum")


def unsorted_segment_max(data, segment_ids, num_segments):
    """tf.unsorted_segment_max: empty segments hold dtype lowest (-3.4028235e38) [TF-internal]."""
    return _seg_fold(data, segment_ids, num_segments, "max")


def _segment_n(data, segment_ids, num_segments):
    # math_ops._unsorted_segment_N: max(unsorted_segment_sum(ones), 1), broadcast over trailing dims
    ids = np.asarray(segment_ids)
    n = np.bincount(ids[ids >= 0], minlength=num_segments).astype(data.dtype)
    n = np.maximum(n, 1)
    return n.reshape((num_segments,) + (1,) * (data.ndim - 1))


def unsorted_segment_mean(data, segment_ids, num_segments):
    """tf.unsorted_segment_mean = unsorted_segment_sum / max(count, 1) [TF-internal math_ops.py]."""
    return unsorted_segment_sum(data, segment_ids, num_segments) / _segment_n(data, segment_ids, num_segments)


def unsorted_segment_sqrt_n(data, segment_ids, num_segments):
    """tf.unsorted_segment_sqrt_n = unsorted_segment_sum / sqrt(max(count, 1)) [TF-internal]."""
    return unsorted_segment_sum(data, segment_ids, num_segments) / np.sqrt(_segment_n(data, segment_ids, num_segments))


def get_aggregation_function(aggregation_fun):
    """utils/utils.py:23-33, same strings, same ValueError."""
    if aggregation_fun in ['sum', 'unsorted_segment_sum']:
        return unsorted_segment_sum
    if aggregation_fun in ['max', 'unsorted_segment_max']:
        return unsorted_segment_max
    if aggregation_fun in ['mean', 'unsorted_segment_mean']:
        return unsorted_segment_mean
    if aggregation_fun in ['sqrt_n', 'unsorted_segment_sqrt_n']:
        return unsorted_segment_sqrt_n
    raise ValueError("Unknown aggregation function '%s'!" % aggregation_fun)


def unsorted_segment_log_softmax(logits, segment_ids, num_segments):
    """dpu_utils.tfutils.unsorted_segment_log_softmax (used at gnns/rgat.py:126-130)
    [dpu_utils-internal, >= 0.1.30]: x - gather(segmax) - gather(log(segsum(exp(x - gather(segmax)))))."""
    max_per_segment = unsorted_segment_max(logits, segment_ids, num_segments)
    recentered = logits - max_per_segment[segment_ids]
    exped = np.exp(recentered)
    per_segment_sums = unsorted_segment_sum(exped, segment_ids, num_segments)
    return recentered - np.log(per_segment_sums)[segment_ids]


# ---- activations (utils/utils.py:36-58) ---------------------------------------------------
def _erf(x):
    try:
        from scipy.special import erf  # has native float32 / float64 loops
        return erf(x).astype(x.dtype)
    except ImportError:
        return np.vectorize(math.erf, otypes=[np.float64])(x).astype(x.dtype) if x.size else x


def tanh(x):
    return np.tanh(x)


def relu(x):
    return np.maximum(x, 0)


def leaky_relu(x, alpha=0.2):
    """tf.nn.leaky_relu default alpha=0.2 [TF-internal]: max(alpha*x, x)."""
    return np.where(x > 0, x, x * np.asarray(alpha, dtype=x.dtype))


def elu(x):
    """tf.nn.elu: x if x > 0 else exp(x) - 1 [TF-internal]."""
    return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1).astype(x.dtype)


def selu(x):
    scale = np.asarray(1.0507009873554804934193349852946, dtype=x.dtype)
    scale_alpha = np.asarray(1.7580993408473768599402175208123, dtype=x.dtype)
    return np.where(x > 0, scale * x, scale_alpha * (np.exp(np.minimum(x, 0)) - 1)).astype(x.dtype)


def gelu(x):
    """utils/utils.py:52-56: x * 0.5 * (1 + erf(x / sqrt(2)))."""
    cdf = np.asarray(0.5, x.dtype) * (np.asarray(1.0, x.dtype) + _erf(x / np.sqrt(np.asarray(2.0, x.dtype))))
    return x * cdf


def get_activation(activation_fun):
    """utils/utils.py:36-58, same strings (case-insensitive), same ValueError."""
    if activation_fun is None:
        return None
    name = activation_fun.lower()
    table = {'linear': None, 'tanh': tanh, 'relu': relu, 'leaky_relu': leaky_relu, 'elu': elu,
             'selu': selu, 'gelu': gelu}
    if name not in table:
        raise ValueError("Unknown activation function '%s'!" % activation_fun)
    return table[name]


def apply_act(fn, x):
    return x if fn is None else fn(x)


def hard_sigmoid(x):
    """Keras hard_sigmoid: clip(0.2*x + 0.5, 0, 1) — GRUCell's recurrent_activation in TF 1.13 [TF-internal]."""
    return np.clip(np.asarray(0.2, x.dtype) * x + np.asarray(0.5, x.dtype), 0, 1)


# ---- node-wise cells / norms --------------------------------------------------------------
def layer_norm_scope(i):
    """Variable scope of the i-th tf.contrib.layers.layer_norm call inside one variable_scope: TF uniquifies repeated
    default scopes as LayerNorm, LayerNorm_1, LayerNorm_2, ...  The layer functions call layer_norm once per TIMESTEP
    (gnns/gnn_film.py:120, rgin.py:139, gnn_edge_mlp.py:119), so every timestep owns its gamma/beta, and the driver's
    inter-layer norm (models/sparse_graph_model.py:192-193) comes after them in the same scope [TF-internal]."""
    return "LayerNorm" if i == 0 else "LayerNorm_%d" % i


def layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm on [V, D] (gnns/rgin.py:139, gnn_film.py:120,
    gnn_edge_mlp.py:119, models/sparse_graph_model.py:193) [TF-internal]: moments over the last
    axis (biased variance), tf.nn.batch_normalization with variance_epsilon=1e-12:
        inv = rsqrt(var + eps) * gamma;  y = x * inv + (beta - mean * inv)."""
    mean = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    var = np.mean(np.square(x - mean), axis=-1, keepdims=True, dtype=x.dtype)
    inv = (np.asarray(1.0, x.dtype) / np.sqrt(var + np.asarray(eps, x.dtype))) * gamma
    return x * inv + (beta - mean * inv)


def simple_rnn_cell(inputs, state, kernel, recurrent_kernel, bias, activation):
    """tf.keras.layers.SimpleRNNCell (utils/utils.py:13-14): act(x K + b + h U) [TF-internal]."""
    return apply_act(activation, inputs @ kernel + bias + state @ recurrent_kernel)


def gru_cell(inputs, state, kernel, recurrent_kernel, bias, activation):
    """tf.keras.layers.GRUCell as of TF 1.13 (utils/utils.py:15-16) [TF-internal]:
    reset_after=False, recurrent_activation=hard_sigmoid, gate order z, r, h:
        z = hs(x K_z + b_z + h U_z); r = hs(x K_r + b_r + h U_r)
        hh = act(x K_h + b_h + (r*h) U_h); h' = z*h + (1-z)*hh."""
    u = state.shape[1]
    xk = inputs @ kernel + bias
    z = hard_sigmoid(xk[:, :u] + state @ recurrent_kernel[:, :u])
    r = hard_sigmoid(xk[:, u:2 * u] + state @ recurrent_kernel[:, u:2 * u])
    hh = apply_act(activation, xk[:, 2 * u:] + (r * state) @ recurrent_kernel[:, 2 * u:])
    return z * state + (np.asarray(1.0, state.dtype) - z) * hh


def mlp(x, weights, name, num_hidden, activation, use_biases=False):
    """utils/utils.py:77-126 with dropout rate 0: `num_hidden` Dense(activation) layers followed
    by one linear Dense; TF variable names <name>/dense[/_i]/kernel [TF-internal naming]."""
    names = ["dense" if i == 0 else "dense_%i" % i for i in range(num_hidden + 1)]
    h = x
    for i, n in enumerate(names):
        b = weights.get("%s/%s/bias" % (name, n)) if use_biases else None
        h = dense(h, weights["%s/%s/kernel" % (name, n)], b)
        if i < num_hidden:
            h = apply_act(activation, h)
    return h
