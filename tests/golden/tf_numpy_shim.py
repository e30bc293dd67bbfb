"""A NumPy stand-in for the TensorFlow 1.x / dpu_utils symbols that the reference's OWN Python code touches, so that the
UNMODIFIED sources under /root/reference (gnns/*.py, utils/utils.py, tasks/ppi_task.py, tasks/qm9_task.py) can be imported and
executed in the build container, where TensorFlow cannot be installed.  TEST INFRASTRUCTURE (used by make_reference_run.py only).

What running the reference over this shim pins, and what it does not:
  * pinned: the reference's COMPOSITION — which ops, in which order, on which operands, with which constants, shapes, variable
    names and concat / segment orders — i.e. everything oracle/gnns.py and oracle/bookkeeping.py restate by hand.  The fixtures
    hold the outputs of the reference's code, not of a transcription of it.
  * not pinned: the semantics of the individual TensorFlow ops.  Every `tf.*` symbol below is a thin adapter over the NumPy
    restatement in oracle/tf_ops.py (each documented there with its [TF-internal] assumptions and cross-checked against PyTorch's
    independent CPU kernels in tests/test_oracle_crosscheck_cpu.py).  A real TF 1.13 run (scripts/dump_tf_golden.py) is still what
    would close that part; DESIGN.md section 6 says so.

Tensors are eager NumPy arrays; variables are created on first use with seeded values and recorded by their TF variable name
(`VARIABLES`, creation order kept), the names following the rules TF 1.x applies (variable_scope prefixes, Keras / tf.layers
default names dense, dense_1, ...; LayerNorm, LayerNorm_1, ... per scope) as far as the reference exercises them.
"""
import contextlib
import sys
import types
from collections import OrderedDict

import numpy as np

from oracle import tf_ops as O

VARIABLES = OrderedDict()          # TF variable name -> float32 array, in creation order
_scope = []                        # tf.variable_scope stack
_unique = {}                       # (scope prefix, base name) -> how many handed out
_rng = [np.random.default_rng(0)]
PLACEHOLDERS = []


def reset(seed: int) -> None:
    VARIABLES.clear()
    _scope.clear()
    _unique.clear()
    _rng[0] = np.random.default_rng(seed)


def _prefix() -> str:
    return "".join(s + "/" for s in _scope)


def _unique_name(base: str) -> str:
    key = (_prefix(), base)
    n = _unique.get(key, 0)
    _unique[key] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


def _make(name: str, shape, kind: str) -> np.ndarray:
    full = _prefix() + name
    if full in VARIABLES:
        return VARIABLES[full]
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    r = _rng[0]
    if kind == "gamma":
        v = 1.0 + 0.1 * r.standard_normal(shape)
    elif kind in ("beta", "bias"):
        v = 0.1 * r.standard_normal(shape)
    else:                                   # kernels / free parameters: O(1) pre-activations for O(1) inputs
        fan_in = shape[0] if len(shape) > 1 else max(shape[0], 1)
        v = r.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
    VARIABLES[full] = v.astype(np.float32)
    return VARIABLES[full]


class _Dense:
    """tf.keras.layers.Dense / tf.layers.Dense: variables <name>/kernel [in, units] (+ <name>/bias), created at the first call."""

    def __init__(self, units, use_bias=True, activation=None, name=None, kernel_initializer=None, **unused):
        self.units, self.use_bias, self.activation = int(units), bool(use_bias), activation
        self.name = name if name is not None else _unique_name("dense")
        self.scope = list(_scope)            # (tf.layers capture the variable scope they were made in; the reference's MLP
        self.kernel = self.bias = None       #  re-enters the same scope at call time, so either rule gives the same names)

    def __call__(self, x):
        if self.kernel is None:
            saved = list(_scope)
            _scope[:] = self.scope
            try:
                self.kernel = _make(self.name + "/kernel", (x.shape[-1], self.units), "kernel")
                self.bias = _make(self.name + "/bias", (self.units,), "bias") if self.use_bias else None
            finally:
                _scope[:] = saved
        return O.dense(x, self.kernel, self.bias, self.activation)


class _Cell:
    def __init__(self, units, activation=None, **unused):
        self.units, self.activation = int(units), activation
        self.name = _unique_name(self.scope_name)
        self.w = None

    def _weights(self, inputs):
        if self.w is None:
            g = self.gates
            self.w = (_make(self.name + "/kernel", (inputs.shape[-1], g * self.units), "kernel"),
                      _make(self.name + "/recurrent_kernel", (self.units, g * self.units), "kernel"),
                      _make(self.name + "/bias", (g * self.units,), "bias"))
        return self.w


class _GRUCell(_Cell):
    scope_name, gates = "gru_cell", 3

    def __call__(self, inputs, states):
        k, u, b = self._weights(inputs)
        out = O.gru_cell(inputs, states[0], k, u, b, self.activation)
        return out, [out]


class _SimpleRNNCell(_Cell):
    scope_name, gates = "simple_rnn_cell", 1

    def __call__(self, inputs, states):
        k, u, b = self._weights(inputs)
        out = O.simple_rnn_cell(inputs, states[0], k, u, b, self.activation)
        return out, [out]


class _LSTMCell(_Cell):
    scope_name, gates = "lstm_cell", 4

    def __call__(self, inputs, states):
        raise NotImplementedError("LSTMCell needs two states; the reference passes one (gnns/ggnn.py:92)")


def _layer_norm(x, **unused):
    scope = _unique_name("LayerNorm")
    beta = _make(scope + "/beta", (x.shape[-1],), "beta")        # (tf.contrib creates beta before gamma)
    gamma = _make(scope + "/gamma", (x.shape[-1],), "gamma")
    return O.layer_norm(x, gamma, beta)


@contextlib.contextmanager
def _variable_scope(name, *unused, **unused_kw):
    _scope.append(name)
    try:
        yield
    finally:
        _scope.pop()


def _get_variable(name=None, shape=None, initializer=None, **unused):
    return _make(name, shape, "kernel")


def _dropout(x, rate=None, keep_prob=None, **unused):
    r = rate if rate is not None else (None if keep_prob is None else 1.0 - keep_prob)
    if r is None or float(r) != 0.0:
        raise NotImplementedError("the shim runs the reference without dropout (rate must be 0)")
    return x


def _shape(x, out_type=None, **unused):
    return np.asarray(np.shape(x), dtype=out_type if out_type is not None else np.int32)


def _count_nonzero(x, **unused):
    return np.int64(np.count_nonzero(x))


def _concat(values, axis=0, **unused):
    """tf.concat; a single TENSOR instead of a list is returned as it is (array_ops.concat wraps a non-list into [values] and
    a one-element concat is an identity): gnns/rgat.py:126 relies on it [TF-internal]."""
    if isinstance(values, np.ndarray):
        return values
    return np.concatenate([np.asarray(v) for v in values], axis=axis)


class _Placeholder:
    """tf.placeholder: a hashable token (the tasks use placeholders as feed-dict keys only)."""

    def __init__(self, dtype=None, shape=None, name=None):
        self.dtype, self.shape, self.name = dtype, shape, name
        PLACEHOLDERS.append(self)

    def __repr__(self):
        return "<placeholder %s>" % self.name


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class RichPath:
    """The part of dpu_utils.utils.RichPath (dpu_utils 0.1.x, a dependency in the reference's requirements.txt that is absent
    here) the tasks use: join + read_by_file_suffix for .json / .npy / .jsonl.gz (decoded as dpu_utils does: json / np.load /
    one JSON document per line, in file order) and a few path predicates."""

    def __init__(self, path):
        self.path = str(path)

    @staticmethod
    def create(path, *unused):
        return RichPath(path)

    def join(self, name):
        import os
        return RichPath(os.path.join(self.path, name))

    def exists(self):
        import os
        return os.path.exists(self.path)

    def is_dir(self):
        import os
        return os.path.isdir(self.path)

    def __str__(self):
        return self.path

    __repr__ = __str__

    def read_by_file_suffix(self):
        import gzip
        import json
        p = self.path
        if p.endswith(".npy"):
            return np.load(p)
        if p.endswith(".jsonl.gz"):
            with gzip.open(p, "rt") as f:
                return [json.loads(line, object_pairs_hook=OrderedDict) for line in f if line.strip()]
        if p.endswith(".json"):
            with open(p) as f:
                return json.load(f, object_pairs_hook=OrderedDict)
        raise ValueError("RichPath shim: unsupported suffix of %r" % p)

    def read_as_jsonl(self):
        return self.read_by_file_suffix()


def install() -> None:
    """Put `tensorflow`, `dpu_utils`, `dpu_utils.utils`, `dpu_utils.tfutils` (and `docopt`-free nothing else) into sys.modules."""
    nn = _module("tensorflow.nn", embedding_lookup=lambda params, ids, **kw: O.embedding_lookup(params, ids),
                 relu=O.relu, leaky_relu=O.leaky_relu, elu=O.elu, selu=O.selu, dropout=_dropout,
                 sigmoid=lambda x: (np.asarray(1.0, x.dtype) / (np.asarray(1.0, x.dtype) + np.exp(-x))),
                 sigmoid_cross_entropy_with_logits=None)
    layers = _module("tensorflow.layers", Dense=_Dense)
    keras_layers = _module("tensorflow.keras.layers", Dense=_Dense, GRUCell=_GRUCell, SimpleRNNCell=_SimpleRNNCell,
                           LSTMCell=_LSTMCell)
    keras = _module("tensorflow.keras", layers=keras_layers)
    contrib = _module("tensorflow.contrib", layers=_module("tensorflow.contrib.layers", layer_norm=_layer_norm))
    initializers = _module("tensorflow.initializers", truncated_normal=lambda **kw: ("truncated_normal", kw))
    summary = _module("tensorflow.summary", scalar=lambda *a, **k: None)
    tf = _module(
        "tensorflow", Tensor=np.ndarray, Variable=np.ndarray, int32=np.int32, int64=np.int64, float32=np.float32, bool=np.bool_,
        nn=nn, layers=layers, keras=keras, contrib=contrib, initializers=initializers, summary=summary,
        concat=_concat,
        reshape=lambda tensor, shape, **kw: np.reshape(tensor, tuple(int(s) for s in shape)),
        expand_dims=lambda x, axis=None, **kw: np.expand_dims(x, axis),
        cast=lambda x, dtype=None, **kw: np.asarray(x).astype(dtype),
        shape=_shape, exp=np.exp, sqrt=lambda x: np.sqrt(np.asarray(x, dtype=np.float32)), tanh=O.tanh, erf=O._erf,
        round=lambda x: np.round(x),                       # (both round half to even)
        count_nonzero=_count_nonzero, einsum=lambda eq, *ops: np.einsum(eq, *ops),
        unsorted_segment_sum=O.unsorted_segment_sum, unsorted_segment_max=O.unsorted_segment_max,
        unsorted_segment_mean=O.unsorted_segment_mean, unsorted_segment_sqrt_n=O.unsorted_segment_sqrt_n,
        variable_scope=_variable_scope, get_variable=_get_variable, placeholder=_Placeholder)
    sys.modules["tensorflow"] = tf
    for m in (nn, layers, keras, keras_layers, contrib, initializers, summary):
        sys.modules[m.__name__] = m
    dpu_utils = _module("dpu_utils.utils", RichPath=RichPath, LocalPath=RichPath)
    tfutils = _module("dpu_utils.tfutils", unsorted_segment_log_softmax=O.unsorted_segment_log_softmax)

    def _not_available(*a, **k):
        raise NotImplementedError("dpu_utils.codeutils is outside the path (VarMisuse data, SURVEY 2a)")
    # (tasks/__init__.py imports every task module; the VarMisuse and citation tasks are only IMPORTED, never run)
    codeutils = _module("dpu_utils.codeutils", split_identifier_into_parts=_not_available, get_language_keywords=_not_available)
    sys.modules["dpu_utils"] = _module("dpu_utils", utils=dpu_utils, tfutils=tfutils, codeutils=codeutils)
    sys.modules["dpu_utils.utils"] = dpu_utils
    sys.modules["dpu_utils.tfutils"] = tfutils
    sys.modules["dpu_utils.codeutils"] = codeutils
