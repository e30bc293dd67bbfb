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

from oracle import model as M
from oracle import tf_ops as O

VARIABLES = OrderedDict()          # TF variable name -> float32 array, in creation order
_scope = []                        # tf.variable_scope stack
_unique = {}                       # (scope prefix, base name) -> how many handed out
_rng = [np.random.default_rng(0)]
PLACEHOLDERS = []
FEEDS = {}                         # placeholder name -> value: an eager "session" — placeholders evaluate to what is fed
NON_TRAINABLE = set()
_keras_uid = {}                    # Keras layer base name -> how many handed out (graph-wide, whatever the scope)


def reset(seed: int) -> None:
    VARIABLES.clear()
    _scope.clear()
    _unique.clear()
    _keras_uid.clear()
    NON_TRAINABLE.clear()
    FEEDS.clear()
    _rng[0] = np.random.default_rng(seed)


class _Tensor(np.ndarray):
    """tf.Tensor is immutable: `x += y` REBINDS x.  models/sparse_graph_model.py:181-185 keeps `t = cur` and then does `cur += ...;
    cur /= 2` — on a plain ndarray that would also change t."""

    def __iadd__(self, other):
        return np.add(self, other)

    def __isub__(self, other):
        return np.subtract(self, other)

    def __imul__(self, other):
        return np.multiply(self, other)

    def __itruediv__(self, other):
        return np.true_divide(self, np.asarray(other, dtype=self.dtype) if np.isscalar(other) else other)


def _t(x):
    return np.asarray(x).view(_Tensor)


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
    if kind == "zeros_int64":
        VARIABLES[full] = np.zeros(shape, np.int64)
        NON_TRAINABLE.add(full)
        return VARIABLES[full]
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
        self.name = name if name is not None else self._default_name()
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
        return _t(O.dense(np.asarray(x), self.kernel, self.bias, self.activation))

    @staticmethod
    def _default_name():
        # tf.layers.Dense: its variables live under variable_scope(default_name="dense"), uniquified inside the ENCLOSING variable scope
        # (utils/utils.py MLP: <name>/dense, <name>/dense_1, ... in every MLP's own scope) [TF-internal]
        return _unique_name("dense")


class _KerasDense(_Dense):
    @staticmethod
    def _default_name():
        # tf.keras.layers.Dense: the layer name is uniquified graph-wide (dense, dense_1, ...) whatever scope it is made in; the
        # variables then sit under the name scopes open at build time (models/sparse_graph_model.py:166 inside "graph_model",
        # tasks/ppi_task.py:176 outside: graph_model/dense/kernel and dense_1/kernel) [TF-internal]
        n = _keras_uid.get("dense", 0)
        _keras_uid["dense"] = n + 1
        return "dense" if n == 0 else "dense_%d" % n


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
    return _t(O.layer_norm(np.asarray(x), gamma, beta))


@contextlib.contextmanager
def _variable_scope(name, *unused, **unused_kw):
    _scope.append(name)
    try:
        yield
    finally:
        _scope.pop()


def _zeros_initializer(*a, **k):
    return "zeros"


def _get_variable(name=None, shape=None, initializer=None, dtype=None, trainable=True, **unused):
    if initializer is _zeros_initializer and dtype is np.int64:
        return _make(name, shape, "zeros_int64")
    return _make(name, shape, "kernel")


class _Dim:
    def __init__(self, v):
        self.value = int(v)


class _Var:
    """What tf.trainable_variables() hands to models/sparse_graph_model.py:155-156 (name, get_shape() -> dims with .value)."""

    def __init__(self, name, array):
        self.name, self._shape = name + ":0", array.shape

    def get_shape(self):
        return [_Dim(d) for d in self._shape]


class _GlobalVar:
    """An entry of the GLOBAL_VARIABLES collection as save_model uses it (sparse_graph_model.py:91-97): .name with TF's ':0'."""

    def __init__(self, name, array):
        self.name, self.value = name + ":0", array


def session_stub():
    """self.sess for save_model: graph.get_collection(GLOBAL_VARIABLES) -> the shim's variables (the model's and the non-trainable
    total_num_graphs; optimizer slots do not exist here: their TF names are TF's own), run(dict) -> the values."""
    graph = types.SimpleNamespace(get_collection=lambda which: [_GlobalVar(n, v) for n, v in VARIABLES.items()])
    return types.SimpleNamespace(graph=graph, run=lambda fetches, **kw: {k: np.array(v.value) for k, v in fetches.items()})


def _trainable_variables():
    return [_Var(n, v) for n, v in VARIABLES.items() if n not in NON_TRAINABLE]


def _dropout(x, rate=None, keep_prob=None, **unused):
    r = rate if rate is not None else (None if keep_prob is None else 1.0 - keep_prob)
    if r is None or float(r) != 0.0:
        raise NotImplementedError("the shim runs the reference without dropout (rate must be 0)")
    return _t(x)


def _reduce_sum(x, axis=None, **unused):
    if isinstance(x, (list, tuple)):                       # tf.reduce_sum(list of scalars): packed first (tasks/qm9_task.py:196)
        x = np.stack([np.asarray(v) for v in x])
    x = np.asarray(x)
    return np.sum(x, axis=axis, dtype=x.dtype)


def _reduce_mean(x, axis=None, **unused):
    x = np.asarray(x)
    return np.mean(x, axis=axis, dtype=x.dtype)


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


def _placeholder(dtype=None, shape=None, name=None):
    """Eager session: a placeholder whose name is in FEEDS IS the fed value, converted to the placeholder's dtype as Session.run
    converts a feed (QM9's float64 features into a float32 placeholder); otherwise a token."""
    if name in FEEDS:
        v = np.asarray(FEEDS[name], dtype=dtype)
        return _t(v) if v.ndim else v[()]
    return _Placeholder(dtype, shape, name)


def _placeholder_with_default(default, shape=None, name=None):
    return FEEDS.get(name, default)


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
                 sigmoid_cross_entropy_with_logits=lambda labels=None, logits=None, **kw: M.sigmoid_cross_entropy_with_logits(
                     np.asarray(logits), np.asarray(labels, dtype=np.asarray(logits).dtype)))
    layers = _module("tensorflow.layers", Dense=_Dense)
    keras_layers = _module("tensorflow.keras.layers", Dense=_KerasDense, GRUCell=_GRUCell, SimpleRNNCell=_SimpleRNNCell,
                           LSTMCell=_LSTMCell)
    keras = _module("tensorflow.keras", layers=keras_layers)
    contrib = _module("tensorflow.contrib", layers=_module("tensorflow.contrib.layers", layer_norm=_layer_norm))
    initializers = _module("tensorflow.initializers", truncated_normal=lambda **kw: ("truncated_normal", kw))
    summary = _module("tensorflow.summary", scalar=lambda *a, **k: None, merge_all=lambda *a, **k: None, FileWriter=object)
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
        variable_scope=_variable_scope, get_variable=_get_variable, placeholder=_placeholder,
        placeholder_with_default=_placeholder_with_default, zeros_initializer=_zeros_initializer,
        zeros_like=lambda x, **kw: _t(np.zeros_like(np.asarray(x))), reduce_sum=_reduce_sum, reduce_mean=_reduce_mean,
        abs=np.abs, square=np.square, squeeze=lambda x, **kw: np.squeeze(np.asarray(x)),
        assign_add=lambda ref, value, **kw: ref + value, trainable_variables=_trainable_variables,
        GraphKeys=types.SimpleNamespace(TRAINABLE_VARIABLES="trainable_variables", GLOBAL_VARIABLES="variables"))
    sys.modules["tensorflow"] = tf
    for m in (nn, layers, keras, keras_layers, contrib, initializers, summary):
        sys.modules[m.__name__] = m
    dpu_utils = _module("dpu_utils.utils", RichPath=RichPath, LocalPath=RichPath,
                        ThreadedIterator=lambda it, *a, **k: iter(it))          # (imported by models/sparse_graph_model.py; never used here)
    tfutils = _module("dpu_utils.tfutils", unsorted_segment_log_softmax=O.unsorted_segment_log_softmax)

    def _not_available(*a, **k):
        raise NotImplementedError("dpu_utils.codeutils is outside the path (VarMisuse data, SURVEY 2a)")
    # (tasks/__init__.py imports every task module; the VarMisuse and citation tasks are only IMPORTED, never run)
    codeutils = _module("dpu_utils.codeutils", split_identifier_into_parts=_not_available, get_language_keywords=_not_available)
    sys.modules["dpu_utils"] = _module("dpu_utils", utils=dpu_utils, tfutils=tfutils, codeutils=codeutils)
    sys.modules["dpu_utils.utils"] = dpu_utils
    sys.modules["dpu_utils.tfutils"] = tfutils
    sys.modules["dpu_utils.codeutils"] = codeutils
