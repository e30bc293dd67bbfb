"""The second backend of tf_numpy_shim.py: the same TensorFlow 1.x / dpu_utils surface on float64 torch tensors, so that the
UNMODIFIED reference sources run under torch.autograd — gradients of the reference's own layer / model code, and its own
__make_train_step (compute_gradients -> per-variable clip_by_norm -> apply_gradients) executed end to end.
TEST INFRASTRUCTURE (used by make_reference_run.py only).

The primitive ops are oracle/torch_ref.py's (the torch mirrors of oracle/tf_ops.py, cross-checked against it in
tests/test_oracle_crosscheck_cpu.py); the optimizer update rules are oracle/optim.py's restatement of TF 1.13's kernels, here in
float64.  As with the NumPy backend: the reference decides WHAT is computed (which variables get gradients, which are clipped with
which norm, which optimizer with which arguments, how the learning rate is normalised), the shim only HOW each op evaluates.
Variable names, scopes and seeded initial values are shared with tf_numpy_shim (same registry), so a float32 NumPy run and a
float64 torch run of one model start from the same values.
"""
import math
import sys
import types

import numpy as np
import torch

import tf_numpy_shim as N
from oracle import torch_ref as R

F64 = torch.float64
SLOTS = {}             # optimizer state per variable name (persists across "session runs" = repeated __make_model calls)
LAST_STEP = {}         # what the last apply_gradients saw: gradients, clipped gradients, learning rate
TVARS = {}             # variable name -> torch leaf tensor (float64, requires_grad); values start as N.VARIABLES' float32 values


def reset(seed: int) -> None:
    N.reset(seed)
    SLOTS.clear()
    LAST_STEP.clear()
    TVARS.clear()


def new_graph() -> None:
    """A new 'session run' over the same variables: name uniquifiers start over (the model is rebuilt), values and slots stay."""
    N._scope.clear()
    N._unique.clear()
    N._keras_uid.clear()


class _T(torch.Tensor):
    """Immutable-tensor semantics for augmented assignment (see tf_numpy_shim._Tensor)."""

    def __iadd__(self, other):
        return torch.add(self, other)

    def __isub__(self, other):
        return torch.sub(self, other)

    def __imul__(self, other):
        return torch.mul(self, other)

    def __itruediv__(self, other):
        return torch.div(self, other)


def _t(x):
    return x.as_subclass(_T) if isinstance(x, torch.Tensor) else x


def _var(name: str, shape, kind: str):
    """The variable `name` in the current scope: created through the NumPy shim's registry (same names, same seeded values)."""
    arr = N._make(name, shape, kind)
    full = N._prefix() + name
    if full not in TVARS:
        if arr.dtype == np.int64:
            TVARS[full] = torch.as_tensor(arr)
        else:
            TVARS[full] = torch.tensor(arr, dtype=F64, requires_grad=True)
    return TVARS[full]


def _act(fn):
    return fn


class _Dense:
    def __init__(self, units, use_bias=True, activation=None, name=None, kernel_initializer=None, **unused):
        self.units, self.use_bias, self.activation = int(units), bool(use_bias), activation
        self.name = name if name is not None else self._default_name()
        self.scope = list(N._scope)
        self.kernel = self.bias = None

    @staticmethod
    def _default_name():
        return N._unique_name("dense")

    def __call__(self, x):
        if self.kernel is None:
            saved = list(N._scope)
            N._scope[:] = self.scope
            try:
                self.kernel = _var(self.name + "/kernel", (x.shape[-1], self.units), "kernel")
                self.bias = _var(self.name + "/bias", (self.units,), "bias") if self.use_bias else None
            finally:
                N._scope[:] = saved
        y = x @ self.kernel
        if self.bias is not None:
            y = y + self.bias
        return _t(y if self.activation is None else self.activation(y))


class _KerasDense(_Dense):
    @staticmethod
    def _default_name():
        return N._KerasDense._default_name()


class _Cell:
    def __init__(self, units, activation=None, **unused):
        self.units, self.activation = int(units), activation
        self.name = N._unique_name(self.scope_name)
        self.w = None

    def _weights(self, inputs):
        if self.w is None:
            g = self.gates
            self.w = (_var(self.name + "/kernel", (inputs.shape[-1], g * self.units), "kernel"),
                      _var(self.name + "/recurrent_kernel", (self.units, g * self.units), "kernel"),
                      _var(self.name + "/bias", (g * self.units,), "bias"))
        return self.w


class _GRUCell(_Cell):
    scope_name, gates = "gru_cell", 3

    def __call__(self, inputs, states):
        K, U, b = self._weights(inputs)
        cur = states[0]
        act = self.activation if self.activation is not None else (lambda v: v)
        u = cur.shape[1]
        xk = inputs @ K + b
        z = R.hard_sigmoid(xk[:, :u] + cur @ U[:, :u])
        r = R.hard_sigmoid(xk[:, u:2 * u] + cur @ U[:, u:2 * u])
        hh = act(xk[:, 2 * u:] + (r * cur) @ U[:, 2 * u:])
        out = _t(z * cur + (1.0 - z) * hh)
        return out, [out]


class _SimpleRNNCell(_Cell):
    scope_name, gates = "simple_rnn_cell", 1

    def __call__(self, inputs, states):
        K, U, b = self._weights(inputs)
        act = self.activation if self.activation is not None else (lambda v: v)
        out = _t(act(inputs @ K + b + states[0] @ U))
        return out, [out]


class _LSTMCell(_Cell):
    scope_name, gates = "lstm_cell", 4

    def __call__(self, inputs, states):
        raise NotImplementedError("LSTMCell needs two states; the reference passes one (gnns/ggnn.py:92)")


def _layer_norm(x, **unused):
    scope = N._unique_name("LayerNorm")
    beta = _var(scope + "/beta", (x.shape[-1],), "beta")
    gamma = _var(scope + "/gamma", (x.shape[-1],), "gamma")
    return _t(R.layer_norm(x, gamma, beta))


def _get_variable(name=None, shape=None, initializer=None, dtype=None, trainable=True, **unused):
    if initializer is N._zeros_initializer and dtype is np.int64:
        return _var(name, shape, "zeros_int64")
    return _var(name, shape, "kernel")


def _to_torch(v, dtype=None):
    if isinstance(v, torch.Tensor):
        return v
    a = np.asarray(v)
    if dtype is np.float32 or a.dtype.kind == "f":
        return torch.as_tensor(a.astype(np.float64))
    return torch.as_tensor(a.astype(np.int64))


def _placeholder(dtype=None, shape=None, name=None):
    if name in N.FEEDS:
        v = N.FEEDS[name]
        if isinstance(v, torch.Tensor):
            return _t(v)
        a = np.asarray(v, dtype=dtype)              # the Session's conversion to the placeholder dtype (float32), then float64
        t = _to_torch(a, dtype)
        return _t(t) if t.dim() else (float(t) if t.dtype.is_floating_point else int(t))
    return N._Placeholder(dtype, shape, name)


def _segment(kind):
    return lambda data=None, segment_ids=None, num_segments=None: _t(R.unsorted_segment(kind, data, segment_ids, int(num_segments)))


def _log_softmax(logits, segment_ids, num_segments):
    """dpu_utils.tfutils.unsorted_segment_log_softmax, as oracle/tf_ops.py states it."""
    n = int(num_segments)
    ids = segment_ids.long()
    mx = R.unsorted_segment("max", logits, ids, n)
    rec = logits - mx[ids]
    sums = R.unsorted_segment("sum", torch.exp(rec), ids, n)
    return rec - torch.log(sums)[ids]


def _cast(x, dtype=None, **unused):
    if isinstance(x, torch.Tensor):
        return x.to(F64) if dtype is np.float32 else x.to(torch.int64)
    return float(x) if dtype is np.float32 else int(x)


def _reduce(fn):
    def run(x, axis=None, **unused):
        if isinstance(x, (list, tuple)):
            x = torch.stack([v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=F64) for v in x])
        return fn(x) if axis is None else fn(x, dim=axis)
    return run


def _clip_by_norm(t, clip_norm, **unused):
    """tf.clip_by_norm as oracle/optim.py states it: (t * clip_norm) / max(||t||_2, clip_norm); zeros stay zeros."""
    l2sum = (t * t).sum()
    l2norm = torch.sqrt(l2sum) if float(l2sum) > 0 else l2sum
    return (t * clip_norm) / torch.clamp(l2norm, min=clip_norm)


class _Optimizer:
    """tf.train.*Optimizer: compute_gradients = torch.autograd.grad, apply_gradients = oracle/optim.py's update rules in float64,
    slots kept per variable NAME across rebuilt graphs."""

    def __init__(self, learning_rate=None, **kw):
        self.lr, self.kw = learning_rate, kw

    def compute_gradients(self, loss, var_list=None):
        grads = torch.autograd.grad(loss, var_list, allow_unused=True)
        LAST_STEP["loss"] = float(loss)
        LAST_STEP["raw_gradients"] = {_name_of(v): (None if g is None else g.detach().clone()) for g, v in zip(grads, var_list)}
        return list(zip(grads, var_list))

    def apply_gradients(self, grads_and_vars, **unused):
        lr = float(self.lr)
        LAST_STEP["learning_rate"] = lr
        LAST_STEP["optimizer"] = type(self).__name__
        LAST_STEP["applied_gradients"] = {}
        with torch.no_grad():
            t = SLOTS.get("__step__", 0) + 1
            SLOTS["__step__"] = t
            for g, var in grads_and_vars:
                name = _name_of(var)
                LAST_STEP["applied_gradients"][name] = None if g is None else g.detach().clone()
                if g is None:
                    continue
                self.update(name, var, g, lr, t)
        return "train_step"


class _Adam(_Optimizer):
    def update(self, name, var, g, lr, t):
        b1, b2, eps = self.kw.get("beta1", 0.9), self.kw.get("beta2", 0.999), self.kw.get("epsilon", 1e-8)
        m, v = SLOTS.setdefault(name, [torch.zeros_like(var), torch.zeros_like(var)])
        lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        m += (g - m) * (1.0 - b1)
        v += (g * g - v) * (1.0 - b2)
        var -= lr_t * m / (torch.sqrt(v) + eps)


class _RMSProp(_Optimizer):
    def update(self, name, var, g, lr, t):
        decay, momentum, eps = self.kw.get("decay", 0.9), self.kw.get("momentum", 0.0), self.kw.get("epsilon", 1e-10)
        ms, mom = SLOTS.setdefault(name, [torch.ones_like(var), torch.zeros_like(var)])
        ms += (g * g - ms) * (1.0 - decay)
        mom.copy_(mom * momentum + lr * g / torch.sqrt(ms + eps))
        var -= mom


class _SGD(_Optimizer):
    def update(self, name, var, g, lr, t):
        var -= lr * g


def _name_of(var):
    for n, v in TVARS.items():
        if v is var:
            return n
    raise KeyError("not a shim variable")


class _Var:
    def __init__(self, name, tensor):
        self.name, self._shape = name + ":0", tuple(tensor.shape)

    def get_shape(self):
        return [N._Dim(d) for d in self._shape]


def trainable():
    return [v for n, v in TVARS.items() if n not in N.NON_TRAINABLE]


def install() -> None:
    m = N._module
    sig = torch.sigmoid
    nn = m("tensorflow.nn", embedding_lookup=lambda params=None, ids=None, **kw: _t(params[ids.long()]),
           relu=torch.relu, leaky_relu=R.activation("leaky_relu"), elu=torch.nn.functional.elu, selu=torch.selu, sigmoid=sig,
           dropout=lambda x, rate=None, **kw: (_t(x) if float(rate) == 0.0 else (_ for _ in ()).throw(NotImplementedError("dropout"))),
           sigmoid_cross_entropy_with_logits=lambda labels=None, logits=None, **kw:
               torch.clamp(logits, min=0) - logits * labels.to(F64) + torch.log1p(torch.exp(-torch.abs(logits))))
    layers = m("tensorflow.layers", Dense=_Dense)
    keras_layers = m("tensorflow.keras.layers", Dense=_KerasDense, GRUCell=_GRUCell, SimpleRNNCell=_SimpleRNNCell, LSTMCell=_LSTMCell)
    keras = m("tensorflow.keras", layers=keras_layers)
    contrib = m("tensorflow.contrib", layers=m("tensorflow.contrib.layers", layer_norm=_layer_norm))
    initializers = m("tensorflow.initializers", truncated_normal=lambda **kw: ("truncated_normal", kw))
    summary = m("tensorflow.summary", scalar=lambda *a, **k: None, merge_all=lambda *a, **k: None, FileWriter=object)
    train = m("tensorflow.train", AdamOptimizer=_Adam, RMSPropOptimizer=_RMSProp, GradientDescentOptimizer=_SGD)
    tf = m(
        "tensorflow", Tensor=torch.Tensor, Variable=torch.Tensor, int32=np.int32, int64=np.int64, float32=np.float32, bool=np.bool_,
        nn=nn, layers=layers, keras=keras, contrib=contrib, initializers=initializers, summary=summary, train=train,
        GraphKeys=types.SimpleNamespace(TRAINABLE_VARIABLES="trainable_variables", GLOBAL_VARIABLES="variables"),
        concat=lambda values, axis=0, **kw: values if isinstance(values, torch.Tensor) else _t(torch.cat(list(values), dim=axis)),
        reshape=lambda tensor, shape, **kw: _t(tensor.reshape(tuple(int(s) for s in shape))),
        expand_dims=lambda x, axis=None, **kw: _t(x.unsqueeze(axis)), cast=_cast,
        shape=lambda x, out_type=None, **kw: [int(s) for s in x.shape], exp=torch.exp,
        sqrt=lambda x: torch.sqrt(x) if isinstance(x, torch.Tensor) else math.sqrt(x), tanh=torch.tanh, erf=torch.erf,
        round=torch.round, count_nonzero=lambda x, **kw: int(torch.count_nonzero(x)),
        einsum=lambda eq, *ops: _t(torch.einsum(eq, *ops)),
        unsorted_segment_sum=_segment("sum"), unsorted_segment_max=_segment("max"), unsorted_segment_mean=_segment("mean"),
        unsorted_segment_sqrt_n=_segment("sqrt_n"),
        variable_scope=N._variable_scope, get_variable=_get_variable, placeholder=_placeholder,
        placeholder_with_default=lambda default, shape=None, name=None: N.FEEDS.get(name, default),
        zeros_initializer=N._zeros_initializer, zeros_like=lambda x, **kw: _t(torch.zeros_like(x)),
        reduce_sum=_reduce(torch.sum), reduce_mean=_reduce(torch.mean), abs=torch.abs, square=torch.square,
        squeeze=lambda x, **kw: x.squeeze(), assign_add=lambda ref, value, **kw: ref + value,
        trainable_variables=lambda: [_Var(n, v) for n, v in TVARS.items() if n not in N.NON_TRAINABLE],
        clip_by_norm=_clip_by_norm, constant=lambda value, dtype=None, **kw: float(value) if dtype is np.float32 else value)
    for name, fn in (("unsorted_segment_sum", "sum"), ("unsorted_segment_max", "max"), ("unsorted_segment_mean", "mean"),
                     ("unsorted_segment_sqrt_n", "sqrt_n")):
        getattr(tf, name).__name__ = name
    sys.modules["tensorflow"] = tf
    for mod in (nn, layers, keras, keras_layers, contrib, initializers, summary, train):
        sys.modules[mod.__name__] = mod
    dpu_utils = m("dpu_utils.utils", RichPath=N.RichPath, LocalPath=N.RichPath, ThreadedIterator=lambda it, *a, **k: iter(it))
    tfutils = m("dpu_utils.tfutils", unsorted_segment_log_softmax=lambda logits=None, segment_ids=None, num_segments=None:
                _log_softmax(logits, segment_ids, num_segments))

    def _not_available(*a, **k):
        raise NotImplementedError("dpu_utils.codeutils is outside the path")
    codeutils = m("dpu_utils.codeutils", split_identifier_into_parts=_not_available, get_language_keywords=_not_available)
    sys.modules["dpu_utils"] = m("dpu_utils", utils=dpu_utils, tfutils=tfutils, codeutils=codeutils)
    sys.modules["dpu_utils.utils"] = dpu_utils
    sys.modules["dpu_utils.tfutils"] = tfutils
    sys.modules["dpu_utils.codeutils"] = codeutils
