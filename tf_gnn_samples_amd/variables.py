"""TF1-style named variables for the PyTorch mirror of the reference's layer functions.

In the reference every gnns.sparse_*_layer() call CREATES its trainable variables as a side
effect under the caller's tf.variable_scope (models/sparse_graph_model.py:177), and the
best-model pickle stores them as {variable.name: ndarray} (models/sparse_graph_model.py:91-107).
Here the variables live in a VariableStore (an nn.Module) under the same hierarchical names,
in TF layout (Dense kernels are [in, out]), so that
  * layer functions take `weights=store.scope("graph_model/gnn_layer_0")` (a read-only mapping
    from the TF-relative name, e.g. "Edge_0_Weight/kernel", to the parameter), and
  * a reference pickle's weight dict can be loaded by name (load_tf_weights).
"""
import math
from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch
from torch import nn


def _init_tensor(shape, init: str, generator: torch.Generator) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    if init == "zeros":
        return torch.zeros(shape)
    if init == "ones":
        return torch.ones(shape)
    if init == "glorot_uniform":
        # Keras / tf.get_variable default for these layers: U(-l, l), l = sqrt(6/(fan_in+fan_out))
        if len(shape) == 1:
            fan_in = fan_out = shape[0]
        else:
            fan_in, fan_out = shape[0], shape[1]
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=generator) * 2.0 - 1.0) * limit
    if init == "orthogonal":
        # Keras recurrent_initializer='orthogonal'
        rows, cols = shape
        a = torch.randn((max(rows, cols), min(rows, cols)), generator=generator)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.t()
        return q[:rows, :cols].contiguous()
    if init.startswith("trunc_normal:"):
        # tf.initializers.truncated_normal(stddev=s): N(0, s) re-drawn beyond 2 s
        std = float(init.split(":", 1)[1])
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=generator)
        return t
    raise ValueError("unknown initializer %r" % init)


class ScopedWeights(Mapping):
    """Read-only view of a VariableStore below a name prefix."""

    def __init__(self, store: "VariableStore", prefix: str):
        self._store, self._prefix = store, prefix.strip("/")

    def _full(self, name):
        return "%s/%s" % (self._prefix, name) if self._prefix else name

    def __getitem__(self, name):
        return self._store[self._full(name)]

    def __iter__(self):
        p = self._prefix + "/" if self._prefix else ""
        return (n[len(p):] for n in self._store.names() if n.startswith(p))

    def __len__(self):
        return sum(1 for _ in self)

    def scope(self, name):
        return ScopedWeights(self._store, self._full(name))


class VariableStore(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        self._params = nn.ParameterList()
        self._index: "OrderedDict[str, int]" = OrderedDict()
        self._generator = torch.Generator().manual_seed(int(seed))

    def create(self, name: str, shape, init: str = "glorot_uniform", trainable: bool = True) -> nn.Parameter:
        name = name.strip("/")
        if name in self._index:
            raise ValueError("variable %s already exists" % name)
        p = nn.Parameter(_init_tensor(shape, init, self._generator).to(torch.float32), requires_grad=trainable)
        self._index[name] = len(self._params)
        self._params.append(p)
        return p

    def create_all(self, prefix: str, specs: Mapping[str, Tuple[tuple, str]]):
        for rel, (shape, init) in specs.items():
            self.create("%s/%s" % (prefix.strip("/"), rel) if prefix else rel, shape, init)

    def __getitem__(self, name: str) -> nn.Parameter:
        try:
            return self._params[self._index[name.strip("/")]]
        except KeyError:
            raise KeyError("no variable named %r (have: %s ...)" % (name, list(self._index)[:8]))

    def __contains__(self, name):
        return name.strip("/") in self._index

    def names(self) -> Iterable[str]:
        return list(self._index.keys())

    def scope(self, prefix: str) -> ScopedWeights:
        return ScopedWeights(self, prefix)

    def num_parameters(self) -> int:
        return sum(int(p.numel()) for p in self._params if p.requires_grad)

    # ---- reference pickle format: {"<name>:0": ndarray} -----------------------------------
    def tf_weights(self) -> Dict[str, np.ndarray]:
        return {n + ":0": self[n].detach().cpu().numpy().copy() for n in self.names()}

    def load_tf_weights(self, weights: Mapping[str, np.ndarray], strict: bool = False, report_unused: bool = True):
        """Assign by variable name like Sparse_Graph_Model.load_weights
        (models/sparse_graph_model.py:109-126): unknown saved names are reported, missing ones
        keep their fresh initialisation.  Returns the set of consumed keys."""
        used = set()
        with torch.no_grad():
            for n in self.names():
                key = n + ":0" if (n + ":0") in weights else n
                if key in weights:
                    w = torch.as_tensor(np.asarray(weights[key]), dtype=torch.float32)
                    if tuple(w.shape) != tuple(self[n].shape):
                        raise ValueError("shape mismatch for %s: %s vs %s" % (n, tuple(w.shape), tuple(self[n].shape)))
                    self[n].copy_(w)
                    used.add(key)
                elif strict:
                    raise KeyError("no saved value for %s" % n)
                else:
                    print('Freshly initializing %s since no saved value was found.' % n)
        if report_unused:
            for k in weights:
                if k not in used:
                    print('Saved weights for %s not used by model.' % k)
        return used
