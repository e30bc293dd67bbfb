"""Shared helpers for the sparse_*_layer mirrors."""
from typing import Mapping, Optional

import torch

from .. import _lib, ops
from ..utils import apply_activation, get_activation


def require_weights(weights, fn_name):
    if weights is None:
        raise ValueError(
            "%s needs `weights=` (mapping from TF variable names to tensors): the TF1 reference creates "
            "its variables as a side effect of the call, the PyTorch mirror takes them explicitly "
            "(see variables.VariableStore / the *_layer_variables helpers)." % fn_name)
    return weights


def concat_edge_kernels(weights: Mapping[str, torch.Tensor], num_edge_types: int, pattern: str, rows=None):
    """[D_in, L*D_out] column-concatenation of the per-edge-type Dense kernels, so that ONE node-side
    GEMM produces every type's transformed states; row v of the result viewed as [L, D_out]
    is (h_v W_0, ..., h_v W_{L-1})."""
    ks = [weights[pattern % l] for l in range(num_edge_types)]
    if rows is not None:
        ks = [k[rows] for k in ks]
    return torch.cat(ks, dim=1)


def reduce_and_activate(X, plan, aggregation: str, activation_function: Optional[str]):
    """seg_gather_reduce with the activation fused as the kernel epilogue when possible."""
    mode = ops.aggregation_mode_id(aggregation)
    act = ops.activation_id(activation_function)
    if act in ops._FUSABLE_ACTS and mode != _lib.AGG_MAX:
        return ops.seg_gather_reduce(X, plan, aggregation, activation_function)
    out = ops.seg_gather_reduce(X, plan, aggregation, None)
    return apply_activation(get_activation(activation_function), out)
