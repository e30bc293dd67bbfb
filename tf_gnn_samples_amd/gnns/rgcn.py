"""sparse_rgcn_layer — MI355X mirror of gnns/rgcn.py:8-117.

    h'_v = sigma( AGG_l AGG_{(u,v) in A_l}  1/(c_{l,v} + 1e-7) * (h_u W_l) )

Same signature, keyword names, defaults and message order as the reference.  What changes
is WHERE the work happens:
  * the per-edge Dense (rgcn.py:96-98, an [E_l, D] x [D, D] MatMul on gathered rows) becomes ONE
    node-side GEMM  H [V, D] @ [W_0 | ... | W_{L-1}]  — identical per-row dot products, 1/degree
    of the flops (library GEMM, dense.lib_gemm);
  * gather (rgcn.py:87-89), degree scale (:100-104), concat (:108), segment reduction
    (:109-112) and activation (:114) are ONE HIP kernel (csrc/seg_reduce.hip) over the
    (target, type)-bucketed CSR built once per batch.
"""
from typing import List, Mapping, Optional

import torch

import os

from .. import _lib, ops
from ..dense import dense
from ..graph import as_rel_graph
from ._common import concat_edge_kernels, reduce_and_activate, require_weights
from .pair import pair_messages_reduce


def aggregate_first_enabled() -> bool:
    """Default order for sum / mean / sqrt_n: gather + reduce the raw states into the (target, type) buckets, then ONE
    GEMM with the stacked kernels (ops.aggregate_then_transform).  RELGNN_RGCN_ORDER=transform_first restores
    transform-then-aggregate."""
    from ..config import settings
    return settings.rgcn_order != "transform_first"


def rgcn_layer_variables(num_edge_types: int, in_dim: int, state_dim: int,
                         use_both_source_and_target: bool = False):
    """TF variable names (relative to the layer's variable scope) -> (shape, initializer):
    one bias-free Dense kernel per edge type, rgcn.py:69-75."""
    fan_in = 2 * in_dim if use_both_source_and_target else in_dim
    return {"Edge_%i_Weight/kernel" % l: ((fan_in, state_dim), "glorot_uniform") for l in range(num_edge_types)}


def sparse_rgcn_layer(node_embeddings: torch.Tensor,
                      adjacency_lists: List[torch.Tensor],
                      type_to_num_incoming_edges: torch.Tensor,
                      state_dim: Optional[int],
                      num_timesteps: int = 1,
                      activation_function: Optional[str] = "tanh",
                      message_aggregation_function: str = "sum",
                      normalize_by_num_incoming: bool = True,
                      use_both_source_and_target: bool = False,
                      *,
                      weights: Mapping[str, torch.Tensor] = None,
                      ) -> torch.Tensor:
    """See gnns/rgcn.py:18-58 for the argument documentation (unchanged).  `weights` maps
    "Edge_%i_Weight/kernel" to the [D_in(*2), state_dim] kernels (TF layout)."""
    weights = require_weights(weights, "sparse_rgcn_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    w = graph.degree_scale(type_to_num_incoming_edges) if normalize_by_num_incoming else None

    cur_node_states = node_embeddings
    mode, act = ops.aggregation_mode_id(message_aggregation_function), ops.activation_id(activation_function)
    if not use_both_source_and_target and graph.wants_pair_tables():
        # many edge types, most (node,type) buckets empty: transform the non-empty ones only (graph.PairTables)
        pairs = graph.pair_tables()
        plan = pairs.plan_transformed(w)
        kernels = [weights["Edge_%i_Weight/kernel" % l] for l in range(L)]
        for _ in range(num_timesteps):
            transformed = ops.typed_linear(cur_node_states, pairs.src, kernels)          # [P_s, state_dim]
            cur_node_states = reduce_and_activate(transformed, plan, message_aggregation_function,
                                                  activation_function)
        return cur_node_states
    if (not use_both_source_and_target and aggregate_first_enabled() and mode != _lib.AGG_MAX and act in ops._FUSABLE_ACTS
            and in_dim % 4 == 0 and state_dim % 4 == 0):
        kernels = [weights["Edge_%i_Weight/kernel" % l] for l in range(L)]                       # L x [D, state_dim]
        for t in range(num_timesteps):
            # (the first timestep is this function's only read of node_embeddings: with the caller's word that nobody else reads
            #  them, the gradient of the activation that produced them rides in this layer's input-gradient product)
            cur_node_states = ops.aggregate_then_transform(cur_node_states, kernels, graph, w,
                                                           message_aggregation_function, activation_function, sole_reader=(t == 0))
        return cur_node_states
    if not use_both_source_and_target:
        plan = graph.plan_transformed(w)
        w_cat = concat_edge_kernels(weights, L, "Edge_%i_Weight/kernel")          # [D, L*state_dim]
        for _ in range(num_timesteps):
            transformed = dense(cur_node_states, w_cat).view(num_nodes * L, state_dim)  # row v*L+l = h_v W_l
            cur_node_states = reduce_and_activate(transformed, plan, message_aggregation_function,
                                                  activation_function)
        return cur_node_states

    # Dense([h_u || h_v]) = h_u W[:D] + h_v W[D:]  (rgcn.py:91-96): both halves node-side,
    # summed per message inside the pair kernel.
    for _ in range(num_timesteps):
        d = cur_node_states.shape[1]
        w_src = concat_edge_kernels(weights, L, "Edge_%i_Weight/kernel", rows=slice(0, d))
        w_tgt = concat_edge_kernels(weights, L, "Edge_%i_Weight/kernel", rows=slice(d, 2 * d))
        p = dense(cur_node_states, w_src).view(num_nodes * L, state_dim)
        q = dense(cur_node_states, w_tgt).view(num_nodes * L, state_dim)
        cur_node_states = pair_messages_reduce(p, q, graph, w, message_aggregation_function,
                                               message_activation=None,
                                               output_activation=activation_function)
    return cur_node_states
