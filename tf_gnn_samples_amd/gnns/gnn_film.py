"""sparse_gnn_film_layer — MI355X mirror of gnns/gnn_film.py:8-122.

    h'_v = LayerNorm( AGG_l AGG_{(u,v) in A_l}  sigma( gamma_{l,v} * (h_u W_l) + beta_{l,v} ) ),
    [gamma_{l,v} | beta_{l,v}] = h_v F_l                                  (gnn_film.py:22-29,102-112)

Node-side: ONE GEMM for all per-type messages  H @ [W_0|..|W_{L-1}]  and ONE for all FiLM weights
H @ [F_0|..|F_{L-1}] (the reference runs the first per EDGE and the second per node and type).
Edge-side: one fused HIP kernel (csrc/edge_fused.hip) for gather + scale + modulate + activation +
segment reduce; gamma/beta are loaded once per (target, type) bucket instead of once per edge.
"""
from typing import List, Mapping, Optional

import torch

from .. import _lib, ops
from ..dense import dense
from ..graph import as_rel_graph
from ..utils import apply_activation, get_activation, layer_norm, layer_norm_scope, layer_norm_variables
from ._common import concat_edge_kernels, require_weights


def gnn_film_layer_variables(num_edge_types: int, in_dim: int, state_dim: int, num_timesteps: int = 1):
    specs = {}
    for l in range(num_edge_types):
        specs["Edge_%i_Weight/kernel" % l] = ((in_dim, state_dim), "glorot_uniform")
        specs["Edge_%i_FiLM_Computations/kernel" % l] = ((in_dim, 2 * state_dim), "glorot_uniform")
    specs.update(layer_norm_variables(state_dim, num_timesteps))      # one LayerNorm scope per timestep
    return specs


def sparse_gnn_film_layer(node_embeddings: torch.Tensor,
                          adjacency_lists: List[torch.Tensor],
                          type_to_num_incoming_edges: torch.Tensor,
                          state_dim: Optional[int],
                          num_timesteps: int = 1,
                          activation_function: Optional[str] = "ReLU",
                          message_aggregation_function: str = "sum",
                          normalize_by_num_incoming: bool = False,
                          *,
                          weights: Mapping[str, torch.Tensor] = None,
                          ) -> torch.Tensor:
    """See gnns/gnn_film.py:17-57.  `weights`: "Edge_%i_Weight/kernel" [D, state_dim],
    "Edge_%i_FiLM_Computations/kernel" [D, 2*state_dim], "LayerNorm[_t]/{gamma,beta}" (one scope per timestep)."""
    weights = require_weights(weights, "sparse_gnn_film_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    mode = ops.aggregation_mode_id(message_aggregation_function)
    ops.activation_id(activation_function)
    w = graph.degree_scale(type_to_num_incoming_edges) if normalize_by_num_incoming else None
    # many-type graphs leave most (node,type) buckets empty: transform only the non-empty ones (graph.PairTables)
    # max aggregation and graphs with hub buckets (RelGraph.has_long_buckets) take the materialised route below: its reduction is
    # the gather-reduce kernel, which splits long buckets into chunked virtual rows (the fused kernel gives one lane group to a node)
    unfused = mode == _lib.AGG_MAX or graph.has_long_buckets
    pairs = graph.pair_tables() if (not unfused and graph.wants_pair_tables()) else None
    if pairs is None:
        w_msg = concat_edge_kernels(weights, L, "Edge_%i_Weight/kernel")               # [D, L*state_dim]
        w_film = concat_edge_kernels(weights, L, "Edge_%i_FiLM_Computations/kernel")   # [D, L*2*state_dim]
    cur_node_states = node_embeddings
    for t in range(num_timesteps):
        if pairs is not None:
            transformed, film = ops.typed_linear_pair(cur_node_states, pairs,
                                                      [weights["Edge_%i_Weight/kernel" % l] for l in range(L)],
                                                      [weights["Edge_%i_FiLM_Computations/kernel" % l] for l in range(L)])
            aggregated = ops.film_messages_reduce(transformed, film, graph, w, message_aggregation_function,
                                                  activation_function, pairs)
            cur_node_states = layer_norm(aggregated, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
            continue
        transformed = dense(cur_node_states, w_msg).view(num_nodes * L, state_dim)      # row v*L+l = h_v W_l
        film = dense(cur_node_states, w_film).view(num_nodes * L, 2 * state_dim)        # row v*L+l = [gamma | beta]
        if unfused:
            # (max backward needs the materialised messages for its tie handling; not a shipped configuration)
            msgs = transformed.index_select(0, graph.key_by_source.long())
            if w is not None:
                msgs = graph.w_original_order(w).unsqueeze(1) * msgs
            fw = film.index_select(0, graph.key_by_target.long())
            msgs = apply_activation(get_activation(activation_function),
                                    fw[:, :state_dim] * msgs + fw[:, state_dim:])
            aggregated = ops.seg_gather_reduce(msgs, graph.plan_messages(), message_aggregation_function, None)
        else:
            aggregated = ops.film_messages_reduce(transformed, film, graph, w, message_aggregation_function,
                                                  activation_function)
        cur_node_states = layer_norm(aggregated, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur_node_states
