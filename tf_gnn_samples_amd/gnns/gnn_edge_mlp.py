"""sparse_gnn_edge_mlp_layer — MI355X mirror of gnns/gnn_edge_mlp.py:7-122.

    h'_v = LayerNorm( AGG_l AGG_{(u,v) in A_l}  sigma( 1/c_{l,v} * MLP_l([h_u || h_v]) ) )

MLP_l = `num_edge_hidden_layers` Dense(elu) layers + one linear Dense (elu is hard-wired,
gnn_edge_mlp.py:76).  With 0 hidden layers the whole message function is ONE fused kernel over
node-side P = H W[:D], Q = H W[D:]; with >= 1 hidden layers only the first layer is node-side, the
rest are per-edge GEMMs on contiguous per-type blocks (they cannot be hoisted past the elu).
"""
from typing import List, Mapping, Optional

import torch

from .. import ops
from ..dense import dense
from ..graph import as_rel_graph
from ..utils import MLP, layer_norm, layer_norm_scope, layer_norm_variables
from ._common import require_weights
from .pair import edge_mlp_messages, pair_messages_reduce


def gnn_edge_mlp_layer_variables(num_edge_types: int, in_dim: int, state_dim: int,
                                 use_target_state_as_input: bool = True, num_edge_hidden_layers: int = 1,
                                 num_timesteps: int = 1):
    specs = {}
    mlp_in = 2 * in_dim if use_target_state_as_input else in_dim
    for l in range(num_edge_types):
        specs.update(MLP.variable_shapes(mlp_in, state_dim, num_edge_hidden_layers, name="Edge_%i_MLP" % l))
    specs.update(layer_norm_variables(state_dim, num_timesteps))      # one LayerNorm scope per timestep
    return specs


def sparse_gnn_edge_mlp_layer(node_embeddings: torch.Tensor,
                              adjacency_lists: List[torch.Tensor],
                              type_to_num_incoming_edges: torch.Tensor,
                              state_dim: Optional[int],
                              num_timesteps: int = 1,
                              activation_function: Optional[str] = "ReLU",
                              message_aggregation_function: str = "sum",
                              normalize_by_num_incoming: bool = False,
                              use_target_state_as_input: bool = True,
                              num_edge_hidden_layers: int = 1,
                              *,
                              weights: Mapping[str, torch.Tensor] = None,
                              ) -> torch.Tensor:
    """See gnns/gnn_edge_mlp.py:19-62.  `weights`: "Edge_%i_MLP/dense[_j]/kernel", "LayerNorm[_t]/{gamma,beta}" (one scope per timestep)."""
    weights = require_weights(weights, "sparse_gnn_edge_mlp_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    ops.aggregation_mode_id(message_aggregation_function)
    ops.activation_id(activation_function)
    w = graph.degree_scale(type_to_num_incoming_edges) if normalize_by_num_incoming else None

    cur_node_states = node_embeddings
    for t in range(num_timesteps):
        d = cur_node_states.shape[1]
        if num_edge_hidden_layers == 0:
            k = [weights["Edge_%i_MLP/dense/kernel" % l] for l in range(L)]
            p = dense(cur_node_states, torch.cat([x[:d] for x in k], dim=1)).view(num_nodes * L, state_dim)
            if use_target_state_as_input:
                q = dense(cur_node_states, torch.cat([x[d:] for x in k], dim=1)).view(num_nodes * L, state_dim)
            else:
                q = torch.zeros_like(p)
            aggregated = pair_messages_reduce(p, q, graph, w, message_aggregation_function,
                                              message_activation=activation_function, output_activation=None)
        else:
            msgs = edge_mlp_messages(cur_node_states, graph, weights, "Edge_%i_MLP", num_edge_hidden_layers, "elu",
                                     use_target_state_as_input)                       # [M, state_dim], type-major
            # scale (:104-108) + activation (:112) are folded into the segment reduce (:113-116)
            aggregated = ops.message_act_reduce(msgs, graph, w, message_aggregation_function, activation_function)
        cur_node_states = layer_norm(aggregated, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur_node_states
