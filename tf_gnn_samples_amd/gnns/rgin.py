"""sparse_rgin_layer — MI355X mirror of gnns/rgin.py:7-142.

    h'_v = LayerNorm( sigma( MLP_aggr( AGG_l AGG_{(u,v) in A_l} sigma(MLP_l(h_u)) ) ) )

Default configuration (edge MLP on the SOURCE state only): MLP_l(h_u) depends on the source node
alone, so it runs node-side — V*L rows instead of M rows, identical per-row arithmetic — and the
edge-side work is the fused gather + segment-reduce kernel.  With use_target_state_as_input the
first layer splits into source/target halves (pair.py); without any edge MLP the raw states are
gathered.
"""
from typing import List, Mapping, Optional

import torch

from .. import ops
from ..dense import dense
from ..graph import as_rel_graph
from ..utils import MLP, apply_activation, get_activation, layer_norm, layer_norm_scope, layer_norm_variables
from ._common import require_weights
from .pair import edge_mlp_messages, pair_messages_reduce


def rgin_layer_variables(num_edge_types: int, in_dim: int, state_dim: int, use_target_state_as_input: bool = False,
                         num_edge_MLP_hidden_layers: Optional[int] = 1, num_aggr_MLP_hidden_layers: Optional[int] = None,
                         num_timesteps: int = 1):
    specs = {}
    mlp_in = 2 * in_dim if use_target_state_as_input else in_dim
    if num_aggr_MLP_hidden_layers is not None:
        agg_in = state_dim if num_edge_MLP_hidden_layers is not None else mlp_in
        specs.update(MLP.variable_shapes(agg_in, state_dim, num_aggr_MLP_hidden_layers, name="Aggregation_MLP"))
    if num_edge_MLP_hidden_layers is not None:
        for l in range(num_edge_types):
            specs.update(MLP.variable_shapes(mlp_in, state_dim, num_edge_MLP_hidden_layers, name="Edge_%i_MLP" % l))
    specs.update(layer_norm_variables(state_dim, num_timesteps))      # one LayerNorm scope per timestep
    return specs


def sparse_rgin_layer(node_embeddings: torch.Tensor,
                      adjacency_lists: List[torch.Tensor],
                      state_dim: Optional[int],
                      num_timesteps: int = 1,
                      activation_function: Optional[str] = "ReLU",
                      message_aggregation_function: str = "sum",
                      use_target_state_as_input: bool = False,
                      num_edge_MLP_hidden_layers: Optional[int] = 1,
                      num_aggr_MLP_hidden_layers: Optional[int] = None,
                      *,
                      weights: Mapping[str, torch.Tensor] = None,
                      ) -> torch.Tensor:
    """See gnns/rgin.py:18-68.  `weights`: "Edge_%i_MLP/dense[_j]/kernel", "Aggregation_MLP/dense[_j]/kernel",
    "LayerNorm[_t]/{gamma,beta}" (one scope per timestep)."""
    weights = require_weights(weights, "sparse_rgin_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    ops.aggregation_mode_id(message_aggregation_function)
    activation_fn = get_activation(activation_function)
    aggregation_MLP = None
    if num_aggr_MLP_hidden_layers is not None:
        aggregation_MLP = MLP(out_size=state_dim, hidden_layers=num_aggr_MLP_hidden_layers,
                              activation_fun=activation_fn, name="Aggregation_MLP", weights=weights)

    cur_node_states = node_embeddings
    for t in range(num_timesteps):
        d = cur_node_states.shape[1]
        if num_edge_MLP_hidden_layers is None:
            # messages are the raw (concatenated) states, no message activation (rgin.py:125-129)
            agg_src = ops.seg_gather_reduce(cur_node_states, graph.plan_untransformed(None), message_aggregation_function, None)
            if use_target_state_as_input:
                agg_tgt = ops.seg_gather_reduce(cur_node_states, graph.plan_target_rows(), message_aggregation_function, None)
                aggregated = torch.cat([agg_src, agg_tgt], dim=1)
            else:
                aggregated = agg_src
        elif not use_target_state_as_input:
            # MLP_l(h_u) on NODES: [V, L, state_dim], then sigma, then gather + reduce
            per_type = []
            for l in range(L):
                mlp_l = MLP(out_size=state_dim, hidden_layers=num_edge_MLP_hidden_layers, activation_fun=activation_fn,
                            name="Edge_%i_MLP" % l, weights=weights)
                per_type.append(mlp_l(cur_node_states))
            transformed = apply_activation(activation_fn, torch.stack(per_type, dim=1)).view(num_nodes * L, state_dim)
            aggregated = ops.seg_gather_reduce(transformed, graph.plan_transformed(None), message_aggregation_function, None)
        elif num_edge_MLP_hidden_layers == 0:
            k = [weights["Edge_%i_MLP/dense/kernel" % l] for l in range(L)]
            p = dense(cur_node_states, torch.cat([x[:d] for x in k], dim=1)).view(num_nodes * L, state_dim)
            q = dense(cur_node_states, torch.cat([x[d:] for x in k], dim=1)).view(num_nodes * L, state_dim)
            aggregated = pair_messages_reduce(p, q, graph, None, message_aggregation_function,
                                              message_activation=activation_function, output_activation=None)
        else:
            msgs = edge_mlp_messages(cur_node_states, graph, weights, "Edge_%i_MLP", num_edge_MLP_hidden_layers,
                                     activation_function, True)
            aggregated = ops.message_act_reduce(msgs, graph, None, message_aggregation_function, activation_function)

        new_node_states = aggregated
        if aggregation_MLP is not None:
            new_node_states = aggregation_MLP(new_node_states)
        new_node_states = apply_activation(activation_fn, new_node_states)
        cur_node_states = layer_norm(new_node_states, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur_node_states
