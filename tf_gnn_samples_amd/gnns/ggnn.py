"""sparse_ggnn_layer — MI355X mirror of gnns/ggnn.py:8-95.

    m_v = AGG_l AGG_{(u,v) in A_l} (h_u W_l)          (no degree scaling, ggnn.py:76-89)
    h'_v = Cell(inputs=m_v, state=h_v)                 (ggnn.py:92; utils/utils.py:10-20)
"""
from typing import List, Mapping, Optional

import torch

from .. import ops
from ..dense import dense_multi
from ..graph import as_rel_graph
from ..utils import gated_unit_variable_shapes, get_gated_unit
from ._common import require_weights


def ggnn_layer_variables(num_edge_types: int, in_dim: int, state_dim: int, gated_unit_type: str = "gru"):
    specs = {"Edge_%i_Weight/kernel" % l: ((in_dim, state_dim), "glorot_uniform") for l in range(num_edge_types)}
    specs.update(gated_unit_variable_shapes(state_dim, state_dim, gated_unit_type))
    return specs


def sparse_ggnn_layer(node_embeddings: torch.Tensor,
                      adjacency_lists: List[torch.Tensor],
                      state_dim: Optional[int],
                      num_timesteps: int = 1,
                      gated_unit_type: str = "gru",
                      activation_function: str = "tanh",
                      message_aggregation_function: str = "sum",
                      *,
                      weights: Mapping[str, torch.Tensor] = None,
                      ) -> torch.Tensor:
    """See gnns/ggnn.py:16-49.  `weights`: "Edge_%i_Weight/kernel" plus the cell's
    "<cell>/kernel", "<cell>/recurrent_kernel", "<cell>/bias" (cell scope: gru_cell / simple_rnn_cell)."""
    from ..utils import GATED_UNIT_SCOPES
    weights = require_weights(weights, "sparse_ggnn_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    ops.aggregation_mode_id(message_aggregation_function)  # ValueError for unknown names, like the reference
    cell_scope = GATED_UNIT_SCOPES.get(gated_unit_type.lower())
    cell_weights = {} if cell_scope is None else {
        k: weights["%s/%s" % (cell_scope, k)] for k in ("kernel", "recurrent_kernel", "bias")
        if gated_unit_type.lower() != 'lstm'}
    gated_cell = get_gated_unit(state_dim, gated_unit_type, activation_function, cell_weights)

    pairs = graph.pair_tables() if graph.wants_pair_tables() else None     # many-type graphs: non-empty buckets only
    if pairs is not None:
        plan = pairs.plan_transformed(None)
        kernels = [weights["Edge_%i_Weight/kernel" % l] for l in range(L)]
    else:
        plan = graph.plan_transformed(None)
        kernels = [weights["Edge_%i_Weight/kernel" % l] for l in range(L)]
    cur_node_states = node_embeddings
    for _ in range(num_timesteps):
        if pairs is not None:
            transformed = ops.typed_linear(cur_node_states, pairs.src, kernels)
        else:
            transformed = dense_multi(cur_node_states, kernels).view(num_nodes * L, state_dim)   # row v*L + l = h_v W_l
        aggregated_messages = ops.seg_gather_reduce(transformed, plan, message_aggregation_function, None)
        cur_node_states = gated_cell(aggregated_messages, [cur_node_states])[0]
    return cur_node_states
