"""sparse_rgdcn_layer — MI355X mirror of gnns/rgdcn.py:8-167 (dynamic per-target convolution kernels).

    h'_{v,c,:} = sigma( AGG_l AGG_{(u,v) in A_l}  1/c_{l,v} * ( h_{u,c,:} @ W_{l,v,c} ) ),
    W_{l,v,c} = reshape( sigma( Dense_{l,c}(h_v or h_{v,c,:}) ), [K, K] )              (rgdcn.py:126-146)

The K x K kernel depends on the TARGET node only, and sum / mean / sqrt_n are linear, so
    sum_e s_e (h_{u_e,c} @ W_{l,v,c}) == ( sum_e s_e h_{u_e,c} ) @ W_{l,v,c}:
the edge-side work is ONE gather + scale + segment-sum of raw source states into the (target, type) buckets
(relgnn_seg_reduce_fwd, seg_stride 1) and the dynamic kernels are applied node-side to the V*L*C aggregated
channel vectors instead of to M*C gathered ones (the reference materialises an [E, K, K] gather per channel and
type, rgdcn.py:140-146).  `max` aggregation does not commute with the kernel application and is not supported.
"""
from typing import List, Mapping, Optional

import torch

from .. import _lib, ops
from ..dense import dense
from ..graph import GatherReducePlan, as_rel_graph
from ..utils import apply_activation, get_activation
from ._common import require_weights


def rgdcn_layer_variables(num_edge_types: int, num_channels: int, channel_dim: int,
                          use_full_state_for_channel_weights: bool = False, tie_channel_weights: bool = False):
    """Dense(units=K*K, no bias) per edge type (and per channel unless tied), rgdcn.py:88-101.  Initialiser in the
    reference: truncated_normal(stddev=1/K^2) — restated as "trunc_normal:<stddev>"."""
    fan_in = num_channels * channel_dim if use_full_state_for_channel_weights else channel_dim
    specs = {}
    for l in range(num_edge_types):
        for c in range(1 if tie_channel_weights else num_channels):
            specs["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, c)] = \
                ((fan_in, channel_dim * channel_dim), "trunc_normal:%r" % (1.0 / channel_dim ** 2))
    return specs


def _bucket_plan(graph, w) -> GatherReducePlan:
    """raw source states gathered into the V*L (target, type) buckets; transposed side merges the types of a source."""
    key = ("HB", None if w is None else w.data_ptr())
    if key not in graph._plans:
        graph._plans[key] = (w, GatherReducePlan(
            rowptr=graph.rowptr_t, stride=1, col=graph.src_t, w=w, num_out=graph.V * graph.L, num_rows_x=graph.V,
            rowptr_b=graph.rowptr_s, stride_b=graph.L, col_b=graph.frow_s, pos_b=graph.pos_t_of_s,
            num_messages=graph.M))
    return graph._plans[key][1]


def sparse_rgdcn_layer(node_embeddings: torch.Tensor,
                       adjacency_lists: List[torch.Tensor],
                       type_to_num_incoming_edges: torch.Tensor,
                       num_channels: int = 8,
                       channel_dim: int = 16,
                       num_timesteps: int = 1,
                       use_full_state_for_channel_weights: bool = False,
                       tie_channel_weights: bool = False,
                       activation_function: Optional[str] = "tanh",
                       message_aggregation_function: str = "sum",
                       normalize_by_num_incoming: bool = True,
                       *,
                       weights: Mapping[str, torch.Tensor] = None,
                       ) -> torch.Tensor:
    """See gnns/rgdcn.py:20-81.  `weights`: "Edge_%i_Channel_%i_Weight_Computation/kernel"
    ([K or C*K, K*K]; only channel 0 when tie_channel_weights)."""
    weights = require_weights(weights, "sparse_rgdcn_layer")
    num_nodes, d = node_embeddings.shape
    C, K = num_channels, channel_dim
    if C * K != d:
        raise ValueError("state dimension must equal num_channels * channel_dim")
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    mode = ops.aggregation_mode_id(message_aggregation_function)
    if mode == _lib.AGG_MAX:
        raise NotImplementedError("sparse_rgdcn_layer: max aggregation does not commute with the per-target kernels")
    activation_fn = get_activation(activation_function)
    w = graph.degree_scale(type_to_num_incoming_edges) if normalize_by_num_incoming else None
    plan = _bucket_plan(graph, w)

    cur_node_states = node_embeddings
    for _ in range(num_timesteps):
        # A[v, l, c, :] = sum_{e in (v,l)} s_e * h_{u_e, c, :}          (one fused gather + scale + segment-sum)
        agg = ops.seg_gather_reduce(cur_node_states, plan, "sum", None).view(num_nodes, L, C, K)
        chunked = cur_node_states.view(num_nodes, C, K)
        new_channels = []
        for c in range(C):
            wc_in = cur_node_states if use_full_state_for_channel_weights else chunked[:, c, :]
            acc = None
            for l in range(L):
                kern = weights["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, 0 if tie_channel_weights else c)]
                dyn = apply_activation(activation_fn, dense(wc_in.contiguous(), kern)).view(num_nodes, K, K)
                msg = torch.bmm(agg[:, l, c, :].unsqueeze(1), dyn).squeeze(1)           # einsum('vi,vij->vj')
                acc = msg if acc is None else acc + msg
            new_channels.append(acc)
        new_states = torch.stack(new_channels, dim=1)                                  # [V, C, K]
        if mode != _lib.AGG_SUM:
            n = graph.messages_per_target().view(num_nodes, 1, 1)
            new_states = new_states / (n if mode == _lib.AGG_MEAN else torch.sqrt(n))
        cur_node_states = apply_activation(activation_fn, new_states).reshape(num_nodes, d)
    return cur_node_states
