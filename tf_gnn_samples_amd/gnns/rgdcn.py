"""sparse_rgdcn_layer — MI355X mirror of gnns/rgdcn.py:8-167 (dynamic per-target convolution kernels).

    h'_{v,c,:} = sigma( AGG_l AGG_{(u,v) in A_l}  1/c_{l,v} * ( h_{u,c,:} @ W_{l,v,c} ) ),
    W_{l,v,c} = reshape( sigma( Dense_{l,c}(h_v or h_{v,c,:}) ), [K, K] )              (rgdcn.py:126-146)

The K x K kernel depends on the TARGET node only, and sum / mean / sqrt_n are linear, so
    sum_e s_e (h_{u_e,c} @ W_{l,v,c}) == ( sum_e s_e h_{u_e,c} ) @ W_{l,v,c}:
the edge-side work is ONE gather + scale + segment-sum of raw source states into the (target, type) buckets
(relgnn_seg_reduce_fwd, seg_stride 1) and the dynamic kernels are applied node-side to the V*L*C aggregated
channel vectors instead of to M*C gathered ones (the reference materialises an [E, K, K] gather per channel and
type, rgdcn.py:140-146): all L*C weight-computation Dense layers are ONE GEMM call and the kernels are applied by one HIP
launch (csrc/rgdcn.hip, relgnn_rgdcn_apply_fwd/bwd).  `max` aggregation does not commute with the kernel application:
it runs the reference's per-message evaluation (see the branch below).
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
    act_id = ops.activation_id(activation_function)
    activation_fn = get_activation(activation_function)
    w = graph.degree_scale(type_to_num_incoming_edges) if normalize_by_num_incoming else None
    kernel = lambda l, c: weights["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, 0 if tie_channel_weights else c)]

    def dynamic_weights_pre(cur):
        """Pre-activation output of every weight-computation Dense (rgdcn.py:134-138) in ONE GEMM call:
        full state: [V, D] @ [D, L*C*K*K] -> [V, L, C, K*K]; per channel: bmm [C, V, K] @ [C, K, L*K*K] -> [C, V, L, K*K]."""
        if use_full_state_for_channel_weights:
            w_cat = torch.cat([kernel(l, c) for l in range(L) for c in range(C)], dim=1)        # [D, L*C*K*K]
            return dense(cur, w_cat), False
        w_c = torch.stack([torch.cat([kernel(l, c) for l in range(L)], dim=1) for c in range(C)])  # [C, K, L*K*K]
        return torch.bmm(cur.view(num_nodes, C, K).transpose(0, 1), w_c), True                  # [C, V, L*K*K]

    cur_node_states = node_embeddings
    if mode == _lib.AGG_MAX:
        # max does not commute with the kernel application: evaluate the reference's per-MESSAGE einsum (rgdcn.py:140-146,
        # an [E, K, K] gather per channel and type) and reduce the materialised messages with the HIP segment-max
        # (equal gradient split among ties, like tf.unsorted_segment_max).  Not a shipped configuration; O(M*K*K) memory.
        src = graph.key_by_source.long() // L                                                    # type-major message order
        tgt_row = graph.key_by_target.long()                                                     # tgt*L + l per message
        w_orig = graph.w_original_order(w).unsqueeze(1) if w is not None else None
        for _ in range(num_timesteps):
            pre, channel_major = dynamic_weights_pre(cur_node_states)
            dyn = apply_activation(activation_fn, pre)
            dyn = dyn.view(C, num_nodes * L, K, K) if channel_major else dyn.view(num_nodes * L, C, K, K)
            chunked = cur_node_states.view(num_nodes, C, K)
            per_channel = []
            for c in range(C):
                kern = (dyn[c] if channel_major else dyn[:, c]).index_select(0, tgt_row)         # [M, K, K]
                msg = torch.bmm(chunked[:, c, :].index_select(0, src).unsqueeze(1), kern).squeeze(1)   # einsum vi,vij->vj
                per_channel.append(msg * w_orig if w_orig is not None else msg)
            msgs = torch.cat(per_channel, dim=1)                                                 # [M, C*K]
            agg = ops.seg_gather_reduce(msgs, graph.plan_messages(), message_aggregation_function, None)
            cur_node_states = apply_activation(activation_fn, agg)
        return cur_node_states

    plan = _bucket_plan(graph, w)
    fused = ops.rgdcn_apply_supported(K, act_id)
    for _ in range(num_timesteps):
        # A[v, l, c, :] = sum_{e in (v,l)} s_e * h_{u_e, c, :}          (one fused gather + scale + segment-sum)
        agg = ops.seg_gather_reduce(cur_node_states, plan, "sum", None)                          # [V*L, C*K]
        pre, channel_major = dynamic_weights_pre(cur_node_states)
        if fused:
            cur_node_states = ops.rgdcn_apply(agg, pre, graph, C, K, message_aggregation_function, activation_function,
                                              activation_function, channel_major)
            continue
        # odd channel widths / gelu: the same arithmetic through library ops
        dyn = apply_activation(activation_fn, pre)
        dyn = dyn.view(C, num_nodes, L, K, K).permute(1, 2, 0, 3, 4) if channel_major else dyn.view(num_nodes, L, C, K, K)
        new_states = torch.einsum('vlci,vlcij->vcj', agg.view(num_nodes, L, C, K), dyn)
        if mode != _lib.AGG_SUM:
            n = graph.messages_per_target().view(num_nodes, 1, 1)
            new_states = new_states / (n if mode == _lib.AGG_MEAN else torch.sqrt(n))
        cur_node_states = apply_activation(activation_fn, new_states).reshape(num_nodes, d)
    return cur_node_states
