"""Messages that depend on BOTH endpoint states (Dense on [h_u || h_v]):
gnns/rgcn.py:91-96 (use_both_source_and_target), gnns/rgin.py:114-119 and
gnns/gnn_edge_mlp.py:95-100 (use_target_state_as_input).

The first Dense layer on the concatenation splits into two node-side GEMMs,
P = H @ W[:D] (rows src*L+l) and Q = H @ W[D:] (rows tgt*L+l); the per-message sum
P[col[p]] + Q[v*L+l] is formed inside the HIP kernel (csrc/edge_fused.hip).
"""
from typing import Mapping, Optional

import torch

from .. import _lib, ops
from ..dense import dense
from ..utils import apply_activation, get_activation


def pair_messages_reduce(p: torch.Tensor, q: torch.Tensor, graph, w: Optional[torch.Tensor],
                         aggregation: str, message_activation: Optional[str],
                         output_activation: Optional[str]) -> torch.Tensor:
    """out[v] = act_out( AGG_p act_msg( w[p] * (P[col[p]] + Q[v*L + l(p)]) ) )

    The fused kernel gives ONE lane group to a target node; a graph that is known to hold hub buckets (RelGraph.has_long_buckets:
    more than ops.LONG_SEGMENT messages in one bucket) takes the materialised route instead, whose reduction is the gather-reduce
    kernel with chunked virtual rows (ops.SplitPlan) — slower per message, but no wave walks 1e4+ messages alone."""
    if ops.aggregation_mode_id(aggregation) == _lib.AGG_MAX or graph.has_long_buckets:
        msgs = ops.pair_materialize(p, q, graph, None)                 # [M, D] type-major order
        if w is not None:
            msgs = graph.w_original_order(w).unsqueeze(1) * msgs
        msgs = apply_activation(get_activation(message_activation), msgs)
        out = ops.seg_gather_reduce(msgs, graph.plan_messages(), aggregation, None)
    else:
        out = ops.pair_messages_reduce_fused(p, q, graph, w, aggregation, message_activation)
    return apply_activation(get_activation(output_activation), out)


def split_first_layer(kernel: torch.Tensor, in_dim: int):
    """Dense kernel on [h_u || h_v] -> (source half, target half)."""
    return kernel[:in_dim], kernel[in_dim:]


def edge_mlp_messages(cur: torch.Tensor, graph, weights: Mapping[str, torch.Tensor], mlp_name_pattern: str,
                      num_hidden_layers: int, hidden_activation: Optional[str], use_target_state_as_input: bool,
                      ) -> torch.Tensor:
    """Per-edge-type MLP with >= 1 hidden layer on [h_u (|| h_v)] -> messages [M, D_out] in the reference's
    type-major message order (utils/utils.py:120-126 applied per edge type, gnns/gnn_edge_mlp.py:102).

    Layer 1 is node-side (split GEMMs) + one gather/add/activation kernel; the remaining layers are
    genuinely per-edge and run as one GEMM per edge type on the contiguous [E_l, D] block."""
    V, d_in = cur.shape
    L = graph.L
    names = ["dense" if i == 0 else "dense_%i" % i for i in range(num_hidden_layers + 1)]
    k0 = [weights["%s/%s/kernel" % (mlp_name_pattern % l, names[0])] for l in range(L)]
    w_src = torch.cat([k[:d_in] for k in k0], dim=1)                                  # [D, L*Dh]
    dh = k0[0].shape[1]
    p = dense(cur, w_src).view(V * L, dh)
    q = None
    if use_target_state_as_input:
        w_tgt = torch.cat([k[d_in:] for k in k0], dim=1)
        q = dense(cur, w_tgt).view(V * L, dh)
    hidden = ops.pair_materialize(p, q, graph, hidden_activation)                     # [M, Dh]
    act_fn = get_activation(hidden_activation)
    offs = graph.type_offsets
    h = hidden
    for i in range(1, num_hidden_layers + 1):
        # one per-edge GEMM per edge type on its contiguous [E_l, D] block, written in place of a concat
        h = ops.blocked_linear(h, offs, [weights["%s/%s/kernel" % (mlp_name_pattern % l, names[i])] for l in range(L)])
        if i < num_hidden_layers:
            h = apply_activation(act_fn, h)
    return h
