"""sparse_rgat_layer — MI355X mirror of gnns/rgat.py:9-141.

    e_{u,l,v,k} = leaky_relu( a_{l,k} . [ (W_l h_u)_k || (W_l h_v)_k ] )
    alpha       = softmax over ALL messages into v (all edge types), per head k
    h'_v        = sigma( concat_k  sum alpha_{u,l,v,k} (W_l h_u)_k )

The attention logit is linear in the two transformed endpoint states, so it decomposes into the
per-node scalars s_src[u,l,k] = a_{l,k}[:Dh] . (W_l h_u)_k and s_tgt[v,l,k] = a_{l,k}[Dh:] . (W_l h_v)_k
(two [V*L, K] tables).  The segmented softmax + weighted sum then needs 4*K bytes per message for the
softmax passes and ONE gather of the source row (csrc/rgat.hip) — instead of the reference's two
[E, D] gathers, [E, K, 2D/K] concat, einsum and K x (5-op log-softmax + segment-sum).
"""
from typing import List, Mapping, Optional

import torch

from .. import ops
from ..dense import dense
from ..graph import as_rel_graph
from ..utils import apply_activation, get_activation
from ._common import concat_edge_kernels, require_weights


def rgat_layer_variables(num_edge_types: int, in_dim: int, state_dim: int):
    specs = {}
    for l in range(num_edge_types):
        specs["Edge_%i_Weight/kernel" % l] = ((in_dim, state_dim), "glorot_uniform")
        # tf.get_variable(shape=(2*state_dim)) without initializer: glorot_uniform (TF1 default) [TF-internal]
        specs["Edge_%i_Attention_Parameters" % l] = ((2 * state_dim,), "glorot_uniform")
    return specs


def _attention_from_segment_ops(T, s_src, s_tgt, graph, num_heads: int, slope: float):
    """rgat.py:98-136 op for op on materialised per-message tensors (reference message order): logits, segmented
    log-softmax over ALL incoming messages of a target per head (dpu_utils unsorted_segment_log_softmax), exp, weighted sum."""
    plan = graph.plan_messages()
    tgt = plan.col_b.long()                                                   # target node of every message
    src_rows, tgt_rows = graph.key_by_source.long(), graph.key_by_target.long()
    logits = torch.nn.functional.leaky_relu(s_src.index_select(0, src_rows) + s_tgt.index_select(0, tgt_rows), slope)   # [M, K]
    seg_max = ops.seg_gather_reduce(logits.detach(), plan, "max", None)                                                  # [V, K]
    shifted = logits - seg_max.index_select(0, tgt)
    e = torch.exp(shifted)
    log_den = torch.log(ops.seg_gather_reduce(e, plan, "sum", None).clamp_min(1e-37))   # (targets without messages: never gathered)
    alpha = torch.exp(shifted - log_den.index_select(0, tgt))                                                            # [M, K]
    M, K = alpha.shape
    D = T.shape[1]
    msgs = (alpha.unsqueeze(2) * T.index_select(0, src_rows).view(M, K, D // K)).reshape(M, D)
    return ops.seg_gather_reduce(msgs, plan, "sum", None)


def sparse_rgat_layer(node_embeddings: torch.Tensor,
                      adjacency_lists: List[torch.Tensor],
                      state_dim: Optional[int],
                      num_heads: int = 4,
                      num_timesteps: int = 1,
                      activation_function: Optional[str] = "tanh",
                      *,
                      weights: Mapping[str, torch.Tensor] = None,
                      ) -> torch.Tensor:
    """See gnns/rgat.py:16-57.  `weights`: "Edge_%i_Weight/kernel" [D, state_dim] and
    "Edge_%i_Attention_Parameters" [2*state_dim] (reshaped (K, 2*state_dim/K): the first state_dim/K entries
    of each head act on the SOURCE half, rgat.py:106-111)."""
    weights = require_weights(weights, "sparse_rgat_layer")
    num_nodes, in_dim = node_embeddings.shape
    if state_dim is None:
        state_dim = in_dim
    per_head_dim = state_dim // num_heads
    if per_head_dim * num_heads != state_dim:
        raise ValueError("state_dim must be divisible by num_heads")
    graph = as_rel_graph(adjacency_lists, num_nodes)
    L = graph.L
    activation_fn = get_activation(activation_function)
    w_cat = concat_edge_kernels(weights, L, "Edge_%i_Weight/kernel")                        # [D, L*state_dim]
    att_flat = torch.stack([weights["Edge_%i_Attention_Parameters" % l] for l in range(L)], dim=0)    # [L, 2D]
    att = att_flat.view(L, num_heads, 2 * per_head_dim)
    # The kernels give every head whole float4 lanes.  A head width that is not a multiple of 4 (state_dim 36 with
    # 2 heads, ...) is zero-padded: zero COLUMNS appended to every head of the edge kernels and zero entries to both
    # halves of the attention parameters change neither the logits nor the weighted sums; the pad columns of the
    # result are dropped again.  (Weights only: the padding costs one small copy per layer call.)
    pad = (-per_head_dim) % 4
    dh = per_head_dim + pad
    width = num_heads * dh
    if pad:
        w_cat = torch.nn.functional.pad(w_cat.view(in_dim, L, num_heads, per_head_dim), (0, pad)).reshape(in_dim, L * width)
        att = torch.cat([torch.nn.functional.pad(att[:, :, :per_head_dim], (0, pad)),
                         torch.nn.functional.pad(att[:, :, per_head_dim:], (0, pad))], dim=2)          # [L, K, 2*dh]
        att_flat = att.reshape(L, 2 * width)
    att_src, att_tgt = att[:, :, :dh], att[:, :, dh:]                                       # [L, K, dh] each
    fused_scores = node_embeddings.is_cuda and ops.rgat_scores_supported(width, num_heads)

    cur_node_states = node_embeddings
    for _ in range(num_timesteps):
        transformed = dense(cur_node_states, w_cat)                                               # [V, L*width]
        if graph.has_long_buckets:
            # hub targets: the attention kernels give one wave to a target; compose rgat.py:98-136 from the segment ops instead,
            # whose gather-reduce kernel splits long buckets into chunked virtual rows (ops.SplitPlan)
            t4 = transformed.view(num_nodes, L, num_heads, dh)
            s_src = (t4 * att_src.unsqueeze(0)).sum(-1).reshape(num_nodes * L, num_heads)
            s_tgt = (t4 * att_tgt.unsqueeze(0)).sum(-1).reshape(num_nodes * L, num_heads)
            aggregated = _attention_from_segment_ops(transformed.view(num_nodes * L, width), s_src, s_tgt, graph, num_heads, 0.2)
        elif fused_scores:   # score tables, softmax and weighted sum as one autograd node (csrc/rgat_scores.hip)
            aggregated = ops.rgat_layer_attention(transformed.view(num_nodes * L, width), att_flat, graph, num_heads, 0.2)
        else:
            t4 = transformed.view(num_nodes, L, num_heads, dh)
            s_src = (t4 * att_src.unsqueeze(0)).sum(-1).reshape(num_nodes * L, num_heads)   # [V*L, K]
            s_tgt = (t4 * att_tgt.unsqueeze(0)).sum(-1).reshape(num_nodes * L, num_heads)
            aggregated = ops.rgat_attention(transformed.view(num_nodes * L, width), s_src, s_tgt, graph, num_heads, 0.2)
        if pad:
            aggregated = aggregated.view(num_nodes, num_heads, dh)[:, :, :per_head_dim].reshape(num_nodes, state_dim)
        cur_node_states = apply_activation(activation_fn, aggregated)
    return cur_node_states
