"""Messages that depend on BOTH endpoint states (Dense on [h_u || h_v]):
gnns/rgcn.py:91-96 (use_both_source_and_target), gnns/rgin.py:114-119 and
gnns/gnn_edge_mlp.py:95-100 (use_target_state_as_input).

The first Dense layer on the concatenation splits into two node-side GEMMs,
P = H @ W[:D] (rows src*L+l) and Q = H @ W[D:] (rows tgt*L+l); the per-message sum
P[col[p]] + Q[v*L+l] is formed inside the HIP kernel.
"""
from typing import Optional

import torch


def pair_messages_reduce(p: torch.Tensor, q: torch.Tensor, graph, w: Optional[torch.Tensor],
                         aggregation: str, message_activation: Optional[str],
                         output_activation: Optional[str]) -> torch.Tensor:
    """out[v] = act_out( AGG_p act_msg( w[p] * (P[col[p]] + Q[v*L + l(p)]) ) )"""
    raise NotImplementedError("pair-message kernels land with csrc/edge_fused.hip")
