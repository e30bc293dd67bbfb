"""NumPy restatement of the reference's six sparse GNN layers, op for op, in the reference's
op order:  per edge type  gather source rows -> edge-wise transform -> (1/in-degree scale) ->
concat over types -> unsorted_segment_{sum,mean,max,sqrt_n} into targets -> node update.

TEST INFRASTRUCTURE; PARITY UNPINNED (oracle/__init__.py).  `weights` maps TF variable names
relative to the layer's variable scope (e.g. "Edge_0_Weight/kernel", Dense kernels laid out
[in, out]) to arrays.  All arithmetic runs in node_embeddings.dtype.
"""
import numpy as np

from . import tf_ops as T
from .tf_ops import SMALL_NUMBER, layer_norm_scope


def _cast(weights, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}


def _targets(adjacency_lists):
    # message_targets = tf.concat([adj[:, 1] for adj in adjacency_lists], 0)  (e.g. gnns/rgcn.py:75-78)
    return np.concatenate([np.asarray(a).reshape(-1, 2)[:, 1] for a in adjacency_lists]).astype(np.int32)


def _inv_degree(type_to_num_incoming_edges, edge_type_idx, edge_targets, dtype):
    # gnns/rgcn.py:100-104: 1.0 / (embedding_lookup(c[l, :], targets) + SMALL_NUMBER), float32 in TF
    c = T.embedding_lookup(np.asarray(type_to_num_incoming_edges[edge_type_idx], dtype=dtype), edge_targets)
    return (np.asarray(1.0, dtype) / (c + np.asarray(SMALL_NUMBER, dtype)))[:, None]


def _rgcn_layer_node_side(cur, adjacency_lists, type_to_num_incoming_edges, weights, normalize, activation_fn):
    """BASELINE-size variant of the sum-aggregating, source-only RGCN message pass (gnns/rgcn.py:84-114): the message
    of edge (u, v) of type l is the row (h_u W_l); it is evaluated once per (l, u) instead of once per edge and the
    [M, D] message tensor is never materialised — the C fold walks the messages in the reference's order
    (type-major, then edge order) and adds scale * row sequentially in fp32.  tests/test_oracle.py checks it against
    the op-for-op path above."""
    dtype = cur.dtype
    V, L = cur.shape[0], len(adjacency_lists)
    table = np.concatenate([T.dense(cur, weights["Edge_%i_Weight/kernel" % l]) for l in range(L)], axis=0)   # [L*V, D]
    rows, scales, ids = [], [], []
    for l, adj in enumerate(adjacency_lists):
        adj = np.asarray(adj).reshape(-1, 2)
        rows.append(adj[:, 0].astype(np.int64) + l * V)
        ids.append(adj[:, 1].astype(np.int32))
        if normalize:
            scales.append(_inv_degree(type_to_num_incoming_edges, l, adj[:, 1], dtype)[:, 0])
    rows, ids = np.ascontiguousarray(np.concatenate(rows)), np.ascontiguousarray(np.concatenate(ids))
    scale = np.ascontiguousarray(np.concatenate(scales)) if normalize else None
    lib = T._clib()
    D = table.shape[1]
    if lib is not None and dtype == np.float32:
        out = np.empty((V, D), np.float32)
        table = np.ascontiguousarray(table)
        lib.oracle_gather_scaled_seg_sum_f32(table.ctypes.data, rows.ctypes.data, scale.ctypes.data if normalize else None,
                                             ids.ctypes.data, len(rows), D, V, out.ctypes.data)
    else:
        msgs = table[rows]
        if normalize:
            msgs = scale[:, None] * msgs
        out = T.unsorted_segment_sum(msgs, ids, V)
    return T.apply_act(activation_fn, out)


def sparse_rgcn_layer(node_embeddings, adjacency_lists, type_to_num_incoming_edges, state_dim,
                      num_timesteps=1, activation_function="tanh", message_aggregation_function="sum",
                      normalize_by_num_incoming=True, use_both_source_and_target=False, *, weights,
                      node_side_transform=False):
    """gnns/rgcn.py:60-117.  node_side_transform=True (sum aggregation, source-only inputs): the BASELINE-size
    evaluation order of _rgcn_layer_node_side."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    if node_side_transform and not use_both_source_and_target and message_aggregation_function in ("sum", "unsorted_segment_sum"):
        cur = node_embeddings
        for _ in range(num_timesteps):
            cur = _rgcn_layer_node_side(cur, adjacency_lists, type_to_num_incoming_edges, weights,
                                        normalize_by_num_incoming, T.get_activation(activation_function))
        return cur
    num_nodes = node_embeddings.shape[0]
    activation_fn = T.get_activation(activation_function)
    aggregate = T.get_aggregation_function(message_aggregation_function)
    message_targets = _targets(adjacency_lists)                                         # :78
    cur = node_embeddings
    for _ in range(num_timesteps):                                                      # :81
        messages_per_type = []
        for l, adj in enumerate(adjacency_lists):                                       # :84
            adj = np.asarray(adj).reshape(-1, 2)
            sources, targets = adj[:, 0], adj[:, 1]
            src_states = T.embedding_lookup(cur, sources)                               # :87-89
            if use_both_source_and_target:                                              # :91-96
                tgt_states = T.embedding_lookup(cur, targets)
                messages = T.dense(np.concatenate([src_states, tgt_states], axis=-1), weights["Edge_%i_Weight/kernel" % l])
            else:
                messages = T.dense(src_states, weights["Edge_%i_Weight/kernel" % l])    # :98
            if normalize_by_num_incoming:                                               # :100-104
                messages = _inv_degree(type_to_num_incoming_edges, l, targets, dtype) * messages
            messages_per_type.append(messages)
        cur_messages = np.concatenate(messages_per_type, axis=0)                        # :108
        aggregated = aggregate(cur_messages, message_targets, num_nodes)                # :109-112
        cur = T.apply_act(activation_fn, aggregated)                                    # :114
    return cur


def sparse_ggnn_layer(node_embeddings, adjacency_lists, state_dim, num_timesteps=1, gated_unit_type="gru",
                      activation_function="tanh", message_aggregation_function="sum", *, weights):
    """gnns/ggnn.py:50-95; cell from utils/utils.py:10-20."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    aggregate = T.get_aggregation_function(message_aggregation_function)
    activation_fn = T.get_activation(activation_function)
    kind = gated_unit_type.lower()
    if kind == 'rnn':
        scope, cell = "simple_rnn_cell", T.simple_rnn_cell
    elif kind == 'gru':
        scope, cell = "gru_cell", T.gru_cell
    elif kind == 'lstm':
        raise NotImplementedError("the reference feeds LSTMCell a single state (ggnn.py:92): it cannot run")
    else:
        raise Exception("Unknown RNN cell type '%s'." % gated_unit_type)
    message_targets = _targets(adjacency_lists)
    cur = node_embeddings
    for _ in range(num_timesteps):
        messages = []
        for l, adj in enumerate(adjacency_lists):
            adj = np.asarray(adj).reshape(-1, 2)
            src_states = T.embedding_lookup(cur, adj[:, 0])                             # :78-79
            messages.append(T.dense(src_states, weights["Edge_%i_Weight/kernel" % l]))  # :80-81
        messages = np.concatenate(messages, axis=0)                                     # :85
        aggregated = aggregate(messages, message_targets, num_nodes)                    # :86-89
        cur = cell(aggregated, cur, weights[scope + "/kernel"], weights[scope + "/recurrent_kernel"],
                   weights[scope + "/bias"], activation_fn)                             # :92
    return cur


def sparse_rgat_layer(node_embeddings, adjacency_lists, state_dim, num_heads=4, num_timesteps=1,
                      activation_function="tanh", *, weights):
    """gnns/rgat.py:58-141."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    if state_dim is None:
        state_dim = node_embeddings.shape[1]
    per_head_dim = state_dim // num_heads
    activation_fn = T.get_activation(activation_function)
    message_targets = _targets(adjacency_lists)                                         # :80
    cur = node_embeddings
    for _ in range(num_timesteps):
        per_type_msgs, per_type_coefs = [], []
        for l, adj in enumerate(adjacency_lists):
            adj = np.asarray(adj).reshape(-1, 2)
            sources, targets = adj[:, 0], adj[:, 1]
            transformed = T.dense(cur, weights["Edge_%i_Weight/kernel" % l])            # :95-96 (on NODES)
            src_t = T.embedding_lookup(transformed, sources).reshape(-1, num_heads, per_head_dim)   # :98-104
            tgt_t = T.embedding_lookup(transformed, targets).reshape(-1, num_heads, per_head_dim)
            pair = np.concatenate([src_t, tgt_t], axis=-1)                              # :106-109  [E, K, 2D/K]
            att = weights["Edge_%i_Attention_Parameters" % l].reshape(num_heads, 2 * per_head_dim)  # :110-111
            coefs = T.leaky_relu(np.einsum('vki,ki->vk', pair, att))                    # :112-115
            per_type_msgs.append(src_t)
            per_type_coefs.append(coefs)
        per_head_messages = np.concatenate(per_type_msgs, axis=0)                       # :120
        per_head_coefs = np.concatenate(per_type_coefs, axis=0)                         # :121
        heads = []
        for k in range(num_heads):                                                      # :124
            att_values = np.exp(T.unsorted_segment_log_softmax(per_head_coefs[:, k], message_targets, num_nodes))  # :126-130
            heads.append(T.unsorted_segment_sum(att_values[:, None] * per_head_messages[:, k, :],
                                                message_targets, num_nodes))            # :133-136
        cur = T.apply_act(activation_fn, np.concatenate(heads, axis=-1))                # :138
    return cur


def sparse_rgin_layer(node_embeddings, adjacency_lists, state_dim, num_timesteps=1, activation_function="ReLU",
                      message_aggregation_function="sum", use_target_state_as_input=False,
                      num_edge_MLP_hidden_layers=1, num_aggr_MLP_hidden_layers=None, *, weights):
    """gnns/rgin.py:69-142."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    activation_fn = T.get_activation(activation_function)
    aggregate = T.get_aggregation_function(message_aggregation_function)
    message_targets = _targets(adjacency_lists)
    cur = node_embeddings
    for t in range(num_timesteps):
        messages_per_type = []
        for l, adj in enumerate(adjacency_lists):
            adj = np.asarray(adj).reshape(-1, 2)
            mlp_in = T.embedding_lookup(cur, adj[:, 0])                                 # :110-112
            if use_target_state_as_input:                                               # :114-119
                mlp_in = np.concatenate([mlp_in, T.embedding_lookup(cur, adj[:, 1])], axis=1)
            if num_edge_MLP_hidden_layers is not None:                                  # :121-124
                messages = T.mlp(mlp_in, weights, "Edge_%i_MLP" % l, num_edge_MLP_hidden_layers, activation_fn)
            else:
                messages = mlp_in
            messages_per_type.append(messages)
        all_messages = np.concatenate(messages_per_type, axis=0)                        # :127
        if num_edge_MLP_hidden_layers is not None:
            all_messages = T.apply_act(activation_fn, all_messages)                     # :128-129
        new_states = aggregate(all_messages, message_targets, num_nodes)                # :130-133
        if num_aggr_MLP_hidden_layers is not None:                                      # :136-137
            new_states = T.mlp(new_states, weights, "Aggregation_MLP", num_aggr_MLP_hidden_layers, activation_fn)
        new_states = T.apply_act(activation_fn, new_states)                             # :138
        cur = T.layer_norm(new_states, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])   # :139
    return cur


def sparse_gnn_film_layer(node_embeddings, adjacency_lists, type_to_num_incoming_edges, state_dim,
                          num_timesteps=1, activation_function="ReLU", message_aggregation_function="sum",
                          normalize_by_num_incoming=False, *, weights):
    """gnns/gnn_film.py:58-122."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    if state_dim is None:
        state_dim = node_embeddings.shape[1]
    activation_fn = T.get_activation(activation_function)
    aggregate = T.get_aggregation_function(message_aggregation_function)
    message_targets = _targets(adjacency_lists)
    cur = node_embeddings
    for t in range(num_timesteps):
        messages_per_type = []
        for l, adj in enumerate(adjacency_lists):
            adj = np.asarray(adj).reshape(-1, 2)
            sources, targets = adj[:, 0], adj[:, 1]
            messages = T.dense(T.embedding_lookup(cur, sources), weights["Edge_%i_Weight/kernel" % l])   # :92-94
            if normalize_by_num_incoming:                                               # :96-100
                messages = _inv_degree(type_to_num_incoming_edges, l, targets, dtype) * messages
            film = T.dense(cur, weights["Edge_%i_FiLM_Computations/kernel" % l])        # :102 (on NODES)
            per_msg = T.embedding_lookup(film, targets)                                 # :103-104
            gamma, beta = per_msg[:, :state_dim], per_msg[:, state_dim:]                # :105-106
            messages_per_type.append(gamma * messages + beta)                           # :108
        all_messages = T.apply_act(activation_fn, np.concatenate(messages_per_type, axis=0))   # :111-112
        aggregated = aggregate(all_messages, message_targets, num_nodes)                # :113-116
        cur = T.layer_norm(aggregated, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])   # :120
    return cur


def sparse_gnn_edge_mlp_layer(node_embeddings, adjacency_lists, type_to_num_incoming_edges, state_dim,
                              num_timesteps=1, activation_function="ReLU", message_aggregation_function="sum",
                              normalize_by_num_incoming=False, use_target_state_as_input=True,
                              num_edge_hidden_layers=1, *, weights):
    """gnns/gnn_edge_mlp.py:63-122 (MLP hidden activation hard-wired to elu, :76)."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    activation_fn = T.get_activation(activation_function)
    aggregate = T.get_aggregation_function(message_aggregation_function)
    message_targets = _targets(adjacency_lists)
    cur = node_embeddings
    for t in range(num_timesteps):
        messages_per_type = []
        for l, adj in enumerate(adjacency_lists):
            adj = np.asarray(adj).reshape(-1, 2)
            sources, targets = adj[:, 0], adj[:, 1]
            mlp_in = T.embedding_lookup(cur, sources)                                   # :91-93
            if use_target_state_as_input:                                               # :95-100
                mlp_in = np.concatenate([mlp_in, T.embedding_lookup(cur, targets)], axis=1)
            messages = T.mlp(mlp_in, weights, "Edge_%i_MLP" % l, num_edge_hidden_layers, T.elu)   # :102
            if normalize_by_num_incoming:                                               # :104-108
                messages = _inv_degree(type_to_num_incoming_edges, l, targets, dtype) * messages
            messages_per_type.append(messages)
        all_messages = T.apply_act(activation_fn, np.concatenate(messages_per_type, axis=0))   # :111-112
        aggregated = aggregate(all_messages, message_targets, num_nodes)                # :113-116
        cur = T.layer_norm(aggregated, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])   # :119
    return cur


def sparse_rgdcn_layer(node_embeddings, adjacency_lists, type_to_num_incoming_edges, num_channels=8, channel_dim=16,
                       num_timesteps=1, use_full_state_for_channel_weights=False, tie_channel_weights=False,
                       activation_function="tanh", message_aggregation_function="sum", normalize_by_num_incoming=True,
                       *, weights):
    """gnns/rgdcn.py:82-167."""
    dtype = node_embeddings.dtype
    weights = _cast(weights, dtype)
    num_nodes = node_embeddings.shape[0]
    activation_fn = T.get_activation(activation_function)
    aggregate = T.get_aggregation_function(message_aggregation_function)
    message_targets = _targets(adjacency_lists)                                              # :106
    cur = node_embeddings
    for _ in range(num_timesteps):
        chunked = cur.reshape(-1, num_channels, channel_dim)                                 # :110-111
        new_chunks = []
        for c in range(num_channels):                                                        # :114
            cur_channel = chunked[:, c, :]
            per_type = []
            for l, adj in enumerate(adjacency_lists):                                        # :119
                adj = np.asarray(adj).reshape(-1, 2)
                sources, targets = adj[:, 0], adj[:, 1]
                src_states = T.embedding_lookup(cur_channel, sources)                        # :122-124
                wc_in = cur if use_full_state_for_channel_weights else cur_channel           # :126-129
                kern = weights["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, 0 if tie_channel_weights else c)]
                edge_weights = T.dense(wc_in, kern, activation=activation_fn)                # :132-134 (Dense WITH activation)
                edge_weights = edge_weights.reshape(-1, channel_dim, channel_dim)            # :135
                w_tgt = T.embedding_lookup(edge_weights, targets)                            # :136-137
                messages = np.einsum('vi,vij->vj', src_states, w_tgt)                        # :140
                if normalize_by_num_incoming:                                                # :141-145
                    messages = _inv_degree(type_to_num_incoming_edges, l, targets, dtype) * messages
                per_type.append(messages)
            msgs = np.concatenate(per_type, axis=0)                                          # :149
            agg = aggregate(msgs, message_targets, num_nodes)                                # :150-153
            new_chunks.append(T.apply_act(activation_fn, agg))                               # :154
        cur = np.concatenate(new_chunks, axis=1)                                             # :158
    return cur
