"""NumPy restatement of the per-layer driver loop and the PPI output head.

TEST INFRASTRUCTURE; PARITY UNPINNED (oracle/__init__.py).
  graph_propagation : models/sparse_graph_model.py:162-202 (dropout: identity at keep-prob 1; with `dropout` = (keep_prob, one
                      keep mask per layer) the masks stand in for TF's random draws, the scaling is tf.nn.dropout's)
  ppi_head_loss     : tasks/ppi_task.py:176-191
  rgcn_ppi_num_parameters : re-derivation of README.md:29 (699 257)
"""
import numpy as np

from . import gnns, tf_ops as T


def graph_propagation(initial_node_features, adjacency_lists, type_to_num_incoming_edges, params, weights,
                      apply_gnn_layer, initial_node_feature_size=None, dropout=None):
    """models/sparse_graph_model.py:162-202.  `weights` uses names relative to "graph_model/":
    "dense/kernel" (input projection, :165-170), "gnn_layer_%i/..." (layer variables, :177) and
    "gnn_layer_%i/Dense/kernel" (:194-200).  `apply_gnn_layer(layer_idx, h, adj, deg, timesteps, layer_weights)`
    plays the role of the model adapter's _apply_gnn_layer (:186-191)."""
    dtype = initial_node_features.dtype
    h_dim = params['hidden_size']
    activation_fn = T.get_activation(params['graph_model_activation_function'])
    if initial_node_features.shape[1] != h_dim:                                         # :165-170
        cur = T.dense(initial_node_features, np.asarray(weights["dense/kernel"], dtype), activation=activation_fn)
    else:
        cur = initial_node_features
    last_residual = np.zeros_like(cur)                                                  # :175
    for layer_idx in range(params['graph_num_layers']):                                 # :176
        scope = "gnn_layer_%i/" % layer_idx
        layer_weights = {k[len(scope):]: v for k, v in weights.items() if k.startswith(scope)}
        # :178-179 tf.nn.dropout(x, rate = 1 - keep_prob) on the layer's INPUT, before the residual step: x / keep_prob where the
        # mask keeps, 0 elsewhere [TF-internal: div(x, keep_prob) * floor(keep_prob + uniform)]; rate 0 is the identity
        if dropout is not None:
            keep_prob, masks = dropout
            cur = cur / np.asarray(keep_prob, dtype) * np.asarray(masks[layer_idx], dtype)
        if layer_idx % params['graph_residual_connection_every_num_layers'] == 0:      # :180-185
            t = cur
            if layer_idx > 0:
                cur = cur + last_residual
                cur = cur / np.asarray(2, dtype)
            last_residual = t
        cur = apply_gnn_layer(layer_idx, cur, adjacency_lists, type_to_num_incoming_edges,
                              params['graph_num_timesteps_per_layer'], layer_weights)   # :186-191
        if params['graph_inter_layer_norm']:                                            # :192-193
            # the LAST LayerNorm scope of the layer's variable scope: the layer's own per-timestep norms come first
            n_ln = sum(1 for k in layer_weights if k.startswith("LayerNorm") and k.endswith("/gamma"))
            ln = T.layer_norm_scope(n_ln - 1)
            cur = T.layer_norm(cur, np.asarray(layer_weights[ln + "/gamma"], dtype),
                               np.asarray(layer_weights[ln + "/beta"], dtype))
        if layer_idx % params['graph_dense_between_every_num_gnn_layers'] == 0:        # :194-200
            cur = T.dense(cur, np.asarray(layer_weights["Dense/kernel"], dtype), activation=activation_fn)
    return cur


def rgcn_apply(params, node_side_transform=False):
    """models/rgcn_model.py:31-44: normalize_by_num_incoming is NOT passed, so the layer default True applies."""
    def apply(layer_idx, h, adj, deg, timesteps, w):
        return gnns.sparse_rgcn_layer(h, adj, deg, params['hidden_size'], num_timesteps=timesteps,
                                      activation_function=params['graph_activation_function'],
                                      message_aggregation_function=params['message_aggregation_function'],
                                      weights={k: v for k, v in w.items() if k.startswith("Edge_")},
                                      node_side_transform=node_side_transform)
    return apply


def sigmoid_cross_entropy_with_logits(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits [TF-internal]: max(x,0) - x*z + log(1 + exp(-|x|))."""
    return np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))


def ppi_head_loss(final_node_representations, target_labels, kernel, bias):
    """tasks/ppi_task.py:176-191: Dense(num_labels, bias) -> sigmoid CE -> sum / num_nodes."""
    logits = T.dense(final_node_representations, kernel, bias)
    total = sigmoid_cross_entropy_with_logits(logits, target_labels).sum(dtype=logits.dtype)
    return total / np.asarray(target_labels.shape[0], logits.dtype), logits


def rgcn_ppi_num_parameters(hidden_size=256, num_layers=3, num_edge_types=3, feature_size=50, num_labels=121,
                            dense_every=10000):
    """README.md:29 'Model has 699257 parameters' for RGCN/PPI (README.md:32 hypers):
    input projection F*h (sparse_graph_model.py:165-170) + per layer L*h*h (rgcn.py:70-74)
    + one h*h 'Dense' for every layer_idx % dense_every == 0 — layer 0 ALWAYS qualifies
    (sparse_graph_model.py:194-200) + output head h*labels + labels (ppi_task.py:176-179)."""
    n = feature_size * hidden_size
    n += num_layers * num_edge_types * hidden_size * hidden_size
    n += sum(1 for i in range(num_layers) if i % dense_every == 0) * hidden_size * hidden_size
    n += hidden_size * num_labels + num_labels
    return n
