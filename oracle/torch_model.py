"""torch-CPU mirror of oracle/model.py (the per-layer driver loop and the PPI head) in any dtype, so that torch.autograd
gives REFERENCE GRADIENTS of the whole model: d loss / d every variable.

TEST INFRASTRUCTURE; PARITY UNPINNED (oracle/__init__.py).  Never imported by the package.
  graph_propagation : models/sparse_graph_model.py:162-202 (dropout keep-prob 1: identity)
  ppi_loss          : tasks/ppi_task.py:176-191
"""
import torch

from . import torch_ref as R
from .tf_ops import layer_norm_scope


def graph_propagation(initial_node_features, adjacency_lists, type_to_num_incoming_edges, params, weights, apply_gnn_layer):
    """`weights`: names relative to "graph_model/" -> tensors (requires_grad where a gradient is wanted);
    `apply_gnn_layer(layer_idx, h, adj, deg, timesteps, layer_weights)` as in oracle/model.py."""
    h_dim = params['hidden_size']
    act = R.activation(params['graph_model_activation_function'])
    if initial_node_features.shape[1] != h_dim:                                         # :165-170
        cur = act(initial_node_features @ weights["dense/kernel"])
    else:
        cur = initial_node_features
    last_residual = torch.zeros_like(cur)                                               # :175
    for layer_idx in range(params['graph_num_layers']):                                 # :176
        scope = "gnn_layer_%i/" % layer_idx
        lw = {k[len(scope):]: v for k, v in weights.items() if k.startswith(scope)}
        if layer_idx % params['graph_residual_connection_every_num_layers'] == 0:      # :180-185
            t = cur
            if layer_idx > 0:
                cur = (cur + last_residual) / 2
            last_residual = t
        cur = apply_gnn_layer(layer_idx, cur, adjacency_lists, type_to_num_incoming_edges,
                              params['graph_num_timesteps_per_layer'], lw)              # :186-191
        if params['graph_inter_layer_norm']:                                            # :192-193
            n_ln = sum(1 for k in lw if k.startswith("LayerNorm") and k.endswith("/gamma"))
            ln = layer_norm_scope(n_ln - 1)
            cur = R.layer_norm(cur, lw[ln + "/gamma"], lw[ln + "/beta"])
        if layer_idx % params['graph_dense_between_every_num_gnn_layers'] == 0:        # :194-200
            cur = act(cur @ lw["Dense/kernel"])
    return cur


def rgcn_apply(params, lean=False, relu_masks=None, pre_activations=None):
    """models/rgcn_model.py:31-44 (normalize_by_num_incoming not passed: the layer default True).  lean=True: the
    BASELINE-size float64 evaluation of oracle/torch_ref.py:sparse_rgcn_layer_lean; relu_masks (lean only): per layer the
    [V, D] bool branch of every ReLU unit as the run under test took it (see sparse_rgcn_layer_lean), pre_activations: a list
    that receives every layer's float64 pre-activation."""
    layer = R.sparse_rgcn_layer_lean if lean else R.sparse_rgcn_layer

    def apply(layer_idx, h, adj, deg, timesteps, w):
        extra = {}
        if lean and relu_masks is not None:
            extra["relu_mask"] = relu_masks[layer_idx]
        if lean and pre_activations is not None:
            extra["pre_activations"] = pre_activations
        return layer(h, adj, deg, params['hidden_size'], num_timesteps=timesteps,
                     activation_function=params['graph_activation_function'],
                     message_aggregation_function=params['message_aggregation_function'],
                     weights={k: v for k, v in w.items() if k.startswith("Edge_")}, **extra)
    return apply


def ppi_loss(final_node_representations, target_labels, kernel, bias):
    """tasks/ppi_task.py:176-191: Dense(num_labels, bias) -> sigmoid cross-entropy -> sum / num_nodes."""
    logits = final_node_representations @ kernel + bias
    total = torch.nn.functional.binary_cross_entropy_with_logits(logits, target_labels, reduction='sum')
    return total / target_labels.shape[0]
