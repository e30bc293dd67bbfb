"""The model adapters of the reference (models/{rgcn,ggnn,rgat,rgin,gnn_film,gnn_edge_mlp,rgdcn}_model.py), table-driven.

In the reference every adapter is a ~45-line class that (i) overrides a few hyper-parameter defaults, (ii) names
itself and (iii) forwards entries of `self.params` as keyword arguments to one `gnns.sparse_*_layer`
(e.g. models/rgcn_model.py:11-22, :24-26, :31-44).  Here one spec per model states exactly those three things and
`build_adapter` turns it into a `Sparse_Graph_Model` subclass with the reference's class name, `default_params()`,
`name()` and `_apply_gnn_layer(node_representations, adjacency_lists, type_to_num_incoming_edges, num_timesteps)`.

NB (reference behaviour kept on purpose): the RGCN and Edge-MLP adapters do not forward a normalisation flag, so the
layer defaults apply (RGCN normalises by in-degree, gnns/rgcn.py:15; Edge-MLP does not, gnns/gnn_edge_mlp.py:15);
GGNN / RGAT / RGIN layers take no in-degree table.
"""
from typing import Any, Callable, Dict, List, NamedTuple, Optional

import torch

from .. import gnns
from .sparse_graph_model import Sparse_Graph_Model


class AdapterSpec(NamedTuple):
    class_name: str
    display_name: Callable[[Dict[str, Any]], str]
    param_overrides: Dict[str, Any]                      # on top of Sparse_Graph_Model.default_params()
    layer_fn: Callable[..., torch.Tensor]
    layer_kwargs: Dict[str, str]                         # layer keyword -> key in self.params
    takes_in_degrees: bool                               # does the layer function take type_to_num_incoming_edges?
    variables: Callable[["Sparse_Graph_Model", int], Dict[str, Any]]
    derived_params: Optional[Callable[[Dict[str, Any]], None]] = None


def _no_dense_no_residual(extra: Dict[str, Any]) -> Dict[str, Any]:
    # RGCN / GGNN / RGAT switch the inter-layer Dense and residuals off by pushing their periods out of reach
    # (which still leaves the Dense after layer 0: models/sparse_graph_model.py:194-200)
    d = {'hidden_size': 128, 'graph_layer_input_dropout_keep_prob': 1.0,
         'graph_dense_between_every_num_gnn_layers': 10000, 'graph_residual_connection_every_num_layers': 10000}
    d.update(extra)
    return d


SPECS: List[AdapterSpec] = [
    AdapterSpec(
        "RGCN_Model", lambda p: "RGCN",
        _no_dense_no_residual({"graph_activation_function": "ReLU", "message_aggregation_function": "sum"}),
        gnns.sparse_rgcn_layer,
        {"state_dim": "hidden_size", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function"},
        True,
        lambda m, d: gnns.rgcn_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'])),
    AdapterSpec(
        "GGNN_Model", lambda p: "GGNN",
        _no_dense_no_residual({'graph_rnn_cell': 'GRU', 'graph_activation_function': "tanh",
                               "message_aggregation_function": "sum"}),
        gnns.sparse_ggnn_layer,
        {"state_dim": "hidden_size", "gated_unit_type": "graph_rnn_cell", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function"},
        False,
        lambda m, d: gnns.ggnn_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'], m.params['graph_rnn_cell'])),
    AdapterSpec(
        "RGAT_Model", lambda p: "RGAT",
        _no_dense_no_residual({'num_heads': 4, 'graph_activation_function': 'tanh'}),
        gnns.sparse_rgat_layer,
        {"state_dim": "hidden_size", "num_heads": "num_heads", "activation_function": "graph_activation_function"},
        False,
        lambda m, d: gnns.rgat_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'])),
    AdapterSpec(
        "RGIN_Model", lambda p: "RGIN",
        {'hidden_size': 128, "graph_activation_function": "ReLU", 'message_aggregation_function': "sum",
         'graph_dense_between_every_num_gnn_layers': 10000, 'graph_inter_layer_norm': True,
         'use_target_state_as_input': False, 'graph_num_edge_MLP_hidden_layers': 1,
         'graph_num_aggr_MLP_hidden_layers': None},
        gnns.sparse_rgin_layer,
        {"state_dim": "hidden_size", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function",
         "use_target_state_as_input": "use_target_state_as_input",
         "num_edge_MLP_hidden_layers": "graph_num_edge_MLP_hidden_layers",
         "num_aggr_MLP_hidden_layers": "graph_num_aggr_MLP_hidden_layers"},
        False,
        lambda m, d: gnns.rgin_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'],
                                               m.params['use_target_state_as_input'],
                                               m.params['graph_num_edge_MLP_hidden_layers'],
                                               m.params['graph_num_aggr_MLP_hidden_layers'],
                                               m.params['graph_num_timesteps_per_layer'])),
    AdapterSpec(
        "GNN_FiLM_Model", lambda p: "GNN-FiLM",
        {"hidden_size": 128, "graph_activation_function": "ReLU", "message_aggregation_function": "sum",
         "normalize_messages_by_num_incoming": False},
        gnns.sparse_gnn_film_layer,
        {"state_dim": "hidden_size", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function",
         "normalize_by_num_incoming": "normalize_messages_by_num_incoming"},
        True,
        lambda m, d: gnns.gnn_film_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'],
                                                   m.params['graph_num_timesteps_per_layer'])),
    AdapterSpec(
        "GNN_Edge_MLP_Model", lambda p: "GNN-Edge-MLP%i" % (p['num_edge_hidden_layers']),
        {'max_nodes_in_batch': 25000, 'hidden_size': 128, "graph_activation_function": "gelu",
         "message_aggregation_function": "sum", 'graph_inter_layer_norm': True, 'use_target_state_as_input': True,
         'num_edge_hidden_layers': 1},
        gnns.sparse_gnn_edge_mlp_layer,
        {"state_dim": "hidden_size", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function",
         "use_target_state_as_input": "use_target_state_as_input", "num_edge_hidden_layers": "num_edge_hidden_layers"},
        True,
        lambda m, d: gnns.gnn_edge_mlp_layer_variables(m.task.num_edge_types, d, m.params['hidden_size'],
                                                       m.params['use_target_state_as_input'],
                                                       m.params['num_edge_hidden_layers'],
                                                       m.params['graph_num_timesteps_per_layer'])),
    AdapterSpec(
        "RGDCN_Model", lambda p: "RGDCN",
        {'max_nodes_in_batch': 25000, 'hidden_size': 128, 'num_channels': 8,
         "use_full_state_for_channel_weights": False, "tie_channel_weights": False,
         "graph_activation_function": "ReLU", "message_aggregation_function": "sum", 'graph_inter_layer_norm': True},
        gnns.sparse_rgdcn_layer,
        {"num_channels": "num_channels", "channel_dim": "channel_dim",
         "use_full_state_for_channel_weights": "use_full_state_for_channel_weights",
         "tie_channel_weights": "tie_channel_weights", "activation_function": "graph_activation_function",
         "message_aggregation_function": "message_aggregation_function"},
        True,
        lambda m, d: gnns.rgdcn_layer_variables(m.task.num_edge_types, m.params['num_channels'], m.params['channel_dim'],
                                                m.params['use_full_state_for_channel_weights'],
                                                m.params['tie_channel_weights']),
        # models/rgdcn_model.py:30: channel_dim is derived before the model is built
        lambda p: p.__setitem__('channel_dim', p['hidden_size'] // p['num_channels'])),
]


def build_adapter(spec: AdapterSpec):
    def default_params(cls):
        params = Sparse_Graph_Model.default_params()
        params.update(spec.param_overrides)
        return params

    def __init__(self, params, task, run_id="run", result_dir=".", device=None):
        if spec.derived_params is not None:
            spec.derived_params(params)
        Sparse_Graph_Model.__init__(self, params, task, run_id, result_dir, device)

    def _gnn_layer_variables(self, in_dim):
        return spec.variables(self, in_dim)

    def _apply_gnn_layer(self, node_representations, adjacency_lists, type_to_num_incoming_edges, num_timesteps):
        kwargs = {kw: self.params[key] for kw, key in spec.layer_kwargs.items()}
        if spec.takes_in_degrees:
            kwargs["type_to_num_incoming_edges"] = type_to_num_incoming_edges
        return spec.layer_fn(node_embeddings=node_representations, adjacency_lists=adjacency_lists,
                             num_timesteps=num_timesteps, weights=self._layer_weights, **kwargs)

    return type(spec.class_name, (Sparse_Graph_Model,), {
        "__doc__": "%s adapter, generated from its AdapterSpec (models/adapters.py)." % spec.class_name,
        "default_params": classmethod(default_params),
        "name": staticmethod(spec.display_name),
        "__init__": __init__,
        "_gnn_layer_variables": _gnn_layer_variables,
        "_apply_gnn_layer": _apply_gnn_layer,
    })


ADAPTER_CLASSES = {spec.class_name: build_adapter(spec) for spec in SPECS}
