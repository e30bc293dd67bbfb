"""RGIN adapter (models/rgin_model.py): default_params :11-24, _apply_gnn_layer :33-49."""
from typing import Any, Dict, List

import torch

from ..gnns import rgin_layer_variables, sparse_rgin_layer
from .sparse_graph_model import Sparse_Graph_Model


class RGIN_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'hidden_size': 128,
            "graph_activation_function": "ReLU",
            'message_aggregation_function': "sum",
            'graph_dense_between_every_num_gnn_layers': 10000,
            'graph_inter_layer_norm': True,
            'use_target_state_as_input': False,
            'graph_num_edge_MLP_hidden_layers': 1,
            'graph_num_aggr_MLP_hidden_layers': None,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "RGIN"

    def _gnn_layer_variables(self, in_dim: int):
        p = self.params
        return rgin_layer_variables(self.task.num_edge_types, in_dim, p['hidden_size'], p['use_target_state_as_input'],
                                    p['graph_num_edge_MLP_hidden_layers'], p['graph_num_aggr_MLP_hidden_layers'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int,
                         ) -> torch.Tensor:
        return sparse_rgin_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            use_target_state_as_input=self.params['use_target_state_as_input'],
            num_edge_MLP_hidden_layers=self.params['graph_num_edge_MLP_hidden_layers'],
            num_aggr_MLP_hidden_layers=self.params['graph_num_aggr_MLP_hidden_layers'],
            weights=self._layer_weights,
        )
