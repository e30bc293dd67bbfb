"""RGAT adapter (models/rgat_model.py): default_params :11-21, _apply_gnn_layer :31-43."""
from typing import Any, Dict, List

import torch

from ..gnns import rgat_layer_variables, sparse_rgat_layer
from .sparse_graph_model import Sparse_Graph_Model


class RGAT_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'hidden_size': 128,
            'num_heads': 4,
            'graph_activation_function': 'tanh',
            'graph_layer_input_dropout_keep_prob': 1.0,
            'graph_dense_between_every_num_gnn_layers': 10000,
            'graph_residual_connection_every_num_layers': 10000,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "RGAT"

    def _gnn_layer_variables(self, in_dim: int):
        return rgat_layer_variables(self.task.num_edge_types, in_dim, self.params['hidden_size'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        return sparse_rgat_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            num_heads=self.params['num_heads'],
            activation_function=self.params['graph_activation_function'],
            weights=self._layer_weights,
        )
