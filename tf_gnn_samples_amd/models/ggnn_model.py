"""GGNN adapter (models/ggnn_model.py): default_params :11-23, _apply_gnn_layer :32-45."""
from typing import Any, Dict, List

import torch

from ..gnns import ggnn_layer_variables, sparse_ggnn_layer
from .sparse_graph_model import Sparse_Graph_Model


class GGNN_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'hidden_size': 128,
            'graph_rnn_cell': 'GRU',  # RNN, GRU, or LSTM
            'graph_activation_function': "tanh",
            "message_aggregation_function": "sum",
            'graph_layer_input_dropout_keep_prob': 1.0,
            'graph_dense_between_every_num_gnn_layers': 10000,
            'graph_residual_connection_every_num_layers': 10000,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "GGNN"

    def _gnn_layer_variables(self, in_dim: int):
        return ggnn_layer_variables(self.task.num_edge_types, in_dim, self.params['hidden_size'],
                                    self.params['graph_rnn_cell'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        return sparse_ggnn_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            gated_unit_type=self.params['graph_rnn_cell'],
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            weights=self._layer_weights,
        )
