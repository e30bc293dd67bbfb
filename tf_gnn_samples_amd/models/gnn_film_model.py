"""GNN-FiLM adapter (models/gnn_film_model.py): default_params :11-19, _apply_gnn_layer :29-43."""
from typing import Any, Dict, List

import torch

from ..gnns import gnn_film_layer_variables, sparse_gnn_film_layer
from .sparse_graph_model import Sparse_Graph_Model


class GNN_FiLM_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            "hidden_size": 128,
            "graph_activation_function": "ReLU",
            "message_aggregation_function": "sum",
            "normalize_messages_by_num_incoming": False,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "GNN-FiLM"

    def _gnn_layer_variables(self, in_dim: int):
        return gnn_film_layer_variables(self.task.num_edge_types, in_dim, self.params['hidden_size'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        return sparse_gnn_film_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            type_to_num_incoming_edges=type_to_num_incoming_edges,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            normalize_by_num_incoming=self.params["normalize_messages_by_num_incoming"],
            weights=self._layer_weights,
        )
