"""RGCN adapter (models/rgcn_model.py): default_params :11-22, name :24-26, _apply_gnn_layer :31-44."""
from typing import Any, Dict, List

import torch

from ..gnns import rgcn_layer_variables, sparse_rgcn_layer
from .sparse_graph_model import Sparse_Graph_Model


class RGCN_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'hidden_size': 128,
            "graph_activation_function": "ReLU",
            "message_aggregation_function": "sum",
            'graph_layer_input_dropout_keep_prob': 1.0,
            'graph_dense_between_every_num_gnn_layers': 10000,
            'graph_residual_connection_every_num_layers': 10000,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "RGCN"

    def _gnn_layer_variables(self, in_dim: int):
        return rgcn_layer_variables(self.task.num_edge_types, in_dim, self.params['hidden_size'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        # NB: like the reference adapter, normalize_by_num_incoming is not passed: layer default True.
        return sparse_rgcn_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            type_to_num_incoming_edges=type_to_num_incoming_edges,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            weights=self._layer_weights,
        )
