"""GNN-Edge-MLP adapter (models/gnn_edge_mlp_model.py): default_params :11-23, name :25-27,
_apply_gnn_layer :32-48."""
from typing import Any, Dict, List

import torch

from ..gnns import gnn_edge_mlp_layer_variables, sparse_gnn_edge_mlp_layer
from .sparse_graph_model import Sparse_Graph_Model


class GNN_Edge_MLP_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'max_nodes_in_batch': 25000,
            'hidden_size': 128,
            "graph_activation_function": "gelu",
            "message_aggregation_function": "sum",
            'graph_inter_layer_norm': True,
            'use_target_state_as_input': True,
            'num_edge_hidden_layers': 1,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "GNN-Edge-MLP%i" % (params['num_edge_hidden_layers'])

    def _gnn_layer_variables(self, in_dim: int):
        p = self.params
        return gnn_edge_mlp_layer_variables(self.task.num_edge_types, in_dim, p['hidden_size'],
                                            p['use_target_state_as_input'], p['num_edge_hidden_layers'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int,
                         ) -> torch.Tensor:
        # NB: like the reference adapter, normalize_by_num_incoming is not passed: layer default False.
        return sparse_gnn_edge_mlp_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            type_to_num_incoming_edges=type_to_num_incoming_edges,
            state_dim=self.params['hidden_size'],
            num_timesteps=num_timesteps,
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            use_target_state_as_input=self.params['use_target_state_as_input'],
            num_edge_hidden_layers=self.params['num_edge_hidden_layers'],
            weights=self._layer_weights,
        )
