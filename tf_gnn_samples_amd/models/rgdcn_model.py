"""RGDCN adapter (models/rgdcn_model.py): default_params :11-23, channel_dim derivation :30, _apply_gnn_layer :33-49."""
from typing import Any, Dict, List

import torch

from ..gnns import rgdcn_layer_variables, sparse_rgdcn_layer
from .sparse_graph_model import Sparse_Graph_Model


class RGDCN_Model(Sparse_Graph_Model):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'max_nodes_in_batch': 25000,
            'hidden_size': 128,
            'num_channels': 8,
            "use_full_state_for_channel_weights": False,
            "tie_channel_weights": False,
            "graph_activation_function": "ReLU",
            "message_aggregation_function": "sum",
            'graph_inter_layer_norm': True,
        })
        return params

    @staticmethod
    def name(params: Dict[str, Any]) -> str:
        return "RGDCN"

    def __init__(self, params: Dict[str, Any], task, run_id: str = "run", result_dir: str = ".", device=None) -> None:
        params['channel_dim'] = params['hidden_size'] // params['num_channels']
        super().__init__(params, task, run_id, result_dir, device)

    def _gnn_layer_variables(self, in_dim: int):
        p = self.params
        return rgdcn_layer_variables(self.task.num_edge_types, p['num_channels'], p['channel_dim'],
                                     p['use_full_state_for_channel_weights'], p['tie_channel_weights'])

    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        return sparse_rgdcn_layer(
            node_embeddings=node_representations,
            adjacency_lists=adjacency_lists,
            type_to_num_incoming_edges=type_to_num_incoming_edges,
            num_channels=self.params['num_channels'],
            channel_dim=self.params['channel_dim'],
            num_timesteps=num_timesteps,
            use_full_state_for_channel_weights=self.params['use_full_state_for_channel_weights'],
            tie_channel_weights=self.params['tie_channel_weights'],
            activation_function=self.params['graph_activation_function'],
            message_aggregation_function=self.params['message_aggregation_function'],
            weights=self._layer_weights,
        )
