"""Model driver + the seven adapters (table-driven, adapters.py) under the reference's class names."""
from .adapters import ADAPTER_CLASSES
from .sparse_graph_model import Sparse_Graph_Model

globals().update(ADAPTER_CLASSES)     # RGCN_Model, GGNN_Model, RGAT_Model, RGIN_Model, GNN_FiLM_Model, GNN_Edge_MLP_Model, RGDCN_Model

# name lookup of the reference's CLI (utils/model_utils.py:32-55), lower-cased; "-"/"_" interchangeable
MODEL_CLASSES = {
    "ggnn": ADAPTER_CLASSES["GGNN_Model"], "gnn_edge_mlp": ADAPTER_CLASSES["GNN_Edge_MLP_Model"],
    "gnn_film": ADAPTER_CLASSES["GNN_FiLM_Model"], "rgat": ADAPTER_CLASSES["RGAT_Model"],
    "rgcn": ADAPTER_CLASSES["RGCN_Model"], "rgdcn": ADAPTER_CLASSES["RGDCN_Model"], "rgin": ADAPTER_CLASSES["RGIN_Model"],
}


def name_to_model_class(name: str):
    """-> (class, extra default overrides).  'GNN-Edge-MLP0' / 'GNN-Edge-MLP1' select the number of hidden layers."""
    key = name.lower().replace("-", "_")
    extra = {}
    if key in ("gnn_edge_mlp0", "gnn_edge_mlp1"):
        extra = {'num_edge_hidden_layers': int(key[-1])}
        key = "gnn_edge_mlp"
    if key not in MODEL_CLASSES:
        raise ValueError("Unknown model '%s'!" % name)
    return MODEL_CLASSES[key], extra


__all__ = ["Sparse_Graph_Model", "MODEL_CLASSES", "name_to_model_class"] + sorted(ADAPTER_CLASSES)
