"""Model adapters (models/__init__.py of the reference)."""
from .sparse_graph_model import Sparse_Graph_Model
from .ggnn_model import GGNN_Model
from .gnn_edge_mlp_model import GNN_Edge_MLP_Model
from .gnn_film_model import GNN_FiLM_Model
from .rgat_model import RGAT_Model
from .rgcn_model import RGCN_Model
from .rgdcn_model import RGDCN_Model
from .rgin_model import RGIN_Model

MODEL_CLASSES = {
    # utils/model_utils.py:32-55 (name_to_model_class), lower-cased names
    "ggnn": GGNN_Model, "gnn_edge_mlp": GNN_Edge_MLP_Model, "gnn-edge-mlp": GNN_Edge_MLP_Model,
    "gnn_film": GNN_FiLM_Model, "gnn-film": GNN_FiLM_Model, "rgat": RGAT_Model, "rgcn": RGCN_Model, "rgdcn": RGDCN_Model, "rgin": RGIN_Model,
}


def name_to_model_class(name: str):
    key = name.lower()
    extra = {}
    if key in ("gnn-edge-mlp0", "gnn_edge_mlp0"):
        key, extra = "gnn_edge_mlp", {'num_edge_hidden_layers': 0}
    if key in ("gnn-edge-mlp1", "gnn_edge_mlp1"):
        key, extra = "gnn_edge_mlp", {'num_edge_hidden_layers': 1}
    if key not in MODEL_CLASSES:
        raise ValueError("Unknown model '%s'!" % name)
    return MODEL_CLASSES[key], extra
