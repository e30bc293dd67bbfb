"""Model driver + the seven adapters (table-driven, adapters.py) under the reference's class names."""
from .adapters import ADAPTER_CLASSES
from .sparse_graph_model import Sparse_Graph_Model

globals().update(ADAPTER_CLASSES)     # RGCN_Model, GGNN_Model, RGAT_Model, RGIN_Model, GNN_FiLM_Model, GNN_Edge_MLP_Model, RGDCN_Model

# name lookup of the reference's CLI (utils/model_utils.py:32-57): exactly its accepted spellings (lower-cased first), its
# parameter overrides and its error text — tests/test_reference_run_cpu.py compares with the table the reference's own function returns
MODEL_CLASSES = {
    "ggnn": ADAPTER_CLASSES["GGNN_Model"], "gnn_edge_mlp": ADAPTER_CLASSES["GNN_Edge_MLP_Model"],
    "gnn_film": ADAPTER_CLASSES["GNN_FiLM_Model"], "rgat": ADAPTER_CLASSES["RGAT_Model"],
    "rgcn": ADAPTER_CLASSES["RGCN_Model"], "rgdcn": ADAPTER_CLASSES["RGDCN_Model"], "rgin": ADAPTER_CLASSES["RGIN_Model"],
}
_MODEL_NAMES = {
    "ggnn": ("ggnn", {}), "ggnn_model": ("ggnn", {}),
    "gnn_edge_mlp": ("gnn_edge_mlp", {}), "gnn-edge-mlp": ("gnn_edge_mlp", {}), "gnn_edge_mlp_model": ("gnn_edge_mlp", {}),
    "gnn_edge_mlp0": ("gnn_edge_mlp", {'num_edge_hidden_layers': 0}), "gnn-edge-mlp0": ("gnn_edge_mlp", {'num_edge_hidden_layers': 0}),
    "gnn_edge_mlp0_model": ("gnn_edge_mlp", {'num_edge_hidden_layers': 0}),
    "gnn_edge_mlp1": ("gnn_edge_mlp", {'num_edge_hidden_layers': 1}), "gnn-edge-mlp1": ("gnn_edge_mlp", {'num_edge_hidden_layers': 1}),
    "gnn_edge_mlp1_model": ("gnn_edge_mlp", {'num_edge_hidden_layers': 1}),
    "gnn_film": ("gnn_film", {}), "gnn-film": ("gnn_film", {}), "gnn_film_model": ("gnn_film", {}),
    "rgat": ("rgat", {}), "rgat_model": ("rgat", {}), "rgcn": ("rgcn", {}), "rgcn_model": ("rgcn", {}),
    "rgdcn": ("rgdcn", {}), "rgdcn_model": ("rgdcn", {}), "rgin": ("rgin", {}), "rgin_model": ("rgin", {}),
}


def name_to_model_class(name: str):
    """-> (class, extra default overrides).  'GNN-Edge-MLP0' / 'GNN-Edge-MLP1' select the number of hidden layers."""
    entry = _MODEL_NAMES.get(name.lower())
    if entry is None:
        raise ValueError("Unknown model type '%s'" % name.lower())
    return MODEL_CLASSES[entry[0]], dict(entry[1])


def restore(saved_model_path: str, result_dir: str, run_id: str = None, device=None):
    """utils/model_utils.py:60-77: rebuild task and model from a best-model pickle (the reference's layout — its own save_model
    writes these, sparse_graph_model.py:90-107) and load its weights.  `device` is this package's addition (default: the model's)."""
    import os
    import pickle
    import time
    from ..tasks import name_to_task_class
    print("Loading model from file %s." % saved_model_path)
    with open(saved_model_path, 'rb') as in_file:
        data_to_load = pickle.load(in_file)
    model_cls, _ = name_to_model_class(data_to_load['model_class'])
    task_cls, _ = name_to_task_class(data_to_load['task_class'])
    if run_id is None:
        run_id = "_".join([task_cls.name(), model_cls.name(data_to_load['model_params']), time.strftime("%Y-%m-%d-%H-%M-%S"),
                           str(os.getpid())])
    task = task_cls(data_to_load['task_params'])
    task.restore_from_metadata(data_to_load['task_metadata'])
    model = model_cls(data_to_load['model_params'], task, run_id, result_dir, device=device)
    model.load_weights(data_to_load['weights'])
    model.log_line("Loaded model from snapshot %s." % saved_model_path)
    return model


__all__ = ["Sparse_Graph_Model", "MODEL_CLASSES", "name_to_model_class", "restore"] + sorted(ADAPTER_CLASSES)
