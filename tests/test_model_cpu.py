"""Host-side model plumbing that needs no GPU: TF variable naming, the reference's best-model pickle layout
(models/sparse_graph_model.py:91-126) and registries."""
import pickle

import numpy as np
import pytest
import torch


def _task():
    from tf_gnn_samples_amd.tasks import PPI_Task
    t = PPI_Task(PPI_Task.default_params())
    t.load_synthetic(2, 1, mean_nodes=60, std_nodes=5, min_nodes=40, max_nodes=80)
    return t


def test_rgcn_variable_names_and_parameter_count():
    from tf_gnn_samples_amd.models import RGCN_Model
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3)
    m = RGCN_Model(p, _task(), device="cpu")
    names = m.variables.names()
    assert m.variables.num_parameters() == 699257                      # README.md:29 of the reference
    assert "graph_model/dense/kernel" in names                        # input projection 50 -> 256
    assert "graph_model/gnn_layer_0/Dense/kernel" in names            # layer 0 ALWAYS gets the Dense (:194-200)
    assert "graph_model/gnn_layer_1/Dense/kernel" not in names
    assert tuple(m.variables["graph_model/gnn_layer_2/Edge_1_Weight/kernel"].shape) == (256, 256)   # TF layout [in, out]
    assert tuple(m.variables["dense_1/kernel"].shape) == (256, 121)


def test_best_model_pickle_round_trip(tmp_path):
    from tf_gnn_samples_amd.models import GNN_FiLM_Model
    task = _task()
    p = GNN_FiLM_Model.default_params()
    p.update(hidden_size=32, graph_num_layers=2)
    a = GNN_FiLM_Model(p, task, device="cpu")
    path = tmp_path / "best_model.pickle"
    a.save_model(str(path))
    data = pickle.load(open(path, "rb"))
    assert set(data) == {"model_class", "task_class", "model_params", "task_params", "task_metadata", "weights"}
    assert data["model_class"] == "GNN-FiLM" and data["task_class"] == "PPI"
    assert all(k.endswith(":0") and isinstance(v, np.ndarray) for k, v in data["weights"].items())
    p2 = dict(p, random_seed=7)
    b = GNN_FiLM_Model(p2, task, device="cpu")
    assert not torch.equal(a.variables["graph_model/gnn_layer_0/Edge_0_Weight/kernel"],
                           b.variables["graph_model/gnn_layer_0/Edge_0_Weight/kernel"])
    b.load_weights(data["weights"])
    for n in a.variables.names():
        assert torch.equal(a.variables[n], b.variables[n]), n


def test_registries_and_unknown_names():
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import name_to_task_class
    assert name_to_model_class("RGCN")[0].__name__ == "RGCN_Model"
    assert name_to_model_class("GNN-Edge-MLP0")[1] == {'num_edge_hidden_layers': 0}
    assert name_to_model_class("rgdcn")[0].__name__ == "RGDCN_Model"
    assert name_to_task_class("qm9").__name__ == "QM9_Task"
    with pytest.raises(ValueError):
        name_to_model_class("GCN")
    with pytest.raises(ValueError):
        name_to_task_class("varmisuse")


def test_every_model_declares_its_variables_on_cpu():
    from tf_gnn_samples_amd.models import MODEL_CLASSES
    task = _task()
    for cls in set(MODEL_CLASSES.values()):
        p = cls.default_params()
        p.update(hidden_size=32, graph_num_layers=2)
        m = cls(p, task, device="cpu")
        assert m.variables.num_parameters() > 0
        # inter-layer norm gets its own scope when the layer already owns a LayerNorm (TF uniquifies the name)
        if p.get('graph_inter_layer_norm') and "graph_model/gnn_layer_0/LayerNorm/gamma" in m.variables.names() \
                and cls.__name__ in ("RGIN_Model", "GNN_Edge_MLP_Model"):
            assert "graph_model/gnn_layer_0/LayerNorm_1/gamma" in m.variables.names()


def test_effective_cpu_count_is_bounded_by_quota_and_affinity():
    import os
    from tf_gnn_samples_amd.parallel import effective_cpu_count
    n = effective_cpu_count()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert n <= max(1, int(int(quota) / int(period)))
    except OSError:
        pass
