"""Upload half of the native batch builder: batches assembled by relgnn_batch_pack in pinned memory and moved with one
async copy equal the numpy iterator's batches bit for bit on the device, stay valid under arena reuse, and drive the
same training metrics through Sparse_Graph_Model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ppi_task(n_train=6, n_valid=3):
    from tf_gnn_samples_amd.tasks import PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(n_train, n_valid, seed=3, mean_nodes=300.0, std_nodes=80.0, min_nodes=100, max_nodes=500)
    return task


def test_native_batches_equal_numpy_batches_on_device(gpu_device):
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch
    from tf_gnn_samples_amd.tasks.batcher import NativeBatcher
    task = _ppi_task(9, 1)
    data = task._loaded_data[DataFold.TRAIN]
    nb = NativeBatcher(task.make_graph_store(data), gpu_device, num_threads=4)
    want = [DeviceBatch(mb, gpu_device) for mb in task.make_minibatch_iterator(data, DataFold.VALIDATION, 900)]
    got = []
    for b in task.make_native_minibatch_iterator(nb, DataFold.VALIDATION, 900):
        # arenas are reused after `depth` batches: snapshot what the consumer would have read
        got.append((b.num_graphs, b.num_nodes, b.num_edges, b.initial_node_features.clone(),
                    [a.clone() for a in b.adjacency_lists], b.type_to_num_incoming_edges.clone(),
                    b.extra['target_labels'].clone(), b.graph_nodes_list.clone()))
    assert len(got) == len(want) >= 3
    for g, w in zip(got, want):
        assert g[:3] == (w.num_graphs, w.num_nodes, w.num_edges)
        assert torch.equal(g[3], w.initial_node_features)
        for a, b in zip(g[4], w.adjacency_lists):
            assert a.dtype == torch.int32 and torch.equal(a, b)
        assert torch.equal(g[5], w.type_to_num_incoming_edges)
        assert torch.equal(g[6], w.extra['target_labels'])
        assert torch.equal(g[7], w.graph_nodes_list)


def test_training_epoch_metrics_do_not_depend_on_the_batcher(gpu_device):
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold
    results = []
    for native in (True, False):
        task = _ppi_task()
        params = RGCN_Model.default_params()
        params.update(hidden_size=64, graph_num_layers=2, max_nodes_in_batch=900, native_batching=native, random_seed=0)
        model = RGCN_Model(params, task, device=gpu_device)
        valid = task._loaded_data[DataFold.VALIDATION]
        loss, res, n, *_ = model._run_epoch("valid", valid, DataFold.VALIDATION, quiet=True)
        loss2, res2, n2, *_ = model._run_epoch("valid", valid, DataFold.VALIDATION, quiet=True)   # arena + store reuse
        assert loss == loss2 and n == n2
        results.append((loss, [r['f1_score'] for r in res], n))
    assert results[0] == results[1]


def test_qm9_native_batches_train_step(gpu_device):
    import gzip, json, os
    from tf_gnn_samples_amd.models import GGNN_Model
    from tf_gnn_samples_amd.tasks import QM9_Task, DataFold
    here = os.path.dirname(os.path.abspath(__file__))
    task = QM9_Task(QM9_Task.default_params())
    with gzip.open(os.path.join(here, "golden", "qm9_valid_256.jsonl.gz"), "rt") as f:
        raw = [json.loads(line) for line in f]
    data = task.load_raw(raw)
    task._loaded_data[DataFold.TRAIN] = data
    task._loaded_data[DataFold.VALIDATION] = data[:64]
    outs = []
    for native in (True, False):
        params = GGNN_Model.default_params()
        params.update(hidden_size=32, graph_num_layers=2, max_nodes_in_batch=1500, native_batching=native, random_seed=0)
        model = GGNN_Model(params, task, device=gpu_device)
        loss, res, n, *_ = model._run_epoch("valid", data[:64], DataFold.VALIDATION, quiet=True)
        outs.append((loss, n, [r['abs_err_task0'] for r in res]))
    assert outs[0] == outs[1]


def test_train_loop_early_stopping_and_restore(gpu_device, tmp_path):
    """Sparse_Graph_Model.train() end to end on the GPU (native input pipeline, deferred metric fetch, best-model
    pickle in the reference's format, :318-371) and test() on the restored weights."""
    import pickle
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold
    task = _ppi_task(8, 3)
    params = RGCN_Model.default_params()
    params.update(hidden_size=64, graph_num_layers=2, max_nodes_in_batch=900, max_epochs=4, patience=2, random_seed=0)
    model = RGCN_Model(params, task, run_id="t", result_dir=str(tmp_path), device=gpu_device)
    first = model._run_epoch("probe", task._loaded_data[DataFold.VALIDATION], DataFold.VALIDATION, quiet=True)[0]
    model.train(quiet=True)
    with open(model.best_model_file, "rb") as f:
        saved = pickle.load(f)
    assert set(saved) >= {"weights", "model_params", "task_params", "model_class"}      # models/sparse_graph_model.py:91-107
    assert all(k.endswith(":0") for k in saved["weights"])
    restored = RGCN_Model(params, task, run_id="r", result_dir=str(tmp_path), device=gpu_device)
    restored.load_weights(saved["weights"])
    loss = restored._run_epoch("valid", task._loaded_data[DataFold.VALIDATION], DataFold.VALIDATION, quiet=True)[0]
    assert np.isfinite(loss) and loss < first          # training improved the validation loss, restore carries it over
    restored.test(task._loaded_data[DataFold.VALIDATION], quiet=True)
