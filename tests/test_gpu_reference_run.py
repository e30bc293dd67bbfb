"""The HIP path against fixtures written by the REFERENCE'S OWN layer code (tests/golden/make_reference_run.py: the unmodified
gnns/*.py executed over a NumPy shim of the TF ops; tests/golden/tf_numpy_shim.py says what that pins).  Same inputs, the variables the
reference created under their TF names, its outputs; the layer functions of the package take the reference's arguments."""
import numpy as np
import pytest
import torch

from test_reference_run_cpu import CASES, call_layer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(len(CASES)), ids=["%s-%d" % (c[0]["function"], n) for n, c in enumerate(CASES)])
def test_hip_layers_reproduce_the_reference_run(gpu_device, i):
    from tf_gnn_samples_amd import gnns
    case, h, adj, deg, weights, want = CASES[i]

    def to_device(x):
        x = np.asarray(x)
        return torch.as_tensor(x, device=gpu_device)

    got = call_layer(gnns, case, h, adj, deg, weights, convert=to_device)
    assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
    err = float(np.abs(got.cpu().numpy() - want).max())
    assert err <= 1e-5 * max(1.0, float(np.abs(want).max())), (case["function"], case["kwargs"], err)


from test_reference_run_cpu import MODEL_CASES, MODEL_Z, build_product_model, build_product_task  # noqa: E402


@pytest.mark.parametrize("i", range(len(MODEL_CASES)), ids=["%s-%s-%d" % (m["model"], m["task"], n) for n, m in enumerate(MODEL_CASES)])
def test_hip_models_reproduce_the_reference_s_forward_model(gpu_device, tmp_path, i):
    """Every model class on the minibatch the reference's iterator built, with the variables the reference's __make_model created:
    final node representations and the task metrics (loss, total_loss, F1 / absolute errors) of the reference's own forward code."""
    from tf_gnn_samples_amd.tasks import DeviceBatch, MinibatchData
    entry, z = MODEL_CASES[i], MODEL_Z
    k = entry["key"]
    task = build_product_task(entry, tmp_path)
    model = build_product_model(entry, task, str(gpu_device))
    with torch.no_grad():
        for n in entry["variables"]:
            model.variables[n].copy_(torch.as_tensor(z["%s/var/%s" % (k, n)], device=gpu_device))
    payload = entry["payload"]
    feed = {'initial_node_features': z[k + "/features"], 'type_to_num_incoming_edges': z[k + "/deg"],
            'graph_nodes_list': z[k + "/graph_nodes_list"], payload: z[k + "/" + payload], 'out_layer_dropout_keep_prob': 1.0,
            'adjacency_lists': [z["%s/adj%d" % (k, l)] for l in range(entry["num_edge_types"])]}
    mb = MinibatchData(feed_dict=feed, num_graphs=entry["num_graphs"], num_nodes=entry["num_nodes"], num_edges=entry["num_edges"])
    batch = DeviceBatch(mb, gpu_device)
    with torch.no_grad():
        final = model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists,
                                                         batch.type_to_num_incoming_edges)
        metrics = model.forward_batch(batch, training=False)
    want = z[k + "/final_node_representations"]
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(final.cpu().numpy() - want).max()) <= 1e-5 * scale
    for name, value in entry["metrics"].items():
        got = float(metrics[name])
        tol = 1e-6 if name == "f1_score" else 2e-5 * max(1.0, abs(value), scale * (entry["num_nodes"] if "total" in name or "abs_err" in name else 1))
        assert abs(got - value) <= tol, (name, got, value)
