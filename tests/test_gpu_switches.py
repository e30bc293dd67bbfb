"""Every non-default value of the route switches that an RGCN training step can reach (tf_gnn_samples_amd.config) computes the same
step: loss and every gradient of a 3-layer RGCN + PPI head on a batch tall enough for the limb kernels (>= 4096 nodes), against the
default settings.  The switches of the FiLM / pair / RGAT kernels, the resident-fold assembly and the data-parallel reducer have
their A/B tests next to those kernels (README.md "Switches" names them)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = [dict(gemm="lib"), dict(gemm="panel"), dict(gemm="torch"), dict(limb="pair"), dict(limb="pair", limb_pair_parts="nn"),
            dict(limb="pair", limb_pair_parts="nt,tn"), dict(limb_cut="0"), dict(weight_limb_cache="0"), dict(tn="lib"),
            dict(rgcn_order="transform_first"), dict(agg_acc="f64"), dict(bwd_overlap="0"), dict(gemm="lib", bwd_overlap="1"),
            dict(act_fusion="0"), dict(act_fusion="0", limb="pair"), dict(gemm="lib", act_fusion="1")]


@pytest.fixture(scope="module")
def problem(gpu_device):
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(3, 1, seed=3)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes >= 4096
    return task, mb


def _step(task, mb, dev):
    from tf_gnn_samples_amd.graph import clear_graph_cache
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DeviceBatch
    clear_graph_cache()
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3, graph_layer_input_dropout_keep_prob=1.0, random_seed=0)
    model = RGCN_Model(p, task, device=str(dev))
    batch = DeviceBatch(mb, dev)
    model.optimizer.zero_grad()
    m = model.forward_batch(batch, training=True)
    m['loss'].backward()
    torch.cuda.synchronize()
    return float(m['loss']), {n: model.variables[n].grad.detach().clone() for n in model.variables.names()}


@pytest.fixture(scope="module")
def default_step(gpu_device, problem):
    from tf_gnn_samples_amd import config
    if config.current() != {name: config.default_of(name) for name in config.current()}:
        pytest.skip("this module compares against the DEFAULT settings: RELGNN_* variables are set in this run")
    return _step(*problem, gpu_device)


@pytest.mark.parametrize("switches", VARIANTS, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_non_default_switch_values_compute_the_same_step(gpu_device, problem, default_step, switches):
    from tf_gnn_samples_amd import config
    loss0, grads0 = default_step
    with config.override(**switches):
        loss, grads = _step(*problem, gpu_device)
    assert abs(loss - loss0) <= 2e-6 * max(1.0, abs(loss0)), (loss, loss0)
    # A ReLU unit whose pre-activation lies within the forward error of zero may take the other branch on another arithmetic or
    # association; its gradient path is a rank-one term of ~1e-5 absolute in the EARLIEST variables (the input projection collects
    # the flips of all three layers) — tests/test_gpu_baseline_size.py measures that effect at up to 4.8e-5 on the exact-fp32 route
    # itself.  So: every gradient within 2e-3 of its largest entry and 1e-3 in relative Frobenius norm (a wrong route is O(1) off).
    for n, g0 in grads0.items():
        diff = (grads[n] - g0).double()
        gmax = max(float(g0.abs().max()), 1e-12)
        assert float(diff.abs().max()) <= 2e-3 * gmax, (n, float(diff.abs().max()), gmax)
        assert float(diff.norm()) <= 1e-3 * max(float(g0.double().norm()), 1e-12), (n, float(diff.norm()), float(g0.double().norm()))
