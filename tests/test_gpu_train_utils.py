"""Fused training-step plumbing (csrc/train_utils.hip) vs the un-fused restatements."""
import numpy as np
import pytest
import torch

from oracle import model as OM

pytestmark = pytest.mark.gpu


def test_fused_clip_adam_matches_unfused_tf_rules(gpu_device):
    from tf_gnn_samples_amd.models.sparse_graph_model import TFStyleOptimizer
    torch.manual_seed(0)
    shapes = [(50, 256), (256, 768), (256, 256), (121,), (1,), (256, 121), (3, 5, 7)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, device=gpu_device)) for s in shapes]
    pa = mk()
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = TFStyleOptimizer(pa, "Adam", 1e-3, 1.0)
    ob = TFStyleOptimizer(pb, "Adam", 1e-3, 1.0)
    for step in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = torch.randn_like(a) * (5.0 if i % 2 == 0 else 0.01)   # some clipped, some not
            a.grad, b.grad = g.clone(), g.clone()
        pb[4].grad = None
        pa[4].grad = None                                              # variable without gradient: untouched
        oa.clip_and_step(lr_scale=0.5)                                 # fused
        ob.clip_gradients(); ob.step(lr_scale=0.5)                     # foreach reference
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert oa.t == ob.t == 4


@pytest.mark.parametrize("name", ["Adam", "RMSProp", "SGD"])
def test_device_optimizer_matches_the_numpy_restatement_of_the_tf_rules(gpu_device, name):
    """The GPU update path (Adam: the two fused multi-tensor HIP launches relgnn_mt_l2norm + relgnn_mt_adam_clip;
    RMSProp / SGD: foreach kernels) against oracle/optim.py — an independent restatement of tf.clip_by_norm and the TF1
    ApplyAdam / ApplyRMSProp / ApplyGradientDescent rules (models/sparse_graph_model.py:227-260)."""
    from oracle import optim as O
    from tf_gnn_samples_amd.models.sparse_graph_model import TFStyleOptimizer
    rng = np.random.default_rng(11)
    shapes = [(50, 256), (256, 768), (121,), (1,), (256, 121), (3, 5, 7)]
    vs = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    params = [torch.nn.Parameter(torch.as_tensor(v, device=gpu_device)) for v in vs]
    opt = TFStyleOptimizer(params, name, 1e-3, 1.0, decay=0.98, momentum=0.85)
    ref = O.make_optimizer(name, vs, 1e-3, decay=0.98, momentum=0.85)
    for step in range(5):
        gs = [(rng.standard_normal(s) * (5.0 if (i + step) % 2 == 0 else 0.01)).astype(np.float32) for i, s in enumerate(shapes)]
        if step == 1:
            gs[3] = None
        for p, g in zip(params, gs):
            p.grad = None if g is None else torch.as_tensor(g, device=gpu_device)
        scale = 0.5 if step == 3 else 1.0
        opt.clip_and_step(scale)
        O.train_step(ref, gs, 1.0, scale)
        for p, r in zip(params, ref.vars):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=3e-6, atol=3e-7)


def test_sigmoid_ce_stats_and_gradient(gpu_device):
    from tf_gnn_samples_amd.tasks.ppi_task import _SigmoidCEStats
    from tf_gnn_samples_amd.utils import micro_f1
    rng = np.random.default_rng(0)
    V, N = 3000, 121
    x = (rng.standard_normal((V, N)) * 3).astype(np.float32)
    x[0, 0] = 0.0
    x[0, 1] = 40.0
    x[0, 2] = -40.0
    z = (rng.random((V, N)) < 0.3).astype(np.float32)
    xd = torch.as_tensor(x, device=gpu_device).requires_grad_(True)
    zd = torch.as_tensor(z, device=gpu_device)
    mean_loss, total_loss, f1, counts = _SigmoidCEStats.apply(xd, zd, 1.0 / V)
    mean_loss.backward()
    stats = [float(total_loss), *[float(c) for c in counts], float(f1)]
    ref = OM.sigmoid_cross_entropy_with_logits(x.astype(np.float64), z.astype(np.float64))
    assert abs(float(stats[0]) - ref.sum()) < 1e-5 * ref.sum()
    assert abs(float(mean_loss) - ref.sum() / V) < 1e-5 * ref.sum() / V
    pred = np.round(1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.int32)
    zi = z.astype(np.int32)
    tp, fp, fn = np.count_nonzero(pred * zi), np.count_nonzero(pred * (zi - 1)), np.count_nonzero((pred - 1) * zi)
    assert [int(stats[1]), int(stats[2]), int(stats[3])] == [tp, fp, fn]
    p, r = tp / (tp + fp), tp / (tp + fn)
    assert abs(float(stats[4]) - 2 * p * r / (p + r)) < 1e-6
    assert abs(float(stats[4]) - float(micro_f1(xd.detach(), zd))) < 1e-6
    gref = (1.0 / (1.0 + np.exp(-x.astype(np.float64))) - z) / V
    assert np.abs(xd.grad.cpu().numpy() - gref).max() < 1e-7
    # the summed loss is differentiable too (DP scales it by the global node count), and both at once
    xd2 = torch.as_tensor(x, device=gpu_device).requires_grad_(True)
    m2, t2, _, _ = _SigmoidCEStats.apply(xd2, zd, 1.0 / V)
    (t2 * (0.5 / V) + m2 * 0.5).backward()
    assert np.abs(xd2.grad.cpu().numpy() - gref).max() < 1e-7


@pytest.mark.parametrize("rows,D", [(1, 64), (7, 128), (5000, 128), (3001, 256), (513, 320), (257, 1000), (100, 32), (64, 6)])
def test_layer_norm_kernel_matches_fp64_reference(gpu_device, rows, D):
    """csrc/layer_norm.hip (tf.contrib.layers.layer_norm semantics: biased variance, eps 1e-12) forward and backward
    against a float64 restatement; D = 6 takes the library fallback (not a multiple of 4)."""
    from tf_gnn_samples_amd.utils import layer_norm
    rng = np.random.default_rng(rows * 31 + D)
    x = (rng.standard_normal((rows, D)) * 2.0 + 0.5).astype(np.float32)
    gamma = (1.0 + 0.2 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(D)).astype(np.float32)
    g = rng.standard_normal((rows, D)).astype(np.float32)
    xd = torch.as_tensor(x, device=gpu_device).requires_grad_(True)
    gd = torch.as_tensor(gamma, device=gpu_device).requires_grad_(True)
    bd = torch.as_tensor(beta, device=gpu_device).requires_grad_(True)
    y = layer_norm(xd, gd, bd)
    y.backward(torch.as_tensor(g, device=gpu_device))
    xr = torch.as_tensor(x, dtype=torch.float64).requires_grad_(True)
    gr = torch.as_tensor(gamma, dtype=torch.float64).requires_grad_(True)
    br = torch.as_tensor(beta, dtype=torch.float64).requires_grad_(True)
    mean = xr.mean(1, keepdim=True)
    var = ((xr - mean) ** 2).mean(1, keepdim=True)
    yr = (xr - mean) / torch.sqrt(var + 1e-12) * gr + br
    yr.backward(torch.as_tensor(g, dtype=torch.float64))
    assert np.abs(y.detach().cpu().numpy() - yr.detach().numpy()).max() < 1e-5
    assert np.abs(xd.grad.cpu().numpy() - xr.grad.numpy()).max() < 1e-5 * max(1.0, float(xr.grad.abs().max()))
    assert np.abs(gd.grad.cpu().numpy() - gr.grad.numpy()).max() < 2e-5 * max(1.0, float(gr.grad.abs().max()))
    assert np.abs(bd.grad.cpu().numpy() - br.grad.numpy()).max() < 2e-5 * max(1.0, float(br.grad.abs().max()))


def test_layer_norm_constant_rows_and_determinism(gpu_device):
    from tf_gnn_samples_amd.utils import layer_norm
    x = torch.full((33, 128), 3.25, device=gpu_device)
    gamma = torch.ones(128, device=gpu_device); beta = torch.full((128,), 0.5, device=gpu_device)
    y = layer_norm(x, gamma, beta)
    assert torch.equal(y, torch.full_like(y, 0.5))          # zero variance: (x - mean) == 0 exactly, eps keeps it finite
    a = torch.randn(1000, 256, device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(0))
    outs = []
    for _ in range(2):
        xa = a.clone().requires_grad_(True); ga = gamma.new_ones(256).requires_grad_(True); ba = gamma.new_zeros(256).requires_grad_(True)
        ya = layer_norm(xa, ga, ba); ya.square().sum().backward()
        outs.append((ya.detach(), xa.grad, ga.grad, ba.grad))
    assert all(torch.equal(p, q) for p, q in zip(*outs))


def test_metrics_readback_is_asynchronous_and_ordered(gpu_device):
    """MetricsReadback: the values are the ones the metrics held when the read-back was created (the copy is enqueued right
    behind the step), several may be in flight at once without sharing a pinned slot, get() is idempotent, mixed dtypes."""
    from tf_gnn_samples_amd.models.sparse_graph_model import MetricsReadback
    x = torch.zeros((), device=gpu_device)
    cnt = torch.zeros((), dtype=torch.int64, device=gpu_device)
    pending = []
    for i in range(6):
        x.add_(1.25)
        cnt.add_(3)
        pending.append(MetricsReadback({"loss": x, "count": cnt, "step": i, "half": x * 0.5}))
    for i, rb in enumerate(pending):
        want = {"loss": 1.25 * (i + 1), "count": 3.0 * (i + 1), "step": i, "half": 0.625 * (i + 1)}
        assert rb.get() == want
        assert rb.get() == want
    assert MetricsReadback({}).get() == {}
    slots = sum(len(v) for v in MetricsReadback._ring.values())
    more = [MetricsReadback({"loss": x}) for _ in range(3)]
    [m.get() for m in more]
    assert sum(len(v) for v in MetricsReadback._ring.values()) <= slots + 3       # slots are recycled after get()
