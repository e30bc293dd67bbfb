"""Execution-model properties of the HIP path: works on a non-default stream, is hipGraph-capturable (no hidden
allocation / sync inside the kernels' launch path once the plan exists), and is bit-deterministic (no atomics)."""
import numpy as np
import pytest
import torch

from oracle import gnns as G
from helpers import degree_table, random_relational_graph, rgcn_weights

pytestmark = pytest.mark.gpu


def _setup(dev, V=400, L=3, D=128, seed=0):
    rng = np.random.default_rng(seed)
    adj = random_relational_graph(rng, V, L, [3000, 400, 0])
    adj[1] = np.concatenate([np.stack([np.arange(V), np.arange(V)], 1).astype(np.int32), adj[1]])
    deg = degree_table(adj, V)
    w = rgcn_weights(rng, L, D, D)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    to = lambda a: torch.as_tensor(a, device=dev)
    return adj, deg, w, h, [to(a) for a in adj], to(deg), {k: to(v) for k, v in w.items()}, to(h)


def test_runs_on_a_side_stream(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    adj, deg, w, h, adj_d, deg_d, w_d, h_d = _setup(gpu_device)
    ref = G.sparse_rgcn_layer(h, adj, deg, 128, 2, "tanh", "sum", weights=w)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = sparse_rgcn_layer(h_d, adj_d, deg_d, 128, 2, "tanh", "sum", weights=w_d)
    side.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5


def test_layer_forward_is_hip_graph_capturable(gpu_device):
    """Once the RelGraph (bucketing, needs a workspace allocation) exists, a layer forward is pure kernel launches on
    the capture stream: it can be captured into a hipGraph and replayed on new node states."""
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer, sparse_gnn_film_layer
    from tf_gnn_samples_amd.graph import RelGraph
    from helpers import glorot
    adj, deg, w, h, adj_d, deg_d, w_d, h_d = _setup(gpu_device, seed=1)
    g = RelGraph(adj_d, h.shape[0])
    g.degree_scale(deg_d)                                   # cached per-message weights
    static_in = h_d.clone()
    with torch.no_grad():
        sparse_rgcn_layer(static_in, g, deg_d, 128, 1, "tanh", "sum", weights=w_d)   # warm-up outside capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = sparse_rgcn_layer(static_in, g, deg_d, 128, 1, "tanh", "sum", weights=w_d)
        for seed in (5, 6):
            x = np.tanh(np.random.default_rng(seed).standard_normal(h.shape)).astype(np.float32)
            static_in.copy_(torch.as_tensor(x, device=gpu_device))
            graph.replay()
            torch.cuda.synchronize()
            ref = G.sparse_rgcn_layer(x, adj, deg, 128, 1, "tanh", "sum", weights=w)
            assert np.abs(static_out.cpu().numpy() - ref).max() < 1e-5


@pytest.mark.parametrize("layer", ["rgcn", "rgat", "film"])
def test_bit_deterministic_forward_and_backward(gpu_device, layer):
    from tf_gnn_samples_amd import gnns as H
    from helpers import glorot
    adj, deg, w, h, adj_d, deg_d, w_d, h_d = _setup(gpu_device, D=64, seed=2)
    rng = np.random.default_rng(3)
    L, D = 3, 64
    if layer == "rgat":
        for l in range(L):
            w_d["Edge_%i_Attention_Parameters" % l] = torch.as_tensor((rng.standard_normal(2 * D) * 0.3).astype(np.float32), device=gpu_device)
        fn = lambda x, ww: H.sparse_rgat_layer(x, adj_d, D, 4, 1, "tanh", weights=ww)
    elif layer == "film":
        for l in range(L):
            w_d["Edge_%i_FiLM_Computations/kernel" % l] = torch.as_tensor(glorot(rng, (D, 2 * D)), device=gpu_device)
        for t in ("LayerNorm", "LayerNorm_1"):
            w_d[t + "/gamma"] = torch.ones(D, device=gpu_device)
            w_d[t + "/beta"] = torch.zeros(D, device=gpu_device)
        fn = lambda x, ww: H.sparse_gnn_film_layer(x, adj_d, deg_d, D, 1, "ReLU", "sum", weights=ww)
    else:
        fn = lambda x, ww: H.sparse_rgcn_layer(x, adj_d, deg_d, D, 1, "tanh", "sum", weights=ww)
    runs = []
    for _ in range(2):
        x = h_d.clone().requires_grad_(True)
        ww = {k: v.clone().requires_grad_(True) for k, v in w_d.items()}
        out = fn(x, ww)
        out.square().sum().backward()
        runs.append((out.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in ww.items() if v.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k


def test_int64_and_noncontiguous_adjacency_inputs(gpu_device):
    from tf_gnn_samples_amd.gnns import sparse_rgcn_layer
    adj, deg, w, h, adj_d, deg_d, w_d, h_d = _setup(gpu_device, seed=4)
    ref = G.sparse_rgcn_layer(h, adj, deg, 128, 1, "ReLU", "mean", weights=w)
    adj64 = [a.to(torch.int64) for a in adj_d]
    wide = [torch.cat([a, a], dim=1)[:, :2] for a in adj64]       # non-contiguous views
    out = sparse_rgcn_layer(h_d, wide, deg_d, 128, 1, "ReLU", "mean", weights=w_d)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5


@pytest.mark.parametrize("which", ["rgcn_ppi", "ggnn_qm9"])
def test_captured_train_step_matches_eager_steps(gpu_device, which):
    """Sparse_Graph_Model.capture_train_step: a whole training step (forward, backward, per-variable clip, Adam with the
    step count in device memory) recorded as ONE hipGraph on a fixed batch.  N replays must train exactly like N eager
    steps (same kernels, same order; the only difference is lr_t evaluated in fp32 on the device)."""
    from tf_gnn_samples_amd.models import GGNN_Model, RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task, QM9_Task
    if which == "rgcn_ppi":
        task = PPI_Task(PPI_Task.default_params())
        task.load_synthetic(3, 1, seed=4, mean_nodes=300, std_nodes=50, min_nodes=100, max_nodes=500, fwd_edges_per_node=6.0)
        cls, extra = RGCN_Model, dict(hidden_size=128, graph_num_layers=2)
        data = task._loaded_data[DataFold.TRAIN]
    else:
        from test_golden_cpu import read_qm9_fixture
        task = QM9_Task(QM9_Task.default_params())
        data = task.load_raw(read_qm9_fixture())
        cls, extra = GGNN_Model, dict(hidden_size=64, graph_num_layers=2, graph_rnn_cell="GRU",
                                      message_aggregation_function="mean")
    mb = next(task.make_minibatch_iterator(list(data), DataFold.VALIDATION, 3000))

    def fresh():
        p = cls.default_params()
        p.update(extra)
        p.update(graph_layer_input_dropout_keep_prob=1.0, random_seed=3)
        return cls(p, task, device=str(gpu_device)), DeviceBatch(mb, gpu_device)

    eager, batch_e = fresh()
    losses_e = [float(eager.train_step(batch_e)['loss'].detach()) for _ in range(6)]
    captured, batch_c = fresh()
    step = captured.capture_train_step(batch_c, warmup_steps=3)           # 3 real steps, then the recorded one
    losses_c = [float(step.replay()['loss']) for _ in range(3)]            # steps 4, 5, 6
    torch.cuda.synchronize()
    assert captured.optimizer.t == eager.optimizer.t == 6
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=2e-5)
    for n in eager.variables.names():
        a, b = eager.variables[n].detach().cpu().numpy(), captured.variables[n].detach().cpu().numpy()
        # Adam normalises every gradient element by its own running magnitude: an element whose gradient is at the fp32
        # noise floor moves by O(lr) in a direction the last bit decides, so a handful of elements may differ by ~lr * 1e-2
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=5e-5, err_msg=n)
        assert float(np.mean(np.abs(b - a) > 1e-6)) < 0.01, n
    assert losses_e[-1] < losses_e[0]


def test_backward_overlap_on_a_side_stream_gives_the_same_bits(gpu_device, monkeypatch):
    """RELGNN_BWD_OVERLAP: the aggregate-first layer's weight gradient on a side stream next to the input gradient's gather is the
    same kernels on the same operands — gradients must be bit-identical to the single-stream order, run after run, and the caching
    allocator must not hand the side stream's buffers out early (many iterations, fresh tensors every time)."""
    from tf_gnn_samples_amd import config, ops
    from tf_gnn_samples_amd.graph import RelGraph
    rng = np.random.default_rng(5)
    V, L, D = 6000, 3, 256
    adj = random_relational_graph(rng, V, L, [60000, 6000, 60000])
    g = RelGraph([torch.as_tensor(a, device=gpu_device) for a in adj], V)
    w = None                                   # unweighted sums (the normalisation weights only change the gather's operand)
    H0 = torch.as_tensor(rng.standard_normal((V, D)).astype(np.float32), device=gpu_device)
    W0 = torch.as_tensor((rng.standard_normal((L, D, D)) * 0.05).astype(np.float32), device=gpu_device)
    gout = torch.as_tensor(rng.standard_normal((V, D)).astype(np.float32), device=gpu_device)

    def grads(overlap):
        monkeypatch.setattr(config.settings, "bwd_overlap", "1" if overlap else "0")
        H, W = H0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        out = ops.aggregate_then_transform(H, W, g, w, "sum", "relu")
        out.backward(gout)
        return H.grad.clone(), W.grad.clone()

    base = grads(False)
    for _ in range(20):
        got = grads(True)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1])
        junk = [torch.empty((V, D), device=gpu_device).normal_() for _ in range(3)]      # churn the allocator between iterations
        del junk



def test_captured_step_on_the_limb_route_and_eager_code_after_replays(gpu_device):
    """A batch tall enough for the limb route (>= 4096 nodes: products from weight limb images, weight gradient on the side
    stream): under capture no image is cached (a replay re-runs kernels, not the host code that would refresh one), and eager
    code behind a replay must see the replayed update (dense.weights_changed), not an image from before it."""
    from tf_gnn_samples_amd import dense as DN
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(3, 1, seed=4)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    from tf_gnn_samples_amd import config
    if not config.settings.limb_gemm:
        pytest.skip("the limb route is what this test is about (RELGNN_GEMM is set to another route in this run)")
    assert mb.num_nodes >= 4096

    def fresh():
        p = RGCN_Model.default_params()
        p.update(hidden_size=256, graph_num_layers=2, graph_layer_input_dropout_keep_prob=1.0, random_seed=3)
        return RGCN_Model(p, task, device=str(gpu_device)), DeviceBatch(mb, gpu_device)

    eager, batch_e = fresh()
    losses_e = [float(eager.train_step(batch_e)['loss'].detach()) for _ in range(6)]
    with torch.no_grad():
        eval_e = float(eager.forward_batch(batch_e, training=False)['loss'])
    captured, batch_c = fresh()
    step = captured.capture_train_step(batch_c, warmup_steps=3)
    with torch.no_grad():
        captured.forward_batch(batch_c, training=False)                    # fills the image cache from the weights after step 3
    losses_c = [float(step.replay()['loss']) for _ in range(3)]
    with torch.no_grad():
        eval_c = float(captured.forward_batch(batch_c, training=False)['loss'])
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=2e-5)
    np.testing.assert_allclose(eval_c, eval_e, rtol=2e-5)                  # (a stale image would give the loss after step 3)
    assert abs(eval_e - losses_e[3]) > 1e-3 * abs(eval_e)


def test_deferred_weight_gradients_only_go_aside_where_nothing_reads_them_early(gpu_device):
    """ops.deferred_weight_gradient_join (train_step): a Dense layer's weight gradient may stay in flight on the side stream until
    join_deferred() only when it goes straight into a leaf parameter that is seen once in the backward pass.  The gradient of a VIEW
    of a parameter (the GRU's recurrent_kernel[:, :2u]) is consumed by the view's backward on the main stream at once, and a
    parameter used twice has its contributions summed there: both must stay on one stream — and give the bits of the plain order."""
    from tf_gnn_samples_amd import config, dense as DN, ops
    torch.manual_seed(0)
    x = torch.randn(6000, 128, device=gpu_device)
    U0 = torch.randn(128, 384, device=gpu_device) * 0.05
    b0 = torch.randn(256, device=gpu_device) * 0.1

    def run(deferred, fn):
        x_ = x.clone().requires_grad_(True)
        U = U0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        y = fn(x_, U, b)
        pend = None
        with config.override(bwd_overlap="1"):
            if deferred:
                with ops.deferred_weight_gradient_join():
                    y.square().sum().backward()
                pend = len(ops._DEFER["pending"])
                ops.join_deferred()
                assert not ops._DEFER["pending"] and not ops._DEFER["targets"] and not ops._DEFER["handed"]
            else:
                y.square().sum().backward()
        torch.cuda.synchronize()
        return pend, [x_.grad.clone(), U.grad.clone(), b.grad.clone()]

    # (1) a leaf kernel seen once: goes aside (one pending join); (2) a view of the parameter: stays; (3) the leaf used twice: the
    # second sight stays and makes the main stream wait
    fns = [
        (lambda x_, U, b: DN.dense(torch.tanh(x_), U, None)[:, :256] + b, 1),
        (lambda x_, U, b: DN.dense(torch.tanh(x_), U[:, :256], b), 0),
        (lambda x_, U, b: DN.dense(torch.tanh(DN.dense(torch.tanh(x_), U, None)[:, :128]), U, None)[:, :256] + b, 1),
    ]
    for fn, expected_pending in fns:
        _, want = run(False, fn)
        for _ in range(3):
            pend, got = run(True, fn)
            assert pend == expected_pending
            for a, b_ in zip(want, got):
                assert torch.equal(a, b_)


def test_deferred_weight_gradients_stay_on_the_main_stream_whenever_autograd_would_touch_them_early(gpu_device):
    """The other ways autograd reads a gradient on the main stream before join_deferred(): a tensor hook on the
    parameter (the accumulator copies), a post-accumulate hook, create_graph, anomaly mode, a second use by a Dense whose input
    needs no gradient (its contribution is summed on the main stream: it must wait for the side stream first).  None of them may go
    aside; all give the bits of the plain order.  And a use the package cannot see (a plain torch op on the same leaf whose
    gradient arrives AFTER the deferred one) is reported by join_deferred() instead of racing silently."""
    from tf_gnn_samples_amd import config, dense as DN, ops
    torch.manual_seed(1)
    x = torch.randn(6000, 128, device=gpu_device)
    U0 = torch.randn(128, 256, device=gpu_device) * 0.05

    def backward(fn, prepare=None, deferred=True, **backward_kw):
        x_ = x.clone().requires_grad_(True)
        U = U0.clone().requires_grad_(True)
        if prepare is not None:
            prepare(U)
        y = fn(x_, U)
        with config.override(bwd_overlap="1"):
            if deferred:
                with ops.deferred_weight_gradient_join():
                    y.square().sum().backward(**backward_kw)
                pend = len(ops._DEFER["pending"])
                ops.join_deferred()
            else:
                pend = None
                y.square().sum().backward(**backward_kw)
        torch.cuda.synchronize()
        return pend, [x_.grad.detach().clone(), U.grad.detach().clone()]

    one = lambda x_, U: DN.dense(torch.tanh(x_), U, None)
    _, want = backward(one, deferred=False)
    pend, got = backward(one)
    assert pend == 1 and all(torch.equal(a, b) for a, b in zip(want, got))
    seen = []
    cases = {
        "tensor hook": dict(prepare=lambda U: U.register_hook(lambda g: g)),
        "post-accumulate hook": dict(prepare=lambda U: U.register_post_accumulate_grad_hook(lambda p: seen.append(float(p.grad.sum())))),
        "create_graph": dict(create_graph=True),
    }
    for name, kw in cases.items():
        pend, got = backward(one, **kw)
        assert pend == 0, name
        assert all(torch.equal(a, b) for a, b in zip(want, got)), name
    assert len(seen) == 1 and seen[0] == float(want[1].sum())
    with torch.autograd.detect_anomaly(check_nan=False):
        pend, got = backward(one)
    assert pend == 0 and all(torch.equal(a, b) for a, b in zip(want, got))

    # second use through a Dense whose input needs no gradient: if the other use comes first in the backward it goes aside and this
    # one is produced on the main stream behind a wait and summed there; if this one comes first, nothing goes aside at all
    const = torch.randn(6000, 128, device=gpu_device)
    two = lambda x_, U: DN.dense(torch.tanh(x_), U, None) + DN.dense(const, U, None)
    _, want2 = backward(two, deferred=False)
    for _ in range(3):
        pend, got = backward(two)
        assert pend in (0, 1)               # (which of the two uses autograd runs first is its choice: the first one may go aside
        assert all(torch.equal(a, b) for a, b in zip(want2, got))   #  only if it is the one whose input needs a gradient)

    # a use the package cannot see: a plain torch product on the same leaf, its gradient accumulated in place on the main stream
    # behind the deferred one -> join_deferred() raises (the order of the two contributions is autograd's: accept either outcome
    # of the ordering, but never a silent pass with pending work and an in-place write)
    hidden = lambda x_, U: DN.dense(torch.tanh(x_), U, None) + torch.sin(const @ U)
    try:
        pend, got = backward(hidden)
    except RuntimeError as e:
        assert "deferred weight-gradient join" in str(e)
    else:
        assert pend == 0                    # (the plain op's gradient arrived first: nothing went aside)
    assert not ops._DEFER["pending"] and not ops._DEFER["targets"] and not ops._DEFER["handed"]
    torch.cuda.synchronize()


@pytest.mark.parametrize("which", ["film_many_types", "ggnn_d128"])
def test_captured_step_of_the_d128_models_reuses_its_limb_images_inside_the_capture(gpu_device, which):
    """Round 6: the 128-column panel products (typed transforms of a 23-type GNN-FiLM model incl. the wave-role kernel for the
    FiLM weights; the GGNN transform through dense_multi and its GRU) take their weights' limb images from a cache — inside a
    captured step from the capture-local one (dense.capture_image_cache: split once in the forward, reused by the backward, dropped
    by the captured optimizer update).  N replays must train like N eager steps, and eager code behind a replay sees the update."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from tf_gnn_samples_amd import config, ops
    from tf_gnn_samples_amd.models import name_to_model_class
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    if not config.settings.limb_gemm:
        pytest.skip("the limb route is what this test is about")
    if which == "film_many_types":
        import bench_other
        task, graphs = bench_other.c5_task_and_graphs(4)
        cls, extra = name_to_model_class("GNN-FiLM")
        hp = dict(hidden_size=128, graph_num_layers=2, graph_dense_between_every_num_gnn_layers=1,
                  graph_residual_connection_every_num_layers=2)
    else:
        task = PPI_Task(PPI_Task.default_params())
        task.load_synthetic(3, 1, seed=4)
        graphs = task._loaded_data[DataFold.TRAIN]
        cls, extra = name_to_model_class("GGNN")
        hp = dict(hidden_size=128, graph_num_layers=2, graph_rnn_cell="GRU", message_aggregation_function="sum")
    mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
    assert mb.num_nodes >= 4096

    def fresh():
        p = cls.default_params()
        p.update(extra)
        p.update(hp)
        p.update(graph_layer_input_dropout_keep_prob=1.0, random_seed=3)
        return cls(p, task, device=str(gpu_device)), DeviceBatch(mb, gpu_device)

    eager, batch_e = fresh()
    losses_e = [float(eager.train_step(batch_e)['loss'].detach()) for _ in range(6)]
    with torch.no_grad():
        eval_e = float(eager.forward_batch(batch_e, training=False)['loss'])
    captured, batch_c = fresh()
    step = captured.capture_train_step(batch_c, warmup_steps=3)
    losses_c = [float(step.replay()['loss']) for _ in range(3)]
    with torch.no_grad():
        eval_c = float(captured.forward_batch(batch_c, training=False)['loss'])
    torch.cuda.synchronize()
    assert step.handover_status() == 0 and ops.handover_status() == 0
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=5e-5)
    np.testing.assert_allclose(eval_c, eval_e, rtol=5e-5)
