"""Data parallelism by graph through the HIP path: two ranks, each runs the package's RGCN model on
its shard of the graphs on a GPU, ONE all-reduce of the flat gradient — the reduced gradient must equal the gradient
of the union batch computed by a single process through the same HIP kernels, and one train_step must leave both
ranks with identical parameters.

Two visible GPUs: backend "nccl" (= RCCL over xGMI), rank r on cuda:r.  One visible GPU (the gpurun box): RCCL
refuses two ranks on one device, so both ranks share cuda:0 and the collective runs over gloo — the compute under
test (librelgnn kernels, reducer, loss scaling) is the same.  tests/test_distributed_cpu.py covers the reducer
logic on CPU with the oracle mirror as compute."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graphs():
    from tf_gnn_samples_amd.tasks.synthetic import make_ppi_shaped_graphs
    return make_ppi_shaped_graphs(6, seed=11, mean_nodes=150, std_nodes=40, min_nodes=60, max_nodes=260,
                                  fwd_edges_per_node=6.0)


def _model(device):
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(1, 1, seed=3, mean_nodes=80, std_nodes=1, min_nodes=60, max_nodes=100)   # sets F / label sizes
    p = RGCN_Model.default_params()
    p.update(hidden_size=64, graph_num_layers=2, graph_layer_input_dropout_keep_prob=1.0, random_seed=5)
    return task, RGCN_Model(p, task, device=str(device))


def _batch(task, graphs, device):
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch
    mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
    return DeviceBatch(mb, device)


def _grads(model, batch):
    model.optimizer.zero_grad()
    m = model.forward_batch(batch, training=True)
    m['loss'].backward()
    return [p.grad.detach().clone() for p in model.optimizer.params]


def _worker(rank, world, port, share_gpu, q, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0" if share_gpu else str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from tf_gnn_samples_amd.parallel import (GradientAllReducer, OverlappedGradientAllReducer, init_distributed,
                                             shard_graphs_by_edges)
    r, local_rank, w = init_distributed(backend="gloo" if share_gpu else "nccl")
    assert (r, w) == (rank, world)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    graphs = _graphs()
    counts = [sum(len(a) for a in g.adjacency_lists) for g in graphs]
    shard = shard_graphs_by_edges(counts, world)[rank]
    task, model = _model(device)
    local = _batch(task, [graphs[i] for i in shard], device)
    reducer = (OverlappedGradientAllReducer(model.optimizer.params, bucket_bytes=16 << 10) if overlap
               else GradientAllReducer(model.optimizer.params))
    if overlap:                              # buckets leave from the backward's hooks
        model.optimizer.zero_grad()
        m = model.forward_batch(local, training=True)
        reducer.arm(float(local.num_nodes))
        m['loss'].backward()
        assert len(reducer.buckets) > 1 and reducer._next > 0      # (buckets left during the backward, in index order)
        reducer.finish()
    else:
        _grads(model, local)
        reducer(float(local.num_nodes))
    reduced = [p.grad.detach().cpu().numpy().copy() for p in model.optimizer.params]
    union = None
    if rank == 0:   # the single-process answer through the same HIP kernels
        union = [g.cpu().numpy() for g in _grads(model, _batch(task, graphs, device))]
    # one full training step with the hook: parameters must stay identical across ranks
    model.optimizer.zero_grad()
    if overlap:
        model.train_step(local, pre_backward=lambda: reducer.arm(float(local.num_nodes)), grad_hook=lambda ps: reducer.finish())
    else:
        model.train_step(local, grad_hook=lambda ps: reducer(float(local.num_nodes)))
    torch.cuda.synchronize()
    params = [p.detach().cpu().numpy().copy() for p in model.optimizer.params]
    q.put((rank, shard, reduced, union, params, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [False, True], ids=["flat", "buckets_during_backward"])
def test_two_ranks_hip_gradient_equals_union_batch(gpu_device, overlap):
    world = 2
    share_gpu = torch.cuda.device_count() < 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, share_gpu, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(i for _, s, *_ in results for i in s) == list(range(6))
    assert all(len(s) > 0 for _, s, *_ in results)
    assert results[0][5] == ("gloo" if share_gpu else "nccl")
    union = results[0][3]
    for _, _, reduced, _, _, _ in results:
        for a, b in zip(reduced, union):
            assert np.abs(a - b).max() <= 2e-6 + 1e-5 * np.abs(b).max(), np.abs(a - b).max()
    for a, b in zip(results[0][4], results[1][4]):
        assert np.array_equal(a, b)       # same reduced gradient + same update rule -> bit-identical parameters
