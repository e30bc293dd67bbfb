"""Where does the time of a training epoch over distinct batches go (input pipeline vs compute)?"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.tasks import DataFold
from tf_gnn_samples_amd.tasks.synthetic import make_ppi_shaped_graphs

device = torch.device("cuda:0"); torch.cuda.set_device(0)
task, mb, batch, gen, local = bench.build_local_batch(0, 1, device)
params = RGCN_Model.default_params()
params.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1, message_aggregation_function="sum",
              graph_activation_function="ReLU", graph_layer_input_dropout_keep_prob=1.0)
so = sys.stdout; sys.stdout = sys.stderr
model = RGCN_Model(params, task, device=str(device))
sys.stdout = so
data = make_ppi_shaped_graphs(64, seed=1)
nodes = sorted(len(g.node_features) for g in data)
model.params['max_nodes_in_batch'] = int(sum(nodes) / 4) + nodes[-1]
for native in (True, False):
    model.params['native_batching'] = native
    for ep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, res, n, *_ = model._run_epoch("e", data, DataFold.TRAIN, quiet=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("native=%s epoch %d: %d steps, %.2f ms/step" % (native, ep, len(res), dt / len(res) * 1e3), flush=True)
# pieces
from tf_gnn_samples_amd.tasks.batcher import NativeBatcher
nb = NativeBatcher(task.make_graph_store(data), device)
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 0
    for b in task.make_native_minibatch_iterator(nb, DataFold.TRAIN, model.params['max_nodes_in_batch']):
        k += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("iterator only: %d batches, %.2f ms/batch" % (k, dt / k * 1e3), flush=True)
