"""Where a distinct-batch pipeline step (bench.py's timed loop) spends its time: host time and GPU time of the batch
assembly (tasks/resident.py) and of the training step, separately, plus the free-running step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.tasks import DataFold

dev = torch.device("cuda:0")
task, fold, gen = bench.build_local_fold(0, 1)
p = RGCN_Model.default_params()
p.update(hidden_size=256, graph_num_layers=3, graph_activation_function="ReLU", graph_layer_input_dropout_keep_prob=1.0)
nodes = sorted(len(g.node_features) for g in fold)
p['max_nodes_in_batch'] = int(sum(nodes) / 4) + nodes[-1]
so = sys.stdout; sys.stdout = sys.stderr
model = RGCN_Model(p, task, device="cuda:0")
sys.stdout = so


def stream():
    while True:
        for b in model._batches(fold, DataFold.TRAIN):
            yield b


it = stream()
for _ in range(12):
    model.train_step(next(it))
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for _ in range(30):
    e0, e1, e2 = ev(), ev(), ev()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    b = next(it)
    t1 = time.perf_counter(); e1.record()
    model.train_step(b)
    t2 = time.perf_counter(); e2.record()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, e0.elapsed_time(e1), e1.elapsed_time(e2), (t3 - t0) * 1e3, b.num_edges))
r = np.median(np.array(rows), axis=0)
print("per step (median of 30, synchronised between steps): assemble host %.3f ms | train_step host enqueue %.3f ms | "
      "assemble GPU %.3f ms | train_step GPU %.3f ms | wall %.3f ms | edges %d" % tuple(r))
# free-running, with the one-step-late fetch like bench.py
pending = None
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 40; edges = 0
up = next(it)
for _ in range(n):
    b = up
    m = model.train_step(b)
    up = next(it)
    if pending is not None:
        float(pending['loss'])
    pending = {k: v.detach() for k, v in m.items() if torch.is_tensor(v)}
    edges += b.num_edges
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
print("free-running: %.3f ms/step, %.1f M edges/s" % (dt, edges / n / dt / 1e3))
