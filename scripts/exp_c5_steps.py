#!/usr/bin/env python
"""Per-step wall times of the C5 fixed-batch training step (bench_other.py's model and batch), 20 steps from a cold process."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench_other as B
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch
dev = torch.device("cuda:0")
task, graphs = B.c5_task_and_graphs(42)
mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
batch = DeviceBatch(mb, dev)
model, _ = B.c5_model(task, dev)
ts = []
for i in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    clear_graph_cache(); model.train_step(batch)
    torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
print("ms per step:", ts, file=sys.stderr)
print("ms per step:", ts)
