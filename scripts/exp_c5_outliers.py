#!/usr/bin/env python
"""Where do the rare 75-290 ms steps of the distinct-batch C5 loop come from?  Per step: host time of batch assembly, of train_step(),
of the closing synchronize, and the caching allocator's device-malloc / free counters (torch.cuda.memory_stats)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import bench as B
from tf_gnn_samples_amd.models import name_to_model_class
from tf_gnn_samples_amd.tasks import DataFold
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
task, fold, _ = B.build_local_fold(0, 1, "C5", {})
cls, extra = name_to_model_class("GNN-FiLM")
p = cls.default_params(); p.update(extra)
p.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1, graph_residual_connection_every_num_layers=2,
         graph_layer_input_dropout_keep_prob=1.0)
nodes = sorted(len(g.node_features) for g in fold)
p['max_nodes_in_batch'] = int(sum(nodes) / max(1, len(fold) // 50)) + nodes[-1]
model = cls(p, task, device=str(dev))
np.random.seed(20240924)


def batches():
    while True:
        for b in model._batches(fold, DataFold.TRAIN):
            yield b


it = batches()
rows = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 70):
    s0 = torch.cuda.memory_stats(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = next(it); t1 = time.perf_counter()
    model.train_step(b); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    s1 = torch.cuda.memory_stats(dev)
    rows.append((i, b.num_nodes, round((t3 - t0) * 1e3, 1), round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1),
                 s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
                 round(s1["reserved_bytes.all.current"] / 2**30, 2), s1["num_alloc_retries"] - s0["num_alloc_retries"]))
print("step nodes total_ms assemble_ms train_step_host_ms sync_ms device_mallocs device_frees reserved_GiB retries")
med = sorted(r[2] for r in rows[8:])[len(rows[8:]) // 2]
for r in rows:
    if r[0] < 8 or r[2] > 1.25 * med or r[6] or r[7]:
        print(*r)
print("median total ms (steps 8..):", med)
