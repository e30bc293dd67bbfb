"""Soak of the many-type path: GNN-FiLM on VarMisuse-shaped graphs (23 edge types, compact pair tables built without host round
trips from the resident fold's per-graph bucket counts), many distinct batches of shuffled epochs; device memory must stay flat,
losses finite, and the tables of every batch must equal the ones built from counts read back from the device."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench_other as BO
from tf_gnn_samples_amd.graph import PairTables, RelGraph
from tf_gnn_samples_amd.tasks import DataFold
dev = torch.device("cuda:0")
task, graphs = BO.c5_task_and_graphs(24)
model, p = BO.c5_model(task, dev)
model.params['max_nodes_in_batch'] = 9000
task._loaded_data[DataFold.TRAIN] = graphs
marks, losses = [], []
EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.time()
steps = 0
for ep in range(EPOCHS):
    for batch in model._batches(graphs, DataFold.TRAIN):
        g = batch.graph
        if steps % 7 == 0 and getattr(g, "pair_counts", None) is not None:      # spot check against the synchronising construction
            fresh = RelGraph([a.clone() for a in batch.adjacency_lists], batch.num_nodes)
            want, got = PairTables(fresh), g.pair_tables()
            assert torch.equal(want.col_t, got.col_t) and torch.equal(want.tgt.node, got.tgt.node) and torch.equal(want.src.bucket_row, got.src.bucket_row)
        m = model.train_step(batch)
        steps += 1
        if steps % 10 == 0:
            losses.append(float(m['loss'].detach()))
    torch.cuda.synchronize()
    marks.append((ep, steps, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
print("epoch/steps/allocated MiB/reserved MiB:", marks, "losses", [round(x, 4) for x in losses[:3]], "...", [round(x, 4) for x in losses[-3:]],
      "wall %.1f s" % (time.time() - t0))
assert all(np.isfinite(losses)), "non-finite loss"
assert marks[-1][3] <= marks[min(3, len(marks) - 1)][3] * 1.1 + 64, "device memory grows"
print("soak ok")
