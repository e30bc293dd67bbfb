"""Which CUDA tensors accumulate over training steps (gc census by shape, two snapshots)."""
import gc, sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
task = PPI_Task(PPI_Task.default_params())
task.load_synthetic(24, 4, seed=0, mean_nodes=600.0, std_nodes=150.0, min_nodes=200, max_nodes=1000)
p = RGCN_Model.default_params(); p.update(hidden_size=128, graph_num_layers=3, max_nodes_in_batch=3000, random_seed=0)
so = sys.stdout; sys.stdout = sys.stderr
model = RGCN_Model(p, task, device="cuda:0")
sys.stdout = so
data = task._loaded_data[DataFold.TRAIN]
def census():
    gc.collect(); torch.cuda.synchronize()
    c = collections.Counter(); b = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                k = (tuple(o.shape), str(o.dtype)); c[k] += 1; b[k] += o.numel() * o.element_size()
        except Exception:
            pass
    return c, b, torch.cuda.memory_allocated() >> 10
for ep in range(50):
    model._run_epoch("e", data, DataFold.TRAIN, quiet=True)
c1, b1, m1 = census()
for ep in range(300):
    model._run_epoch("e", data, DataFold.TRAIN, quiet=True)
c2, b2, m2 = census()
print("allocated KiB", m1, "->", m2, " tensors", sum(c1.values()), "->", sum(c2.values()))
grow = [(k, c2[k] - c1[k], (b2[k] - b1[k]) >> 10) for k in c2 if c2[k] > c1.get(k, 0)]
for k, n, kb in sorted(grow, key=lambda x: -x[2])[:15]:
    print(k, "+%d tensors" % n, "+%d KiB" % kb)
import tf_gnn_samples_amd.graph as G
print("pending checks", len(G._PENDING_CHECKS), "graph cache", len(getattr(G, "_GRAPH_CACHE", {})))
from tf_gnn_samples_amd.models.sparse_graph_model import MetricsReadback
print("readback ring", {k: len(v) for k, v in MetricsReadback._ring.items()})
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
