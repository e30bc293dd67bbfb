"""C5 at full size on ONE MI355X: the whole ~10 M-message VarMisuse-shaped batch (23 edge types, D=128, 10 layers) that
BASELINE.json shards over 8 GPUs.  Memory sizing check for the 288 GB HBM; argv[1] = number of graphs (336 ~ 10 M)."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.models import name_to_model_class
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graphs

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 336
dev = torch.device("cuda:0")
graphs = make_varmisuse_shaped_graphs(n_graphs, seed=0)
task = PPI_Task(PPI_Task.default_params())
task._PPI_Task__num_edge_types = 23; task._PPI_Task__initial_node_feature_size = 128; task._PPI_Task__num_labels = 1
mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
batch = DeviceBatch(mb, dev)
cls, extra = name_to_model_class("GNN-FiLM")
p = cls.default_params(); p.update(extra)
p.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1, graph_residual_connection_every_num_layers=2)
so = sys.stdout; sys.stdout = sys.stderr
model = cls(p, task, device="cuda:0")
sys.stdout = so
def step():
    clear_graph_cache()
    return model.train_step(batch)
for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
print(json.dumps({"config": "C5 whole batch on one GPU", "graphs": n_graphs, "nodes": mb.num_nodes, "edges": mb.num_edges,
                  "train_ms": round(ms, 2), "train_edges_per_s": round(mb.num_edges / ms * 1e3),
                  "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
