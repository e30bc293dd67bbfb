"""C2 with a 4x larger batch on one GPU (64 PPI-shaped graphs, ~7.4 M edges): does the step scale linearly?"""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tf_gnn_samples_amd.dense import enable_gemm_autotuning
from tf_gnn_samples_amd.graph import RelGraph
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
tuned = enable_gemm_autotuning()
task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(n_graphs, 1, seed=0)
mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
batch = DeviceBatch(mb, dev)
p = RGCN_Model.default_params()
p.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1, message_aggregation_function="sum",
         graph_activation_function="ReLU", graph_layer_input_dropout_keep_prob=1.0)
so = sys.stdout; sys.stdout = sys.stderr
model = RGCN_Model(p, task, device="cuda:0")
sys.stdout = so
side = torch.cuda.Stream()
state = {"g": None}
def step():
    batch.graph = state["g"] if state["g"] is not None else RelGraph.build_on_stream(batch.adjacency_lists, batch.num_nodes, side)
    state["g"] = RelGraph.build_on_stream(batch.adjacency_lists, batch.num_nodes, side)
    return model.train_step(batch)
for _ in range(20):
    step()
enable_gemm_autotuning(tune=False)
for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
print(json.dumps({"config": "C2 RGCN PPI-shaped, %d graphs on one GPU" % n_graphs, "nodes": mb.num_nodes, "edges": mb.num_edges,
                  "train_ms": round(ms, 3), "train_edges_per_s": round(mb.num_edges / ms * 1e3), "gemm_autotuned": bool(tuned),
                  "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
