import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from tf_gnn_samples_amd.tasks.batcher import NativeBatcher

def tm(nb, ids, tag):
    nb.pack(ids); nb.pack(ids); torch.cuda.synchronize()
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); p = nb.pack_host(ids, i % 2); t1 = time.perf_counter()
        nb.upload(p, i % 2); torch.cuda.synchronize(); ts.append((t1 - t0) * 1e3)
    print(tag, " ".join("%.2f" % x for x in ts), flush=True)

device = torch.device("cuda:0")
torch.cuda.set_device(0)
task, mb, batch, gen, local = bench.build_local_batch(0, 1, device)
nb = NativeBatcher(task.make_graph_store(local), device)
ids = np.arange(len(local))
tm(nb, ids, "fresh")
from tf_gnn_samples_amd.models import RGCN_Model
params = RGCN_Model.default_params()
params.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1, max_nodes_in_batch=10**9, graph_layer_input_dropout_keep_prob=1.0)
model = RGCN_Model(params, task, device=device)
for _ in range(20):
    model.train_step(batch)
torch.cuda.synchronize()
tm(nb, ids, "after 20 train steps")
print("h2d old path %.2f ms" % bench.time_h2d(mb, device))
tm(nb, ids, "after time_h2d")
from tf_gnn_samples_amd.dense import enable_gemm_autotuning
enable_gemm_autotuning()
for _ in range(5):
    model.train_step(batch)
torch.cuda.synchronize()
tm(nb, ids, "after tunableop")
print("threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
