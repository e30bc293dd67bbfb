"""Host enqueue time vs GPU time of one C2 training step (is the step host-bound?)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from tf_gnn_samples_amd.graph import RelGraph, clear_graph_cache
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.dense import enable_gemm_autotuning

device = torch.device("cuda:0"); torch.cuda.set_device(0)
tuned = enable_gemm_autotuning()
task, mb, batch, gen, local = bench.build_local_batch(0, 1, device)
params = RGCN_Model.default_params()
params.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1, message_aggregation_function="sum",
              graph_activation_function="ReLU", graph_layer_input_dropout_keep_prob=1.0)
so = sys.stdout; sys.stdout = sys.stderr
model = RGCN_Model(params, task, device=str(device))
sys.stdout = so
side = torch.cuda.Stream()

def step(mode):
    if mode == "serial":
        clear_graph_cache(); batch.graph = None
    elif mode == "cached":
        batch.graph = None            # graph cache hit: no bucketing at all
    else:
        batch.graph = RelGraph.build_on_stream(batch.adjacency_lists, batch.num_nodes, side)
    return model.train_step(batch)

for _ in range(25):
    step("serial")
enable_gemm_autotuning(tune=False)
if os.environ.get("SINGLE_THREAD_AUTOGRAD"):
    torch.autograd.set_multithreading_enabled(False)
hi = torch.cuda.Stream(priority=-1)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for mode in ("serial", "side", "cached", "side-hiprio-main"):
    if mode == "side-hiprio-main":
        hi.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hi)
        mode = "side"
    for _ in range(5):
        step(mode)
    torch.cuda.synchronize()
    host = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(mode); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(mode)
    torch.cuda.synchronize()
    print("%-7s host enqueue median %.3f ms   pipelined step %.3f ms" % (mode, np.median(host) * 1e3, (time.perf_counter() - t0) / 50 * 1e3))
