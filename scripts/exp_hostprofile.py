"""Where does the HOST time of one training step go? (cProfile over 30 steps, no per-step sync)"""
import cProfile, pstats, sys, time, io
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.models import RGCN_Model
dev = torch.device("cuda:0")
task, mb, batch, gen, local = bench.build_local_batch(0, 1, dev)
p = RGCN_Model.default_params(); p.update(hidden_size=256, graph_num_layers=3)
sys.stdout = sys.stderr
model = RGCN_Model(p, task, device="cuda:0")
sys.stdout = sys.__stdout__
def step():
    clear_graph_cache()
    return model.train_step(batch)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("enqueue ms/step %.3f   total ms/step %.3f" % (t_enq / 30 * 1e3, t_all / 30 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
