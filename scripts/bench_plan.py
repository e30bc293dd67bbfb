import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tf_gnn_samples_amd.graph import RelGraph
dev = torch.device("cuda:0")
task, mb, batch, gen, local = bench.build_local_batch(0, 1, dev)
for _ in range(5): g = RelGraph(batch.adjacency_lists, mb.num_nodes, validate=False)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); s.record()
for _ in range(50): g = RelGraph(batch.adjacency_lists, mb.num_nodes, validate=False)
e.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print("RelGraph build: GPU %.1f us, host enqueue %.1f us" % (s.elapsed_time(e) / 50 * 1e3, (t1 - t0) / 50 * 1e6))
