"""Launch ONLY the dominant kernel (RGCN layer-forward gather + 1/deg scale + segment-sum + ReLU on the C2 batch)
a few times: the target of the rocprofv3 PMC passes (HBM traffic per launch)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
task, mb, batch, gen, local = bench.build_local_batch(0, 1, dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ms, alg_bytes, M = bench.time_segment_kernel(batch, 256, iters)
print("kernel avg ms %.4f  algorithmic bytes %d  messages %d  -> %.1f GB/s" % (ms, alg_bytes, M, alg_bytes / ms / 1e6))
