"""Soak of the wave-role product kernel (relgnn_limb_gemm_xf32_pc: the default forward products): many C2-sized training steps; the
hand-over status word must stay 0 (no poll of any launch gave up)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tf_gnn_samples_amd import ops
from tf_gnn_samples_amd.models import RGCN_Model
from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
task = PPI_Task(PPI_Task.default_params())
task.load_synthetic(48, 4, seed=0)                       # C2-sized graphs: batches of ~16 graphs, V ~ 36 k: the wave-role kernel's regime
p = RGCN_Model.default_params(); p.update(hidden_size=256, graph_num_layers=3, random_seed=0)
model = RGCN_Model(p, task, device="cuda:0")
data = task._loaded_data[DataFold.TRAIN]
t0 = time.time(); steps = 0
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    loss, res, n, *_ = model._run_epoch("e", data, DataFold.TRAIN, quiet=True)
    steps += len(res)
torch.cuda.synchronize()
print("steps", steps, "loss %.4f" % loss, "hand-over status", ops.handover_status(), "wall %.1f s" % (time.time() - t0))
