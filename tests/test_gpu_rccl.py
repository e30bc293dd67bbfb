"""The data-parallel path's collective on real hardware, as far as ONE GPU allows: a 1-rank "nccl" (= RCCL) process group
bound to cuda:0 runs the GradientAllReducer round trip.  (World size 2 is covered on CPU with gloo,
tests/test_distributed_cpu.py; RCCL refuses two ranks on one device.)  Runs in a subprocess so that the process
group cannot leak into other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
os.environ.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
from tf_gnn_samples_amd.parallel import GradientAllReducer
# init_distributed() would wait for rank 1; build the same kind of group with one member
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ps = [torch.nn.Parameter(torch.randn(5, 7, device="cuda")), torch.nn.Parameter(torch.randn(11, device="cuda"))]
for p in ps:
    p.grad = torch.randn_like(p)
want = [p.grad.clone() for p in ps]
red = GradientAllReducer(ps)
red(123.0)                      # world size 1: early return, gradients untouched
assert all(torch.equal(p.grad, w) for p, w in zip(ps, want))
flat = torch.cat([w.reshape(-1) for w in want]) * 3.0
dist.all_reduce(flat); dist.barrier(); torch.cuda.synchronize()
assert torch.allclose(flat, torch.cat([w.reshape(-1) for w in want]) * 3.0)
print("RCCL_OK", dist.get_backend())
dist.destroy_process_group()
'''


def test_rccl_process_group_on_one_gpu(gpu_device):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, REPO=repo)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "RCCL_OK nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
