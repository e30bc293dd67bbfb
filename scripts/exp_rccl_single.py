"""RCCL sanity on the one GPU we have: a 1-rank "nccl" process group bound to cuda:0, one all-reduce and a barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1024, dtype=torch.float32, device="cuda")
dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()
print("rccl ok", float(x.sum()), dist.get_backend())
dist.destroy_process_group()
