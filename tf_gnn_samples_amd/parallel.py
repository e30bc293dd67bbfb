"""Data parallelism BY GRAPH across the GPUs of one node.

A minibatch of the reference is a disjoint union of graphs (tasks/ppi_task.py:220-233): no edge
crosses graphs, so message passing never communicates.  Whole graphs are assigned to ranks
(balanced by edge count), every rank builds its own local disjoint-union batch exactly as the
single-GPU batcher would, and the ONLY collective is one all-reduce(SUM) of the flat fp32
gradient per step over RCCL/xGMI (one process per GPU, torch.distributed backend "nccl" == RCCL).

Objective equivalence with the single-batch reference loss (tasks/ppi_task.py:183-191,
loss = total_loss / num_nodes_in_batch): rank r holds g_r = d(total_r / n_r); the global gradient
is sum_r n_r * g_r / sum_r n_r.  The weight n_r rides in the last slot of the same flat buffer, so
there is exactly one collective.  Per-variable clip_by_norm (models/sparse_graph_model.py:253-260)
is applied AFTER the all-reduce.
"""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_graphs_by_edges(edge_counts: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of whole graphs to ranks, balancing sum_l E_l.
    Deterministic (ties broken by graph index / lowest rank); each shard keeps ascending graph order."""
    order = sorted(range(len(edge_counts)), key=lambda i: (-int(edge_counts[i]), i))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(edge_counts[i])
    return [sorted(s) for s in shards]


def effective_cpu_count() -> int:
    """CPUs this process may actually keep busy: the cgroup CPU quota (v2 cpu.max, v1 cfs quota) and the affinity
    mask, whichever is smaller.  os.cpu_count() reports the host's hardware threads (256 on the MI355X boxes) even when
    the container is capped at 16 CPUs; running more busy threads than the quota gets the whole process throttled for
    the rest of the 100 ms scheduler period (observed: 30-60 ms stalls in host-side batching)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p_ > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return max(1, n)


def init_distributed(backend: str = None):
    """One process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            # bind the communicator to this rank's GPU up front (no device guessing inside barrier()/collectives)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        except TypeError:        # older torch without device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradientAllReducer:
    """Weighted gradient average across ranks with ONE all-reduce of one flat fp32 buffer."""

    def __init__(self, params: Sequence[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    @torch.no_grad()
    def __call__(self, local_weight: float):
        """grad <- sum_r w_r * grad_r / sum_r w_r  (w_r = local_weight, e.g. nodes in the local batch)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        self.flat[:-1].mul_(float(local_weight))
        self.flat[-1] = float(local_weight)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat[:-1].div_(self.flat[-1])
        # the reduced gradients stay where they are: every .grad becomes a view of the flat buffer (no unpack copies;
        # the next backward allocates fresh .grad tensors after zero_grad() and this buffer is overwritten by the pack)
        for p, v in zip(self.params, self.views):
            p.grad = v


class OverlappedGradientAllReducer(GradientAllReducer):
    """The same weighted average, with the collective cut into buckets that leave while the backward is still running
    (RELGNN_ALLREDUCE=overlap in bench.py; the flat one-collective form stays the default until an N > 1 run has timed both).

    Buckets are contiguous ranges of the same flat buffer, filled from its END (the backward produces the last layers'
    gradients first): a parameter's post-accumulate hook counts its bucket down, a complete bucket is packed (one
    multi-tensor copy + one scale by the local weight) and all-reduced asynchronously.  The local weight n_r — known before the
    backward — is reduced by its own tiny collective at arm() time; finish() (the training step's grad_hook) launches what
    is left (parameters that received no gradient count as zeros), waits, divides by sum_r n_r and points every .grad at its
    slice, exactly like the flat form.  For two ranks the result is bit-identical to the flat form (a sum of two values has
    one order); for more ranks the reduction order of an element may depend on where the library cuts its buffer.

    Collectives must be issued in the SAME sequence on every rank, and the order in which autograd completes buckets is not a
    property of the model: which parameters get a gradient at all, and when, follows the autograd graph of THIS rank's batch
    (hub routes, pair tables and typed panels are picked per batch).  So buckets leave strictly in index order, as in DDP: a
    complete bucket is launched only when every bucket before it has been — otherwise it waits for a later hook or for finish(),
    which launches whatever is left, again in index order.  Two ranks whose backward passes complete their buckets in different
    orders (tests/test_distributed_cpu.py) therefore still pair bucket b with bucket b."""

    def __init__(self, params: Sequence[torch.nn.Parameter], bucket_bytes: int = 1 << 20, group=None):
        super().__init__(params, group)
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += p.numel()
        self.buckets = []                       # (lo, hi, [param indices]) in the order the backward completes them
        members, hi = [], off
        for i in reversed(range(len(self.params))):
            members.append(i)
            if (hi - offs[i]) * 4 >= bucket_bytes or i == 0:
                self.buckets.append((offs[i], hi, members[::-1]))
                members, hi = [], offs[i]
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def close(self) -> None:
        """Remove the post-accumulate hooks (a reducer that is dropped while its parameters live on; with the hooks in place
        ops.deferred_targets_ok keeps every weight gradient on the main stream)."""
        for h in self._handles:
            h.remove()
        self._handles = []

    def _make_hook(self, i):
        def hook(_param):
            if self._armed:
                self._left[self.bucket_of[i]] -= 1
                self._launch_ready_prefix()
        return hook

    def _launch_ready_prefix(self) -> None:
        while self._next < len(self.buckets) and self._left[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @torch.no_grad()
    def arm(self, local_weight: float) -> None:
        """Call before the backward of the step."""
        if not self._active():
            return
        self._w = float(local_weight)
        self._left = [len(idx) for _, _, idx in self.buckets]
        self._next = 0                          # buckets [0, _next) have been launched: strictly in index order on every rank
        self.flat[-1] = self._w
        self._works = [dist.all_reduce(self.flat[-1:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        self._armed = True

    @torch.no_grad()
    def _launch(self, b: int) -> None:
        lo, hi, idx = self.buckets[b]
        grads = [self.params[i].grad if self.params[i].grad is not None else torch.zeros_like(self.params[i]) for i in idx]
        torch._foreach_copy_([self.views[i] for i in idx], grads)
        self.flat[lo:hi].mul_(self._w)
        self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @torch.no_grad()
    def finish(self) -> None:
        """The training step's grad_hook: after the backward, before clipping."""
        if not self._armed:
            return
        self._armed = False
        while self._next < len(self.buckets):   # what the hooks left: incomplete buckets (parameters without a gradient) included
            self._launch(self._next)
            self._next += 1
        for w in self._works:
            w.wait()
        self._works = []
        self.flat[:-1].div_(self.flat[-1])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def __call__(self, local_weight: float):
        """Without arm() before the backward this is the flat form's call: everything at once."""
        if not self._armed:
            self.arm(local_weight)
        self.finish()
