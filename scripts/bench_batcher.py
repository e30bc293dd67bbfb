"""Host-side batch builder timings on the C2 batch: pack into pageable vs pinned arenas, thread counts, upload."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from tf_gnn_samples_amd.tasks import PPI_Task, DataFold
from tf_gnn_samples_amd.tasks.batcher import NativeBatcher

task = PPI_Task(PPI_Task.default_params())
task.load_synthetic(16, 1, seed=0)
graphs = task._loaded_data[DataFold.TRAIN]
store = task.make_graph_store(graphs)
ids = np.arange(16, dtype=np.int64)
lay = store.layout(ids)
nbytes = int(lay[2])
print("arena bytes", nbytes, "cpus", os.cpu_count())

def t(fn, n=10):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3

pageable = np.zeros(nbytes, np.uint8)
pinned = torch.empty(nbytes, dtype=torch.uint8)
if torch.cuda.is_available():
    pinned = pinned.pin_memory()
src = np.ones(nbytes, np.uint8)
print("numpy copy pageable->pageable %.2f ms" % t(lambda: np.copyto(pageable, src)))
print("numpy copy pageable->pinned   %.2f ms" % t(lambda: np.copyto(pinned.numpy(), src)))
print("layout() %.3f ms" % t(lambda: store.layout(ids)))
for th in (1, 2, 4, 8, 16, 32):
    a = t(lambda: store.pack_into(ids, lay, pageable.ctypes.data, nbytes, th))
    b = t(lambda: store.pack_into(ids, lay, pinned.data_ptr(), nbytes, th))
    print("threads %2d: pack pageable %.2f ms  pinned %.2f ms" % (th, a, b))
mbs = task.make_minibatch_iterator(graphs, DataFold.VALIDATION, 10 ** 9)
print("numpy iterator one batch %.1f ms" % t(lambda: next(task.make_minibatch_iterator(graphs, DataFold.VALIDATION, 10 ** 9)), 3))
if torch.cuda.is_available():
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    def up():
        dev.copy_(pinned, non_blocking=True); torch.cuda.synchronize()
    print("upload pinned->hbm %.2f ms" % t(up))
