"""Which weight-gradient products does a C3 (GGNN / QM9) step run, and through which route?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tf_gnn_samples_amd import dense
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.models import name_to_model_class
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, QM9_Task
from test_golden_cpu import read_qm9_fixture

seen = {}
orig = dense.matmul_tn_splitk
def logged(a, b):
    key = (tuple(a.shape), a.stride(), tuple(b.shape), b.stride())
    seen[key] = seen.get(key, 0) + 1
    return orig(a, b)
dense.matmul_tn_splitk = logged
orig_lib = dense.lib_gemm
libseen = {}
def logged_lib(layout, a, b, *args, **kw):
    key = (layout, tuple(a.shape), a.stride(), tuple(b.shape), b.stride())
    libseen[key] = libseen.get(key, 0) + 1
    return orig_lib(layout, a, b, *args, **kw)
dense.lib_gemm = logged_lib
task = QM9_Task(QM9_Task.default_params())
samples = task.load_raw(read_qm9_fixture() * 11)
mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, 50000))
batch = DeviceBatch(mb, torch.device("cuda:0"))
cls, extra = name_to_model_class("GGNN")
p = cls.default_params(); p.update(hidden_size=128, graph_num_layers=6, graph_rnn_cell="GRU", message_aggregation_function="mean")
so = sys.stdout; sys.stdout = sys.stderr
model = cls(p, task, device="cuda:0")
sys.stdout = so
clear_graph_cache(); model.train_step(batch); torch.cuda.synchronize()
print("matmul_tn_splitk:")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]): print("  ", v, k)
print("lib_gemm:")
for k, v in sorted(libseen.items(), key=lambda kv: -kv[1]): print("  ", v, k)
