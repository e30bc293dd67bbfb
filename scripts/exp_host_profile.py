"""cProfile of the host side of one training step (C2 RGCN and C3 GGNN): where does the enqueue time go?"""
import cProfile, pstats, io, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import numpy as np, torch
import bench
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.models import RGCN_Model, name_to_model_class
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, QM9_Task

device = torch.device("cuda:0"); torch.cuda.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
so = sys.stdout; sys.stdout = sys.stderr
if which == "C2":
    task, mb, batch, gen, local = bench.build_local_batch(0, 1, device)
    params = RGCN_Model.default_params()
    params.update(hidden_size=256, graph_num_layers=3, graph_num_timesteps_per_layer=1, message_aggregation_function="sum",
                  graph_activation_function="ReLU", graph_layer_input_dropout_keep_prob=1.0)
    model = RGCN_Model(params, task, device=str(device))
else:
    from test_golden_cpu import read_qm9_fixture
    task = QM9_Task(QM9_Task.default_params())
    samples = task.load_raw(read_qm9_fixture() * 11)
    mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, 50000))
    batch = DeviceBatch(mb, device)
    cls, extra = name_to_model_class("GGNN")
    p = cls.default_params(); p.update(hidden_size=128, graph_num_layers=6, graph_rnn_cell="GRU", message_aggregation_function="mean")
    model = cls(p, task, device=str(device))
sys.stdout = so

def step():
    clear_graph_cache()
    return model.train_step(batch)
for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print("%s step %.3f ms" % (which, (time.perf_counter() - t0) / 30 * 1e3))
host = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); host.append(time.perf_counter() - t0)
print("host enqueue median %.3f ms" % (np.median(host) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
