"""Timing of the non-headline BASELINE.json configs on one MI355X (informational; the contract line is bench.py).
  C3  GGNN on QM9 graphs (real molecules from tests/golden tiled to a 50k-node batch), GRU, mean/max, D=128, 6 layers
  C4  RGAT on the C2 PPI-shaped batch, D=256, 4 heads, 3 layers
  C5  GNN-FiLM on a VarMisuse-shaped batch (~1.25M edges = one rank's share), 23 edge types, D=128, 10 layers
  also RGIN / GNN-Edge-MLP0/1 / GNN-FiLM / RGDCN (16 channels x 16) on the C2 batch.
Each line: forward+backward+optimizer step time, edges/s, and forward-only time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from tf_gnn_samples_amd.graph import clear_graph_cache
from tf_gnn_samples_amd.models import name_to_model_class
from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task, QM9_Task
from tf_gnn_samples_amd.tasks.synthetic import make_varmisuse_shaped_graphs

dev = torch.device("cuda:0")
# RELGNN_TUNE_GEMMS=1: let PyTorch TunableOp pick the library GEMM solution per shape during the priming steps, as
# bench.py does for the headline config (the batch is fixed here, so every shape recurs).
TUNE = bool(os.environ.get("RELGNN_TUNE_GEMMS"))
if TUNE:
    from tf_gnn_samples_amd.dense import enable_gemm_autotuning
    TUNE = enable_gemm_autotuning()


def run(name, model, batch, mb, steps=20, prime=15):
    def step():
        clear_graph_cache()
        return model.train_step(batch)
    for _ in range(prime):
        step()
    if TUNE:
        with torch.no_grad():
            clear_graph_cache(); model.forward_batch(batch, training=False)
        torch.cuda.synchronize()
        enable_gemm_autotuning(tune=False)          # freeze the choices for the timed steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    with torch.no_grad():
        for _ in range(3):
            clear_graph_cache(); model.forward_batch(batch, training=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            clear_graph_cache(); model.forward_batch(batch, training=False)
        torch.cuda.synchronize()
        fms = (time.perf_counter() - t0) / steps * 1e3
    if TUNE:
        enable_gemm_autotuning(tune=True)
    graph_ms = None
    if os.environ.get("RELGNN_CAPTURE", "1") != "0":
        # the same step recorded as ONE hipGraph on this fixed batch (model.capture_train_step)
        try:
            cap = model.capture_train_step(batch)
            for _ in range(3):
                cap.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                cap.replay()
            torch.cuda.synchronize()
            graph_ms = round((time.perf_counter() - t0) / steps * 1e3, 3)
        except Exception as e:
            graph_ms = "capture failed: %r" % (e,)
    print(json.dumps({"config": name + (" [GEMMs autotuned]" if TUNE else ""), "nodes": mb.num_nodes, "edges": mb.num_edges, "graphs": mb.num_graphs,
                      "train_ms": round(ms, 3), "train_edges_per_s": round(mb.num_edges / ms * 1e3),
                      "train_ms_hipgraph": graph_ms,
                      "fwd_ms": round(fms, 3), "fwd_edges_per_s": round(mb.num_edges / fms * 1e3)}), flush=True)


def quiet_model(cls, p, task):
    so = sys.stdout; sys.stdout = sys.stderr
    try:
        return cls(p, task, device="cuda:0")
    finally:
        sys.stdout = so


which = sys.argv[1:] or ["C3", "C4", "C5", "RGIN", "MLP0", "MLP1"]

if "C3" in which:
    from test_golden_cpu import read_qm9_fixture
    task = QM9_Task(QM9_Task.default_params())
    raw = read_qm9_fixture()
    samples = task.load_raw(raw * 11)                      # 2816 molecules ~ 50k nodes
    mb = next(task.make_minibatch_iterator(list(samples), DataFold.VALIDATION, 50000))
    batch = DeviceBatch(mb, dev)
    for agg in ("mean", "max"):
        cls, extra = name_to_model_class("GGNN")
        p = cls.default_params(); p.update(hidden_size=128, graph_num_layers=6, graph_rnn_cell="GRU", message_aggregation_function=agg)
        run("C3 GGNN/QM9 GRU %s D=128 6 layers" % agg, quiet_model(cls, p, task), batch, mb)

if any(w in which for w in ("C2", "C4", "RGIN", "MLP0", "MLP1", "FILM", "RGDCN")):
    task = PPI_Task(PPI_Task.default_params()); task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, dev)
    for key, mname in (("C2", "RGCN"), ("C4", "RGAT"), ("RGIN", "RGIN"), ("MLP0", "GNN-Edge-MLP0"), ("MLP1", "GNN-Edge-MLP1"),
                       ("FILM", "GNN-FiLM"), ("RGDCN", "RGDCN")):
        if key in which:
            cls, extra = name_to_model_class(mname)
            p = cls.default_params(); p.update(extra); p.update(hidden_size=256, graph_num_layers=3)
            if key == "RGDCN":
                p.update(num_channels=16)       # channel_dim = 256 / 16 = 16
            run("%s %s on C2 PPI-shaped batch D=256 3 layers" % (key, mname), quiet_model(cls, p, task), batch, mb, steps=10, prime=8)

if "C5" in which or "VM" in which:
    graphs = make_varmisuse_shaped_graphs(42, seed=0)       # ~ one rank's share of the 10M-edge batch
    task = PPI_Task(PPI_Task.default_params())
    task._PPI_Task__num_edge_types = 23; task._PPI_Task__initial_node_feature_size = 128; task._PPI_Task__num_labels = 1
    mb = next(task.make_minibatch_iterator(list(graphs), DataFold.VALIDATION, 10 ** 9))
    batch = DeviceBatch(mb, dev)
    cls, extra = name_to_model_class("GNN-FiLM")
    p = cls.default_params(); p.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1,
                                       graph_residual_connection_every_num_layers=2)
    if "C5" in which:
        run("C5 GNN-FiLM VarMisuse-shaped 23 types D=128 10 layers (1 rank share)", quiet_model(cls, p, task), batch, mb, steps=5, prime=5)
    if "VM" in which:     # the other models of the reference's VarMisuse runs on the same batch (tasks/default_hypers/VarMisuse_*.json)
        for mname in ("RGCN", "GGNN"):
            cls, extra = name_to_model_class(mname)
            p = cls.default_params(); p.update(extra)
            p.update(hidden_size=128, graph_num_layers=10, graph_dense_between_every_num_gnn_layers=1,
                     graph_residual_connection_every_num_layers=2)
            run("VM %s VarMisuse-shaped 23 types D=128 10 layers" % mname, quiet_model(cls, p, task), batch, mb, steps=5, prime=5)
