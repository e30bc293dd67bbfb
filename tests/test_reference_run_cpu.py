"""Fixtures written by the REFERENCE'S OWN code (tests/golden/make_reference_run.py: the unmodified /root/reference sources executed
in the build container over a NumPy shim of the TF symbols they touch) against

  * the oracle's hand restatements (oracle/gnns.py, oracle/bookkeeping.py, oracle/model.py) — which this pins as TRANSCRIPTIONS of
    gnns/*.py, utils/utils.py, tasks/ppi_task.py and tasks/qm9_task.py: op order, operands, constants, variable names, concat and
    segment orders, batch packing, loaders.  (The semantics of the individual TF ops stay the oracle's: tf_numpy_shim.py header.)
  * the product's host-side loaders and batch builders, bit for bit.

The GPU kernels against the same layer fixtures: tests/test_gpu_reference_run.py.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import bookkeeping
from oracle import gnns as G
from oracle import tf_ops as T

GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))


def _load(name):
    z = np.load(GOLDEN / name)
    return z, json.loads(bytes(z["manifest"]).decode())


def layer_cases():
    z, manifest = _load("reference_run_layers.npz")
    for case in manifest:
        k = case["key"]
        adj = [z["%s/adj%d" % (k, l)] for l in range(case["num_edge_types"])]
        weights = {n: z["%s/var/%s" % (k, n)] for n in case["variables"]}
        yield case, z[k + "/h"], adj, z[k + "/deg"], weights, z[k + "/out"]


def call_layer(module, case, h, adj, deg, weights, convert=lambda x: x):
    """module.<function>(...) with the reference's argument order (the oracle and the product share it) + weights=."""
    kw = dict(case["kwargs"])
    fn = getattr(module, case["function"])
    if case["function"] == "sparse_rgdcn_layer":
        return fn(convert(h), [convert(a) for a in adj], convert(deg), weights={k: convert(v) for k, v in weights.items()}, **kw)
    state_dim = kw.pop("state_dim")
    args = (convert(h), [convert(a) for a in adj]) + ((convert(deg),) if case["takes_degrees"] else ()) + (state_dim,)
    return fn(*args, weights={k: convert(v) for k, v in weights.items()}, **kw)


CASES = list(layer_cases())


@pytest.mark.parametrize("i", range(len(CASES)), ids=["%s-%d" % (c[0]["function"], n) for n, c in enumerate(CASES)])
def test_oracle_layers_are_the_reference_layers(i):
    case, h, adj, deg, weights, want = CASES[i]
    got = call_layer(G, case, h, adj, deg, weights)
    assert got.dtype == np.float32 and got.shape == want.shape
    # same primitive ops on both sides: a faithful transcription gives the same bits (one ulp of slack for a re-associated scale)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * max(1.0, float(np.abs(want).max())))
    # every variable the reference created is one the restatement reads under the same TF name, and nothing else
    used = set()

    class Spy(dict):
        def __getitem__(self, k):
            used.add(k)
            return dict.__getitem__(self, k)

        def get(self, k, d=None):
            used.add(k)
            return dict.get(self, k, d)

    spy = Spy(weights)
    import unittest.mock as mock
    with mock.patch.object(G, "_cast", lambda w, dtype: spy):
        call_layer(G, case, h, adj, deg, weights)
    assert used >= set(weights), sorted(set(weights) - used)


def test_layer_fixture_covers_every_layer_function_and_branch():
    fns = {c[0]["function"] for c in CASES}
    assert fns == {"sparse_rgcn_layer", "sparse_ggnn_layer", "sparse_rgat_layer", "sparse_rgin_layer", "sparse_gnn_film_layer",
                   "sparse_gnn_edge_mlp_layer", "sparse_rgdcn_layer"}
    aggs = {c[0]["kwargs"].get("message_aggregation_function") for c in CASES}
    assert {"sum", "mean", "max", "sqrt_n"} <= aggs
    acts = {str(c[0]["kwargs"].get("activation_function")).lower() for c in CASES}
    assert {"tanh", "relu", "elu", "selu", "gelu", "leaky_relu"} <= acts


# ------------------------------------------------------------------------------------------------------------------------------
# tasks: loaders and batch builders
# ------------------------------------------------------------------------------------------------------------------------------
def _samples(z, prefix, payload):
    out = []
    for g in range(int(z[prefix + "/count"])):
        adj = []
        l = 0
        while "%s/g%d/adj%d" % (prefix, g, l) in z.files:
            adj.append(z["%s/g%d/adj%d" % (prefix, g, l)])
            l += 1
        out.append((adj, z["%s/g%d/deg" % (prefix, g)], z["%s/g%d/features" % (prefix, g)], z["%s/g%d/%s" % (prefix, g, payload)]))
    return out


def _same_sample(got_adj, got_deg, got_feat, got_payload, want):
    adj, deg, feat, payload = want
    assert len(got_adj) == len(adj)
    for a, b in zip(got_adj, adj):
        np.testing.assert_array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2))
    np.testing.assert_array_equal(np.asarray(got_deg), deg)
    np.testing.assert_array_equal(np.asarray(got_feat), feat)
    np.testing.assert_array_equal(np.asarray(got_payload), payload)


def _batches(z, key, L, payload):
    out = []
    for b in range(int(z[key + "/count"])):
        p = "%s/b%d/" % (key, b)
        out.append(dict(sizes=z[p + "sizes"], features=z[p + "features"], deg=z[p + "deg"], gnl=z[p + "graph_nodes_list"],
                        payload=z[p + payload], keep=float(z[p + "keep_prob"]), adj=[z[p + "adj%d" % l] for l in range(L)]))
    return out


def _same_batch(feed, sizes, want, payload_key):
    assert [int(s) for s in sizes] == [int(s) for s in want["sizes"]]
    # (the reference feeds what np.array() makes of its lists — float64 for QM9's JSON numbers — into a float32 placeholder
    #  (sparse_graph_task.py:120-122): the session rounds it to float32; the product builds float32 directly)
    np.testing.assert_array_equal(np.asarray(feed["initial_node_features"], dtype=np.float32), want["features"].astype(np.float32))
    np.testing.assert_array_equal(np.asarray(feed["type_to_num_incoming_edges"]), want["deg"])
    got_gnl = np.asarray(feed["graph_nodes_list"])
    assert got_gnl.dtype == np.int32
    np.testing.assert_array_equal(got_gnl, want["gnl"])
    np.testing.assert_array_equal(np.asarray(feed[payload_key], dtype=np.float32), np.asarray(want["payload"], dtype=np.float32))
    for a, b in zip(feed["adjacency_lists"], want["adj"]):
        a = np.asarray(a)
        assert a.shape == b.shape and (a.shape[0] > 0 or a.dtype == np.int32)      # (empty types: zeros((0, 2), int32))
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("which", range(4))
def test_ppi_loader_and_batches_are_the_reference_s(tmp_path, which):
    from make_reference_run import write_ppi_dir
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    z, manifest = _load("reference_run_tasks.npz")
    entry = manifest["ppi"][which]
    write_ppi_dir(str(tmp_path), manifest["ppi_dir_seed"])
    self_loops, tie, L = entry["add_self_loop_edges"], entry["tie_fwd_bkwd_edges"], entry["num_edge_types"]
    p = PPI_Task.default_params()
    p.update(add_self_loop_edges=self_loops, tie_fwd_bkwd_edges=tie, out_layer_dropout_keep_prob=0.8)
    task = PPI_Task(p)
    task.load_data(str(tmp_path))
    assert task.num_edge_types == L and task.initial_node_feature_size == entry["initial_node_feature_size"]
    assert task.get_metadata() == {k: v for k, v in entry["metadata"].items() if k in task.get_metadata()} or \
        all(task.get_metadata().get(k) == v for k, v in entry["metadata"].items() if k in ("num_edge_types", "initial_node_feature_size", "num_labels"))
    product = {"train": list(task._loaded_data[DataFold.TRAIN]), "valid": list(task._loaded_data[DataFold.VALIDATION]),
               "test": list(task.load_eval_data_from_path(str(tmp_path)))}
    oracle = {}
    for name in ("train", "valid", "test"):
        links = json.load(open(tmp_path / ("%s_graph.json" % name)))["links"]
        feats, labels, gid = (np.load(tmp_path / ("%s_%s.npy" % (name, s))) for s in ("feats", "labels", "graph_id"))
        oracle[name] = bookkeeping.ppi_graphs_from_dgl_arrays(links, feats, labels, gid, self_loops, tie)
        want = _samples(z, "%s/%s" % (entry["prefix"], name), "node_labels")
        assert len(product[name]) == len(want) == len(oracle[name])
        for g, o, w in zip(product[name], oracle[name], want):
            _same_sample(g.adjacency_lists, g.type_to_node_to_num_incoming_edges, g.node_features, g.node_labels, w)
            _same_sample(o[0], o[1], o[2], o[3], w)
    folds = {"train": DataFold.TRAIN, "valid": DataFold.VALIDATION, "test": DataFold.TEST}
    for b in entry["batches"]:
        want = _batches(z, b["key"], L, "target_labels")
        # the oracle's packer on the order the reference's np.random.shuffle left (recorded), the product's iterator under the seed
        ordered = [oracle[b["fold"]][j] for j in b["order_after_shuffle"]]
        samples = [bookkeeping.GraphSample(o[0], o[1], o[2], o[3]) for o in ordered]
        got = list(bookkeeping.pack_batches(samples, L, b["max_nodes_per_batch"]))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            _same_batch(g, (g["num_graphs"], g["num_nodes"], g["num_edges"]), w, "target_labels")
        data = list(product[b["fold"]])
        np.random.seed(b["numpy_seed"])
        mbs = list(task.make_minibatch_iterator(data, folds[b["fold"]], b["max_nodes_per_batch"]))
        assert len(mbs) == len(want)
        for mb, w in zip(mbs, want):
            _same_batch(mb.feed_dict, (mb.num_graphs, mb.num_nodes, mb.num_edges), w, "target_labels")
            assert float(mb.feed_dict["out_layer_dropout_keep_prob"]) == w["keep"]


@pytest.mark.parametrize("which", range(4))
def test_qm9_loader_and_batches_are_the_reference_s(which):
    import gzip
    from tf_gnn_samples_amd.tasks import DataFold, QM9_Task
    z, manifest = _load("reference_run_tasks.npz")
    entry = manifest["qm9"][which]
    self_loops, tie = entry["add_self_loop_edges"], entry["tie_fwd_bkwd_edges"]
    with gzip.open(GOLDEN / "qm9_valid_256.jsonl.gz", "rt") as f:
        raw = [json.loads(line) for line in f]
    p = QM9_Task.default_params()
    p.update(add_self_loop_edges=self_loops, tie_fwd_bkwd_edges=tie, task_ids=entry["task_ids"])
    task = QM9_Task(p)
    if "reference_raises" in entry:
        # The reference cannot load this configuration at all (tasks/qm9_task.py:140-145 appends the backward lists to the list it
        # is enumerating and runs one edge type past the in-degree table: IndexError on the first molecule).  The product computes
        # what the loop evidently means (it enumerates a snapshot) — an extension, not parity; DESIGN.md section 5.
        assert entry["reference_raises"] == "IndexError" and not tie
        train = task.load_raw(raw[:40])
        assert task.num_edge_types == entry["num_edge_types"] or task.num_edge_types % 2 == 0
        assert len(train) == 40 and len(train[0].adjacency_lists) == task.num_edge_types
        return
    train = task.load_raw(raw[:40])                      # the reference loads train first, then valid (load_data :77-79)
    valid = task.load_raw(raw)[:48]
    L = entry["num_edge_types"]
    assert task.num_edge_types == L and task.initial_node_feature_size == entry["initial_node_feature_size"]
    for name, samples in (("train", train), ("valid", valid)):
        want = _samples(z, "%s/%s" % (entry["prefix"], name), "target_values")
        assert len(samples) == len(want)
        for g, w in zip(samples, want):
            _same_sample(g.adjacency_lists, g.type_to_node_to_num_incoming_edges, g.node_features, g.target_values, w)
    # the oracle's restatement of the per-molecule conversion, on the raw triples
    for d, w in zip(raw[:48], _samples(z, entry["prefix"] + "/valid", "target_values")):
        o_adj, o_deg = bookkeeping.qm9_graph_to_adjacency_lists(d["graph"], len(d["node_features"]), L, self_loops, tie)
        for a, b in zip(o_adj, w[0]):
            np.testing.assert_array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2))
        np.testing.assert_array_equal(o_deg, w[1])
    folds = {"train": (DataFold.TRAIN, train), "valid": (DataFold.VALIDATION, valid)}
    for b in entry["batches"]:
        want = _batches(z, b["key"], L, "target_values")
        fold, data = folds[b["fold"]]
        data = list(data)
        np.random.seed(b["numpy_seed"])
        mbs = list(task.make_minibatch_iterator(data, fold, b["max_nodes_per_batch"]))
        assert len(mbs) == len(want)
        for mb, w in zip(mbs, want):
            _same_batch(mb.feed_dict, (mb.num_graphs, mb.num_nodes, mb.num_edges), w, "target_values")


def test_micro_f1_and_activation_table_are_the_reference_s():
    import torch
    from tf_gnn_samples_amd.utils import micro_f1
    z, manifest = _load("reference_run_tasks.npz")
    assert manifest["constants"] == dict(SMALL_NUMBER=T.SMALL_NUMBER, BIG_NUMBER=1e7)
    # utils/utils.py:60-74 (round(sigmoid) on ints, counts, precision / recall in float64, cast to float32) — the product's torch
    # restatement on the same logits / labels, incl. logits of exactly 0 (sigmoid = 0.5 rounds to 0: half to even)
    got = micro_f1(torch.from_numpy(z["micro_f1/logits"]), torch.from_numpy(z["micro_f1/labels"]))
    assert np.float32(float(got)) == np.float32(z["micro_f1/value"])
    x = z["activations/x"]
    for name in manifest["activations"]:
        np.testing.assert_array_equal(T.apply_act(T.get_activation(name), x), z["activations/" + name])


# ------------------------------------------------------------------------------------------------------------------------------
# whole forward models: Sparse_Graph_Model.__make_model of the reference (all but the optimizer) on one minibatch
# ------------------------------------------------------------------------------------------------------------------------------
def model_cases():
    z, manifest = _load("reference_run_models.npz")
    return z, [m for m in manifest if m["key"].startswith("model")], next(m for m in manifest if m["key"] == "readme_rgcn_ppi")


def build_product_task(entry, tmp_path):
    """The package's task with the fixture's task parameters and the metadata the reference's loader derived from the same files."""
    import gzip
    from make_reference_run import write_ppi_dir
    from tf_gnn_samples_amd.tasks import PPI_Task, QM9_Task
    cls = PPI_Task if entry["task"] == "PPI" else QM9_Task
    p = cls.default_params()
    p.update({k: v for k, v in entry["task_params"].items() if k in p})
    task = cls(p)
    if entry["task"] == "PPI":
        write_ppi_dir(str(tmp_path), 11)
        task.load_data(str(tmp_path))
    else:
        with gzip.open(GOLDEN / "qm9_valid_256.jsonl.gz", "rt") as f:
            raw = [json.loads(line) for line in f]
        task.load_raw(raw[:40])
        task.load_raw(raw)
    assert task.num_edge_types == entry["num_edge_types"]
    return task


def build_product_model(entry, task, device):
    from tf_gnn_samples_amd import models
    cls = getattr(models, entry["model"])
    p = cls.default_params()
    p.update(entry["model_params"])
    return cls(p, task, device=device)


MODEL_Z, MODEL_CASES, README_CASE = model_cases()


@pytest.mark.parametrize("i", range(len(MODEL_CASES)), ids=["%s-%s-%d" % (m["model"], m["task"], n) for n, m in enumerate(MODEL_CASES)])
def test_model_variable_inventory_is_the_reference_s(tmp_path, capsys, i):
    """Names (TF variable names incl. the graph-wide dense / dense_1 numbering and the per-scope LayerNorm / MLP numbering), shapes,
    creation order and the logged parameter count of every model class, as the reference's own __make_model produced them."""
    entry = MODEL_CASES[i]
    task = build_product_task(entry, tmp_path)
    model = build_product_model(entry, task, "cpu")
    names = model.variables.names()
    # (creation order is not part of the contract — checkpoints are dictionaries by name, sparse_graph_model.py:90-126 — and differs
    #  where TF builds a layer at its first call: e.g. RGAT's attention parameters exist before any Dense kernel)
    assert {n: list(model.variables[n].shape) for n in names} == dict(zip(entry["variables"], entry["variable_shapes"]))
    assert len(names) == len(entry["variables"])
    logged = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Model has")]
    assert logged == [l for l in entry["logged"] if l.startswith("Model has")]


def test_readme_parameter_count_comes_out_of_the_reference_s_own_code(tmp_path, capsys):
    """README.md:29 'Model has 699257 parameters.': logged by the reference's __make_model run over the shim at the README's
    hyper-parameters — a known answer of the reference that the shim's variable bookkeeping has to reproduce — and by the package."""
    from oracle import model as OM
    assert README_CASE["logged"] == ["Model has 699257 parameters."]
    assert OM.rgcn_ppi_num_parameters() == 699257
    shapes = dict(zip(README_CASE["variables"], README_CASE["variable_shapes"]))
    assert shapes["graph_model/dense/kernel"] == [50, 256] and shapes["dense_1/kernel"] == [256, 121]
    assert "graph_model/gnn_layer_0/Dense/kernel" in shapes and "graph_model/gnn_layer_1/Dense/kernel" not in shapes


@pytest.mark.parametrize("i", [n for n, m in enumerate(MODEL_CASES) if m["model"] == "RGCN_Model"])
def test_oracle_driver_loop_and_ppi_head_are_the_reference_s(i):
    from oracle import model as OM
    entry, z = MODEL_CASES[i], MODEL_Z
    k = entry["key"]
    W = {n[len("graph_model/"):]: z["%s/var/%s" % (k, n)] for n in entry["variables"] if n.startswith("graph_model/")}
    adj = [z["%s/adj%d" % (k, l)] for l in range(entry["num_edge_types"])]
    final = OM.graph_propagation(z[k + "/features"], adj, z[k + "/deg"], entry["model_params"], W, OM.rgcn_apply(entry["model_params"]))
    want = z[k + "/final_node_representations"]
    np.testing.assert_allclose(final, want, rtol=0, atol=2e-7 * max(1.0, float(np.abs(want).max())))
    # the head is the graph's second unnamed Keras Dense (dense_1) when there is an input projection, the first (dense) when
    # hidden_size equals the feature size (sparse_graph_model.py:165-172)
    head = "dense_1" if "graph_model/dense/kernel" in entry["variables"] else "dense"
    assert (head == "dense") == (entry["model_params"]["hidden_size"] == z[k + "/features"].shape[1])
    loss, _ = OM.ppi_head_loss(final, z[k + "/target_labels"], z["%s/var/%s/kernel" % (k, head)], z["%s/var/%s/bias" % (k, head)])
    assert abs(float(loss) - float(z[k + "/metric/loss"])) <= 1e-6 * abs(float(z[k + "/metric/loss"]))


def test_default_hyper_parameters_and_name_tables_are_the_reference_s():
    """default_params() of every model / task class, model.name(params), name_to_model_class / name_to_task_class
    (utils/model_utils.py) incl. their error texts — as returned by the reference's own classes."""
    from tf_gnn_samples_amd import models, tasks
    _, manifest = _load("reference_run_models.npz")
    d = next(m for m in manifest if m["key"] == "defaults")
    for cls_name, want in d["models"].items():
        cls = getattr(models, cls_name)
        got = cls.default_params()
        assert {k: got.get(k, "<missing>") for k in want} == want, cls_name
        assert cls.name(got) == d["display_names"][cls_name]
    for cls_name, want in d["tasks"].items():
        cls = getattr(tasks, cls_name)
        got = cls.default_params()
        assert {k: got.get(k, "<missing>") for k in want["params"]} == want["params"], cls_name
        assert cls.name() == want["name"] and cls.default_data_path() == want["data_path"]
    for name, (cls_name, extra) in d["model_names"].items():
        if cls_name == "ValueError":
            with pytest.raises(ValueError) as e:
                models.name_to_model_class(name)
            assert str(e.value) == extra
        else:
            cls, got_extra = models.name_to_model_class(name)
            assert cls.__name__ == cls_name and got_extra == extra, name
    for name, (cls_name, extra) in d["task_names"].items():
        if cls_name == "ValueError":
            with pytest.raises(ValueError) as e:
                tasks.name_to_task_class(name)
            assert str(e.value) == extra
        else:
            cls, got_extra = tasks.name_to_task_class(name)
            assert cls.__name__ == cls_name and got_extra == extra, name


def test_unknown_names_raise_what_the_reference_raises():
    """utils/utils.py:19-20, 33-34, 57-58 executed: exception types and texts, the spellings of 'no activation', the aliases of the
    aggregation functions — in the package's utils and in the op layer that picks the kernels."""
    import torch
    from tf_gnn_samples_amd import ops, utils
    _, manifest = _load("reference_run_tasks.npz")
    e = manifest["errors"]
    with pytest.raises(ValueError) as err:
        utils.get_activation("swish")
    assert [type(err.value).__name__, str(err.value)] == e["get_activation('swish',)"]
    for arg, key in (("median", "get_aggregation_function('median',)"), (None, "get_aggregation_function(None,)")):
        with pytest.raises(ValueError) as err:
            utils.get_aggregation_function(arg)
        assert [type(err.value).__name__, str(err.value)] == e[key]
        with pytest.raises(ValueError) as err:
            ops.aggregation_mode_id(arg)
        assert str(err.value) == e[key][1]
    with pytest.raises(Exception) as err:
        utils.get_gated_unit(8, "xyz", "tanh", {})
    assert type(err.value) is Exception and str(err.value) == e["get_gated_unit(8, 'xyz', 'tanh')"][1]
    for n in manifest["no_activation"]:
        assert utils.get_activation(n) is None
    for alias, fn in manifest["aggregation_aliases"].items():
        assert ops.aggregation_mode_id(alias) == ops.aggregation_mode_id(fn.replace("unsorted_segment_", ""))
        assert utils.get_aggregation_function(alias).__name__ == fn


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's sources under torch.autograd (tests/golden/tf_torch_shim.py): gradients and the train step
# ------------------------------------------------------------------------------------------------------------------------------
def autograd_fixture():
    z, manifest = _load("reference_run_autograd.npz")
    return z, manifest


AUTOGRAD_Z, AUTOGRAD = autograd_fixture()
NON_SMOOTH = ("relu", "leaky_relu", "selu")


@pytest.mark.parametrize("i", range(len(CASES)), ids=["%s-%d" % (c[0]["function"], n) for n, c in enumerate(CASES)])
def test_torch_mirrors_differentiate_like_the_reference_s_own_code(i):
    """oracle/torch_ref.py supplies the float64 reference GRADIENTS of the GPU tests: its layer functions against d loss / d (node
    states, every variable) of the reference's layer code run under autograd — same primitives on both sides, so a faithful mirror
    agrees to rounding."""
    import torch
    from oracle import torch_ref as R
    case, h, adj, deg, weights, _ = CASES[i]
    k, z = case["key"], AUTOGRAD_Z
    entry = AUTOGRAD["layers"][i]
    assert entry["function"] == case["function"] and entry["variables"] == case["variables"]
    ht = torch.tensor(h.astype(np.float64), requires_grad=True)
    wt = {n: torch.tensor(v.astype(np.float64), requires_grad=True) for n, v in weights.items()}
    kw = dict(case["kwargs"])
    fn = getattr(R, case["function"])
    adj_t, deg_t = [torch.as_tensor(a.astype(np.int64)) for a in adj], torch.as_tensor(deg.astype(np.float64))
    if case["function"] == "sparse_rgdcn_layer":
        out = fn(ht, adj_t, deg_t, weights=wt, **kw)
    else:
        state_dim = kw.pop("state_dim")
        out = fn(ht, adj_t, deg_t, state_dim, weights=wt, **kw) if case["takes_degrees"] else fn(ht, adj_t, state_dim, weights=wt, **kw)
    np.testing.assert_allclose(out.detach().numpy(), z[k + "/out64"], rtol=0, atol=1e-12 * max(1.0, float(np.abs(z[k + "/out64"]).max())))
    loss = (out * torch.as_tensor(z[k + "/cotangent"])).sum()
    names = case["variables"]
    grads = torch.autograd.grad(loss, [ht] + [wt[n] for n in names], allow_unused=True)
    for name, g in zip(["h"] + names, grads):
        want = z["%s/grad/%s" % (k, "h" if name == "h" else "var/" + name)]
        got = np.zeros_like(want) if g is None else g.numpy()
        assert np.abs(got - want).max() <= 1e-10 * max(1.0, float(np.abs(want).max())), (case["function"], name)


@pytest.mark.parametrize("i", range(len(AUTOGRAD["train"])), ids=["%s-%s" % (t["model"], t["steps"][0]["optimizer"]) for t in AUTOGRAD["train"]])
def test_oracle_clip_and_optimizers_reproduce_the_reference_s_train_step(i):
    """sparse_graph_model.py:227-260 executed by the reference (compute_gradients over the trainable variables, tf.clip_by_norm of
    every gradient with clamp_gradient_norm, the optimizer the parameters name, the learning rate scaled by num_graphs /
    lr_for_num_graphs_per_batch): oracle/optim.py's float32 classes, fed the recorded raw gradients, must land on the recorded
    variables after each of the two steps."""
    from oracle import optim
    t, z = AUTOGRAD["train"][i], AUTOGRAD_Z
    k, names, mp = t["key"], t["variables"], t["model_params"]
    lr_scale = 1.0 if mp.get("lr_for_num_graphs_per_batch") is None else t["num_graphs"] / mp["lr_for_num_graphs_per_batch"]
    assert abs(t["steps"][0]["learning_rate"] - mp["learning_rate"] * lr_scale) <= 1e-12
    kind = mp["optimizer"].lower()
    assert t["steps"][0]["optimizer"] == {"adam": "_Adam", "rmsprop": "_RMSProp", "sgd": "_SGD"}[kind]
    init = [z["%s/initial/%s" % (k, n)] for n in names]
    if kind == "adam":
        opt = optim.Adam(init, mp["learning_rate"])
    elif kind == "rmsprop":
        opt = optim.RMSProp(init, mp["learning_rate"], decay=mp["learning_rate_decay"], momentum=mp["momentum"])
    else:
        opt = optim.GradientDescent(init, mp["learning_rate"])
    for step in range(2):
        raw = [z["%s/step%d/raw_gradient/%s" % (k, step, n)] for n in names]
        unused = set(t["steps"][step]["without_gradient"])
        clipped = [None if n in unused else optim.clip_by_norm(g, mp["clamp_gradient_norm"]) for n, g in zip(names, raw)]
        for n, c in zip(names, clipped):
            want = z["%s/step%d/applied_gradient/%s" % (k, step, n)]
            if c is not None:
                assert np.abs(c - want).max() <= 2e-6 * max(1e-30, float(np.abs(want).max())), n
                assert float(np.sqrt((want ** 2).sum())) <= mp["clamp_gradient_norm"] * (1 + 1e-9)
        opt.apply_gradients(clipped, lr_scale=lr_scale)
        for n, v in zip(names, opt.vars):
            want = z["%s/step%d/variable_after/%s" % (k, step, n)]
            assert np.abs(v - want).max() <= 3e-6 * max(1.0, float(np.abs(want).max())), (step, n)


@pytest.mark.parametrize("entry", [m for m in MODEL_CASES if m.get("checkpoint")], ids=lambda m: "%s-%s" % (m["task"], m["model"]))
def test_checkpoints_written_by_the_reference_s_save_model_restore_into_the_package(entry, capsys, tmp_path):
    """tests/golden/reference_run_checkpoints/*.pickle come out of the reference's own save_model (sparse_graph_model.py:90-107); the
    package's restore() (utils/model_utils.py:60-77) rebuilds task and model from them: class names through the name tables, metadata,
    every model variable by its TF name with ':0' — nothing freshly initialised, nothing but the step counter left over."""
    import pickle
    from tf_gnn_samples_amd import models
    path = GOLDEN / entry["checkpoint"]
    data = pickle.load(open(path, "rb"))
    assert set(data) == {"model_class", "task_class", "model_params", "task_params", "task_metadata", "weights"}
    assert set(data["weights"]) == {n + ":0" for n in entry["variables"]} | {"total_num_graphs:0"}
    model = models.restore(str(path), str(tmp_path), device="cpu")
    out = capsys.readouterr().out
    assert "Freshly initializing" not in out
    assert [l for l in out.splitlines() if "not used by model" in l] in ([], ["Saved weights for total_num_graphs:0 not used by model."])
    assert type(model).__name__ == entry["model"] and type(model.task).__name__ == entry["task"] + "_Task"
    assert model.task.num_edge_types == entry["num_edge_types"]
    k = entry["key"]
    for n in entry["variables"]:
        np.testing.assert_array_equal(model.variables[n].detach().numpy(), MODEL_Z["%s/var/%s" % (k, n)])


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json's single-GPU configurations at full size, through the reference's own model code (run_c2_full_size)
# ------------------------------------------------------------------------------------------------------------------------------
def baseline_reference_run(which):
    """(fixture arrays of configuration `which`, its manifest, regenerated variables, the package's task, the batch as the package
    builds it) for which in c2 (RGCN, configs[1]), c4 (RGAT, configs[3]), c3 (GGNN on real molecules, configs[2])."""
    import gzip
    from make_reference_run import regenerate_variables
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task, QM9_Task
    zz, mm = _load("reference_run_baseline_size.npz")
    m = mm[which]
    z = {k[len(which) + 1:]: zz[k] for k in zz.files if k.startswith(which + "/")}
    W = regenerate_variables(m["variables"], m["variable_shapes"], m["variable_seed"])
    for n, s in m["variable_checksums"].items():
        assert float(np.asarray(W[n], np.float64).sum()) == s, n
    W = {n: np.array(v) for n, v in W.items()}
    if which == "c3":
        task = QM9_Task(QM9_Task.default_params())
        with gzip.open(GOLDEN / "qm9_valid_256.jsonl.gz", "rt") as f:
            raw = [json.loads(line) for line in f]
        data = task.load_raw(raw)
        mb = next(task.make_minibatch_iterator(list(data), DataFold.VALIDATION, 10 ** 9))
    else:
        task = PPI_Task(PPI_Task.default_params())
        task.load_synthetic(16, 1, seed=0)
        mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    assert (mb.num_nodes, mb.num_edges, mb.num_graphs) == (m["num_nodes"], m["num_edges"], m["num_graphs"])
    return z, m, W, task, mb


def _oracle_adapter(which, p):
    """models/{rgcn,rgat,ggnn}_model.py:_apply_gnn_layer restated for oracle.model.graph_propagation."""
    from oracle import model as OM
    if which == "c2":
        return OM.rgcn_apply(p, node_side_transform=True)
    if which == "c4":
        return lambda i, h, adj, deg, steps, w: G.sparse_rgat_layer(h, adj, p['hidden_size'], num_heads=p['num_heads'], num_timesteps=steps,
                                                                   activation_function=p['graph_activation_function'], weights=w)
    return lambda i, h, adj, deg, steps, w: G.sparse_ggnn_layer(
        h, adj, p['hidden_size'], num_timesteps=steps, gated_unit_type=p['graph_rnn_cell'],
        activation_function=p['graph_activation_function'], message_aggregation_function=p['message_aggregation_function'],
        weights={k: v for k, v in w.items() if not k.startswith(("LayerNorm", "Dense"))})


@pytest.mark.parametrize("which", ["c2", "c3"] + (["c4"] if __import__("os").environ.get("RELGNN_TEST_SLOW") else []))
def test_oracle_at_baseline_size_is_the_reference_s_model_code_at_baseline_size(which):
    """The oracle's driver loop (for C2 in its BASELINE-size evaluation order: node-side transform, C fold in the reference's message
    order) on the full batches against what the reference's own model code computed there (for C2 / C4: per-edge work over 1.85 M
    messages): 96 sampled rows, every row's norm and every column's sum of the final node representations; for C2 the loss too.
    (c4 — the NumPy RGAT over 1.85 M messages, 2.5 minutes — only with RELGNN_TEST_SLOW=1; the HIP path is held to the c4 fixture
    directly in tests/test_gpu_reference_run.py.)"""
    from oracle import model as OM
    z, m, W, task, mb = baseline_reference_run(which)
    if which == "c2":
        assert m["logged"] == ["Model has 699257 parameters."]
    fd = mb.feed_dict
    p = m["model_params"]
    Wg = {n[len("graph_model/"):]: v for n, v in W.items() if n.startswith("graph_model/")}
    final = OM.graph_propagation(fd['initial_node_features'].astype(np.float32), fd['adjacency_lists'],
                                 fd['type_to_num_incoming_edges'].astype(np.float32), p, Wg, _oracle_adapter(which, p))
    scale = max(1.0, m["final_abs_max"])
    assert np.abs(final[z["rows"]] - z["final_rows"]).max() <= 2e-6 * scale
    assert np.abs(np.sqrt((final.astype(np.float64) ** 2).sum(1)) - z["final_row_l2"]).max() <= 1e-5 * scale
    assert np.abs(final.astype(np.float64).sum(0) - z["final_column_sum"]).max() <= 1e-3 * scale
    if which == "c2":
        loss, logits = OM.ppi_head_loss(final, fd['target_labels'].astype(np.float32), W["dense_1/kernel"], W["dense_1/bias"])
        assert abs(float(loss) - m["metrics"]["loss"]) <= 2e-6 * m["metrics"]["loss"]
