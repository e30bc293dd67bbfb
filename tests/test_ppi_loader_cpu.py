"""DGL-PPI file loader (reference: tasks/ppi_task.py:76-162, file set :87-90): a synthetic data
directory in the DGL ppi.zip layout is written to disk and loaded through PPI_Task.load_data /
load_eval_data_from_path; graphs, shifted node ids, edge order, self-loop / backward edge types and the per-type
in-degree tables must equal the line-by-line restatement of the reference loader (oracle/bookkeeping.py), bit for bit,
for every combination of the two task parameters."""
import json

import numpy as np
import pytest

from oracle import bookkeeping


def _write_fold(path, name, rng, sizes, graph_ids, F=6, C=4):
    n = int(sum(sizes))
    gid = np.concatenate([np.full(s, g, np.int64) for s, g in zip(sizes, graph_ids)])
    feats = rng.standard_normal((n, F)).astype(np.float32)
    labels = (rng.random((n, C)) < 0.4).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    links = []
    for k, s in enumerate(sizes):          # links of different graphs interleaved in the file, duplicates + self links kept
        for _ in range(3 * s):
            links.append({"source": int(starts[k] + rng.integers(0, s)), "target": int(starts[k] + rng.integers(0, s))})
    order = rng.permutation(len(links))
    links = [links[i] for i in order]
    links.append(dict(links[0]))           # a duplicated edge
    with open(path / ("%s_graph.json" % name), "w") as f:
        json.dump({"directed": False, "multigraph": False, "links": links, "nodes": [{"id": i} for i in range(n)]}, f)
    np.save(path / ("%s_feats.npy" % name), feats)
    np.save(path / ("%s_labels.npy" % name), labels)
    np.save(path / ("%s_graph_id.npy" % name), gid)
    return links, feats, labels, gid


@pytest.mark.parametrize("self_loops,tie", [(True, False), (True, True), (False, False), (False, True)])
def test_dgl_ppi_files_load_like_the_reference(tmp_path, self_loops, tie):
    from tf_gnn_samples_amd.tasks import DataFold, PPI_Task
    rng = np.random.default_rng(3)
    folds = {}
    for name, sizes, gids in (("train", [7, 1, 12, 5], [5, 9, 2, 21]), ("valid", [4, 9], [23, 24]), ("test", [6], [1])):
        folds[name] = _write_fold(tmp_path, name, rng, sizes, gids)
    p = PPI_Task.default_params()
    p.update(add_self_loop_edges=self_loops, tie_fwd_bkwd_edges=tie)
    task = PPI_Task(p)
    task.load_data(str(tmp_path))
    loaded = {"train": task._loaded_data[DataFold.TRAIN], "valid": task._loaded_data[DataFold.VALIDATION],
              "test": task.load_eval_data_from_path(str(tmp_path))}
    L = 1 + int(self_loops) + int(not tie)
    assert task.num_edge_types == L and task.initial_node_feature_size == 6 and task.num_labels == 4
    for name, (links, feats, labels, gid) in folds.items():
        want = bookkeeping.ppi_graphs_from_dgl_arrays(links, feats, labels, gid, self_loops, tie)
        got = loaded[name]
        assert len(got) == len(want)
        for g, (adj, deg, f, l) in zip(got, want):
            assert len(g.adjacency_lists) == L
            for a, b in zip(g.adjacency_lists, adj):
                np.testing.assert_array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2))
            np.testing.assert_array_equal(np.asarray(g.type_to_node_to_num_incoming_edges), deg)
            np.testing.assert_array_equal(g.node_features, f)
            np.testing.assert_array_equal(g.node_labels, l)
    # the loaded graphs batch like any others (ids shifted to start at 0 in every graph)
    mb = next(task.make_minibatch_iterator(list(loaded["train"]), DataFold.VALIDATION, 10 ** 6))
    assert mb.num_graphs == 4 and mb.num_nodes == 25
    for a in mb.feed_dict["adjacency_lists"]:
        assert a.dtype == np.int32 and (a.size == 0 or (a.min() >= 0 and a.max() < 25))
