#!/usr/bin/env python
"""Regenerates the fixtures under tests/golden/.  Run from the repo root IN THE BUILD CONTAINER
(it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

  qm9_valid_256.jsonl.gz   the first 256 molecules of the reference's data/qm9/valid.jsonl.gz (CC0, see the
                           reference's data/qm9/LICENSE), verbatim lines: REAL graphs for config C3.
  layers_small.npz         seeded inputs + weights + the ORACLE's float32 outputs for all six layers on a small
                           3-edge-type graph.  The reference itself cannot run here (no TensorFlow), so these
                           are oracle outputs, not reference outputs: they pin the oracle against silent drift and
                           give the GPU tests a fixed vector.  (Outputs of the reference's OWN code, run over a shim of
                           the TF symbols it touches: make_reference_run.py -> reference_run_*.npz.)
"""
import gzip
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import gnns as G  # noqa: E402
from helpers import degree_table, glorot, random_relational_graph, rgcn_weights  # noqa: E402

OUT = Path(__file__).resolve().parent


def make_qm9(n=256):
    src = Path("/root/reference/data/qm9/valid.jsonl.gz")
    with gzip.open(src, "rt") as f, gzip.open(OUT / "qm9_valid_256.jsonl.gz", "wt") as g:
        for i, line in enumerate(f):
            if i >= n:
                break
            g.write(line)


def layer_cases(seed=1234, V=48, D=16, L=3, K=4):
    rng = np.random.default_rng(seed)
    adj = random_relational_graph(rng, V, L, [160, 0, 40])
    adj[1] = np.stack([np.arange(V), np.arange(V)], 1).astype(np.int32)   # self loops, like PPI type 1
    deg = degree_table(adj, V)
    h = np.tanh(rng.standard_normal((V, D))).astype(np.float32)
    ln = {"LayerNorm/gamma": (1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
          "LayerNorm/beta": (0.1 * rng.standard_normal(D)).astype(np.float32)}
    # the layers below run 2 timesteps: the reference's tf.contrib.layers.layer_norm call of the SECOND timestep owns
    # its own variables (TF scope LayerNorm_1).  Drawn from a separate stream so that every other fixture entry stays
    # what it was.
    rng_ln = np.random.default_rng(seed + 1)
    ln["LayerNorm_1/gamma"] = (1 + 0.1 * rng_ln.standard_normal(D)).astype(np.float32)
    ln["LayerNorm_1/beta"] = (0.1 * rng_ln.standard_normal(D)).astype(np.float32)
    w = {}
    w["rgcn"] = rgcn_weights(rng, L, D, D)
    w["ggnn"] = dict(rgcn_weights(rng, L, D, D), **{"gru_cell/kernel": glorot(rng, (D, 3 * D)),
                                                    "gru_cell/recurrent_kernel": glorot(rng, (D, 3 * D)),
                                                    "gru_cell/bias": (0.1 * rng.standard_normal(3 * D)).astype(np.float32)})
    w["rgat"] = rgcn_weights(rng, L, D, D)
    for l in range(L):
        w["rgat"]["Edge_%i_Attention_Parameters" % l] = (0.3 * rng.standard_normal(2 * D)).astype(np.float32)
    w["film"] = dict(rgcn_weights(rng, L, D, D), **ln)
    for l in range(L):
        w["film"]["Edge_%i_FiLM_Computations/kernel" % l] = glorot(rng, (D, 2 * D))
    w["rgin"] = dict(ln)
    w["edge_mlp"] = dict(ln)
    for l in range(L):
        w["rgin"]["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (D, D))
        w["rgin"]["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
        w["edge_mlp"]["Edge_%i_MLP/dense/kernel" % l] = glorot(rng, (2 * D, D))
        w["edge_mlp"]["Edge_%i_MLP/dense_1/kernel" % l] = glorot(rng, (D, D))
    out = {
        "rgcn": G.sparse_rgcn_layer(h, adj, deg, D, 2, "ReLU", "sum", weights=w["rgcn"]),
        "ggnn": G.sparse_ggnn_layer(h, adj, D, 2, "gru", "tanh", "mean", weights=w["ggnn"]),
        "rgat": G.sparse_rgat_layer(h, adj, D, K, 2, "tanh", weights=w["rgat"]),
        "film": G.sparse_gnn_film_layer(h, adj, deg, D, 2, "ReLU", "sum", weights=w["film"]),
        "rgin": G.sparse_rgin_layer(h, adj, D, 2, "ReLU", "sum", weights=w["rgin"]),
        "edge_mlp": G.sparse_gnn_edge_mlp_layer(h, adj, deg, D, 2, "gelu", "sum", weights=w["edge_mlp"]),
    }
    flat = {"h": h, "deg": deg, "num_heads": np.int32(K)}
    for l, a in enumerate(adj):
        flat["adj_%d" % l] = a
    for layer, ws in w.items():
        for k, v in ws.items():
            flat["w|%s|%s" % (layer, k)] = v
    for layer, o in out.items():
        flat["out|%s" % layer] = o.astype(np.float32)
    np.savez_compressed(OUT / "layers_small.npz", **flat)


if __name__ == "__main__":
    make_qm9()
    layer_cases()
    print("wrote", sorted(p.name for p in OUT.iterdir()))
