#!/usr/bin/env python
"""Pins the oracle (and through it the HIP path) to the REAL reference: runs the UNMODIFIED layer functions of
microsoft/tf-gnn-samples (gnns/*.py, utils/utils.py) under TensorFlow 1.13 + dpu_utils on the committed fixture
tests/golden/layers_small.npz — same inputs, same weights, assigned to the variables the reference creates by NAME —
and writes

    tests/golden/tf_layers_small.npz      out|<layer> : what TF computed;  meta|* : versions, variable lists
    tests/golden/tf_semantics.npz         single-op vectors for every TF-internal assumption oracle/tf_ops.py and
                                          oracle/optim.py make (GRUCell step, hard_sigmoid, layer_norm, leaky_relu,
                                          unsorted_segment_{max,mean,sqrt_n} incl. empty segments,
                                          unsorted_segment_log_softmax, clip_by_norm, Adam / RMSProp / SGD steps)

tests/test_tf_golden.py consumes both files when they exist (oracle AND HIP path against TF, 1e-5 abs) and reports
"PARITY UNPINNED" while they do not.

This cannot run in the build container or on the GPU box (Python 3.10, no TensorFlow wheel, no network).  On any
machine with the reference's requirements.txt (tensorflow-gpu>=1.13.1 or tensorflow==1.13.*/1.15.*, dpu-utils>=0.1.30,
Python <= 3.7):

    git clone https://github.com/microsoft/tf-gnn-samples /path/to/reference
    python scripts/dump_tf_golden.py --reference /path/to/reference
    git add tests/golden/tf_layers_small.npz tests/golden/tf_semantics.npz

The script only READS the reference checkout; nothing of it is copied.  A variable the reference creates that the
fixture has no value for (or vice versa) is an ERROR: the name lists are part of what is being pinned.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"

# layer -> (function name, positional builder, kwargs) exactly as tests/test_golden_cpu.py::oracle_layer_outputs calls the oracle
CASES = {
    "rgcn": ("sparse_rgcn_layer", True, dict(num_timesteps=2, activation_function="ReLU", message_aggregation_function="sum")),
    "ggnn": ("sparse_ggnn_layer", False, dict(num_timesteps=2, gated_unit_type="gru", activation_function="tanh",
                                              message_aggregation_function="mean")),
    "rgat": ("sparse_rgat_layer", False, dict(num_timesteps=2, activation_function="tanh")),
    "film": ("sparse_gnn_film_layer", True, dict(num_timesteps=2, activation_function="ReLU", message_aggregation_function="sum")),
    "rgin": ("sparse_rgin_layer", False, dict(num_timesteps=2, activation_function="ReLU", message_aggregation_function="sum")),
    "edge_mlp": ("sparse_gnn_edge_mlp_layer", True, dict(num_timesteps=2, activation_function="gelu",
                                                         message_aggregation_function="sum")),
}


def load_fixture():
    z = np.load(GOLDEN / "layers_small.npz")
    adj = [z["adj_%d" % l] for l in range(3)]
    weights = {}
    for k in z.files:
        if k.startswith("w|"):
            _, layer, name = k.split("|", 2)
            weights.setdefault(layer, {})[name] = z[k]
    return z["h"], adj, z["deg"], int(z["num_heads"]), weights


def run_layers(tf, gnns, out):
    h, adj, deg, K, weights = load_fixture()
    D = h.shape[1]
    for layer, (fn_name, takes_deg, kwargs) in CASES.items():
        g = tf.Graph()
        with g.as_default():
            ph_h = tf.placeholder(tf.float32, [None, D])
            ph_adj = [tf.placeholder(tf.int32, [None, 2]) for _ in adj]
            ph_deg = tf.placeholder(tf.float32, [len(adj), None])          # fed as float32 (tasks/sparse_graph_task.py:144)
            args = dict(node_embeddings=ph_h, adjacency_lists=ph_adj, state_dim=D, **kwargs)
            if takes_deg:
                args["type_to_num_incoming_edges"] = ph_deg
            if layer == "rgat":
                args["num_heads"] = K
            with tf.variable_scope(layer):
                result = getattr(gnns, fn_name)(**args)
            variables = g.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)
            names = sorted(v.name[len(layer) + 1:-2] for v in variables)      # strip "<layer>/" and ":0"
            have = sorted(weights[layer])
            if names != have:
                raise SystemExit("variable names differ for %s:\n  reference creates %s\n  fixture holds     %s"
                                 % (layer, names, have))
            with tf.Session(graph=g) as sess:
                sess.run(tf.global_variables_initializer())
                for v in variables:
                    value = weights[layer][v.name[len(layer) + 1:-2]]
                    sess.run(v.assign(value.reshape(v.shape.as_list())))
                feed = {ph_h: h, ph_deg: deg.astype(np.float32)}
                feed.update({p: a.reshape(-1, 2) for p, a in zip(ph_adj, adj)})
                out["out|" + layer] = sess.run(result, feed_dict=feed).astype(np.float32)
            out["meta|variables|" + layer] = np.array(names)


def run_semantics(tf, out):
    """One vector per [TF-internal] assumption of oracle/tf_ops.py / oracle/optim.py; inputs are stored next to outputs."""
    from dpu_utils.tfutils import unsorted_segment_log_softmax
    rng = np.random.default_rng(7)
    g = tf.Graph()
    with g.as_default(), tf.Session(graph=g) as sess:
        x = (rng.standard_normal((6, 5)) * 2).astype(np.float32)
        ids = np.array([0, 2, 2, 5, 0, 2], np.int32)            # segments 1, 3, 4 empty
        out["in|x"], out["in|ids"] = x, ids
        for name, fn in (("sum", tf.unsorted_segment_sum), ("max", tf.unsorted_segment_max),
                         ("mean", tf.unsorted_segment_mean), ("sqrt_n", tf.unsorted_segment_sqrt_n)):
            out["seg|" + name] = sess.run(fn(x, ids, 6))
        out["seg|max_negative_id"] = sess.run(tf.unsorted_segment_max(x, np.array([0, -1, 2, 5, 0, 2], np.int32), 6))
        out["seg|log_softmax"] = sess.run(unsorted_segment_log_softmax(x[:, 0], ids, 6))
        probe = np.array([-3.0, -2.5, -1.0, -1e-3, 0.0, 1e-3, 1.0, 2.5, 3.0], np.float32)
        out["in|probe"] = probe
        out["act|leaky_relu"] = sess.run(tf.nn.leaky_relu(probe))
        out["act|elu"], out["act|selu"] = sess.run(tf.nn.elu(probe)), sess.run(tf.nn.selu(probe))
        out["act|hard_sigmoid"] = sess.run(tf.keras.backend.hard_sigmoid(tf.constant(probe)))
        out["act|gelu_erf"] = sess.run(probe * 0.5 * (1.0 + tf.erf(probe / tf.sqrt(2.0))))
        deg = np.array([0.0, 1.0, 2.0, 3.0, 7.0], np.float32)
        out["in|deg"], out["misc|inv_degree"] = deg, sess.run(1.0 / (tf.constant(deg) + 1e-7))
        # layer norm incl. a constant row and a tiny-variance row (variance_epsilon)
        ln_x = np.stack([rng.standard_normal(5), np.ones(5), np.array([0, 2e-6, 0, 2e-6, 1e-6])]).astype(np.float32)
        ph = tf.placeholder(tf.float32, [None, 5])
        ln = tf.contrib.layers.layer_norm(ph)
        sess.run(tf.global_variables_initializer())
        out["in|ln_x"], out["ln|out"] = ln_x, sess.run(ln, {ph: ln_x})
        # Keras cells: one step, weights assigned by name
        for kind, cls in (("gru", tf.keras.layers.GRUCell), ("rnn", tf.keras.layers.SimpleRNNCell)):
            cell = cls(4, activation=tf.tanh)
            xin, hin = rng.standard_normal((3, 4)).astype(np.float32), np.tanh(rng.standard_normal((3, 4))).astype(np.float32)
            px, phh = tf.placeholder(tf.float32, [None, 4]), tf.placeholder(tf.float32, [None, 4])
            res = cell(px, [phh])[0]
            sess.run(tf.variables_initializer(cell.variables))
            for v in cell.variables:
                val = (rng.standard_normal(v.shape.as_list()) * 0.5).astype(np.float32)
                sess.run(v.assign(val))
                out["cell|%s|%s" % (kind, v.name.split("/")[-1][:-2])] = val
            out["cell|%s|x" % kind], out["cell|%s|h" % kind] = xin, hin
            out["cell|%s|out" % kind] = sess.run(res, {px: xin, phh: hin})
        # optimizers: 3 steps each on one variable with a per-variable clip_by_norm(1.0)
        grads = [(rng.standard_normal((4, 3)) * s).astype(np.float32) for s in (5.0, 0.01, 0.3)]
        out["opt|grads"] = np.stack(grads)
        out["opt|clip"] = np.stack([sess.run(tf.clip_by_norm(gr, 1.0)) for gr in grads])
        for name, make in (("adam", lambda: tf.train.AdamOptimizer(1e-3)),
                           ("rmsprop", lambda: tf.train.RMSPropOptimizer(1e-3, decay=0.98, momentum=0.85)),
                           ("sgd", lambda: tf.train.GradientDescentOptimizer(1e-3))):
            with tf.variable_scope("opt_" + name):
                var = tf.get_variable("w", initializer=np.ones((4, 3), np.float32))
                pg = tf.placeholder(tf.float32, [4, 3])
                opt = make()
                step = opt.apply_gradients([(tf.clip_by_norm(pg, 1.0), var)])
            sess.run(tf.variables_initializer([var] + opt.variables()))
            traj = []
            for gr in grads:
                sess.run(step, {pg: gr})
                traj.append(sess.run(var))
            out["opt|%s" % name] = np.stack(traj)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of microsoft/tf-gnn-samples (read only)")
    args = ap.parse_args()
    try:
        import tensorflow as tf
        import dpu_utils  # noqa: F401
    except ImportError as e:
        raise SystemExit("needs the reference's environment (TensorFlow 1.13-1.15 + dpu_utils): %s" % e)
    if not tf.__version__.startswith("1."):
        raise SystemExit("TensorFlow %s found; the reference is TF1 graph-mode code (README.md:16: 1.13.1)" % tf.__version__)
    sys.path.insert(0, args.reference)
    import gnns  # the reference's package, unmodified
    layers, sem = {}, {}
    run_layers(tf, gnns, layers)
    run_semantics(tf, sem)
    for d in (layers, sem):
        d["meta|tensorflow"] = np.array(tf.__version__)
        d["meta|numpy"] = np.array(np.__version__)
    np.savez_compressed(GOLDEN / "tf_layers_small.npz", **layers)
    np.savez_compressed(GOLDEN / "tf_semantics.npz", **sem)
    print("wrote", GOLDEN / "tf_layers_small.npz", "and", GOLDEN / "tf_semantics.npz")


if __name__ == "__main__":
    main()
