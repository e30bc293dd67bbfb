"""Shared input builders for the tests (seeded, small enough for the NumPy oracle)."""
import numpy as np

from oracle import bookkeeping


def random_relational_graph(rng, num_nodes, num_edge_types, edges_per_type, empty_types=(), heavy_tail=True):
    """L adjacency lists [E_l, 2] int32 with duplicates, self edges and (optionally) empty types."""
    adj = []
    for l in range(num_edge_types):
        e = 0 if l in empty_types else int(edges_per_type if np.isscalar(edges_per_type) else edges_per_type[l])
        src = rng.integers(0, num_nodes, size=e)
        if heavy_tail and e:
            w = rng.lognormal(0.0, 1.0, size=num_nodes)
            tgt = rng.choice(num_nodes, size=e, p=w / w.sum())
        else:
            tgt = rng.integers(0, num_nodes, size=e)
        adj.append(np.stack([src, tgt], axis=1).astype(np.int32).reshape(-1, 2))
    return adj


def glorot(rng, shape):
    limit = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def rgcn_weights(rng, num_edge_types, in_dim, out_dim):
    return {"Edge_%i_Weight/kernel" % l: glorot(rng, (in_dim, out_dim)) for l in range(num_edge_types)}


def degree_table(adj, num_nodes):
    return bookkeeping.in_degree_table(adj, num_nodes)


# ---- parity bookkeeping ---------------------------------------------------------------------------------------------
PARITY_TOL = 1e-5        # north_star: "within 1e-5 abs on fp32 node states"
_PARITY_LOG = []


def parity_errors(out, ref):
    """(max abs error, max abs error / max |ref|) of a HIP result against the oracle."""
    try:
        import torch
        if torch.is_tensor(out):
            out = out.detach().cpu().numpy()
    except ImportError:
        pass
    ref = np.asarray(ref)
    abs_err = float(np.abs(np.asarray(out, dtype=np.float64) - ref.astype(np.float64)).max()) if ref.size else 0.0
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    return abs_err, (abs_err / scale if scale > 0 else 0.0)


def assert_parity(out, ref, *, strict_abs, what="", tol=PARITY_TOL):
    """strict_abs=True  : the layer's states are bounded by construction (1/in-degree-normalised sums, softmax-weighted
                          sums, GRU/tanh outputs): fail on the north-star 1e-5 ABSOLUTE error, whatever max|ref| is.
       strict_abs=False : un-normalised sums (GGNN/RGIN/FiLM/Edge-MLP defaults) grow past O(1) and fp32 carries 1e-7
                          RELATIVE precision: fail on abs <= tol where max|ref| <= 1, else on
                          the error relative to max|ref|.  Both numbers are recorded either way."""
    abs_err, rel_err = parity_errors(out, ref)
    _PARITY_LOG.append({"what": what, "abs": abs_err, "rel": rel_err, "strict_abs": bool(strict_abs),
                        "max_ref": float(np.abs(ref).max()) if np.size(ref) else 0.0})
    if strict_abs:
        assert abs_err <= tol, "%s: max abs error %.3e > %.1e (rel %.3e)" % (what, abs_err, tol, rel_err)
    else:
        assert min(abs_err, rel_err) <= tol, "%s: abs %.3e, rel %.3e > %.1e" % (what, abs_err, rel_err, tol)
    return abs_err, rel_err


def parity_log():
    return list(_PARITY_LOG)


def layer_norm_weights(D, num_timesteps=2, rng=None):
    """gamma/beta for the per-timestep LayerNorm scopes of the FiLM / RGIN / Edge-MLP layers (LayerNorm, LayerNorm_1, ...:
    the reference calls tf.contrib.layers.layer_norm once per timestep).  rng=None: identity parameters."""
    from oracle.tf_ops import layer_norm_scope
    w = {}
    for t in range(num_timesteps):
        s = layer_norm_scope(t)
        w[s + "/gamma"] = np.ones(D, np.float32) if rng is None else (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
        w[s + "/beta"] = np.zeros(D, np.float32) if rng is None else (0.1 * rng.standard_normal(D)).astype(np.float32)
    return w


def set_switch(monkeypatch, env_name: str, value):
    """Set the route switch that `env_name` (RELGNN_*) initialises — tf_gnn_samples_amd.config.settings, read at call time —
    for the rest of the test (value None: back to its default).  The environment itself is read once, at import."""
    from tf_gnn_samples_amd import config
    name = config.attribute_of(env_name)
    monkeypatch.setattr(config.settings, name, config.default_of(name) if value is None else value)
