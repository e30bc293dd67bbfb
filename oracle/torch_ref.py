"""torch-CPU mirror of oracle/gnns.py (same op order: gather -> per-edge transform -> scale ->
concat -> index_add segment reduce -> node update) so that torch.autograd provides reference
GRADIENTS, and so that the reference's CPU op chain — including its per-edge matmul — can be
timed multi-threaded as the CPU baseline.

TEST INFRASTRUCTURE; PARITY UNPINNED (oracle/__init__.py).  Never imported by the package.
"""
import math

import torch

from .tf_ops import layer_norm_scope

SMALL_NUMBER = 1e-7


def _lowest(dtype):
    return torch.finfo(dtype).min


def unsorted_segment(kind, data, segment_ids, num_segments):
    """tf.unsorted_segment_{sum,mean,sqrt_n,max} with TF's gradient semantics (max: gradient split
    equally among ties, as torch's scatter_reduce(amax) does)."""
    ids = segment_ids.long()
    if kind == "max":
        out = torch.full((num_segments,) + tuple(data.shape[1:]), _lowest(data.dtype), dtype=data.dtype)
        idx = ids.view(-1, *([1] * (data.dim() - 1))).expand_as(data)
        return out.scatter_reduce(0, idx, data, reduce="amax", include_self=True)
    out = torch.zeros((num_segments,) + tuple(data.shape[1:]), dtype=data.dtype).index_add(0, ids, data)
    if kind == "sum":
        return out
    n = torch.bincount(ids, minlength=num_segments).clamp(min=1).to(data.dtype)
    n = n.view(-1, *([1] * (data.dim() - 1)))
    return out / n if kind == "mean" else out / torch.sqrt(n)


_KINDS = {"sum": "sum", "unsorted_segment_sum": "sum", "max": "max", "unsorted_segment_max": "max",
          "mean": "mean", "unsorted_segment_mean": "mean", "sqrt_n": "sqrt_n", "unsorted_segment_sqrt_n": "sqrt_n"}


def aggregation(name):
    if name not in _KINDS:
        raise ValueError("Unknown aggregation function '%s'!" % name)
    kind = _KINDS[name]
    return lambda data, ids, n: unsorted_segment(kind, data, ids, n)


def activation(name):
    if name is None:
        return lambda x: x
    n = name.lower()
    table = {
        'linear': lambda x: x, 'tanh': torch.tanh, 'relu': torch.relu,
        'leaky_relu': lambda x: torch.nn.functional.leaky_relu(x, 0.2),
        'elu': torch.nn.functional.elu, 'selu': torch.selu,
        'gelu': lambda x: x * (0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))),
    }
    if n not in table:
        raise ValueError("Unknown activation function '%s'!" % name)
    return table[n]


def layer_norm(x, gamma, beta, eps=1e-12):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def hard_sigmoid(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def mlp(x, w, name, num_hidden, act):
    names = ["dense" if i == 0 else "dense_%i" % i for i in range(num_hidden + 1)]
    h = x
    for i, n in enumerate(names):
        h = h @ w["%s/%s/kernel" % (name, n)]
        if i < num_hidden:
            h = act(h)
    return h


def _targets(adj):
    return torch.cat([a[:, 1] for a in adj]).long()


def _inv_deg(deg, l, targets, dtype):
    return (1.0 / (deg[l].to(dtype)[targets.long()] + SMALL_NUMBER)).unsqueeze(-1)


def sparse_rgcn_layer(h, adj, deg, state_dim, num_timesteps=1, activation_function="tanh",
                      message_aggregation_function="sum", normalize_by_num_incoming=True,
                      use_both_source_and_target=False, *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    tgts = _targets(adj)
    cur = h
    for _ in range(num_timesteps):
        msgs = []
        for l, a in enumerate(adj):
            src = cur.index_select(0, a[:, 0].long())
            if use_both_source_and_target:
                src = torch.cat([src, cur.index_select(0, a[:, 1].long())], dim=-1)
            m = src @ weights["Edge_%i_Weight/kernel" % l]
            if normalize_by_num_incoming:
                m = _inv_deg(deg, l, a[:, 1], cur.dtype) * m
            msgs.append(m)
        cur = act(agg(torch.cat(msgs, 0), tgts, V))
    return cur


def sparse_ggnn_layer(h, adj, state_dim, num_timesteps=1, gated_unit_type="gru", activation_function="tanh",
                      message_aggregation_function="sum", *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    tgts = _targets(adj)
    kind = gated_unit_type.lower()
    scope = {"gru": "gru_cell", "rnn": "simple_rnn_cell"}[kind]
    K, U, b = weights[scope + "/kernel"], weights[scope + "/recurrent_kernel"], weights[scope + "/bias"]
    cur = h
    for _ in range(num_timesteps):
        msgs = [cur.index_select(0, a[:, 0].long()) @ weights["Edge_%i_Weight/kernel" % l] for l, a in enumerate(adj)]
        m = agg(torch.cat(msgs, 0), tgts, V)
        if kind == "rnn":
            cur = act(m @ K + b + cur @ U)
        else:
            u = cur.shape[1]
            xk = m @ K + b
            z = hard_sigmoid(xk[:, :u] + cur @ U[:, :u])
            r = hard_sigmoid(xk[:, u:2 * u] + cur @ U[:, u:2 * u])
            hh = act(xk[:, 2 * u:] + (r * cur) @ U[:, 2 * u:])
            cur = z * cur + (1.0 - z) * hh
    return cur


def sparse_rgat_layer(h, adj, state_dim, num_heads=4, num_timesteps=1, activation_function="tanh", *, weights):
    act = activation(activation_function)
    V = h.shape[0]
    D = state_dim if state_dim is not None else h.shape[1]
    dh = D // num_heads
    tgts = _targets(adj)
    cur = h
    for _ in range(num_timesteps):
        msgs, coefs = [], []
        for l, a in enumerate(adj):
            t = cur @ weights["Edge_%i_Weight/kernel" % l]
            s = t.index_select(0, a[:, 0].long()).reshape(-1, num_heads, dh)
            g = t.index_select(0, a[:, 1].long()).reshape(-1, num_heads, dh)
            att = weights["Edge_%i_Attention_Parameters" % l].reshape(num_heads, 2 * dh)
            coefs.append(torch.nn.functional.leaky_relu(torch.einsum('vki,ki->vk', torch.cat([s, g], -1), att), 0.2))
            msgs.append(s)
        msgs, coefs = torch.cat(msgs, 0), torch.cat(coefs, 0)
        heads = []
        for k in range(num_heads):
            x = coefs[:, k]
            mx = unsorted_segment("max", x, tgts, V)
            rec = x - mx[tgts]
            ssum = unsorted_segment("sum", torch.exp(rec), tgts, V)
            a_val = torch.exp(rec - torch.log(ssum)[tgts])
            heads.append(unsorted_segment("sum", a_val.unsqueeze(-1) * msgs[:, k, :], tgts, V))
        cur = act(torch.cat(heads, -1))
    return cur


def sparse_rgin_layer(h, adj, state_dim, num_timesteps=1, activation_function="ReLU",
                      message_aggregation_function="sum", use_target_state_as_input=False,
                      num_edge_MLP_hidden_layers=1, num_aggr_MLP_hidden_layers=None, *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    tgts = _targets(adj)
    cur = h
    for t in range(num_timesteps):
        msgs = []
        for l, a in enumerate(adj):
            x = cur.index_select(0, a[:, 0].long())
            if use_target_state_as_input:
                x = torch.cat([x, cur.index_select(0, a[:, 1].long())], 1)
            if num_edge_MLP_hidden_layers is not None:
                x = mlp(x, weights, "Edge_%i_MLP" % l, num_edge_MLP_hidden_layers, act)
            msgs.append(x)
        m = torch.cat(msgs, 0)
        if num_edge_MLP_hidden_layers is not None:
            m = act(m)
        new = agg(m, tgts, V)
        if num_aggr_MLP_hidden_layers is not None:
            new = mlp(new, weights, "Aggregation_MLP", num_aggr_MLP_hidden_layers, act)
        cur = layer_norm(act(new), weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur


def sparse_gnn_film_layer(h, adj, deg, state_dim, num_timesteps=1, activation_function="ReLU",
                          message_aggregation_function="sum", normalize_by_num_incoming=False, *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    D = state_dim if state_dim is not None else h.shape[1]
    tgts = _targets(adj)
    cur = h
    for t in range(num_timesteps):
        msgs = []
        for l, a in enumerate(adj):
            m = cur.index_select(0, a[:, 0].long()) @ weights["Edge_%i_Weight/kernel" % l]
            if normalize_by_num_incoming:
                m = _inv_deg(deg, l, a[:, 1], cur.dtype) * m
            film = (cur @ weights["Edge_%i_FiLM_Computations/kernel" % l]).index_select(0, a[:, 1].long())
            msgs.append(film[:, :D] * m + film[:, D:])
        new = agg(act(torch.cat(msgs, 0)), tgts, V)
        cur = layer_norm(new, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur


def sparse_gnn_edge_mlp_layer(h, adj, deg, state_dim, num_timesteps=1, activation_function="ReLU",
                              message_aggregation_function="sum", normalize_by_num_incoming=False,
                              use_target_state_as_input=True, num_edge_hidden_layers=1, *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    tgts = _targets(adj)
    cur = h
    for t in range(num_timesteps):
        msgs = []
        for l, a in enumerate(adj):
            x = cur.index_select(0, a[:, 0].long())
            if use_target_state_as_input:
                x = torch.cat([x, cur.index_select(0, a[:, 1].long())], 1)
            m = mlp(x, weights, "Edge_%i_MLP" % l, num_edge_hidden_layers, torch.nn.functional.elu)
            if normalize_by_num_incoming:
                m = _inv_deg(deg, l, a[:, 1], cur.dtype) * m
            msgs.append(m)
        new = agg(act(torch.cat(msgs, 0)), tgts, V)
        cur = layer_norm(new, weights[layer_norm_scope(t) + "/gamma"], weights[layer_norm_scope(t) + "/beta"])
    return cur


def sparse_rgdcn_layer(h, adj, deg, num_channels=8, channel_dim=16, num_timesteps=1,
                       use_full_state_for_channel_weights=False, tie_channel_weights=False, activation_function="tanh",
                       message_aggregation_function="sum", normalize_by_num_incoming=True, *, weights):
    act, agg = activation(activation_function), aggregation(message_aggregation_function)
    V = h.shape[0]
    tgts = _targets(adj)
    cur = h
    for _ in range(num_timesteps):
        chunked = cur.reshape(-1, num_channels, channel_dim)
        outs = []
        for c in range(num_channels):
            ch = chunked[:, c, :]
            per_type = []
            for l, a in enumerate(adj):
                src = ch.index_select(0, a[:, 0].long())
                inp = cur if use_full_state_for_channel_weights else ch
                kern = weights["Edge_%i_Channel_%i_Weight_Computation/kernel" % (l, 0 if tie_channel_weights else c)]
                ew = act(inp @ kern).reshape(-1, channel_dim, channel_dim).index_select(0, a[:, 1].long())
                m = torch.einsum('vi,vij->vj', src, ew)
                if normalize_by_num_incoming:
                    m = _inv_deg(deg, l, a[:, 1], cur.dtype) * m
                per_type.append(m)
            outs.append(act(agg(torch.cat(per_type, 0), tgts, V)))
        cur = torch.cat(outs, 1)
    return cur


def sparse_rgcn_layer_lean(h, adj, deg, state_dim, num_timesteps=1, activation_function="tanh",
                           message_aggregation_function="sum", normalize_by_num_incoming=True, *, weights,
                           relu_mask=None, pre_activations=None):
    """BASELINE-size variant of sparse_rgcn_layer above for REFERENCE GRADIENTS in float64 (sum aggregation, source-only
    inputs): the same function, evaluated as  act(sum_l A_l (h W_l))  with A_l the sparse [V, V] matrix holding
    1/(c_{l,v} + 1e-7) at (v, u) for every edge (u, v) of type l (duplicates summed).  Op for op the chain above keeps two
    [M, D] float64 tensors per layer alive for autograd (7.6 GB per layer at the C2 batch); this one keeps [V, D] tensors.
    In float64 the association of the sums is immaterial at the 1e-12 level (tests/test_oracle_crosscheck_cpu.py checks it
    against the op-for-op path); the 1/(c + 1e-7) scale is evaluated in float32 like the reference's and then widened, so
    that both sides multiply by the same number.

    relu_mask ([V, D] bool, ReLU layers with one timestep only): the layer is evaluated as  mask * pre-activation  — the ReLU with
    the BRANCH of every unit prescribed from outside (the float32 run under test) instead of decided by the sign of the float64
    pre-activation.  A unit whose pre-activation lies within the float32 forward error of zero takes the other branch in the two
    arithmetics, and its whole gradient path (a rank-one term of every weight gradient upstream) then exists in one run and not in
    the other: a difference that measures where a knife edge fell, not the accuracy of any product.  With the branches prescribed
    the two runs differentiate the SAME piecewise-linear function.  pre_activations (a list): the float64 pre-activation is
    appended, so that the caller can count the units whose branch differs."""
    if _KINDS.get(message_aggregation_function) != "sum":
        raise ValueError("sparse_rgcn_layer_lean: sum aggregation only")
    act = activation(activation_function)
    if relu_mask is not None:
        if activation_function.lower() != "relu" or num_timesteps != 1:
            raise ValueError("sparse_rgcn_layer_lean: relu_mask needs a ReLU layer with one timestep")
        gate = relu_mask.to(h.dtype)
        act = lambda z: z * gate
    V = h.shape[0]
    mats = []
    for l, a in enumerate(adj):
        a = a.long()
        if normalize_by_num_incoming:
            scale = (1.0 / (deg[l].to(torch.float32)[a[:, 1]] + torch.tensor(SMALL_NUMBER, dtype=torch.float32))).to(h.dtype)
        else:
            scale = torch.ones(a.shape[0], dtype=h.dtype)
        mats.append(torch.sparse_coo_tensor(torch.stack([a[:, 1], a[:, 0]]), scale, (V, V)).coalesce().to_sparse_csr())
    cur = h
    for _ in range(num_timesteps):
        total = None
        for l, A in enumerate(mats):
            part = torch.sparse.mm(A, cur @ weights["Edge_%i_Weight/kernel" % l])
            total = part if total is None else total + part
        if pre_activations is not None:
            pre_activations.append(total.detach())
        cur = act(total)
    return cur
