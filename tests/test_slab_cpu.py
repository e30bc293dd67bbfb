"""tasks/slab.py on the CPU: the sliced-ELL message lists hold exactly the bucketed CSR they were built from."""
import numpy as np
import pytest
import torch

from tf_gnn_samples_amd.tasks import slab as S


def _random_fold(rng, G, L, max_nodes=40):
    """rowptr / ids / w of a fold bucketed as one disjoint union; edges stay inside their graph."""
    nodes = rng.integers(1, max_nodes, size=G)
    node_off = np.concatenate([[0], np.cumsum(nodes)]).astype(np.int64)
    N = int(node_off[-1])
    lens = np.zeros(N * L, dtype=np.int64)
    buckets = []
    for g in range(G):
        for v in range(node_off[g], node_off[g + 1]):
            for l in range(L):
                n = int(rng.choice([0, 0, 1, 2, 5, 70])) if rng.random() < 0.9 else int(rng.integers(0, 200))
                lens[v * L + l] = n
                buckets.append(rng.integers(node_off[g], node_off[g + 1], size=n))
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate(buckets).astype(np.int32) if buckets else np.zeros(0, np.int32)
    w = rng.random(len(ids)).astype(np.float32)
    return node_off, rowptr, ids, w


@pytest.mark.parametrize("G,L,weighted", [(1, 1, True), (5, 3, True), (9, 2, False), (3, 23, True)])
def test_ell_lists_reproduce_the_bucketed_csr(G, L, weighted):
    rng = np.random.default_rng(G * 10 + L)
    node_off, rowptr, ids, w = _random_fold(rng, G, L)
    d = S._build_direction(rowptr, ids, w if weighted else None, node_off, L, "cpu")
    sb, sl, so = d.slice_base.numpy(), d.slice_len.numpy(), d.slice_off.numpy()
    bucket, blen = d.slice_bucket.numpy(), d.slice_blen.numpy()
    ell = d.ell_id.numpy().view(np.uint16)
    assert (d.ell_w is None) == (not weighted)
    seen = 0
    for g in range(G):
        n_b = (node_off[g + 1] - node_off[g]) * L
        assert sb[g + 1] - sb[g] == (n_b + 63) // 64
        order = []
        for q in range(sb[g], sb[g + 1]):
            assert sl[q] == blen[q * 64:(q + 1) * 64].max()                     # steps: the longest bucket
            for lane in range(64):
                b = bucket[q * 64 + lane]
                if b < 0:
                    assert blen[q * 64 + lane] == 0
                    continue
                order.append(b)
                s = node_off[g] * L + b                                   # bucket of the union
                n = rowptr[s + 1] - rowptr[s]
                assert blen[q * 64 + lane] == n
                at = so[q] + (np.arange(n) // 8) * 512 + lane * 8 + np.arange(n) % 8
                assert np.array_equal(ell[at].astype(np.int64) + node_off[g], ids[rowptr[s]:rowptr[s + 1]])   # bucket order kept
                if weighted:
                    assert np.array_equal(d.ell_w.numpy()[at], w[rowptr[s]:rowptr[s + 1]])
                k = np.arange(n, (sl[q] + 7) // 8 * 8)                    # entries past the bucket's end (whole chunks are stored)
                pad = so[q] + (k // 8) * 512 + lane * 8 + k % 8
                assert (ell[pad] == node_off[g + 1] - node_off[g]).all()
                if weighted:
                    assert (d.ell_w.numpy()[pad] == 0).all()
                seen += n
        assert sorted(order) == list(range(n_b))                          # every bucket of the graph exactly once
        lens_sorted = [rowptr[node_off[g] * L + b + 1] - rowptr[node_off[g] * L + b] for b in order]
        assert lens_sorted == sorted(lens_sorted, reverse=True)           # decreasing length: a wave's lanes run alike
    assert seen == len(ids) == d.messages and d.entries == int((((sl.astype(np.int64) + 7) // 8 * 8) * 64).sum())
    assert len(ell) == d.entries + 2 * 8 * 64                             # two chunks the kernel may prefetch past the last slice


def test_batch_table_ranks_graphs_by_work():
    t = S.batch_table(np.array([7, 2, 9]), np.array([0, 10, 30, 35]), np.array([10, 20, 5]), np.array([100, 500, 500]))
    assert t.tolist() == [[2, 10, 20], [9, 30, 5], [7, 0, 10]]


def test_messages_that_cross_graphs_are_refused():
    node_off = np.array([0, 3, 6], dtype=np.int64)
    rowptr = np.array([0, 1, 1, 1, 1, 1, 1], dtype=np.int64)
    with pytest.raises(ValueError, match="crosses graphs"):
        S._build_direction(rowptr, np.array([4], dtype=np.int32), None, node_off, 1, "cpu")
