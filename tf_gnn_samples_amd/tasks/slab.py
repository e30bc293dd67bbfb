"""Sliced-ELLPACK message lists of a data fold's graphs for the LDS-tiled gather (csrc/slab_gather.hip).

A batch is a disjoint union of graphs and no edge crosses graphs (tasks/ppi_task.py:220-233): the buckets of graph g gather rows
of graph g's slab of the node-state table only.  relgnn_slab_gather_f32 stages 8-column slices of that slab in LDS and lets every
lane fold one (node, type) bucket sequentially — the reference's summation order — reading its rows from LDS.  For that the message
list of a graph is stored per direction (by target: gnns/rgcn.py:87-112 forward; by source: its gradient) as

    buckets of the graph in order of decreasing length, 64 per slice, stored in whole chunks of CHUNK = 8 steps (slice_len: its
    longest bucket); entry k of lane i of slice q at  slice_off[q] + 512 (k // 8) + 8 i + k % 8 :  graph-LOCAL row id (uint16) and
    weight (float32) — a lane's 8 entries of a chunk are contiguous (one 16-byte and two 16-byte loads); entries past a bucket's
    end name row `nodes of the graph` (a row of zeros the kernel keeps behind the slab) with weight 0, and the lists end in two
    chunks of such padding (the kernel prefetches two chunks ahead without a bounds test)

which is a property of the graph, not of the batch (ids are graph-local): built ONCE per fold from the fold's bucketing
(tasks/resident.py) and shared by every batch.  A batch contributes a K x 3 table (fold graph, first node in the batch, nodes),
heaviest graph first.
"""
from typing import Optional

import numpy as np
import torch

LANES = 64
CHUNK = 8          # relgnn_slab_gather_chunk()


class SlabDirection:
    """Device arrays of one direction (by target / by source) for all graphs of the fold."""
    __slots__ = ("slice_base", "slice_len", "slice_off", "slice_bucket", "slice_blen", "ell_id", "ell_w", "max_bucket",
                 "entries", "messages", "lane_steps")


def _build_direction(rowptr: np.ndarray, ids: np.ndarray, w: Optional[np.ndarray], node_off: np.ndarray, L: int, device) -> SlabDirection:
    """rowptr [N*L+1], ids [M] (fold-global row ids per bucketed message), w [M] or None, node_off [G+1]."""
    G = len(node_off) - 1
    N = int(node_off[-1])
    B = N * L
    lens = np.diff(rowptr).astype(np.int64)
    bucket_graph = np.repeat(np.arange(G), np.diff(node_off) * L)
    # buckets of a graph by decreasing length, ties in bucket order: one global stable sort on (graph, -length)
    order = np.lexsort((-lens, bucket_graph))
    per_graph = (np.diff(node_off) * L).astype(np.int64)
    slices_per_graph = (per_graph + LANES - 1) // LANES
    slice_base = np.concatenate([[0], np.cumsum(slices_per_graph)]).astype(np.int64)
    S = int(slice_base[-1])
    first_bucket = np.concatenate([[0], np.cumsum(per_graph)])[:-1]
    rank = np.arange(B) - np.repeat(first_bucket, per_graph)                # position of a bucket in its graph's sorted list
    g_sorted = bucket_graph[order]
    slot = (slice_base[g_sorted] * LANES + rank).astype(np.int64)          # lane slot (slice * 64 + lane) of the sorted bucket
    slice_bucket = np.full(S * LANES, -1, dtype=np.int32)
    slice_blen = np.zeros(S * LANES, dtype=np.int32)
    local_bucket = order - np.repeat(node_off[:-1] * L, per_graph)[order]   # bucket id relative to the graph's first bucket
    slice_bucket[slot] = local_bucket.astype(np.int32)
    slice_blen[slot] = lens[order].astype(np.int32)
    slice_len = slice_blen.reshape(S, LANES).max(axis=1).astype(np.int32) if S else np.zeros(0, np.int32)
    stored = (slice_len.astype(np.int64) + CHUNK - 1) // CHUNK * CHUNK          # a slice is stored in whole chunks
    slice_off = np.concatenate([[0], np.cumsum(stored * LANES)])
    entries = int(slice_off[-1])
    slice_graph = np.repeat(np.arange(G), slices_per_graph)
    # every entry starts as padding: the zero row behind its graph's slab
    ell_id = np.concatenate([np.repeat(np.diff(node_off)[slice_graph], stored * LANES),
                             np.zeros(2 * CHUNK * LANES, dtype=np.int64)]).astype(np.uint16)
    ell_w = np.zeros(entries + 2 * CHUNK * LANES, dtype=np.float32) if w is not None else None
    M = len(ids)
    if M:
        slot_of_bucket = np.empty(B, dtype=np.int64)
        slot_of_bucket[order] = slot
        msg_bucket = np.repeat(np.arange(B), lens)
        k = np.arange(M) - rowptr[:-1][msg_bucket]
        q, lane = slot_of_bucket[msg_bucket] // LANES, slot_of_bucket[msg_bucket] % LANES
        dst = slice_off[q] + (k // CHUNK) * (CHUNK * LANES) + lane * CHUNK + k % CHUNK
        local = ids.astype(np.int64) - node_off[bucket_graph[msg_bucket]]
        if local.min() < 0 or local.max() >= 65536 or (local >= np.diff(node_off)[bucket_graph[msg_bucket]]).any():
            raise ValueError("slab plan: a message crosses graphs or a graph has more than 65535 nodes")
        ell_id[dst] = local.astype(np.uint16)
        if w is not None:
            ell_w[dst] = w
    d = SlabDirection()
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)
    d.slice_base = dev(slice_base, torch.int32)
    d.slice_len, d.slice_off = dev(slice_len, torch.int32), dev(slice_off[:-1], torch.int64)
    d.slice_bucket, d.slice_blen = dev(slice_bucket, torch.int32), dev(slice_blen, torch.int32)
    d.ell_id = torch.as_tensor(ell_id.view(np.int16), device=device)       # (torch has no uint16 arithmetic: raw bits)
    d.ell_w = None if ell_w is None else torch.as_tensor(ell_w, device=device)
    d.max_bucket = int(lens.max()) if B else 0
    d.entries, d.messages = entries, M
    d.lane_steps = int(slice_len.astype(np.int64).sum()) * LANES       # what the kernel folds: messages + in-slice imbalance
    return d


class SlabFold:
    """Both directions for a fold bucketed as ONE disjoint union (tasks/resident.py: RelGraph over the whole fold)."""

    def __init__(self, fold_graph, node_off: np.ndarray, w_t: Optional[torch.Tensor], w_s: Optional[torch.Tensor], device):
        L = fold_graph.L
        host = lambda t: t.detach().cpu().numpy()
        self.L, self.node_off = L, np.asarray(node_off, dtype=np.int64)
        self.max_nodes = int(np.diff(self.node_off).max()) if len(self.node_off) > 1 else 0
        self.by_target = _build_direction(host(fold_graph.rowptr_t).astype(np.int64), host(fold_graph.src_t),
                                          None if w_t is None else host(w_t), self.node_off, L, device)
        self.by_source = _build_direction(host(fold_graph.rowptr_s).astype(np.int64), host(fold_graph.tgt_s),
                                          None if w_s is None else host(w_s), self.node_off, L, device)
        self.graph_messages = None


class SlabBatch:
    """What a batch's RelGraph carries: the fold's lists + this batch's graph table (device int64 [K, 3], heaviest first)."""
    __slots__ = ("fold", "desc", "num_graphs", "max_nodes", "w_t", "w_s")

    def __init__(self, fold: SlabFold, desc: torch.Tensor, num_graphs: int, max_nodes: int, w_t, w_s):
        self.fold, self.desc, self.num_graphs, self.max_nodes = fold, desc, int(num_graphs), int(max_nodes)
        self.w_t, self.w_s = w_t, w_s       # the batch's per-message scale tensors the lists' weights stand for (identity check)


def batch_table(graph_ids: np.ndarray, node_off_b: np.ndarray, nodes: np.ndarray, work: np.ndarray) -> np.ndarray:
    """[K, 3] int64 (fold graph, first node in the batch, nodes), heaviest (most messages) first; ties in batch order."""
    order = np.argsort(-work, kind="stable")
    return np.stack([graph_ids[order], node_off_b[:-1][order], nodes[order]], axis=1).astype(np.int64)
