"""A data fold RESIDENT in HBM: graphs, features, labels, degree tables and the bucketing of the whole fold.

The reference rebuilds every minibatch from Python lists on the host and feeds it through sess.run
(tasks/ppi_task.py:197-256, models/sparse_graph_model.py:272-293); tasks/batcher.py already replaces that by a C++
packer + one H2D copy per batch.  An MI355X has 288 GB of HBM: the data folds of this code base (PPI 57 k nodes /
1.6 M edges, QM9 2.4 M nodes, even VarMisuse-sized folds) fit many times over, so the fold can simply live on the
device.  A batch is then a list of graph ids, and

  * its tensors are gathered on the device from the fold's flat arrays by relgnn_batch_gather (same packing rules as
    relgnn_batch_pack: adjacency + node offset, payload / degree rows concatenated on the node axis, graph index per
    node) — one C call, four streaming kernels;
  * its (target,type)/(source,type) bucketing is NOT recomputed: the fold was bucketed once as one big disjoint union,
    and relgnn_plan_assemble re-bases the per-graph slices of those arrays (include/relgnn.h section 10) —
    three streaming kernels instead of two radix sorts per batch, bit-identical arrays.

Nothing crosses PCIe per batch except the K-entry offset tables.  This is the input pipeline of bench.py's headline
(distinct batches of a shuffled epoch); its `same_batch` figure instead re-buckets one resident batch per step.
"""
from typing import Iterator, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .. import config as _cfg
from ..graph import RelGraph
from .batcher import GraphStore
from .sparse_graph_task import DeviceBatch


def _dev_i64(a, device):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=device)


class ResidentDataset:
    def __init__(self, store: GraphStore, device, features: str = "initial_node_features",
                 constants: Optional[dict] = None):
        self.store, self.device = store, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RelGnnLibraryError("a resident data fold lives in HBM; got device %s" % self.device)
        self.features, self.constants = features, dict(constants or {})
        G, L = store.num_graphs, store.num_edge_types
        dev = self.device
        self.node_off = store.node_off                                           # host int64 [G+1]
        self.edge_off = np.stack(store.edge_off) if L else np.zeros((0, G + 1), np.int64)   # host [L, G+1]
        per_graph_msgs = (self.edge_off[:, 1:] - self.edge_off[:, :-1]).sum(0) if L else np.zeros(G, np.int64)
        self.msg_off = np.concatenate([[0], np.cumsum(per_graph_msgs)]).astype(np.int64)
        self.type_off = np.concatenate([[0], np.cumsum(self.edge_off[:, -1])]).astype(np.int64)
        N, M = int(self.node_off[-1]), int(self.msg_off[-1])
        if N * max(L, 1) >= 2 ** 31 - 1 or M >= 2 ** 31 - 1:
            raise ValueError("data fold too large for one int32-indexed union; use the host batcher")
        # flat arrays -> HBM
        self.payload_d = [torch.as_tensor(a, device=dev) for a in store.payload]
        self.graph_payload_d = {k: torch.as_tensor(v, device=dev) for k, v in store.graph_payload.items()}
        self.deg_d = torch.stack([torch.as_tensor(a, device=dev) for a in store.deg]) if L else torch.zeros((0, N), device=dev)
        self.adj_d = [torch.as_tensor(a, device=dev) for a in store.adj]         # [E_l, 2] int32, graph-local ids
        self.node_off_d, self.msg_off_d = _dev_i64(self.node_off, dev), _dev_i64(self.msg_off, dev)
        self.edge_off_d, self.type_off_d = _dev_i64(self.edge_off, dev), _dev_i64(self.type_off, dev)
        # bucket the WHOLE fold once as one disjoint union (global node ids = local + node offset of the graph)
        union = []
        for l in range(L):
            counts = torch.as_tensor(self.edge_off[l, 1:] - self.edge_off[l, :-1], device=dev)
            shift = torch.repeat_interleave(self.node_off_d[:-1], counts).to(torch.int32)
            union.append((self.adj_d[l] + shift.unsqueeze(1)).contiguous())
        g = RelGraph(union, N, validate=True)                                    # node-id range check, once
        self._has_hubs = g.has_long_buckets          # a bucket of the fold too long for one wave: split per batch too
        # non-empty (node, type) buckets per graph and type, by target and by source: what graph.PairTables would otherwise read
        # back from the device for every batch (a host sync per batch makes a many-type step host-bound: C5 45 ms vs 39 on the GPU)
        self.pair_counts = None
        if L >= 8:
            cnt = np.zeros((2, G, L), dtype=np.int64)
            for l in range(L):
                a = np.asarray(store.adj[l]).reshape(-1, 2)
                gidx = np.repeat(np.arange(G), np.diff(self.edge_off[l]))
                for side in (0, 1):                   # side 0: by target (column 1 of the adjacency list), 1: by source
                    key = gidx.astype(np.int64) * (N + 1) + a[:, 1 - side].astype(np.int64) + self.node_off[gidx]
                    cnt[side, :, l] = np.bincount(np.unique(key) // (N + 1), minlength=G)
            self.pair_counts = cnt
        self.plan_d = dict(rowptr_t=g.rowptr_t, perm_t=g.perm_t, col_t=g.col_t, rowptr_s=g.rowptr_s, perm_s=g.perm_s,
                           frow_s=g.frow_s, pos_t_of_s=g.pos_t_of_s)
        # per-message 1/(in-degree + 1e-7) of the whole fold, by-target and by-source order: graph properties, copied
        # per batch by relgnn_plan_assemble instead of being recomputed (relgnn_degree_scale + two gathers per batch)
        self.w_t_d = g.degree_scale(self.deg_d) if M > 0 else torch.zeros(0, device=dev)
        self.w_s_d = g.w_by_source(self.w_t_d) if M > 0 else torch.zeros(0, device=dev)
        # operands of relgnn_batch_gather: the fold's adjacency as ONE type-major [M_fold, 2] list, payload rows as
        # 4-byte elements (anything else is gathered with index_select)
        self.adj_flat_d = torch.cat(self.adj_d).contiguous() if L else torch.zeros((0, 2), dtype=torch.int32, device=dev)
        self._fast = [i for i, t in enumerate(self.payload_d) if t.element_size() == 4 and t.is_contiguous()]
        self._fast_cols = [int(self.payload_d[i][0].numel()) if self.payload_d[i].shape[0] else 1 for i in self._fast]
        self._num_nodes_fold = N
        # pinned staging ring for the per-batch offset tables (one small H2D per batch); a slot is rewritten only after
        # the copy that read it has run
        self._stage = [None] * 4
        self._stage_done = [None] * 4
        self._stage_at = 0

    def _staging(self, n: int) -> torch.Tensor:
        k = self._stage_at
        self._stage_at = (k + 1) % len(self._stage)
        if self._stage_done[k] is not None:
            self._stage_done[k].synchronize()
        buf = self._stage[k]
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1024) * 2, dtype=torch.int64).pin_memory()
            self._stage[k] = buf
        return buf, k

    # ---- one batch ----------------------------------------------------------------------------
    def assemble(self, graph_ids: Sequence[int], lean: bool = True) -> DeviceBatch:
        """One batch from graph ids.  lean (default): only what every layer's fused gather kernels read is produced now —
        payload rows, degree tables, graph slot per node, both row-pointer arrays, source / target node and
        1/(in-degree + 1e-7) scale per bucketed message (16 bytes per message).  The batch's adjacency lists and the six
        permutation-type arrays of the bucketing (RelGraph._LAZY_ARRAYS) are gathered on their first read, from the same
        offset tables (the RGCN / GGNN sum paths never read them).  lean=False produces everything at once."""
        import ctypes
        lib = _lib.load_library()
        st = _lib.current_stream()
        dev, L, G = self.device, self.store.num_edge_types, self.store.num_graphs
        ids = np.ascontiguousarray(np.asarray(graph_ids, dtype=np.int64))
        K = len(ids)
        nodes = self.node_off[ids + 1] - self.node_off[ids]
        edges = self.edge_off[:, ids + 1] - self.edge_off[:, ids]                # [L, K]
        node_off_b = np.concatenate([[0], np.cumsum(nodes)])
        edge_off_b = np.concatenate([np.zeros((L, 1), np.int64), np.cumsum(edges, axis=1)], axis=1)     # [L, K+1]
        msg_off_b = np.concatenate([[0], np.cumsum(edges.sum(0))])
        type_off_b = np.concatenate([[0], np.cumsum(edge_off_b[:, -1])])
        V, M = int(node_off_b[-1]), int(msg_off_b[-1])
        # ONE small H2D: [ids | node_off_b | msg_off_b | type_off_b | edge_off_b]
        sizes = [K, K + 1, K + 1, L + 1, L * (K + 1)]
        total = sum(sizes)
        host, slot = self._staging(total)
        hn = host.numpy()
        at = 0
        for part in (ids, node_off_b, msg_off_b, type_off_b, edge_off_b.reshape(-1)):
            hn[at:at + part.size] = part
            at += part.size
        tab = host[:total].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._stage_done[slot] = ev
        o = np.cumsum([0] + sizes)
        ids_d, node_off_bd, msg_off_bd, type_off_bd, edge_off_bd = (tab[o[i]:o[i + 1]] for i in range(5))
        type_off = [int(x) for x in type_off_b]
        i32 = lambda n: torch.empty(int(n), dtype=torch.int32, device=dev)

        # ---- tensors of the batch: one C call, a handful of gather kernels (packing rules of relgnn_batch_pack) ----
        names = self.store.payload_names
        payload = {}
        out_fast = []
        for i in self._fast:
            src = self.payload_d[i]
            out_fast.append(torch.empty((V,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev))
            payload[names[i]] = out_fast[-1]
        deg = torch.empty((L, V), dtype=torch.float32, device=dev)
        n2g = torch.empty(V, dtype=torch.int32, device=dev)
        nf = len(self._fast)
        arr = ctypes.c_void_p * max(nf, 1)

        def gather(n_payloads, with_node_tables, adj_flat):
            _lib.check(lib.relgnn_batch_gather(
                _lib.ptr(ids_d), K, L, G, _lib.ptr(node_off_bd), _lib.ptr(edge_off_bd), _lib.ptr(type_off_bd),
                _lib.ptr(self.node_off_d), _lib.ptr(self.edge_off_d), _lib.ptr(self.type_off_d), V, M, self._num_nodes_fold,
                n_payloads, arr(*[self.payload_d[i].data_ptr() for i in self._fast]),
                (ctypes.c_int32 * max(nf, 1))(*self._fast_cols), arr(*[t.data_ptr() for t in out_fast]),
                _lib.ptr(self.deg_d), _lib.ptr(deg) if with_node_tables else None, _lib.ptr(self.adj_flat_d),
                _lib.ptr(adj_flat), _lib.ptr(n2g) if with_node_tables else None, _lib.current_stream()),
                "relgnn_batch_gather")

        def split_types(adj_flat):
            return [adj_flat[type_off[l]:type_off[l + 1]] for l in range(L)]

        adj_flat = None if lean else torch.empty((M, 2), dtype=torch.int32, device=dev)
        gather(nf, True, adj_flat)
        if len(self._fast) < len(names):          # payloads that are not 4-byte rows
            node_idx = (torch.arange(V, device=dev) - node_off_bd[n2g.long()] + self.node_off_d[ids_d[n2g.long()]])
            for i, name in enumerate(names):
                if i not in self._fast:
                    payload[name] = self.payload_d[i].index_select(0, node_idx)
        for name, flat in self.graph_payload_d.items():
            payload[name] = flat.index_select(0, ids_d)

        # ---- bucketing: re-based slices of the fold's arrays ----
        S = V * L
        rowptr_t, rowptr_s, tgt_s, src_t = i32(S + 1), i32(S + 1), i32(M), i32(M)
        w_t = torch.empty(M, dtype=torch.float32, device=dev)
        w_s = torch.empty(M, dtype=torch.float32, device=dev)
        d = self.plan_d

        def plan(full):
            """full=False: row pointers, src_t, tgt_s, scales.  full=True: additionally the six permutation-type arrays
            (the lean outputs are rewritten with the same values)."""
            six = {k: (i32(M) if full else None) for k in RelGraph._LAZY_ARRAYS}
            _lib.check(lib.relgnn_plan_assemble(
                _lib.ptr(ids_d), K, L, G, _lib.ptr(node_off_bd), _lib.ptr(msg_off_bd), _lib.ptr(edge_off_bd),
                _lib.ptr(type_off_bd), _lib.ptr(self.node_off_d), _lib.ptr(self.msg_off_d), _lib.ptr(self.edge_off_d),
                _lib.ptr(self.type_off_d), V, M,
                _lib.ptr(d["rowptr_t"]), _lib.ptr(d["perm_t"]), _lib.ptr(d["col_t"]), _lib.ptr(d["rowptr_s"]),
                _lib.ptr(d["perm_s"]), _lib.ptr(d["frow_s"]), _lib.ptr(d["pos_t_of_s"]),
                _lib.ptr(rowptr_t), _lib.ptr(six["perm_t"]), _lib.ptr(six["col_t"]), _lib.ptr(six["inv_perm_t"]),
                _lib.ptr(rowptr_s), _lib.ptr(six["perm_s"]), _lib.ptr(six["frow_s"]), _lib.ptr(tgt_s),
                _lib.ptr(six["pos_t_of_s"]), _lib.ptr(self.w_t_d), _lib.ptr(self.w_s_d), _lib.ptr(src_t), _lib.ptr(w_t),
                _lib.ptr(w_s), _lib.current_stream()), "relgnn_plan_assemble")
            return six

        state = {"adj": None if lean else split_types(adj_flat)}

        def adjacency():
            """the batch's adjacency lists (tasks/ppi_task.py:228), gathered when first read"""
            if state["adj"] is None:
                tab.record_stream(torch.cuda.current_stream(dev))        # `tab` was allocated on the assembling stream
                flat = torch.empty((M, 2), dtype=torch.int32, device=dev)
                gather(0, False, flat)
                state["adj"] = split_types(flat)
            return state["adj"]

        def complete():
            tab.record_stream(torch.cuda.current_stream(dev))
            return {"adjacency_lists": adjacency(), **plan(True)}

        six = plan(not lean)
        if lean:
            graph = RelGraph.from_arrays(None, V, rowptr_t=rowptr_t, rowptr_s=rowptr_s, tgt_s=tgt_s,
                                         edge_counts=[type_off[l + 1] - type_off[l] for l in range(L)], complete=complete)
        else:
            graph = RelGraph.from_arrays(state["adj"], V, rowptr_t=rowptr_t, rowptr_s=rowptr_s, tgt_s=tgt_s, **six)
        graph.preset_degree_scale(deg, src_t, w_t, w_s)
        if self.pair_counts is not None:
            graph.pair_counts = (self.pair_counts[0, ids].sum(0).tolist(), self.pair_counts[1, ids].sum(0).tolist())
        if self._has_hubs:
            graph.split_long_segments()
        batch = DeviceBatch.from_tensors(
            num_graphs=K, num_nodes=V, num_edges=M, initial_node_features=payload[self.features],
            adjacency_lists=adjacency if lean else state["adj"],
            type_to_num_incoming_edges=deg, graph_nodes_list=n2g,
            extra={**{k: v for k, v in payload.items() if k != self.features}, **self.constants})
        batch.graph = graph
        return batch

    def iterate(self, graph_ids: Sequence[int], max_nodes_per_batch: int) -> Iterator[DeviceBatch]:
        """Batches of one epoch (the training loop asks for batch i+1 right after it has enqueued step i), assembled on the CALLER's
        stream: the lean assembly is ~85 us of streaming kernels; on a side stream under the previous step they stretched whichever
        GEMM they met from 113 to 229 us (2.38 vs 2.44 ms per step, round 2) — that form was a switch until round 6."""
        for ids in self.store.split_batches(graph_ids, max_nodes_per_batch):
            yield self.assemble(ids)
