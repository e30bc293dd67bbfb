"""Native batch builder: flat dataset store + C++ packing into one pinned arena + one async upload.

Same batching semantics as the reference's numpy loops (tasks/ppi_task.py:209-256, tasks/qm9_task.py:212-261;
input contract tasks/sparse_graph_task.py:139-149), restated in include/relgnn.h section 9:
  * graphs are taken in order while node_offset + |V_g| < max_nodes_per_batch (strict);
  * adjacency lists are shifted by the node offset, degree tables / features / labels concatenated on the node axis;
  * an edge type without edges is an empty [0, 2] int32 list.
The reference hides its Python batching behind a ThreadedIterator (models/sparse_graph_model.py:272); here the
dataset is flattened ONCE (GraphStore), a batch is a list of graph ids, relgnn_batch_pack() assembles it with a few
host threads directly in pinned memory, and the whole batch crosses PCIe as ONE copy on a side stream while the
previous batch computes (NativeBatcher.iterate: double-buffered arenas, packing on a background thread).
"""
import ctypes
import queue
import threading
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .sparse_graph_task import DeviceBatch

_LAY_V, _LAY_M, _LAY_BYTES, _LAY_DEG, _LAY_N2G, _LAY_FIXED = 0, 1, 2, 3, 4, 5


def _addr(a: np.ndarray) -> int:
    return a.ctypes.data


class GraphStore:
    """All graphs of one data fold, flattened (host memory, immutable).

    payloads: name -> (attribute of the graph sample holding per-node rows, numpy dtype); every payload becomes a
    [V, ...] device tensor of the batch (torch dtype derived from the numpy dtype)."""

    def __init__(self, graphs: Sequence, num_edge_types: int, payloads: Dict[str, tuple],
                 graph_payloads: Optional[Dict[str, tuple]] = None):
        G, L = len(graphs), int(num_edge_types)
        self.num_graphs, self.num_edge_types = G, L
        counts = np.array([len(g.node_features) for g in graphs], dtype=np.int64)
        self.node_off = np.zeros(G + 1, np.int64)
        np.cumsum(counts, out=self.node_off[1:])
        self.edge_off, self.adj, self.deg = [], [], []
        for l in range(L):
            lists = [np.asarray(g.adjacency_lists[l]).reshape(-1, 2) for g in graphs]
            off = np.zeros(G + 1, np.int64)
            np.cumsum([len(a) for a in lists], out=off[1:])
            flat = np.concatenate(lists).astype(np.int32) if G and off[-1] else np.zeros((0, 2), np.int32)
            self.edge_off.append(off)
            self.adj.append(np.ascontiguousarray(flat))
            d = [np.asarray(g.type_to_node_to_num_incoming_edges)[l] for g in graphs]
            self.deg.append(np.ascontiguousarray(np.concatenate(d).astype(np.float32)) if G else np.zeros(0, np.float32))
        self.payload_names, self.payload, self.payload_tail, self.payload_dtype = [], [], [], []
        for name, (attr, dtype) in payloads.items():
            rows = [np.asarray(getattr(g, attr)) for g in graphs]
            flat = np.ascontiguousarray(np.concatenate(rows, axis=0).astype(dtype)) if G else np.zeros((0, 1), dtype)
            self.payload_names.append(name)
            self.payload.append(flat)
            self.payload_tail.append(tuple(flat.shape[1:]))
            self.payload_dtype.append(torch.from_numpy(np.zeros(1, dtype)).dtype)
        # per-GRAPH rows (e.g. QM9 regression targets): gathered in Python (a few KB), appended to the arena tail
        self.graph_payload = {}
        for name, (attr, dtype) in (graph_payloads or {}).items():
            rows = np.asarray([np.asarray(getattr(g, attr)) for g in graphs], dtype=dtype)
            self.graph_payload[name] = np.ascontiguousarray(rows.reshape(G, -1))
        P = len(self.payload)
        self.row_bytes = np.array([a.strides[0] if a.ndim > 1 else a.itemsize for a in self.payload], np.int64)
        # pointer tables handed to the C ABI (kept alive with the arrays they point into)
        self._edge_off_ptrs = (ctypes.c_void_p * max(L, 1))(*[_addr(a) for a in self.edge_off])
        self._adj_ptrs = (ctypes.c_void_p * max(L, 1))(*[_addr(a) for a in self.adj])
        self._deg_ptrs = (ctypes.c_void_p * max(L, 1))(*[_addr(a) for a in self.deg])
        self._payload_ptrs = (ctypes.c_void_p * max(P, 1))(*[_addr(a) for a in self.payload])
        self.layout_len = int(_lib.load_library().relgnn_batch_layout_len(L, P))

    # ---- thin wrappers over the C ABI ------------------------------------------------------
    def count_fitting(self, graph_ids: np.ndarray, first: int, max_nodes: int) -> int:
        n = int(_lib.load_library().relgnn_batch_count(_addr(self.node_off), _addr(graph_ids), len(graph_ids), first,
                                                       int(max_nodes)))
        if n < 0:
            raise ValueError("relgnn_batch_count: bad argument")
        return n

    def layout(self, graph_ids: np.ndarray) -> np.ndarray:
        lay = np.zeros(self.layout_len, np.int64)
        _lib.check(_lib.load_library().relgnn_batch_layout(
            self.num_edge_types, len(graph_ids), _addr(graph_ids), _addr(self.node_off), self._edge_off_ptrs,
            len(self.payload), _addr(self.row_bytes), _addr(lay)), "relgnn_batch_layout")
        return lay

    def pack_into(self, graph_ids: np.ndarray, lay: np.ndarray, arena_addr: int, arena_bytes: int, num_threads: int):
        _lib.check(_lib.load_library().relgnn_batch_pack(
            self.num_edge_types, len(graph_ids), _addr(graph_ids), _addr(self.node_off), self._edge_off_ptrs,
            self._adj_ptrs, self._deg_ptrs, len(self.payload), self._payload_ptrs, _addr(self.row_bytes), _addr(lay),
            arena_addr, arena_bytes, int(num_threads)), "relgnn_batch_pack")

    def split_batches(self, graph_ids: Sequence[int], max_nodes_per_batch: int) -> List[np.ndarray]:
        ids = np.ascontiguousarray(np.asarray(graph_ids, dtype=np.int64))
        out, at = [], 0
        while at < len(ids):
            n = self.count_fitting(ids, at, max_nodes_per_batch)
            if n == 0:
                raise ValueError("graph %d does not fit max_nodes_per_batch=%d" % (int(ids[at]), max_nodes_per_batch))
            out.append(ids[at:at + n].copy())
            at += n
        return out

    def views(self, arena: torch.Tensor, lay: np.ndarray):
        """Typed views of one packed arena (host or device uint8 tensor)."""
        L, P = self.num_edge_types, len(self.payload)
        V = int(lay[_LAY_V])

        def section(off, nbytes, dtype, shape):
            return arena[off:off + nbytes].view(dtype).view(shape)

        payload = {}
        for p, name in enumerate(self.payload_names):
            off = int(lay[_LAY_FIXED + p])
            payload[name] = section(off, V * int(self.row_bytes[p]), self.payload_dtype[p], (V,) + self.payload_tail[p])
        deg = section(int(lay[_LAY_DEG]), L * V * 4, torch.float32, (L, V))
        n2g = section(int(lay[_LAY_N2G]), V * 4, torch.int32, (V,))
        adj = []
        for l in range(L):
            E = int(lay[_LAY_FIXED + P + L + l])
            adj.append(section(int(lay[_LAY_FIXED + P + l]), E * 8, torch.int32, (E, 2)))
        return payload, deg, n2g, adj


class NativeBatcher:
    """Packs batches of a GraphStore into pinned arenas and uploads each with a single async copy."""

    def __init__(self, store: GraphStore, device, num_threads: Optional[int] = None, depth: int = 2,
                 features: str = "initial_node_features", constants: Optional[dict] = None, bucket: bool = True):
        self.store, self.device = store, torch.device(device)
        from ..parallel import effective_cpu_count
        self.num_threads = int(num_threads or max(1, min(8, effective_cpu_count() // 2)))   # leave the quota's other half to torch
        self.depth = max(2, int(depth))
        self.features = features
        self.constants = dict(constants or {})
        # bucket: also run the (target,type)/(source,type) bucketing of the batch on the copy stream right behind its
        # upload, so that it overlaps with the previous batch's compute (the batch then carries `.graph`)
        self.bucket = bool(bucket) and self.device.type == "cuda"
        self._host = [None] * self.depth      # pinned uint8 arenas
        self._dev = [None] * self.depth       # device uint8 arenas
        self._h2d_done = [None] * self.depth  # event: upload out of host arena k finished
        self._released = [None] * self.depth  # event: every consumer kernel of device arena k's last batch is enqueued
        self._slot = 0
        self._last = None
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _arena(self, pool, k, nbytes, pinned):
        cur = pool[k]
        if cur is None or cur.numel() < nbytes:
            cap = max(nbytes, 1 << 20) * 5 // 4
            if pinned:
                t = torch.empty(cap, dtype=torch.uint8)
                cur = t.pin_memory() if self.device.type == "cuda" else t
            else:
                cur = torch.empty(cap, dtype=torch.uint8, device=self.device)
            pool[k] = cur
        return cur

    def pack_host(self, graph_ids: np.ndarray, slot: int):
        """CPU half: layout + C++ packing into host arena `slot` (waits for that arena's previous upload)."""
        graph_ids = np.ascontiguousarray(np.asarray(graph_ids, dtype=np.int64))
        lay = self.store.layout(graph_ids)
        nbytes = int(lay[_LAY_BYTES])
        tail = {}                                           # name -> (offset, rows, cols, dtype) of per-graph rows
        for name, flat in self.store.graph_payload.items():
            tail[name] = (nbytes, len(graph_ids), flat.shape[1], flat.dtype)
            nbytes += (len(graph_ids) * flat.shape[1] * flat.itemsize + 255) // 256 * 256
        if self._h2d_done[slot] is not None:
            self._h2d_done[slot].synchronize()
        host = self._arena(self._host, slot, nbytes, pinned=True)
        self.store.pack_into(graph_ids, lay, host.data_ptr(), host.numel(), self.num_threads)
        host_np = host.numpy()
        for name, (off, rows, cols, dtype) in tail.items():
            dst = host_np[off:off + rows * cols * dtype.itemsize].view(dtype).reshape(rows, cols)
            np.take(self.store.graph_payload[name], graph_ids, axis=0, out=dst)
        return graph_ids, lay, host, nbytes, tail

    def upload(self, packed, slot: int) -> DeviceBatch:
        """GPU half: one H2D copy of the arena on the copy stream; the current stream waits for it."""
        graph_ids, lay, host, nbytes, tail = packed
        if self.device.type != "cuda":
            dev = host[:nbytes].clone()
        else:
            before = self._dev[slot]
            dev = self._arena(self._dev, slot, nbytes, pinned=False)
            cur = torch.cuda.current_stream(self.device)
            if dev is not before:
                # A fresh block from the caching allocator (first use, or the arena grew): the allocator hands out
                # memory whose previous owner's kernels may still be QUEUED on the current stream (the training loop
                # runs the host ahead of the GPU on purpose).  The copy stream must not write into it before those
                # kernels have run, and the allocator must know the block is used on the copy stream too.
                fresh = torch.cuda.Event()
                fresh.record(cur)
                self._copy_stream.wait_event(fresh)
                dev.record_stream(self._copy_stream)
                self._released[slot] = None
            # the arena's previous batch must be fully consumed before it is overwritten; NOT a wait on everything
            # enqueued so far, or the upload could never overlap with the batch that is computing right now
            if self._released[slot] is not None:
                self._copy_stream.wait_event(self._released[slot])
            with torch.cuda.stream(self._copy_stream):
                dev[:nbytes].copy_(host[:nbytes], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            self._h2d_done[slot] = ev
            cur.wait_stream(self._copy_stream)
        payload, deg, n2g, adj = self.store.views(dev, lay)
        for name, (off, rows, cols, dtype) in tail.items():   # per-graph rows: [n_graphs, cols]
            tdt = torch.from_numpy(np.zeros(1, dtype)).dtype
            payload[name] = dev[off:off + rows * cols * dtype.itemsize].view(tdt).view(rows, cols)
        graph = None
        if self.bucket and int(lay[_LAY_M]) > 0:
            from ..graph import RelGraph
            graph = RelGraph.build_on_stream(adj, int(lay[_LAY_V]), self._copy_stream)
        batch = DeviceBatch.from_tensors(
            num_graphs=len(graph_ids), num_nodes=int(lay[_LAY_V]), num_edges=int(lay[_LAY_M]),
            initial_node_features=payload[self.features], adjacency_lists=adj, type_to_num_incoming_edges=deg,
            graph_nodes_list=n2g,
            extra={**{k: v for k, v in payload.items() if k != self.features}, **self.constants})
        batch.graph = graph
        batch._arena_slot = slot
        return batch

    def release(self, batch: DeviceBatch):
        """The consumer has enqueued all work that reads `batch` (on the current stream): its arena may be reused once
        that work has run.  iterate() / pack() call this themselves when the next batch is requested."""
        slot = getattr(batch, "_arena_slot", None)
        if slot is not None and self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._released[slot] = ev

    def pack(self, graph_ids) -> DeviceBatch:
        """One batch, synchronously packed, asynchronously uploaded.  The batch stays valid until `depth` more
        batches have been produced by this batcher (its arena is then reused)."""
        slot = self._slot
        self._slot = (slot + 1) % self.depth
        if self._last is not None:
            self.release(self._last)
        self._last = self.upload(self.pack_host(graph_ids, slot), slot)
        return self._last

    def iterate(self, graph_ids: Sequence[int], max_nodes_per_batch: int) -> Iterator[DeviceBatch]:
        """All batches of `graph_ids` in order; batch i+1 is packed on a background thread while batch i is consumed
        (the reference's ThreadedIterator, models/sparse_graph_model.py:272)."""
        batches = self.store.split_batches(graph_ids, max_nodes_per_batch)
        q = queue.Queue()
        # A host arena may be re-packed only after the consumer has ISSUED the upload of the batch it holds (pack_host
        # then waits for that copy to finish): one token per arena, taken by the producer, returned by the consumer.
        # The producer therefore runs at most `depth` batches ahead, packing while the consumer enqueues GPU work.
        slot_free = [threading.Semaphore(1) for _ in range(self.depth)]
        stop = threading.Event()

        def produce():
            try:
                for i, ids in enumerate(batches):
                    slot = i % self.depth
                    while not slot_free[slot].acquire(timeout=0.05):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    q.put((slot, self.pack_host(ids, slot)))
                q.put(None)
            except BaseException as e:   # surface packing errors in the consumer
                q.put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, packed = item
                batch = self.upload(packed, slot)
                slot_free[slot].release()
                yield batch
                self.release(batch)      # the consumer is back: everything reading `batch` is enqueued
        finally:                         # also runs when the consumer abandons the iterator early
            stop.set()
            t.join()
