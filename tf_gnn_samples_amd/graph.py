"""Batched relational graph container: the (edge type, target)-bucketed CSR pair.

Input contract restated from the reference (tasks/sparse_graph_task.py:139-149):
  adjacency_lists: list of L int32 tensors [E_l, 2]; adjacency_lists[l][e] = [source, target]
  (messages flow column 0 -> column 1, gnns/rgcn.py:85-86); duplicated edges, self edges and
  empty edge types ([0, 2], tasks/ppi_task.py:248-249) are all legal.

The reference concatenates messages type-major (gnns/rgcn.py:78,108) and scatters them with
tf.unsorted_segment_*.  Here the message list is bucketed ONCE per batch by a stable sort on
(target, type) and on (source, type); every layer / timestep / backward pass then runs
gather+reduce kernels over those buckets (librelgnn: seg_reduce.hip) without atomics and
in the reference's per-target message order.

All index arithmetic happens on the device through the C ABI (relgnn_relational_keys,
relgnn_segment_plan, relgnn_gather_*): it is integer work and is tested bit-exact against
the NumPy oracle.
"""
import ctypes
import os
from collections import OrderedDict
from typing import List, Optional, Sequence

import torch

from . import _lib


_ZERO_FLAGS = {}


def _i32(n, device):
    return torch.empty(int(n), dtype=torch.int32, device=device)


class GatherReducePlan:
    """Everything relgnn_seg_reduce_fwd needs for  out[s] = REDUCE_p w[p] * X[col[p]]
    plus the transposed bucketing used for the gradient w.r.t. X."""

    def __init__(self, *, rowptr, stride, col, w, num_out, num_rows_x, rowptr_b, stride_b, col_b,
                 pos_b, num_messages):
        # `col` / `pos_b` may be zero-argument callables: resolved on first read (arrays a lean RelGraph has deferred)
        self.rowptr, self.stride, self._col, self.w = rowptr, int(stride), col, w
        self.num_out, self.num_rows_x = int(num_out), int(num_rows_x)
        self.rowptr_b, self.stride_b, self.col_b, self._pos_b = rowptr_b, int(stride_b), col_b, pos_b
        self.num_messages = int(num_messages)
        self._w_bwd = {}

    @property
    def col(self):
        if callable(self._col):
            self._col = self._col()
        return self._col

    @property
    def pos_b(self):
        if callable(self._pos_b):
            self._pos_b = self._pos_b()
        return self._pos_b

    def w_bwd(self, mode: int):
        """Per-message weights in the TRANSPOSED order for the gradient of sum/mean/sqrt_n:
        w_b[q] = w[p(q)] * f(n_segment(p(q))), f = 1, 1/max(n,1), 1/sqrt(max(n,1))."""
        if mode in self._w_bwd:
            return self._w_bwd[mode]
        lib = _lib.load_library()
        st = _lib.current_stream()
        M = self.num_messages
        if self.w is None and mode == _lib.AGG_SUM:
            res = None
        elif M == 0:
            res = torch.empty(0, dtype=torch.float32, device=self.rowptr.device)
        else:
            if mode == _lib.AGG_SUM:
                scale_f = self.w
            else:
                scale_f = torch.empty(M, dtype=torch.float32, device=self.rowptr.device)
                _lib.check(lib.relgnn_segment_counts_scale(
                    _lib.ptr(self.rowptr), self.num_out, self.stride, mode, _lib.ptr(self.w),
                    _lib.ptr(scale_f), st), "relgnn_segment_counts_scale")
            res = torch.empty(M, dtype=torch.float32, device=self.rowptr.device)
            _lib.check(lib.relgnn_gather_f32(_lib.ptr(scale_f), _lib.ptr(self.pos_b), M,
                                             _lib.ptr(res), st), "relgnn_gather_f32")
        self._w_bwd[mode] = res
        return res


def build_segment_plan(keys: torch.Tensor, num_segments: int, want_sorted_keys: bool = False):
    """Stable bucketing of messages by segment id: returns (rowptr [S+1], perm [M], sorted_keys|None)."""
    lib = _lib.load_library()
    M = keys.numel()
    dev = keys.device
    rowptr = _i32(num_segments + 1, dev)
    perm = _i32(M, dev)
    sorted_keys = _i32(M, dev) if want_sorted_keys else None
    ws_bytes = lib.relgnn_segment_plan_workspace_bytes(M, num_segments)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.relgnn_segment_plan(_lib.ptr(keys), M, num_segments, _lib.ptr(rowptr),
                                       _lib.ptr(perm), _lib.ptr(sorted_keys), _lib.ptr(ws), ws_bytes,
                                       _lib.current_stream()), "relgnn_segment_plan")
    return rowptr, perm, sorted_keys


class RelGraph:
    """(edge type, target)- and (edge type, source)-bucketed view of one batched graph."""

    def __init__(self, adjacency_lists: Sequence[torch.Tensor], num_nodes: int, validate=True):
        """validate=True : read the device-side range-check flag now (one host sync) and raise ValueError
                            for node ids outside [0, num_nodes), as TF-CPU raises InvalidArgumentError;
           validate="deferred": keep the flag on the device; call .check() (or check_pending_graph_errors())
                            at the next natural sync point — lets the host run ahead of the GPU;
           validate=False: never read it."""
        lib = _lib.load_library()
        if len(adjacency_lists) == 0:
            raise ValueError("need at least one edge type")
        dev = adjacency_lists[0].device
        adj = []
        for a in adjacency_lists:
            if a.dim() != 2 or a.shape[1] != 2:
                raise ValueError("adjacency lists must have shape [E, 2], got %s" % (tuple(a.shape),))
            if a.dtype != torch.int32:
                a = a.to(torch.int32)
            adj.append(a.contiguous())
        self.adjacency_lists = adj
        self.L = L = len(adj)
        self.V = V = int(num_nodes)
        self.edge_counts = [int(a.shape[0]) for a in adj]
        self.M = M = sum(self.edge_counts)
        self.device = dev
        if V * L >= 2 ** 31 - 1 or M >= 2 ** 31 - 1:
            raise ValueError("graph too large for int32 indices")
        st = _lib.current_stream()

        key_t, key_s = _i32(M, dev), _i32(M, dev)
        node_t, node_s = _i32(M, dev), _i32(M, dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        h_adj = (ctypes.c_void_p * L)(*[_lib.ptr(a) if a.shape[0] else None for a in adj])    # refuses host tensors
        h_cnt = (ctypes.c_int64 * L)(*self.edge_counts)
        _lib.check(lib.relgnn_relational_keys_all(h_adj, h_cnt, L, V, _lib.ptr(key_t), _lib.ptr(key_s), _lib.ptr(node_t),
                                                  _lib.ptr(node_s), _lib.ptr(err), st), "relgnn_relational_keys_all")
        self._key_t, self._key_s = key_t, key_s

        S = V * L
        ws_bytes = lib.relgnn_relational_plan_workspace_bytes(M, V)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        # by (target, type): position p -> original message perm_t[p]; gathers row col_t[p] = src*L + l
        self.rowptr_t, self.perm_t, self.col_t = _i32(S + 1, dev), _i32(M, dev), _i32(M, dev)
        inv_t = _i32(M, dev)
        _lib.check(lib.relgnn_relational_plan(_lib.ptr(node_t), _lib.ptr(key_t), _lib.ptr(key_s), M, V, L,
                                              _lib.ptr(self.rowptr_t), _lib.ptr(self.perm_t), _lib.ptr(self.col_t), None,
                                              _lib.ptr(inv_t), None, None, _lib.ptr(ws), ws_bytes, st),
                   "relgnn_relational_plan")
        # by (source, type): position q -> original message perm_s[q]; (target, type) row frow_s[q] = tgt*L + l,
        # target node tgt_s[q]; pos_t_of_s[q] = by-target position of the same message
        self.rowptr_s, self.perm_s = _i32(S + 1, dev), _i32(M, dev)
        self.frow_s, self.tgt_s, self.pos_t_of_s = _i32(M, dev), _i32(M, dev), _i32(M, dev)
        _lib.check(lib.relgnn_relational_plan(_lib.ptr(node_s), _lib.ptr(key_s), _lib.ptr(key_t), M, V, L,
                                              _lib.ptr(self.rowptr_s), _lib.ptr(self.perm_s), _lib.ptr(self.frow_s),
                                              _lib.ptr(self.tgt_s), None, _lib.ptr(inv_t), _lib.ptr(self.pos_t_of_s),
                                              _lib.ptr(ws), ws_bytes, st), "relgnn_relational_plan")
        self.inv_perm_t = inv_t
        self._src_t = None
        self._plans = {}
        self._scales = OrderedDict()
        self._err_flag = err
        self._checked = False
        if validate is True:
            self.check()
        elif validate == "deferred":
            # only the 4-byte flag is kept alive, not the graph (a training loop may build thousands of graphs
            # between two fetches)
            _PENDING_CHECKS.append((self._err_flag, self.V))
            if len(_PENDING_CHECKS) > 4096:
                check_pending_graph_errors()

    # index arrays / adjacency lists a producer may leave out (tasks/resident.py lean assembly) and fill on first use
    _LAZY_ARRAYS = ("perm_t", "col_t", "inv_perm_t", "perm_s", "frow_s", "pos_t_of_s")

    @classmethod
    def from_arrays(cls, adjacency_lists, num_nodes: int, *, rowptr_t, rowptr_s, tgt_s, perm_t=None, col_t=None,
                    inv_perm_t=None, perm_s=None, frow_s=None, pos_t_of_s=None, edge_counts=None, complete=None,
                    device=None) -> "RelGraph":
        """A RelGraph whose bucketing was produced elsewhere (tasks/resident.py: slices of a fold-level bucketing
        re-based by relgnn_plan_assemble).  The arrays must be what __init__ would compute; the node-id range check is
        the producer's job.
        Lean form: the six _LAZY_ARRAYS (and `adjacency_lists`, then give `edge_counts`) may be None when `complete` is a
        callable returning a dict with them ({"adjacency_lists": [...], "perm_t": ..., ...}); it runs on the first read
        of any of them (the sum / mean / sqrt_n layers of the RGCN / GGNN path never do)."""
        self = cls.__new__(cls)
        lazy = {k: v for k, v in dict(perm_t=perm_t, col_t=col_t, inv_perm_t=inv_perm_t, perm_s=perm_s, frow_s=frow_s,
                                      pos_t_of_s=pos_t_of_s).items()}
        missing = [k for k, v in lazy.items() if v is None]
        if (missing or adjacency_lists is None) and complete is None:
            raise ValueError("from_arrays: %s missing and no `complete` callable" % (missing or "adjacency_lists"))
        if adjacency_lists is not None:
            adj = [a if a.dtype == torch.int32 else a.to(torch.int32) for a in adjacency_lists]
            self.adjacency_lists = adj
            edge_counts = [int(a.shape[0]) for a in adj]
        elif edge_counts is None:
            raise ValueError("from_arrays: edge_counts is required when adjacency_lists is deferred")
        self.L = L = len(edge_counts)
        self.V = V = int(num_nodes)
        self.edge_counts = [int(e) for e in edge_counts]
        self.M = M = sum(self.edge_counts)
        self.device = dev = rowptr_t.device
        # the per-message keys (tgt*L+l, src*L+l in type-major order) are only read by the pair / materialised-message
        # paths: computed on first use (relgnn_relational_keys_all), not per batch
        self._key_t = self._key_s = None
        err = _ZERO_FLAGS.get(dev)            # never written for a graph whose producer validated it: one shared word
        if err is None:
            err = _ZERO_FLAGS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rowptr_t, self.rowptr_s, self.tgt_s = rowptr_t, rowptr_s, tgt_s
        for k, v in lazy.items():
            if v is not None:
                setattr(self, k, v)
        self._complete = complete
        self._src_t = None
        self._plans = {}
        self._scales = OrderedDict()
        self._err_flag = err
        self._checked = True
        return self

    def __getattr__(self, name):
        # only reached when normal lookup fails: a deferred array of a lean from_arrays() graph
        if name in RelGraph._LAZY_ARRAYS or name == "adjacency_lists":
            complete = self.__dict__.get("_complete")
            if complete is not None:
                self.__dict__["_complete"] = None
                for k, v in complete().items():
                    self.__dict__.setdefault(k, v)
                if name in self.__dict__:
                    return self.__dict__[name]
        raise AttributeError("%s has no attribute %r" % (type(self).__name__, name))

    def _ensure_keys(self):
        if self._key_t is None:
            lib = _lib.load_library()
            L, M, dev = self.L, self.M, self.device
            key_t, key_s, node_t, node_s = _i32(M, dev), _i32(M, dev), _i32(M, dev), _i32(M, dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            h_adj = (ctypes.c_void_p * L)(*[_lib.ptr(a) if a.shape[0] else None for a in self.adjacency_lists])
            h_cnt = (ctypes.c_int64 * L)(*self.edge_counts)
            _lib.check(lib.relgnn_relational_keys_all(h_adj, h_cnt, L, self.V, _lib.ptr(key_t), _lib.ptr(key_s),
                                                      _lib.ptr(node_t), _lib.ptr(node_s), _lib.ptr(err),
                                                      _lib.current_stream()), "relgnn_relational_keys_all")
            self._key_t, self._key_s = key_t, key_s

    @property
    def key_by_target(self):
        """tgt*L + l of every message in the reference's type-major order (gnns/rgcn.py:78,108)."""
        self._ensure_keys()
        return self._key_t

    @property
    def key_by_source(self):
        self._ensure_keys()
        return self._key_s

    def preset_degree_scale(self, type_to_num_incoming_edges: torch.Tensor, src_t, w_t, w_s):
        """Arrays the producer of a from_arrays() graph already holds (tasks/resident.py copies them from fold-level
        arrays inside relgnn_plan_assemble): the source node per by-target position and the per-message
        1/(in-degree + 1e-7) scales for THIS degree table, in by-target and by-source order."""
        t = type_to_num_incoming_edges
        self._src_t = src_t
        self._scales[(t.data_ptr(), t._version, tuple(t.shape))] = (t, w_t)
        plan = self.plan_transformed(w_t)
        plan._w_bwd[_lib.AGG_SUM] = w_s
        self._plans[("w_s", w_t.data_ptr())] = (w_t, w_s)
        self._preset = (src_t, w_t, w_s)

    # ---- bucketing on a side stream (input pipeline) --------------------------------------------
    ready_event = None

    @classmethod
    def build_on_stream(cls, adjacency_lists, num_nodes: int, stream, validate="deferred") -> "RelGraph":
        """Bucket a batch on `stream` (e.g. the copy stream that just uploaded it) while the previous batch computes
        on the main stream.  The consumer calls wait_ready() before the first kernel that reads the graph."""
        with torch.cuda.stream(stream):
            g = cls(adjacency_lists, num_nodes, validate=validate)
            g.ready_event = torch.cuda.Event()
            g.ready_event.record(stream)
        return g

    def wait_ready(self):
        """Make the current stream wait for a graph built by build_on_stream (no host sync) and tell the caching
        allocator that the graph's arrays are now used on this stream."""
        ev, self.ready_event = self.ready_event, None
        if ev is None:
            return self
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        d = self.__dict__                     # (deferred arrays of a lean graph are not touched: they do not exist yet)
        for t in (self._key_t, self._key_s, self.rowptr_t, self.rowptr_s, self.tgt_s, self._err_flag,
                  *[d.get(k) for k in RelGraph._LAZY_ARRAYS], *d.get("adjacency_lists", ()), *d.get("_preset", ())):
            if t is not None:
                t.record_stream(cur)
        return self

    def check(self):
        """Host sync: raise if the device-side validation saw a node id outside [0, V); the same round trip reads the
        longest bucket and splits long ones (split_long_segments)."""
        if not self._checked:
            self._checked = True
            longest = self._longest_buckets()
            flags = torch.cat([self._err_flag.to(torch.int64).reshape(1), longest]).tolist()
            if flags[0] != 0:
                # TF-CPU raises InvalidArgumentError for out-of-range gather / segment ids.
                raise ValueError("adjacency list holds a node id outside [0, %d)" % self.V)
            self._attach_split_plans(flags[1:])

    # ---- hubs: buckets too long for one wave -----------------------------------------------------
    def _bucketings(self):
        return ((self.rowptr_t, 1), (self.rowptr_t, self.L), (self.rowptr_s, 1), (self.rowptr_s, self.L))

    def _longest_buckets(self) -> torch.Tensor:
        """longest bucket of the four bucketings the kernels use: (target, type), target, (source, type), source"""
        if self.M == 0:
            return torch.zeros(4, dtype=torch.int64, device=self.device)
        out = []
        for rp, stride in self._bucketings():
            b = rp[0::stride]
            out.append((b[1:] - b[:-1]).max().to(torch.int64).reshape(1))
        return torch.cat(out)

    def _attach_split_plans(self, longest, threshold: int = None) -> bool:
        from . import ops
        threshold = ops.LONG_SEGMENT if threshold is None else threshold
        any_long = False
        for (rp, stride), n in zip(self._bucketings(), longest):
            if n > threshold:
                plans = getattr(rp, "_relgnn_split", None)
                if plans is None:
                    plans = rp._relgnn_split = {}
                if stride not in plans:
                    plans[stride] = ops.SplitPlan(rp, stride, self.V * self.L // stride, threshold)
                any_long = True
        self.has_long_buckets = any_long
        return any_long

    has_long_buckets = False

    def split_long_segments(self, threshold: int = None) -> bool:
        """One host round trip: find buckets longer than `threshold` messages (default ops.LONG_SEGMENT) and route the
        gather / reduce launches that walk them through chunked virtual rows (ops.SplitPlan).  Called by check() for
        validated graphs and by tasks/resident.py for batches of a fold that is known to hold such hubs; graphs built
        with validate="deferred" keep one wave per bucket (PPI-, QM9- and VarMisuse-shaped data: longest bucket < 2 k)."""
        return self._attach_split_plans(self._longest_buckets().tolist(), threshold)

    # ---- derived index arrays -----------------------------------------------------------
    @property
    def src_t(self):
        """source NODE of each by-target position (row into an untransformed [V, D] table)."""
        if self._src_t is None:
            lib = _lib.load_library()
            self._src_t = _i32(self.M, self.device)
            _lib.check(lib.relgnn_gather_div_i32(_lib.ptr(self.key_by_source), _lib.ptr(self.perm_t),
                                                 self.M, self.L, _lib.ptr(self._src_t),
                                                 _lib.current_stream()), "relgnn_gather_div_i32")
        return self._src_t

    def degree_scale(self, type_to_num_incoming_edges: torch.Tensor) -> torch.Tensor:
        """w[p] = 1/(type_to_num_incoming_edges[l, v] + 1e-7) for by-target position p
        (gnns/rgcn.py:100-104); cached per degree tensor."""
        t = type_to_num_incoming_edges
        key = (t.data_ptr(), t._version, tuple(t.shape))
        hit = self._scales.get(key)
        if hit is not None:
            return hit[1]
        if tuple(t.shape) != (self.L, self.V):
            raise ValueError("type_to_num_incoming_edges must have shape [%d, %d]" % (self.L, self.V))
        tt = t.to(torch.float32).contiguous()
        w = torch.empty(self.M, dtype=torch.float32, device=self.device)
        lib = _lib.load_library()
        if self.M > 0:           # an edge-free batch has no message to scale
            _lib.check(lib.relgnn_degree_scale(_lib.ptr(tt), _lib.ptr(self.rowptr_t), self.L, self.V, 1e-7,
                                           _lib.ptr(w), _lib.current_stream()), "relgnn_degree_scale")
        self._scales[key] = (t, w)  # keep `t` alive so the data_ptr key cannot be recycled
        while len(self._scales) > 4:
            self._scales.popitem(last=False)
        return w

    def w_by_source(self, w: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """per-message weights re-ordered from by-target positions to by-(source,type) positions."""
        if w is None:
            return None
        key = ("w_s", w.data_ptr())
        if key not in self._plans:
            lib = _lib.load_library()
            ws = torch.empty_like(w)
            _lib.check(lib.relgnn_gather_f32(_lib.ptr(w), _lib.ptr(self.pos_t_of_s), self.M, _lib.ptr(ws),
                                             _lib.current_stream()), "relgnn_gather_f32")
            self._plans[key] = (w, ws)
        return self._plans[key][1]

    def w_original_order(self, w: torch.Tensor) -> torch.Tensor:
        """per-message weights in the reference's type-major message order."""
        key = ("w_o", w.data_ptr())
        if key not in self._plans:
            lib = _lib.load_library()
            wo = torch.empty_like(w)
            _lib.check(lib.relgnn_gather_f32(_lib.ptr(w), _lib.ptr(self.inv_perm_t), self.M, _lib.ptr(wo),
                                             _lib.current_stream()), "relgnn_gather_f32")
            self._plans[key] = (w, wo)
        return self._plans[key][1]

    def messages_per_target(self) -> torch.Tensor:
        """n_v = number of incoming messages of node v over all edge types, float32 [V], clamped to >= 1
        (the N of tf.unsorted_segment_mean / sqrt_n)."""
        if "n_v" not in self._plans:
            rp = self.rowptr_t
            n = (rp[self.L::self.L] - rp[:-1:self.L]).clamp(min=1).to(torch.float32)
            self._plans["n_v"] = n
        return self._plans["n_v"]

    @property
    def iota(self):
        """[0, 1, ..., M-1] int32 (identity gather for per-message tables already in by-target order)."""
        if "iota" not in self._plans:
            self._plans["iota"] = torch.arange(self.M, dtype=torch.int32, device=self.device)
        return self._plans["iota"]

    @property
    def tgt_t(self):
        """target NODE of each by-target position."""
        if "tgt_t" not in self._plans:
            lib = _lib.load_library()
            t = _i32(self.M, self.device)
            _lib.check(lib.relgnn_gather_div_i32(_lib.ptr(self.key_by_target), _lib.ptr(self.perm_t), self.M, self.L,
                                                 _lib.ptr(t), _lib.current_stream()), "relgnn_gather_div_i32")
            self._plans["tgt_t"] = t
        return self._plans["tgt_t"]

    # ---- plans ----------------------------------------------------------------------------
    def _deferred(self, name: str):
        """A zero-argument getter of one of this graph's (possibly lazily produced) arrays for a plan's deferred fields.
        Holds the graph WEAKLY: the plan lives in self._plans, and a strong reference back would make graph <-> plan a
        cycle that only the cyclic collector frees — every batch's index arrays (tens of MB at C2) would then outlive
        the step until a full collection happens to run."""
        have = self.__dict__.get(name)
        if have is not None:                  # already materialised (every graph but a lean from_arrays() one): hand the
            return have                       # tensor over, so that a plan used in backward does not need the graph alive
        import weakref
        ref = weakref.ref(self)

        def get():
            g = ref()
            if g is None:
                raise RuntimeError("the RelGraph of this plan is gone")
            return getattr(g, name)
        return get

    def plan_transformed(self, w: Optional[torch.Tensor] = None) -> GatherReducePlan:
        """Messages gathered from a per-(node, type) table T [V*L, D] (row = src*L + l), reduced
        over ALL edge types into the target node.  w: optional by-target per-message weights."""
        key = ("T", None if w is None else w.data_ptr())
        if key not in self._plans:
            self._plans[key] = (w, GatherReducePlan(
                rowptr=self.rowptr_t, stride=self.L, col=self._deferred("col_t"), w=w, num_out=self.V,
                num_rows_x=self.V * self.L, rowptr_b=self.rowptr_s, stride_b=1, col_b=self.tgt_s,
                pos_b=self._deferred("pos_t_of_s"), num_messages=self.M))
        return self._plans[key][1]

    def plan_untransformed(self, w: Optional[torch.Tensor] = None) -> GatherReducePlan:
        """Messages gathered straight from the node states H [V, D] (row = src)."""
        key = ("H", None if w is None else w.data_ptr())
        if key not in self._plans:
            self._plans[key] = (w, GatherReducePlan(
                rowptr=self.rowptr_t, stride=self.L, col=self.src_t, w=w, num_out=self.V,
                num_rows_x=self.V, rowptr_b=self.rowptr_s, stride_b=self.L, col_b=self.tgt_s,
                pos_b=self._deferred("pos_t_of_s"), num_messages=self.M))
        return self._plans[key][1]


    def plan_messages(self) -> GatherReducePlan:
        """Messages materialised as an [M, D] tensor in the reference's type-major order (the operand of
        tf.unsorted_segment_* in the reference), reduced into their target nodes."""
        key = ("MSG",)
        if key not in self._plans:
            dev = self.device
            if "tgt_orig" not in self._plans:
                lib = _lib.load_library()
                t = _i32(self.M, dev)
                ident = torch.arange(self.M, dtype=torch.int32, device=dev)
                _lib.check(lib.relgnn_gather_div_i32(_lib.ptr(self.key_by_target), _lib.ptr(ident), self.M, self.L,
                                                     _lib.ptr(t), _lib.current_stream()), "relgnn_gather_div_i32")
                self._plans["tgt_orig"] = t
            self._plans[key] = (None, GatherReducePlan(
                rowptr=self.rowptr_t, stride=self.L, col=self.perm_t, w=None, num_out=self.V, num_rows_x=self.M,
                rowptr_b=torch.arange(self.M + 1, dtype=torch.int32, device=dev), stride_b=1,
                col_b=self._plans["tgt_orig"], pos_b=self.inv_perm_t, num_messages=self.M))
        return self._plans[key][1]

    def plan_target_rows(self) -> GatherReducePlan:
        """Every message carries its TARGET node's own row (the h_v half of [h_u || h_v] when no edge MLP
        follows, gnns/rgin.py:114-125): out[v] = AGG over v's messages of X[v]."""
        key = ("TGT",)
        if key not in self._plans:
            self._plans[key] = (None, GatherReducePlan(
                rowptr=self.rowptr_t, stride=self.L, col=self.tgt_t, w=None, num_out=self.V, num_rows_x=self.V,
                rowptr_b=self.rowptr_t, stride_b=self.L, col_b=self.tgt_t,
                pos_b=torch.arange(self.M, dtype=torch.int32, device=self.device), num_messages=self.M))
        return self._plans[key][1]

    # ---- compact (node, type) pair tables -------------------------------------------------------
    def pair_tables(self) -> "PairTables":
        """Compact numbering of the NON-EMPTY (node, type) buckets (see PairTables); built once, one host sync."""
        if "pairs" not in self._plans:
            self._plans["pairs"] = PairTables(self)
        return self._plans["pairs"]

    def wants_pair_tables(self) -> bool:
        """Node-side per-type transforms over all V*L (node,type) rows waste work when most buckets are empty
        (VarMisuse-shaped graphs: 23 edge types, ~2/3 of the buckets empty).  Few-type graphs (PPI: every bucket
        non-empty) keep the dense [V*L, D] tables and one big GEMM.  RELGNN_PAIR_TABLES=0/1 overrides."""
        from .config import settings
        if settings.pair_tables != "auto":
            return settings.pair_tables == "1"
        if self.L < 8 or self.M == 0:
            return False
        known = getattr(self, "pair_counts", None)
        if known is not None:                      # (a resident fold counted them per graph: decided without building anything)
            return (sum(known[0]) + sum(known[1])) < 0.6 * (2 * self.V * self.L)
        pt = self.pair_tables()
        return (pt.tgt.num_pairs + pt.src.num_pairs) < 0.6 * (2 * self.V * self.L)

    @property
    def type_offsets(self):
        """start of every edge type in the type-major message list (python ints), length L+1."""
        offs = [0]
        for e in self.edge_counts:
            offs.append(offs[-1] + e)
        return offs


# rows per GEMM batch entry of the compact tables; every type's row block is padded to a multiple
PAIR_CHUNK = 512


# Small index tables that are computed on the host (numpy) go to the device through a ring of pinned staging buffers with an
# asynchronous copy: torch.as_tensor(numpy, device=...) copies from pageable memory and blocks the host until everything queued on
# the stream has run — one such call per batch is enough to stop the host from running ahead of the GPU (C5: host-bound, 45 ms).
_UPLOAD_SLOTS, _UPLOAD_BYTES = 8, 1 << 20
_upload_ring = {"bufs": [None] * _UPLOAD_SLOTS, "done": [None] * _UPLOAD_SLOTS, "at": 0}


def _upload(arr, device, dtype=None) -> torch.Tensor:
    """numpy array -> device tensor without a host / stream synchronisation (small arrays; larger ones: a plain copy)."""
    import numpy as np
    arr = np.ascontiguousarray(arr)
    device = torch.device(device)
    if (device.type != "cuda" or arr.nbytes > _UPLOAD_BYTES or arr.nbytes == 0
            or torch.cuda.is_current_stream_capturing()):
        # (under stream capture: the ring's event.synchronize() is not capturable and a replay would re-read a staging slot that
        #  has been overwritten since — a plain copy from the array's own memory instead)
        t = torch.as_tensor(arr, device=device)
        return t if dtype is None else t.to(dtype)
    k = _upload_ring["at"]
    _upload_ring["at"] = (k + 1) % _UPLOAD_SLOTS
    if _upload_ring["done"][k] is not None:
        _upload_ring["done"][k].synchronize()          # (the copy that read this slot eight uploads ago: long done)
    if _upload_ring["bufs"][k] is None:
        _upload_ring["bufs"][k] = torch.empty(_UPLOAD_BYTES, dtype=torch.uint8).pin_memory()
    stage = _upload_ring["bufs"][k][:arr.nbytes]
    stage.numpy()[:] = arr.view(np.uint8).reshape(-1)
    out = stage.to(device, non_blocking=True).view(torch.from_numpy(arr[:0]).dtype).view(arr.shape)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _upload_ring["done"][k] = ev
    return out if dtype is None else out.to(dtype)


_NONZERO_STATIC_OK = {}
_VALIDATE_KNOWN_COUNTS = False     # debug (tests set it): compare host-side counts with the device's, one synchronisation each


def _nonzero_known(mask: torch.Tensor, size: int) -> torch.Tensor:
    """torch.nonzero(mask).view(-1) for a 1-D mask whose number of set entries the HOST already knows: no device -> host round trip."""
    key = mask.device.type
    ok = _NONZERO_STATIC_OK.get(key)
    if ok is None:                  # (older builds expose nonzero_static for CPU tensors only: probe once per device type)
        try:
            probe = torch.nonzero_static(torch.ones(2, dtype=torch.bool, device=mask.device), size=2)
            ok = probe.numel() == 2
        except (AttributeError, RuntimeError, NotImplementedError):
            ok = False
        _NONZERO_STATIC_OK[key] = ok
    if not ok:
        return torch.nonzero(mask).view(-1)
    out = torch.nonzero_static(mask, size=int(size), fill_value=-1).view(-1)
    if _VALIDATE_KNOWN_COUNTS:      # debug: the host-side count against the device's (one host synchronisation)
        have = int(mask.sum())
        if have != int(size):
            raise ValueError("_nonzero_known: the producer of the graph said %d set entries, the mask has %d" % (int(size), have))
    return out


class SidePairs:
    """One side (sources or targets) of the compact pair numbering.

    Rows are TYPE-MAJOR and every type's block is padded to a multiple of PAIR_CHUNK rows, so the whole table is a
    batch of [PAIR_CHUNK, D] tiles that each belong to ONE edge type: the per-type transforms become a single batched
    GEMM with the weight picked per tile (ops.typed_linear).  Padding rows gather an all-zero input row, are never
    referenced by a message, and cost <= L * PAIR_CHUNK rows.

    P            number of table rows (including padding); num_pairs = non-empty (node, type) buckets
    bucket_row   [V*L] int32  node-major bucket (node*L + type) -> row, -1 for empty buckets
    node         [P]   int64  node of each row (ascending node id inside a type); V for padding rows
    offsets      L+1 python ints: padded start of every type's block
    chunk_type   [P / PAIR_CHUNK] int64: edge type of every tile
    node_rowptr  [V+1] int32, node_col [num_pairs] int32: CSR node -> its rows (for summing per-pair gradients back
                 into the node in ascending type order, the order the dense GEMM would add them)"""

    def __init__(self, rowptr: torch.Tensor, V: int, L: int, chunk: int = PAIR_CHUNK):
        """(counts per type are read back by PairTables — one host sync — unless the producer of the graph knows them: a resident
        fold counts the non-empty buckets of every graph once, tasks/resident.py -> RelGraph.pair_counts)"""
        dev = rowptr.device
        self.V, self.L, self.chunk = V, L, chunk
        nonempty = (rowptr[1:] > rowptr[:-1]).view(V, L)
        by_type = nonempty.t().contiguous()                                   # [L, V]
        # position inside the type's block: ONE flat scan (a row-wise cumsum of an [L, V] tensor runs one workgroup per row —
        # 227 us for the 23 types of a VarMisuse-shaped batch, twice per batch), minus the pairs of the types before
        if V > 0:
            flat = torch.cumsum(by_type.view(-1), 0, dtype=torch.int32)
            ends = flat[V - 1::V]                                             # [L] pairs up to and including type l
            counts = torch.diff(ends, prepend=ends.new_zeros(1))
            rank = flat.view(L, V) - (ends - counts + 1).unsqueeze(1)
        else:
            rank = torch.zeros((L, 0), dtype=torch.int32, device=dev)
            counts = torch.zeros(L, dtype=torch.int32, device=dev)
        padded = (counts + (chunk - 1)) // chunk * chunk
        starts = torch.cumsum(padded, 0, dtype=torch.int32) - padded          # [L]
        ids = torch.where(by_type, rank + starts.unsqueeze(1), torch.full_like(rank, -1))
        self.bucket_row = ids.t().contiguous().view(-1)                       # [V*L], node-major
        self._counts_dev = torch.stack([counts, padded])                      # read once by PairTables (one sync)
        self._by_type, self._ids, self._nonempty = by_type, ids, nonempty
        per_node = nonempty.sum(1, dtype=torch.int32)
        self.node_rowptr = torch.zeros(V + 1, dtype=torch.int32, device=dev)
        torch.cumsum(per_node, 0, dtype=torch.int32, out=self.node_rowptr[1:])
        self._dw_plans = {}

    def _finish(self, counts: List[int], padded: List[int]):
        dev = self.bucket_row.device
        self.offsets = [0]
        for c in padded:
            self.offsets.append(self.offsets[-1] + int(c))
        self.P = self.offsets[-1]
        self.num_pairs = int(sum(counts))
        self.type_counts = [int(c) for c in counts]
        import numpy as np
        # (every size below is known on the host: no device -> host round trip, no pageable copy)
        self.node_col = self.bucket_row[_nonzero_known(self._nonempty.view(-1), self.num_pairs)].contiguous()   # node-major order
        node = torch.full((self.P,), self.V, dtype=torch.int64, device=dev)
        where = _nonzero_known(self._by_type.view(-1), self.num_pairs)        # type-major positions of real pairs
        node[self._ids.view(-1)[where].long()] = where % max(self.V, 1)
        self.node = node
        self.pad_rows = _nonzero_known(node == self.V, self.P - self.num_pairs)   # <= L * chunk rows
        chunks = [int(p) // self.chunk for p in padded]
        self.chunk_counts = chunks
        self.chunk_type = _upload(np.repeat(np.arange(self.L, dtype=np.int64), chunks), dev)
        del self._by_type, self._ids, self._nonempty

    def panel_indices(self):
        """(row -> node as int32 with -1 for padding rows, tile -> edge type as int32): the index operands of
        relgnn_panel_gemm_f32 (gathered rows, per-tile kernel)."""
        cached = getattr(self, "_panel_idx", None)
        if cached is None:
            node32 = torch.where(self.node == self.V, torch.full_like(self.node, -1), self.node).to(torch.int32).contiguous()
            cached = self._panel_idx = (node32, self.chunk_type.to(torch.int32).contiguous())
        return cached

    def weight_grad_plan(self, num_sub_rows: int):
        """CSR that sums per-tile partial weight gradients [num_tiles * K, 1024] (K = num_sub_rows 1024-float slices
        of one [Din, Dout] partial) into [L * K, 1024]: out row (l, j) <- rows (tile * K + j) for the tiles of type l."""
        plan = self._dw_plans.get(num_sub_rows)
        if plan is None:
            import numpy as np
            K = num_sub_rows
            tile_start = np.concatenate([[0], np.cumsum(self.chunk_counts)])
            rowptr = np.zeros(self.L * K + 1, np.int64)
            cols = []
            for l in range(self.L):
                n = self.chunk_counts[l]
                tiles = np.arange(tile_start[l], tile_start[l] + n)
                for j in range(K):
                    rowptr[l * K + j + 1] = rowptr[l * K + j] + n
                cols.append((tiles[None, :] * K + np.arange(K)[:, None]).reshape(-1))
            col = np.concatenate(cols) if cols else np.zeros(0, np.int64)
            dev = self.bucket_row.device
            plan = (_upload(rowptr.astype(np.int32), dev), _upload(col.astype(np.int32), dev))
            self._dw_plans[num_sub_rows] = plan
        return plan


class PairTables:
    """Compact row numbering for per-(node, type) tables over the non-empty buckets only.

    tgt : SidePairs of the by-(target, type) buckets (rows of FiLM-weight / target-side tables)
    src : SidePairs of the by-(source, type) buckets (rows of transformed-message tables)
    col_t  [M] int32  src-table row gathered by each by-target message   (replaces RelGraph.col_t)
    frow_s [M] int32  tgt-table row of each by-source message             (replaces RelGraph.frow_s)"""

    def __init__(self, g: "RelGraph"):
        self.tgt = SidePairs(g.rowptr_t, g.V, g.L)
        self.src = SidePairs(g.rowptr_s, g.V, g.L)
        known = getattr(g, "pair_counts", None)      # (non-empty buckets per type, by target / by source) from the graph's producer
        if known is not None:
            chunk = self.tgt.chunk
            counts = [[list(c), [(int(x) + chunk - 1) // chunk * chunk for x in c]] for c in known]
        else:
            counts = torch.stack([self.tgt._counts_dev, self.src._counts_dev]).tolist()   # the one host sync
        self.tgt._finish(*counts[0])
        self.src._finish(*counts[1])
        self.P_t, self.P_s = self.tgt.P, self.src.P
        self.col_t = self.src.bucket_row[g.col_t.long()].contiguous()
        self.frow_s = self.tgt.bucket_row[g.frow_s.long()].contiguous()
        self._graph = g
        self._by_row = None
        self._plans = {}

    def node_csr_both(self):
        """(rowptr [V+1], col [num_pairs_src + num_pairs_tgt]) int32: node -> its rows in the by-source table FOLLOWED by its rows in
        the by-target table, the latter numbered behind the former (row + P_s) — the CSR of ONE reduction that sums the per-row
        input gradients of both typed transforms of a layer into the nodes (ops._TypedLinearPair)."""
        both = getattr(self, "_both", None)
        if both is None:
            a, b = self.src, self.tgt
            V, dev = a.V, a.node_rowptr.device
            ca, cb = torch.diff(a.node_rowptr), torch.diff(b.node_rowptr)
            rowptr = torch.zeros(V + 1, dtype=torch.int32, device=dev)
            torch.cumsum(ca + cb, 0, dtype=torch.int32, out=rowptr[1:])
            nodes = torch.arange(V, device=dev)
            col = torch.empty(a.num_pairs + b.num_pairs, dtype=torch.int32, device=dev)
            for side, cnt, lead, shift in ((a, ca, None, 0), (b, cb, ca, a.P)):
                n = side.num_pairs
                if n == 0:
                    continue
                node_of = torch.repeat_interleave(nodes, cnt.long(), output_size=n)
                dest = rowptr[node_of].long() + (torch.arange(n, device=dev) - side.node_rowptr[node_of].long())
                if lead is not None:
                    dest = dest + lead[node_of].long()
                col[dest] = side.node_col + shift
            both = self._both = (rowptr, col)
        return both

    def _messages_by_source_row(self):
        """Stable bucketing of the messages by the compact row of their (source, type) bucket: the transposed plan
        whose output rows ARE the compact rows (the table is type-major, the by-source order of RelGraph node-major)."""
        if self._by_row is None:
            g = self._graph
            lib = _lib.load_library()
            st = _lib.current_stream()
            keys = self.src.bucket_row[g.key_by_source.long()].contiguous()            # row of every original message
            rowptr, perm, _ = build_segment_plan(keys, self.P_s)
            tgt, pos_t = _i32(g.M, g.device), _i32(g.M, g.device)
            _lib.check(lib.relgnn_gather_div_i32(_lib.ptr(g.key_by_target), _lib.ptr(perm), g.M, g.L, _lib.ptr(tgt), st),
                       "relgnn_gather_div_i32")
            _lib.check(lib.relgnn_gather_i32(_lib.ptr(g.inv_perm_t), _lib.ptr(perm), g.M, _lib.ptr(pos_t), st),
                       "relgnn_gather_i32")
            self._by_row = (rowptr, tgt, pos_t)
        return self._by_row

    def plan_transformed(self, w: Optional[torch.Tensor] = None) -> GatherReducePlan:
        """RelGraph.plan_transformed for a compact source table T [P_s, D]: messages gather row col_t[p]; the gradient
        is reduced straight into the compact rows (padding rows own no message and come out zero)."""
        key = None if w is None else w.data_ptr()
        if key not in self._plans:
            g = self._graph
            rowptr_b, tgt_b, pos_b = self._messages_by_source_row()
            self._plans[key] = (w, GatherReducePlan(
                rowptr=g.rowptr_t, stride=g.L, col=self.col_t, w=w, num_out=g.V, num_rows_x=self.P_s,
                rowptr_b=rowptr_b, stride_b=1, col_b=tgt_b, pos_b=pos_b, num_messages=g.M))
        return self._plans[key][1]


# ---- cache: the layer functions receive raw adjacency lists on every call ------------------
_GRAPH_CACHE: "OrderedDict[tuple, RelGraph]" = OrderedDict()
_GRAPH_CACHE_SIZE = 8


_PENDING_CHECKS: List[tuple] = []


def check_pending_graph_errors():
    """Validate every RelGraph built with validate="deferred" since the last call (one host sync each)."""
    pending, _PENDING_CHECKS[:] = list(_PENDING_CHECKS), []
    if not pending:
        return
    flags = torch.cat([f for f, _ in pending])
    bad = torch.nonzero(flags).flatten().tolist()          # one host sync for all pending graphs
    if bad:
        raise ValueError("adjacency list holds a node id outside [0, %d)" % pending[bad[0]][1])


def as_rel_graph(adjacency_lists, num_nodes: int, validate=True) -> RelGraph:
    """RelGraph for these adjacency tensors (built once per batch, reused by every layer)."""
    if isinstance(adjacency_lists, RelGraph):
        if adjacency_lists.V != num_nodes:
            raise ValueError("RelGraph was built for %d nodes, got %d" % (adjacency_lists.V, num_nodes))
        return adjacency_lists.wait_ready()
    key = tuple((a.data_ptr(), tuple(a.shape), a._version, str(a.dtype)) for a in adjacency_lists) + (int(num_nodes),)
    g = _GRAPH_CACHE.get(key)
    if g is not None:
        _GRAPH_CACHE.move_to_end(key)
        return g
    g = RelGraph(adjacency_lists, num_nodes, validate=validate)
    g._cache_refs = list(adjacency_lists)  # keep the keyed storage alive while cached
    _GRAPH_CACHE[key] = g
    while len(_GRAPH_CACHE) > _GRAPH_CACHE_SIZE:
        _GRAPH_CACHE.popitem(last=False)
    return g


def clear_graph_cache():
    _GRAPH_CACHE.clear()
