"""Input contract of the hot path (mirror of tasks/sparse_graph_task.py:10-20,139-149).

A minibatch is ONE disjoint-union graph:
  initial_node_features       float32 [V, F]
  adjacency_lists             L x int32 [E_l, 2], rows = [source, target]
  type_to_num_incoming_edges  float32 [L, V]   (float despite the reference docstring, :144-145)
"""
from enum import Enum
from typing import Any, Dict, NamedTuple, Optional

import numpy as np
import torch


class DataFold(Enum):
    TRAIN = 0
    VALIDATION = 1
    TEST = 2


class MinibatchData(NamedTuple):
    feed_dict: Dict[str, Any]
    num_graphs: int
    num_nodes: int
    num_edges: int


class DeviceBatch:
    """The feed dict of one minibatch, resident in HBM (the reference copies host->device on every
    sess.run, models/sparse_graph_model.py:293; here it is done once, explicitly)."""

    def __init__(self, mb: MinibatchData, device, pin: bool = False):
        fd = mb.feed_dict
        self.num_graphs, self.num_nodes, self.num_edges = mb.num_graphs, mb.num_nodes, mb.num_edges

        def put(a, dtype):
            t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)
            if pin and device != "cpu":
                t = t.pin_memory()
            return t.to(device, non_blocking=pin)

        self.initial_node_features = put(fd['initial_node_features'], torch.float32)
        self.adjacency_lists = [put(np.asarray(a).reshape(-1, 2), torch.int32) for a in fd['adjacency_lists']]
        self.type_to_num_incoming_edges = put(fd['type_to_num_incoming_edges'], torch.float32)
        self.graph_nodes_list = put(fd['graph_nodes_list'], torch.int32) if fd.get('graph_nodes_list') is not None else None
        self.extra = {k: v for k, v in fd.items() if k not in (
            'initial_node_features', 'adjacency_lists', 'type_to_num_incoming_edges', 'graph_nodes_list')}
        for k, v in list(self.extra.items()):
            if isinstance(v, np.ndarray):
                self.extra[k] = put(v, torch.float32 if v.dtype.kind == 'f' else torch.int64)

    @classmethod
    def from_tensors(cls, num_graphs, num_nodes, num_edges, initial_node_features, adjacency_lists,
                     type_to_num_incoming_edges, graph_nodes_list=None, extra=None) -> "DeviceBatch":
        """A batch whose tensors already live on the device (tasks/batcher.py: views of one uploaded arena)."""
        self = cls.__new__(cls)
        self.num_graphs, self.num_nodes, self.num_edges = int(num_graphs), int(num_nodes), int(num_edges)
        self.initial_node_features = initial_node_features
        # a zero-argument callable defers the lists (tasks/resident.py: the layers get the batch's bucketing, the lists are
        # gathered only if something reads them)
        self._adjacency = adjacency_lists if callable(adjacency_lists) else list(adjacency_lists)
        self.type_to_num_incoming_edges = type_to_num_incoming_edges
        self.graph_nodes_list = graph_nodes_list
        self.extra = dict(extra or {})
        return self

    @property
    def adjacency_lists(self):
        if callable(self._adjacency):
            self._adjacency = list(self._adjacency())
        return self._adjacency

    @adjacency_lists.setter
    def adjacency_lists(self, value):
        self._adjacency = value

    ready_event = None

    def wait_ready(self) -> "DeviceBatch":
        """A batch assembled on a side stream (tasks/resident.py) carries the event recorded behind its last kernel: make
        the CURRENT stream wait for it (no host sync) and tell the caching allocator that the batch's tensors are now
        used on this stream."""
        ev, self.ready_event = self.ready_event, None
        if ev is None:
            return self
        cur = torch.cuda.current_stream(self.initial_node_features.device)
        cur.wait_event(ev)
        tensors = [self.initial_node_features, self.type_to_num_incoming_edges, self.graph_nodes_list,
                   *([] if callable(self._adjacency) else self._adjacency),
                   *[v for v in self.extra.values() if torch.is_tensor(v)]]
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(cur)
        return self


class Sparse_Graph_Task:
    """Minimal task interface used by Sparse_Graph_Model (tasks/sparse_graph_task.py:23-254)."""

    @classmethod
    def default_params(cls):
        return {}

    @staticmethod
    def name() -> str:
        raise NotImplementedError()

    def __init__(self, params: Dict[str, Any]):
        self.params = params
        self._loaded_data = {}  # type: Dict[DataFold, Any]

    @property
    def num_edge_types(self) -> int:
        raise NotImplementedError()

    @property
    def initial_node_feature_size(self) -> int:
        raise NotImplementedError()

    def get_metadata(self) -> Dict[str, Any]:
        return {}

    def output_variable_scope(self, model_has_input_projection: bool) -> str:
        """Absolute TF variable scope of the task's output variables ("" = the graph's root scope)."""
        return ""

    # ---- native batching (tasks/batcher.py); tasks override the payload tables / post-processing ----
    NODE_PAYLOADS = {"initial_node_features": ("node_features", np.float32)}
    GRAPH_PAYLOADS: Dict[str, tuple] = {}

    def make_graph_store(self, data):
        """Flatten one data fold once for the C++ batch builder (include/relgnn.h section 9)."""
        from .batcher import GraphStore
        return GraphStore(data, self.num_edge_types, self.NODE_PAYLOADS, self.GRAPH_PAYLOADS)

    def _finish_native_batch(self, batch: "DeviceBatch") -> "DeviceBatch":
        return batch

    def make_native_minibatch_iterator(self, batcher, data_fold: DataFold, max_nodes_per_batch: int,
                                       rng: Optional[np.random.RandomState] = None):
        """Same batches as make_minibatch_iterator, assembled by relgnn_batch_pack and already on the device.
        TRAIN shuffles the graph order (the reference shuffles the data list in place, tasks/ppi_task.py:204-206)."""
        ids = np.arange(batcher.store.num_graphs)
        if data_fold == DataFold.TRAIN:
            (rng or np.random).shuffle(ids)
            batcher.constants['out_layer_dropout_keep_prob'] = self.params.get('out_layer_dropout_keep_prob', 1.0)
        else:
            batcher.constants['out_layer_dropout_keep_prob'] = 1.0
        for batch in batcher.iterate(ids, max_nodes_per_batch):
            yield self._finish_native_batch(batch)

    def restore_from_metadata(self, metadata: Dict[str, Any]) -> None:
        pass
