"""PPI task mirror: data layout, batch builder, output head (tasks/ppi_task.py).

In scope for the hot path: the BATCH BUILDER (index bookkeeping, tasks/ppi_task.py:197-256), which
must be bit-exact, and enough of the output head (:165-194) to run the reference's training
step around the path.  The DGL-PPI file loader (:76-162) is provided for completeness; without
the dataset (none on the box) `load_synthetic` fills the folds with PPI-shaped graphs.
"""
import json
import os
from typing import Any, Dict, Iterator, List, Optional

import numpy as np
import torch

from .. import config as _cfg
from ..dense import dense
from ..utils import micro_f1
from .sparse_graph_task import DataFold, MinibatchData, Sparse_Graph_Task
from .synthetic import GraphSample, make_ppi_shaped_graphs


class _SigmoidCEStats(torch.autograd.Function):
    """(mean loss, summed loss, micro-F1) of one batch in one pass over the logits (csrc/train_utils.hip: sum of sigmoid-CE
    losses, true_pos, false_pos, false_neg, F1).  Separate outputs instead of one indexed stats vector: indexing, dividing
    and their autograd mirrors were five ~5 us kernels per step around two real ones."""

    @staticmethod
    def forward(ctx, logits, labels, inv_n: float):
        from .. import _lib
        lib = _lib.load_library()
        logits, labels = logits.contiguous(), labels.contiguous()
        stats = torch.empty(6, dtype=torch.float32, device=logits.device)
        nbytes = lib.relgnn_sigmoid_ce_stats_workspace_bytes()
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=logits.device)
        _lib.check(lib.relgnn_sigmoid_ce_stats(_lib.ptr(logits), _lib.ptr(labels), logits.numel(), float(inv_n),
                                               _lib.ptr(stats), _lib.ptr(ws), nbytes, _lib.current_stream()),
                   "relgnn_sigmoid_ce_stats")
        ctx.save_for_backward(logits, labels)
        ctx.inv_n = float(inv_n)
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None, not as a zero tensor
        mean, total, f1 = stats[5], stats[0], stats[4]        # (the mean is the kernel's own float32 product)
        counts = stats[1:4]                                   # true_pos, false_pos, false_neg
        ctx.mark_non_differentiable(f1, counts)
        return mean, total, f1, counts

    @staticmethod
    def backward(ctx, g_mean, g_total, g_f1, g_counts):
        from .. import _lib
        lib = _lib.load_library()
        logits, labels = ctx.saved_tensors
        if g_mean is None and g_total is None:
            return None, None, None

        def scalar(g):                            # the incoming gradients stay on the device: the kernel reads them
            return None if g is None else g.reshape(1).to(torch.float32).contiguous()
        g_mean, g_total = scalar(g_mean), scalar(g_total)
        rows, cols = logits.shape if logits.dim() == 2 else (1, logits.numel())
        if logits.dim() == 2 and cols % 16 and _cfg.settings.limb_gemm and _cfg.settings.head_pad == "1":
            # rows of the next multiple of 16 floats, zeros behind the labels: the head's input-gradient product then runs on the limb
            # route with the ReLU' of the last GNN layer in its epilogue (dense.mark_zero_padded) instead of a K = 121 library
            # product + a pass over [V, 256]
            from ..dense import mark_zero_padded
            ld = (cols + 15) // 16 * 16
            buf = torch.empty((rows, ld), dtype=torch.float32, device=logits.device)
            _lib.check(lib.relgnn_sigmoid_ce_bwd_padded(_lib.ptr(logits), _lib.ptr(labels), rows, cols, _lib.ptr(g_mean), ctx.inv_n,
                                                        _lib.ptr(g_total), _lib.ptr(buf), ld, _lib.current_stream()),
                       "relgnn_sigmoid_ce_bwd_padded")
            return mark_zero_padded(buf[:, :cols], ld), None, None
        gl = torch.empty_like(logits)
        _lib.check(lib.relgnn_sigmoid_ce_bwd(_lib.ptr(logits), _lib.ptr(labels), logits.numel(), _lib.ptr(g_mean), ctx.inv_n,
                                             _lib.ptr(g_total), _lib.ptr(gl), _lib.current_stream()), "relgnn_sigmoid_ce_bwd")
        return gl, None, None


class PPI_Task(Sparse_Graph_Task):
    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'add_self_loop_edges': True,
            'tie_fwd_bkwd_edges': False,
            'out_layer_dropout_keep_prob': 1.0,
        })
        return params

    @staticmethod
    def name() -> str:
        return "PPI"

    @staticmethod
    def default_data_path() -> str:
        return "data/ppi"

    def __init__(self, params: Dict[str, Any]):
        super().__init__(params)
        self.__num_edge_types = 3
        self.__initial_node_feature_size = 0
        self.__num_labels = 0

    def get_metadata(self) -> Dict[str, Any]:
        return {'num_edge_types': self.__num_edge_types,
                'initial_node_feature_size': self.__initial_node_feature_size,
                'num_labels': self.__num_labels}

    def restore_from_metadata(self, metadata: Dict[str, Any]) -> None:
        self.__num_edge_types = metadata['num_edge_types']
        self.__initial_node_feature_size = metadata['initial_node_feature_size']
        self.__num_labels = metadata['num_labels']

    @property
    def num_edge_types(self) -> int:
        return self.__num_edge_types

    @property
    def initial_node_feature_size(self) -> int:
        return self.__initial_node_feature_size

    @property
    def num_labels(self) -> int:
        return self.__num_labels

    # -------------------- Data --------------------
    def _edge_type_layout(self):
        # tasks/ppi_task.py:99-106
        n = 1
        self_loop = bkwd = None
        if self.params['add_self_loop_edges']:
            self_loop = n
            n += 1
        if not self.params['tie_fwd_bkwd_edges']:
            bkwd = n
            n += 1
        return 0, self_loop, bkwd, n

    def load_synthetic(self, num_train_graphs: int = 16, num_valid_graphs: int = 2, seed: int = 0, **gen) -> None:
        """PPI-shaped stand-in for load_data (no dataset on the box): default task params only."""
        if not self.params['add_self_loop_edges'] or self.params['tie_fwd_bkwd_edges']:
            raise ValueError("synthetic PPI-shaped data uses the default [fwd, self_loop, bkwd] edge types")
        self._loaded_data[DataFold.TRAIN] = make_ppi_shaped_graphs(num_train_graphs, seed=seed, **gen)
        self._loaded_data[DataFold.VALIDATION] = make_ppi_shaped_graphs(num_valid_graphs, seed=seed + 1, **gen)
        g = self._loaded_data[DataFold.TRAIN][0]
        self.__num_edge_types = len(g.adjacency_lists)
        self.__initial_node_feature_size = g.node_features.shape[1]
        self.__num_labels = g.node_labels.shape[1]

    def load_data(self, path: str) -> None:
        self._loaded_data[DataFold.TRAIN] = self._load_fold(path, "train")
        self._loaded_data[DataFold.VALIDATION] = self._load_fold(path, "valid")

    def load_eval_data_from_path(self, path: str):
        return self._load_fold(path, "test")

    def _load_fold(self, data_dir: str, data_name: str) -> List[GraphSample]:
        """DGL ppi.zip layout ({fold}_graph.json, _feats.npy, _labels.npy, _graph_id.npy),
        semantics of tasks/ppi_task.py:76-162, vectorised: per graph, node ids are shifted to start
        at 0, edge order is the file's link order, self loops ascend by node id."""
        with open(os.path.join(data_dir, "%s_graph.json" % data_name)) as f:
            graph_json = json.load(f)
        feats = np.load(os.path.join(data_dir, "%s_feats.npy" % data_name))
        labels = np.load(os.path.join(data_dir, "%s_labels.npy" % data_name))
        graph_id = np.load(os.path.join(data_dir, "%s_graph_id.npy" % data_name))
        self.__initial_node_feature_size = feats.shape[-1]
        self.__num_labels = labels.shape[-1]
        fwd_t, self_t, bkwd_t, n_types = self._edge_type_layout()
        self.__num_edge_types = n_types

        # graphs in order of first appearance; node offset = first node id of the graph
        order, first = [], {}
        for node_id, gid in enumerate(graph_id):
            if gid not in first:
                first[gid] = node_id
                order.append(gid)
        links = np.array([(e['source'], e['target']) for e in graph_json['links']], dtype=np.int64).reshape(-1, 2)
        link_gid = graph_id[links[:, 0]] if len(links) else np.zeros(0, graph_id.dtype)
        graphs = []
        for gid in order:
            nodes = np.nonzero(graph_id == gid)[0]
            n = len(nodes)
            off = first[gid]
            e = links[link_gid == gid] - off
            adj = [None] * n_types
            adj[fwd_t] = e.astype(np.int64)
            if self_t is not None:
                adj[self_t] = np.stack([np.arange(n), np.arange(n)], axis=1)
            if bkwd_t is not None:
                adj[bkwd_t] = np.ascontiguousarray(e[:, ::-1]).astype(np.int64)
            deg = np.stack([np.bincount(a[:, 1], minlength=n) if len(a) else np.zeros(n, np.int64) for a in adj])
            graphs.append(GraphSample(adjacency_lists=adj, type_to_node_to_num_incoming_edges=deg,
                                      node_features=feats[nodes], node_labels=labels[nodes]))
        return graphs

    # -------------------- Output head (tasks/ppi_task.py:165-194) --------------------
    def output_variable_scope(self, model_has_input_projection: bool) -> str:
        # tasks/ppi_task.py:176-179: an unnamed tf.keras.layers.Dense; Keras numbers unnamed layers per graph, and the
        # model's unnamed input projection (models/sparse_graph_model.py:165-170) already took "dense"
        return "dense_1" if model_has_input_projection else "dense"

    def output_variables(self, hidden_size: int):
        # unnamed Keras Dense with bias (:176-179); TF auto-names it after the model's input projection
        return {"kernel": ((hidden_size, self.__num_labels), "glorot_uniform"), "bias": ((self.__num_labels,), "zeros")}

    def compute_task_metrics(self, final_node_representations: torch.Tensor, batch, weights) -> Dict[str, torch.Tensor]:
        labels = batch.extra['target_labels']
        per_node_logits = dense(final_node_representations, weights["kernel"], weights["bias"])
        num_nodes_in_batch = labels.shape[0]
        if per_node_logits.is_cuda:
            # loss sum + F1 counts + F1: one pass
            loss, total_loss, f1, _ = _SigmoidCEStats.apply(per_node_logits, labels, 1.0 / float(num_nodes_in_batch))
            return {'loss': loss, 'total_loss': total_loss, 'f1_score': f1}
        else:
            total_loss = torch.nn.functional.binary_cross_entropy_with_logits(per_node_logits, labels, reduction='sum')
            f1 = micro_f1(per_node_logits.detach(), labels)
        return {'loss': total_loss / float(num_nodes_in_batch), 'total_loss': total_loss, 'f1_score': f1}

    NODE_PAYLOADS = {"initial_node_features": ("node_features", np.float32), "target_labels": ("node_labels", np.float32)}

    # -------------------- Minibatching (tasks/ppi_task.py:197-256) --------------------
    def make_minibatch_iterator(self, data: List[GraphSample], data_fold: DataFold,
                                max_nodes_per_batch: int, rng: Optional[np.random.RandomState] = None
                                ) -> Iterator[MinibatchData]:
        """Disjoint-union packing, same semantics as the reference: graphs are taken in order while
        node_offset + |V_g| < max_nodes_per_batch (strict), adjacency lists shifted by the node offset,
        degree tables concatenated along the node axis, empty edge types -> zeros((0, 2), int32)."""
        if data_fold == DataFold.TRAIN:
            (rng or np.random).shuffle(data)
            out_keep = self.params['out_layer_dropout_keep_prob']
        else:
            out_keep = 1.0
        i = 0
        while i < len(data):
            start = i
            node_offset = 0
            offsets = []
            while i < len(data) and node_offset + len(data[i].node_features) < max_nodes_per_batch:
                offsets.append(node_offset)
                node_offset += len(data[i].node_features)
                i += 1
            if i == start:
                raise ValueError("graph %d (%d nodes) does not fit max_nodes_per_batch=%d"
                                 % (i, len(data[i].node_features), max_nodes_per_batch))
            chunk = data[start:i]
            adjacency = []
            num_edges = 0
            for l in range(self.num_edge_types):
                parts = [np.asarray(g.adjacency_lists[l]).reshape(-1, 2) + off for g, off in zip(chunk, offsets)]
                a = np.concatenate(parts).astype(np.int32) if parts else np.zeros((0, 2), np.int32)
                if a.shape[0] == 0:
                    a = np.zeros((0, 2), dtype=np.int32)
                num_edges += a.shape[0]
                adjacency.append(a)
            feed = {
                'initial_node_features': np.concatenate([np.asarray(g.node_features) for g in chunk], axis=0),
                'type_to_num_incoming_edges': np.concatenate(
                    [np.asarray(g.type_to_node_to_num_incoming_edges) for g in chunk], axis=1),
                'graph_nodes_list': np.concatenate(
                    [np.full([len(g.node_features)], k, dtype=np.int32) for k, g in enumerate(chunk)]),
                'target_labels': np.concatenate([np.asarray(g.node_labels) for g in chunk], axis=0).astype(np.float32),
                'out_layer_dropout_keep_prob': out_keep,
                'adjacency_lists': adjacency,
            }
            yield MinibatchData(feed_dict=feed, num_graphs=len(chunk), num_nodes=node_offset, num_edges=num_edges)

    def early_stopping_metric(self, task_metric_results: List[Dict[str, Any]], num_graphs: int) -> float:
        return float(np.sum([float(m['total_loss']) for m in task_metric_results]) / num_graphs)

    def pretty_print_epoch_task_metrics(self, task_metric_results: List[Dict[str, Any]], num_graphs: int) -> str:
        return "Avg MicroF1: %.3f" % (float(np.average([float(m['f1_score']) for m in task_metric_results])),)
