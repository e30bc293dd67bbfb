"""QM9 task mirror (tasks/qm9_task.py): jsonl.gz loader (:86-147), batch builder (:200-261) and the
gated-regression head with per-graph unsorted_segment_sum pooling (:150-197) — the second user of the
segment-sum kernel in the reference.

Edge types (defaults add_self_loop_edges=True, tie_fwd_bkwd_edges=True): type 0 = self loops, types 1..4 = bond
types with both directions in one list (:114-133); each graph's adjacency lists are SORTED (:135)."""
import gzip
import json
from typing import Any, Dict, Iterator, List, NamedTuple, Optional

import numpy as np
import torch

from .. import ops
from ..dense import dense
from .sparse_graph_task import DataFold, MinibatchData, Sparse_Graph_Task


class QM9GraphSample(NamedTuple):
    adjacency_lists: List[np.ndarray]
    type_to_node_to_num_incoming_edges: np.ndarray
    node_features: List[List[float]]
    target_values: List[float]


class QM9_Task(Sparse_Graph_Task):
    # tasks/qm9_task.py:22-26
    CHEMICAL_ACC_NORMALISING_FACTORS = [0.066513725, 0.012235489, 0.071939046,
                                        0.033730778, 0.033486113, 0.004278493,
                                        0.001330901, 0.004165489, 0.004128926,
                                        0.00409976, 0.004527465, 0.012292586,
                                        0.037467458]

    @classmethod
    def default_params(cls):
        params = super().default_params()
        params.update({
            'task_ids': [0],
            'add_self_loop_edges': True,
            'tie_fwd_bkwd_edges': True,
            'use_graph': True,
            'activation_function': "tanh",
            'out_layer_dropout_keep_prob': 1.0,
        })
        return params

    @staticmethod
    def name() -> str:
        return "QM9"

    @staticmethod
    def default_data_path() -> str:
        return "data/qm9"

    def __init__(self, params: Dict[str, Any]):
        super().__init__(params)
        self.__num_edge_types = 0
        self.__annotation_size = 0

    def get_metadata(self) -> Dict[str, Any]:
        return {'num_edge_types': self.__num_edge_types, 'annotation_size': self.__annotation_size}

    def restore_from_metadata(self, metadata: Dict[str, Any]) -> None:
        self.__num_edge_types = metadata['num_edge_types']
        self.__annotation_size = metadata['annotation_size']

    @property
    def num_edge_types(self) -> int:
        return self.__num_edge_types

    @property
    def initial_node_feature_size(self) -> int:
        return self.__annotation_size

    # -------------------- Data Loading --------------------
    @staticmethod
    def read_jsonl_gz(path: str, max_graphs: Optional[int] = None) -> List[dict]:
        out = []
        with gzip.open(path, "rt") as f:
            for line in f:
                out.append(json.loads(line))
                if max_graphs is not None and len(out) >= max_graphs:
                    break
        return out

    def load_data(self, path: str, max_graphs: Optional[int] = None) -> None:
        import os
        for fold, fname in ((DataFold.TRAIN, "train.jsonl.gz"), (DataFold.VALIDATION, "valid.jsonl.gz")):
            p = os.path.join(path, fname)
            if os.path.exists(p):
                self._loaded_data[fold] = self.load_raw(self.read_jsonl_gz(p, max_graphs))

    def load_eval_data_from_path(self, path: str):
        return self.load_raw(self.read_jsonl_gz(path))

    def load_raw(self, data: List[dict]) -> List[QM9GraphSample]:
        """tasks/qm9_task.py:86-112."""
        num_fwd_edge_types = 0
        for g in data:
            num_fwd_edge_types = max(num_fwd_edge_types, max([e[1] for e in g['graph']]))
        if self.params['add_self_loop_edges']:
            num_fwd_edge_types += 1
        self.__num_edge_types = max(self.__num_edge_types,
                                    num_fwd_edge_types * (1 if self.params['tie_fwd_bkwd_edges'] else 2))
        self.__annotation_size = max(self.__annotation_size, len(data[0]["node_features"][0]))
        return [QM9GraphSample(*self._graph_to_adjacency_lists(d['graph'], len(d["node_features"])),
                               node_features=d["node_features"],
                               target_values=[d["targets"][task_id][0] for task_id in self.params['task_ids']])
                for d in data]

    def _graph_to_adjacency_lists(self, graph, num_nodes: int):
        """tasks/qm9_task.py:114-147: raw triples (src, bond type e in 1..4, dst)."""
        L = self.__num_edge_types
        lists = [[] for _ in range(L)]
        deg = np.zeros((L, num_nodes))
        for src, e, dest in graph:
            t = e if self.params['add_self_loop_edges'] else e - 1
            lists[t].append((src, dest))
            deg[t, dest] += 1
            if self.params['tie_fwd_bkwd_edges']:
                lists[t].append((dest, src))
                deg[t, src] += 1
        if self.params['add_self_loop_edges']:
            for node in range(num_nodes):
                deg[0, node] = 1
                lists[0].append((node, node))
        adj = [np.array(sorted(a), dtype=np.int32) if len(a) > 0 else np.zeros((0, 2), dtype=np.int32) for a in lists]
        if not self.params['tie_fwd_bkwd_edges']:
            half = L // 2
            adj = adj[:half]
            for t, a in enumerate(list(adj)):
                adj.append(np.array(sorted((y, x) for (x, y) in a), dtype=np.int32).reshape(-1, 2))
                for (x, y) in a:
                    # Reference behaviour kept bit for bit (tasks/qm9_task.py:144-145): the count goes to y, the target
                    # of the FORWARD edge, although the reversed edge (y -> x) lands on x.  So with
                    # tie_fwd_bkwd_edges=False the backward types' table is NOT the true in-degree of their adjacency
                    # lists; it only reaches layers that normalise by it (RGCN default, FiLM/Edge-MLP when asked).
                    deg[half + t][y] += 1
        return adj, deg

    # -------------------- Output head (tasks/qm9_task.py:150-197) --------------------
    def output_variables(self, hidden_size: int):
        specs = {}
        for task_id in self.params['task_ids']:
            s = "out_layer_task%i" % task_id
            specs[s + "/regression_gate/dense/kernel"] = ((hidden_size + self.__annotation_size, 1), "glorot_uniform")
            specs[s + "/regression_gate/dense/bias"] = ((1,), "zeros")
            specs[s + "/regression/dense/kernel"] = ((hidden_size, 1), "glorot_uniform")
            specs[s + "/regression/dense/bias"] = ((1,), "zeros")
        return specs

    def compute_task_metrics(self, final_node_representations: torch.Tensor, batch, weights) -> Dict[str, torch.Tensor]:
        metrics = {}
        losses = []
        num_graphs = batch.num_graphs
        targets = batch.extra['target_values']                                   # [tasks, G]
        for internal_id, task_id in enumerate(self.params['task_ids']):
            w = weights.scope("out_layer_task%i" % task_id) if hasattr(weights, "scope") else weights
            # (dense(): the [hidden, 1] weight gradients go through the streaming kernel — as plain `@` autograd handed
            # them to the library as [V, hidden]^T @ [V, 1] products, 191 us each on a 50 k-node batch)
            per_node_outputs = dense(final_node_representations, w["regression/dense/kernel"], w["regression/dense/bias"])
            gate_input = torch.cat([final_node_representations, batch.initial_node_features], dim=-1)
            gate = torch.sigmoid(dense(gate_input, w["regression_gate/dense/kernel"], w["regression_gate/dense/bias"]))
            per_node_gated_outputs = gate * per_node_outputs
            # Sum up all nodes per graph: the HIP segment-sum kernel (2nd call-site family, :185-187)
            per_graph_outputs = ops.unsorted_segment_sum(per_node_gated_outputs, batch.graph_nodes_list, num_graphs).squeeze(-1)
            per_graph_errors = per_graph_outputs - targets[internal_id, :]
            metrics['abs_err_task%i' % task_id] = per_graph_errors.abs().sum()
            losses.append((0.5 * per_graph_errors ** 2).mean())
        metrics['loss'] = torch.stack(losses).sum()
        metrics['total_loss'] = metrics['loss'] * float(num_graphs)
        return metrics

    NODE_PAYLOADS = {"initial_node_features": ("node_features", np.float32)}
    GRAPH_PAYLOADS = {"target_values": ("target_values", np.float32)}

    def _finish_native_batch(self, batch):
        batch.extra['target_values'] = batch.extra['target_values'].t()          # [tasks, G] (:254)
        return batch

    # -------------------- Minibatching (tasks/qm9_task.py:200-261) --------------------
    def make_minibatch_iterator(self, data: List[QM9GraphSample], data_fold: DataFold, max_nodes_per_batch: int,
                                rng: Optional[np.random.RandomState] = None) -> Iterator[MinibatchData]:
        if data_fold == DataFold.TRAIN:
            (rng or np.random).shuffle(data)
            out_keep = self.params['out_layer_dropout_keep_prob']
        else:
            out_keep = 1.0
        i = 0
        while i < len(data):
            start, node_offset, offsets = i, 0, []
            while i < len(data) and node_offset + len(data[i].node_features) < max_nodes_per_batch:
                offsets.append(node_offset)
                node_offset += len(data[i].node_features)
                i += 1
            if i == start:
                raise ValueError("graph %d does not fit max_nodes_per_batch=%d" % (i, max_nodes_per_batch))
            chunk = data[start:i]
            adjacency, num_edges = [], 0
            for l in range(self.num_edge_types):
                a = np.concatenate([np.asarray(g.adjacency_lists[l]).reshape(-1, 2) + off for g, off in zip(chunk, offsets)])
                a = a.astype(np.int32) if a.shape[0] else np.zeros((0, 2), dtype=np.int32)
                num_edges += a.shape[0]
                adjacency.append(a)
            feed = {
                'initial_node_features': np.concatenate([np.asarray(g.node_features, dtype=np.float32) for g in chunk], axis=0),
                'type_to_num_incoming_edges': np.concatenate([g.type_to_node_to_num_incoming_edges for g in chunk], axis=1),
                'graph_nodes_list': np.concatenate([np.full([len(g.node_features)], k, dtype=np.int32)
                                                    for k, g in enumerate(chunk)]),
                'target_values': np.transpose(np.array([g.target_values for g in chunk], dtype=np.float32), axes=[1, 0]),
                'out_layer_dropout_keep_prob': out_keep,
                'adjacency_lists': adjacency,
            }
            yield MinibatchData(feed_dict=feed, num_graphs=len(chunk), num_nodes=node_offset, num_edges=num_edges)

    def early_stopping_metric(self, task_metric_results, num_graphs: int) -> float:
        return float(np.sum([float(m['total_loss']) for m in task_metric_results]) / num_graphs)

    def pretty_print_epoch_task_metrics(self, task_metric_results, num_graphs: int) -> str:
        maes = {t: 0.0 for t in self.params['task_ids']}
        for r in task_metric_results:
            for t in self.params['task_ids']:
                maes[t] += float(r['abs_err_task%i' % t]) / float(num_graphs)
        maes_str = " ".join("%i:%.5f" % (t, maes[t]) for t in self.params['task_ids'])
        err_str = " ".join("%i:%.5f" % (t, maes[t] / self.CHEMICAL_ACC_NORMALISING_FACTORS[t]) for t in self.params['task_ids'])
        return "MAEs: %s | Error Ratios: %s" % (maes_str, err_str)
