"""Model driver mirror (models/sparse_graph_model.py of the reference).

Hot-path scope: the per-layer loop of __build_graph_propagation_model
(:162-202) and the abstract hook _apply_gnn_layer (:204-225), same names and argument meaning.
The training-step plumbing around it (optimizer with per-variable clip_by_norm :227-260, epoch
loop with throughput counters :263-311) is restated minimally so that the reference's own
"edges/sec" number (README.md:34-35) can be measured end to end.

TF1 builds a static graph once and feeds numpy batches through sess.run; here the variables
live in a VariableStore under the reference's TF names, batches live in HBM (DeviceBatch), and a
step is eager PyTorch-ROCm around the librelgnn HIP kernels.
"""
import os
import pickle
import time
from abc import ABC, abstractmethod
from typing import Any, Dict, Iterable, List, Optional

import numpy as np
import torch

from ..dense import dense
from ..graph import as_rel_graph, check_pending_graph_errors
from ..tasks import DataFold, DeviceBatch, Sparse_Graph_Task
from ..utils import apply_activation, get_activation, layer_norm, layer_norm_scope
from ..variables import VariableStore


class TFStyleOptimizer:
    """compute_gradients -> per-variable tf.clip_by_norm -> apply_gradients
    (models/sparse_graph_model.py:227-260) with TF1 update rules [TF-internal]:
      Adam    : lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; var -= lr_t*m/(sqrt(v)+eps), eps=1e-8
      RMSProp : ms = d*ms+(1-d)*g^2; mom = momentum*mom + lr*g/sqrt(ms+eps); var -= mom, eps=1e-10, ms0=1
      SGD     : var -= lr*g
    """

    def __init__(self, params: List[torch.nn.Parameter], name: str, learning_rate: float, clamp_gradient_norm: float,
                 decay: float = 0.98, momentum: float = 0.85):
        self.params = [p for p in params if p.requires_grad]
        self.name = name.lower()
        if self.name not in ('sgd', 'rmsprop', 'adam'):
            raise Exception('Unknown optimizer "%s".' % name)
        self.lr, self.clip, self.decay, self.momentum = learning_rate, clamp_gradient_norm, decay, momentum
        self.t = 0
        if self.name == 'adam':
            self.m = [torch.zeros_like(p) for p in self.params]
            self.v = [torch.zeros_like(p) for p in self.params]
        elif self.name == 'rmsprop':
            self.ms = [torch.ones_like(p) for p in self.params]
            self.mom = [torch.zeros_like(p) for p in self.params]

    def _fused_adam_available(self):
        return self.name == 'adam' and len(self.params) > 0 and all(p.is_cuda for p in self.params)

    @torch.no_grad()
    def clip_and_step(self, lr_scale: float = 1.0, device_step_count: bool = False):
        """clip_by_norm per variable + update.  Adam on the GPU: two fused multi-tensor HIP launches
        (relgnn_mt_l2norm, relgnn_mt_adam_clip) instead of ~30 elementwise kernels.
        device_step_count=True (hipGraph capture): the step count and lr_t live in device memory
        (relgnn_adam_step_size / relgnn_mt_adam_clip_devlr); the caller keeps self.t in step."""
        if not self._fused_adam_available():
            if device_step_count:
                raise RuntimeError("a captured training step needs the fused device-side Adam update")
            self.clip_gradients()
            self.step(lr_scale)
            return
        import ctypes
        from .. import _lib
        lib = _lib.load_library()
        st = _lib.current_stream()
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return
        b1, b2, eps = 0.9, 0.999, 1e-8
        dev_state = None
        if device_step_count:
            dev_state = self._device_state()
            _lib.check(lib.relgnn_adam_step_size(_lib.ptr(dev_state), self.lr * lr_scale, b1, b2, st), "relgnn_adam_step_size")
        else:
            self.t += 1
            lr_t = self.lr * lr_scale * (1 - b2 ** self.t) ** 0.5 / (1 - b1 ** self.t)
        for c0 in range(0, len(idx), _lib.MT_MAX):
            chunk = idx[c0:c0 + _lib.MT_MAX]
            n = len(chunk)
            grads = [self.params[i].grad if self.params[i].grad.is_contiguous() else self.params[i].grad.contiguous()
                     for i in chunk]
            arr = ctypes.c_void_p * n
            h_g = arr(*[g.data_ptr() for g in grads])
            h_p = arr(*[self.params[i].data_ptr() for i in chunk])
            h_m = arr(*[self.m[i].data_ptr() for i in chunk])
            h_v = arr(*[self.v[i].data_ptr() for i in chunk])
            h_n = (ctypes.c_int64 * n)(*[self.params[i].numel() for i in chunk])
            norms = torch.empty(n, dtype=torch.float32, device=self.params[chunk[0]].device)
            ws_bytes = lib.relgnn_mt_l2norm_workspace_bytes()
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=norms.device)
            _lib.check(lib.relgnn_mt_l2norm(h_g, h_n, n, _lib.ptr(norms), _lib.ptr(ws), ws_bytes, st), "relgnn_mt_l2norm")
            if dev_state is not None:
                _lib.check(lib.relgnn_mt_adam_clip_devlr(h_p, h_g, h_m, h_v, h_n, n, _lib.ptr(norms), float(self.clip),
                                                         dev_state[1:].data_ptr(), b1, b2, eps, st), "relgnn_mt_adam_clip_devlr")
            else:
                _lib.check(lib.relgnn_mt_adam_clip(h_p, h_g, h_m, h_v, h_n, n, _lib.ptr(norms), float(self.clip), lr_t,
                                                   b1, b2, eps, st), "relgnn_mt_adam_clip")
        from ..dense import weights_changed
        weights_changed()                  # (written through raw pointers: the tensors' version counters did not move)

    def _device_state(self):
        """[steps taken, lr_t] in device memory, seeded from the host step count."""
        if getattr(self, "_dev_state", None) is None:
            self._dev_state = torch.tensor([float(self.t), 0.0], dtype=torch.float32, device=self.params[0].device)
        return self._dev_state

    def sync_device_step_count(self):
        """Call before switching to device-side counting (the host count may have advanced since the last time)."""
        if getattr(self, "_dev_state", None) is not None:
            self._dev_state[0] = float(self.t)

    @torch.no_grad()
    def clip_gradients(self):
        """tf.clip_by_norm per variable: g * clip / max(||g||, clip)."""
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        norms = torch.stack(torch._foreach_norm(grads))
        scales = self.clip / torch.clamp(norms, min=self.clip)       # == 1 where ||g|| <= clip
        torch._foreach_mul_(grads, list(scales.unbind(0)))

    @torch.no_grad()
    def step(self, lr_scale: float = 1.0):
        ps = [p for p in self.params if p.grad is not None]
        gs = [p.grad for p in ps]
        if not ps:
            return
        lr = self.lr * lr_scale
        self.t += 1
        if self.name == 'sgd':
            torch._foreach_add_(ps, gs, alpha=-lr)
        elif self.name == 'adam':
            b1, b2, eps = 0.9, 0.999, 1e-8
            idx = [i for i, p in enumerate(self.params) if p.grad is not None]
            m = [self.m[i] for i in idx]
            v = [self.v[i] for i in idx]
            torch._foreach_mul_(m, b1)
            torch._foreach_add_(m, gs, alpha=1 - b1)
            torch._foreach_mul_(v, b2)
            torch._foreach_addcmul_(v, gs, gs, value=1 - b2)
            lr_t = lr * (1 - b2 ** self.t) ** 0.5 / (1 - b1 ** self.t)
            denom = torch._foreach_sqrt(v)
            torch._foreach_add_(denom, eps)
            torch._foreach_addcdiv_(ps, m, denom, value=-lr_t)
        else:
            eps = 1e-10
            idx = [i for i, p in enumerate(self.params) if p.grad is not None]
            ms = [self.ms[i] for i in idx]
            mom = [self.mom[i] for i in idx]
            torch._foreach_mul_(ms, self.decay)
            torch._foreach_addcmul_(ms, gs, gs, value=1 - self.decay)
            denom = torch._foreach_add(ms, eps)
            torch._foreach_sqrt_(denom)
            torch._foreach_mul_(mom, self.momentum)
            torch._foreach_addcdiv_(mom, gs, denom, value=lr)
            torch._foreach_sub_(ps, mom)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    # ---- optimizer state under TF's GLOBAL_VARIABLES names (the reference pickles those too, :91-107) ----
    def _slots(self):
        """(TF slot suffix, tensors) per optimizer [TF-internal slot names: Adam -> '<var>/Adam', '<var>/Adam_1';
        RMSProp -> '<var>/RMSProp' (mean square), '<var>/RMSProp_1' (momentum)]."""
        if self.name == 'adam':
            return [("Adam", self.m), ("Adam_1", self.v)]
        if self.name == 'rmsprop':
            return [("RMSProp", self.ms), ("RMSProp_1", self.mom)]
        return []

    def tf_slot_weights(self, names: List[str]) -> Dict[str, np.ndarray]:
        assert len(names) == len(self.params)
        out = {}
        for suffix, tensors in self._slots():
            for n, t in zip(names, tensors):
                out["%s/%s:0" % (n, suffix)] = t.detach().cpu().numpy().copy()
        if self.name == 'adam':      # TF 1.13 keeps beta^(t+1) in two non-trainable scalars
            out["beta1_power:0"] = np.float32(0.9 ** (self.t + 1))
            out["beta2_power:0"] = np.float32(0.999 ** (self.t + 1))
        return out

    @torch.no_grad()
    def load_tf_slots(self, weights, names: List[str]) -> set:
        """Restore the slots a reference pickle carries; returns the consumed keys.  The step count is recovered from
        beta1_power (= 0.9^(t+1))."""
        used = set()
        for suffix, tensors in self._slots():
            for n, t in zip(names, tensors):
                key = "%s/%s:0" % (n, suffix)
                if key in weights:
                    t.copy_(torch.as_tensor(np.asarray(weights[key]), dtype=torch.float32))
                    used.add(key)
        if self.name == 'adam' and "beta1_power:0" in weights:
            self.t = self._steps_from_beta_powers(weights.get("beta1_power:0"), weights.get("beta2_power:0"))
            used.update(k for k in ("beta1_power:0", "beta2_power:0") if k in weights)
        return used

    @staticmethod
    def _steps_from_beta_powers(beta1_power, beta2_power) -> int:
        """Step count t from TF's float32 scalars beta1_power = 0.9^(t+1), beta2_power = 0.999^(t+1).  0.9^(t+1) leaves the
        normal float32 range after ~830 steps and is exactly 0.0 after ~985, so it identifies t only for short runs; past
        that 0.999^(t+1) does (normal up to ~8.7e4 steps); when both have underflowed every bias correction is 1 to fp32
        precision and any large t reproduces the update rule."""
        import math
        tiny = float(np.finfo(np.float32).tiny)
        for power, beta in ((beta1_power, 0.9), (beta2_power, 0.999)):
            if power is None:
                continue
            p = float(np.asarray(power).reshape(-1)[0])
            if math.isfinite(p) and tiny <= p < 1.0:
                return max(0, int(round(math.log(p) / math.log(beta))) - 1)
            if p >= 1.0:
                return 0
        return 10 ** 6


class MetricsReadback:
    """The metric scalars of ONE step on their way to the host: packed into one device vector and copied into pinned
    memory by an asynchronous D2H enqueued right behind the step's own kernels.  get() waits for THAT copy only.

    float(tensor) / .item() is a stream-ordered copy on the current stream: called one step late it still waits for
    everything enqueued since — the whole NEXT step — and the GPU then idles until the host has come back and enqueued
    new work (measured: a ~0.2 ms bubble per 2.9 ms C2 step).  The reference has the same round trip once per
    sess.run (models/sparse_graph_model.py:293)."""
    _ring: Dict[Any, list] = {}

    def __init__(self, metrics: Dict[str, Any]):
        self._host_values = {k: v for k, v in metrics.items() if not (torch.is_tensor(v) and v.is_cuda)}
        dev = [(k, v.detach()) for k, v in metrics.items() if torch.is_tensor(v) and v.is_cuda]
        self._names = [k for k, _ in dev]
        self._event = self._slot = None
        self._status_at = None
        if dev:
            device = dev[0][1].device
            # one small kernel when the metrics share a dtype (the usual case: fp32 scalars), converted only if they differ
            dtype = dev[0][1].dtype if all(v.dtype == dev[0][1].dtype for _, v in dev) else torch.float64
            parts = [v.reshape(()) if v.dtype == dtype else v.reshape(()).to(dtype) for _, v in dev]
            # The hand-over status word of the device (ops.handover_word: a wave-role product kernel that gave up on an LDS
            # hand-over has written wrong numbers and says so there) rides in the same copy: its int32 bits as one more fp32
            # entry of the packed vector (a view, no launch of its own); get() raises when it is not zero.
            from .. import ops as _ops
            word = _ops._HANDOVER_WORDS.get(device.index)
            if word is not None:
                if dtype == torch.float32:
                    parts.append(word[0].view(torch.float32))
                else:
                    parts.append(word[0].to(dtype))
                self._status_at = len(dev)
            packed = torch.stack(parts)
            ring = MetricsReadback._ring.setdefault((device, len(parts), dtype), [])
            # a pinned slot is taken until its reader has consumed it (get()) or dropped it
            slot = next((b for b in ring if not b[1]), None)
            if slot is None:
                slot = [torch.empty(len(parts), dtype=dtype).pin_memory(), False]
                ring.append(slot)
            slot[1] = True
            slot[0].copy_(packed, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(device))
            self._slot = slot
            self._device = device

    def get(self) -> Dict[str, float]:
        out = {k: (float(v) if torch.is_tensor(v) else v) for k, v in self._host_values.items()}
        if self._event is not None:
            self._event.synchronize()
            host = self._slot[0]
            status = 0
            if self._status_at is not None:
                status = (int(host[self._status_at:self._status_at + 1].view(torch.int32)[0]) if host.dtype == torch.float32
                          else int(host[self._status_at]))
            out.update(zip(self._names, host.tolist()))
            self._event = None
            self._slot[1] = False
            self._values = {k: out[k] for k in self._names}
            if status:
                from .. import ops as _ops
                _ops._HANDOVER_WORDS[self._device.index][0].zero_()      # reported once; later steps start clean
                _ops.raise_on_handover(status)
        elif self._names:
            out.update(self._values)
        return out

    def __del__(self):
        slot = getattr(self, "_slot", None)
        if slot is not None and getattr(self, "_event", None) is not None:
            try:
                self._event.synchronize()          # the copy must not land in a slot somebody else has taken
            except Exception:
                pass
            slot[1] = False


class CapturedTrainStep:
    """A training step recorded as a hipGraph on one fixed batch (Sparse_Graph_Model.capture_train_step)."""

    def __init__(self, model, graph, metrics, batch):
        self.model, self.graph, self.metrics, self.batch = model, graph, metrics, batch
        self._expected_t = model.optimizer.t

    def replay(self) -> Dict[str, torch.Tensor]:
        """One more training step: a single graph launch.  The returned tensors are overwritten by the next replay.
        The graph reads the Adam step count from device memory; if the host count moved since the last replay (an eager
        train_step(), load_weights()), the device copy is re-seeded first, so lr_t never comes from a stale count.
        (lr_t of a replayed step is computed in float32 on the device, of an eager step in double on the host: the two
        agree to ~1e-7 relative, not bit for bit.)"""
        opt = self.model.optimizer
        if opt.t != self._expected_t:
            opt.sync_device_step_count()
        self.graph.replay()
        from ..dense import weights_changed
        weights_changed()                  # the replayed update rewrote the parameters: limb images kept by eager code are stale
        opt.t += 1
        self._expected_t = opt.t
        return self.metrics

    def handover_status(self) -> int:
        """The device's hand-over status word (ops.handover_status: synchronises).  A replay loop that reads the metric tensors
        itself instead of going through MetricsReadback checks this where it syncs anyway."""
        from .. import ops as _ops
        return _ops.handover_status(device=self.model.device)


class Sparse_Graph_Model(ABC):
    """Abstract superclass of all graph models (reference docstring: models/sparse_graph_model.py:16-20)."""

    @classmethod
    def default_params(cls):
        # models/sparse_graph_model.py:22-45
        return {
            'max_nodes_in_batch': 50000,
            'graph_num_layers': 8,
            'graph_num_timesteps_per_layer': 1,
            'graph_layer_input_dropout_keep_prob': 0.8,
            'graph_dense_between_every_num_gnn_layers': 1,
            'graph_model_activation_function': 'tanh',
            'graph_residual_connection_every_num_layers': 2,
            'graph_inter_layer_norm': False,
            'max_epochs': 10000,
            'patience': 25,
            'optimizer': 'Adam',
            'learning_rate': 0.001,
            'learning_rate_decay': 0.98,
            'lr_for_num_graphs_per_batch': None,
            'momentum': 0.85,
            'clamp_gradient_norm': 1.0,
            'random_seed': 0,
        }

    @staticmethod
    @abstractmethod
    def name(params: Dict[str, Any]) -> str:
        raise NotImplementedError()

    def __init__(self, params: Dict[str, Any], task: Sparse_Graph_Task, run_id: str = "run",
                 result_dir: str = ".", device: Optional[str] = None) -> None:
        self.params = params
        self.task = task
        self.run_id = run_id
        self.result_dir = result_dir
        self.device = torch.device(device if device is not None else "cuda")
        self._native_batchers = {}
        self.training = False
        torch.manual_seed(params['random_seed'])
        np.random.seed(params['random_seed'])
        if self.device.type == "cuda":
            # the hand-over status block of this device (ops.handover_word): exists before any step can be captured into a hipGraph;
            # read back with every step's metrics (MetricsReadback) and by handover_status()
            from .. import ops as _ops
            self.handover_word = _ops.handover_word(self.device)
        self.variables = VariableStore(seed=params['random_seed'])
        self.__make_model()
        self.variables.to(self.device)
        self.__make_train_step()

    @property
    def log_file(self):
        return os.path.join(self.result_dir, "%s.log" % self.run_id)

    @property
    def best_model_file(self):
        return os.path.join(self.result_dir, "%s_best_model.pickle" % self.run_id)

    def log_line(self, msg):
        try:
            with open(self.log_file, 'a') as log_fh:
                log_fh.write(msg + '\n')
        except OSError:
            pass
        print(msg)

    # -------------------- Model Saving/Loading (reference pickle layout :91-126) --------------------
    def save_model(self, path: str) -> None:
        data_to_save = {
            "model_class": self.name(self.params),
            "task_class": self.task.name(),
            "model_params": self.params,
            "task_params": self.task.params,
            "task_metadata": self.task.get_metadata(),
            # every GLOBAL_VARIABLE of the TF graph: the model variables and the optimizer's slots
            "weights": dict(self.variables.tf_weights(), **self.optimizer.tf_slot_weights(self._trainable_names())),
        }
        with open(path, 'wb') as out_file:
            pickle.dump(data_to_save, out_file, pickle.HIGHEST_PROTOCOL)

    def _trainable_names(self) -> List[str]:
        return [n for n in self.variables.names() if self.variables[n].requires_grad]

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """:109-126: assign every variable the pickle names (model variables and optimizer slots), freshly initialise
        (= keep the initial value of) the ones it does not, report saved entries nothing consumed."""
        used = self.variables.load_tf_weights(weights, report_unused=False)
        used |= self.optimizer.load_tf_slots(weights, self._trainable_names())
        from ..dense import weights_changed
        weights_changed()                  # (copy_ moves the version counters too; said once more for limb images kept per step)
        for var_name in weights:
            if var_name not in used:
                print('Saved weights for %s not used by model.' % var_name)

    # -------------------- Model Construction --------------------
    @abstractmethod
    def _gnn_layer_variables(self, in_dim: int) -> Dict[str, Any]:
        """TF-relative variable specs of ONE gnn layer (what the reference's layer function would
        create under variable_scope('gnn_layer_%i'))."""
        raise NotImplementedError()

    def __make_model(self):
        p = self.params
        h_dim = p['hidden_size']
        vs = self.variables
        if self.task.initial_node_feature_size != h_dim:          # :165-170, unnamed Keras Dense
            vs.create("graph_model/dense/kernel", (self.task.initial_node_feature_size, h_dim))
        self._inter_norm_name = []
        for layer_idx in range(p['graph_num_layers']):            # :176-200
            scope = "graph_model/gnn_layer_%i" % layer_idx
            specs = dict(self._gnn_layer_variables(h_dim))
            if p['graph_inter_layer_norm']:
                own = sum(1 for k in specs if k.startswith("LayerNorm") and k.endswith("/gamma"))
                ln = layer_norm_scope(own)          # TF uniquifies: the scope after the layer's own per-timestep norms
                specs[ln + "/beta"] = ((h_dim,), "zeros")
                specs[ln + "/gamma"] = ((h_dim,), "ones")
                self._inter_norm_name.append(ln)
            else:
                self._inter_norm_name.append(None)
            if layer_idx % p['graph_dense_between_every_num_gnn_layers'] == 0:
                specs["Dense/kernel"] = ((h_dim, h_dim), "glorot_uniform")
            vs.create_all(scope, specs)
        # The task names its own output variables (absolute TF scope): PPI's head is an UNNAMED Keras Dense, which TF
        # auto-names after the model's unnamed input projection ("dense" -> "dense_1", tasks/ppi_task.py:176-179);
        # QM9's heads live under variable_scope("out_layer_task%i") at the root (tasks/qm9_task.py:163-176).
        has_projection = self.task.initial_node_feature_size != h_dim
        self._task_scope = self.task.output_variable_scope(has_projection)
        vs.create_all(self._task_scope, self.task.output_variables(h_dim))
        self.log_line("Model has %i parameters." % vs.num_parameters())

    def compute_final_node_representations(self, initial_node_features: torch.Tensor,
                                           adjacency_lists, type_to_num_incoming_edges: torch.Tensor,
                                           dropout_keep_prob: float = 1.0) -> torch.Tensor:
        """__build_graph_propagation_model, models/sparse_graph_model.py:162-202."""
        p = self.params
        activation_fn = get_activation(p['graph_model_activation_function'])
        w = self.variables.scope("graph_model")
        num_nodes = initial_node_features.shape[0]
        # bucketed once, shared by every layer; the index range check is read back at the next fetch
        graph = as_rel_graph(adjacency_lists, num_nodes, validate="deferred")
        from ..dense import dense_act, vouch_sole_consumer
        from ..ops import activation_id
        act_id = activation_id(p['graph_model_activation_function'])
        num_layers, res_every = p['graph_num_layers'], p['graph_residual_connection_every_num_layers']

        def only_the_next_layer_reads(next_layer_idx: int) -> bool:
            """Is the layer function of iteration `next_layer_idx` the only reader of the tensor handed to it?  (One of the two words
            that let a layer fold a non-idempotent activation gradient — tanh' of the Dense below it — into its input-gradient
            product: dense.py, "activation gradients folded into the product that feeds them"; the other one is the layer's own:
            that it reads its input once.)  At a residual step the tensor is read by the average instead (and kept for the next
            residual step); at layer 0 it is kept for the first residual step too."""
            if next_layer_idx >= num_layers or dropout_keep_prob < 1.0:
                return False
            if next_layer_idx % res_every == 0:
                return next_layer_idx == 0 and res_every >= num_layers
            return True

        if self.task.initial_node_feature_size != p['hidden_size']:
            cur_node_representations = dense_act(initial_node_features, w["dense/kernel"], None, act_id,
                                                 sole_consumer=only_the_next_layer_reads(0))
        else:
            cur_node_representations = initial_node_features
        last_residual_representations = None          # (the reference's zeros_like is overwritten at layer 0 before any use)
        for layer_idx in range(p['graph_num_layers']):
            self._layer_weights = w.scope('gnn_layer_%i' % layer_idx)
            if dropout_keep_prob < 1.0:
                cur_node_representations = torch.nn.functional.dropout(
                    cur_node_representations, p=1.0 - dropout_keep_prob, training=True)
            if layer_idx % p['graph_residual_connection_every_num_layers'] == 0:
                t = cur_node_representations
                if layer_idx > 0:
                    cur_node_representations = (cur_node_representations + last_residual_representations) / 2
                last_residual_representations = t
            cur_node_representations = self._apply_gnn_layer(
                cur_node_representations, graph, type_to_num_incoming_edges, p['graph_num_timesteps_per_layer'])
            if p['graph_inter_layer_norm']:
                ln = self._inter_norm_name[layer_idx]
                cur_node_representations = layer_norm(cur_node_representations,
                                                      self._layer_weights[ln + "/gamma"], self._layer_weights[ln + "/beta"])
            if layer_idx % p['graph_dense_between_every_num_gnn_layers'] == 0:
                # (the layer's / the norm's output is referenced nowhere else: this Dense is its only reader)
                cur_node_representations = dense_act(vouch_sole_consumer(cur_node_representations, True),
                                                     self._layer_weights["Dense/kernel"], None, act_id,
                                                     sole_consumer=only_the_next_layer_reads(layer_idx + 1), sole_reader=True)
            else:
                vouch_sole_consumer(cur_node_representations, only_the_next_layer_reads(layer_idx + 1))
        return vouch_sole_consumer(cur_node_representations, False)     # (a task head may read it more than once)

    @abstractmethod
    def _apply_gnn_layer(self,
                         node_representations: torch.Tensor,
                         adjacency_lists: List[torch.Tensor],
                         type_to_num_incoming_edges: torch.Tensor,
                         num_timesteps: int) -> torch.Tensor:
        """Run a GNN layer on a graph (models/sparse_graph_model.py:204-225; same arguments).
        The current layer's variables are available as self._layer_weights."""
        raise Exception("Models have to implement _apply_gnn_layer!")

    def __make_train_step(self):
        p = self.params
        self.optimizer = TFStyleOptimizer(list(self.variables.parameters()), p['optimizer'], p['learning_rate'],
                                          p['clamp_gradient_norm'], decay=p['learning_rate_decay'],
                                          momentum=p['momentum'])

    # -------------------- Training Loop --------------------
    def forward_batch(self, batch: DeviceBatch, training: bool):
        keep = self.params['graph_layer_input_dropout_keep_prob'] if training else 1.0
        batch.wait_ready()               # assembled on a side stream (tasks/resident.py)? wait on the GPU, not the host
        # a batch from the input pipeline may carry its bucketing, built on a side stream (tasks/batcher.py)
        graph = getattr(batch, "graph", None)
        final = self.compute_final_node_representations(
            batch.initial_node_features, graph if graph is not None else batch.adjacency_lists,
            batch.type_to_num_incoming_edges, keep)
        return self.task.compute_task_metrics(final, batch, self.variables.scope(self._task_scope))

    def capture_train_step(self, batch: DeviceBatch, warmup_steps: int = 3):
        """One full training step on a FIXED batch as a hipGraph (forward, backward, clip, Adam: ~150 launches replayed
        with one host call).  For workloads whose step is host-enqueue bound (C3: 153 k messages per batch) and whose
        batch shapes repeat.  Runs `warmup_steps` real steps first (allocator / plan caches), then captures; returns a
        CapturedTrainStep whose replay() performs exactly one more step and returns the (static) metric tensors."""
        from ..graph import as_rel_graph as _as_graph
        if self.device.type != "cuda" or self.optimizer.name != 'adam':
            raise RuntimeError("capture_train_step needs a GPU model trained with Adam")
        batch.wait_ready()
        if getattr(batch, "graph", None) is None:           # the bucketing is part of the fixed batch, not of the step
            batch.graph = _as_graph(batch.adjacency_lists, batch.num_nodes, validate=True)
        opt = self.optimizer
        opt.sync_device_step_count()
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup_steps):
                self.train_step(batch, device_step_count=True)
                opt.t += 1
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad()
        from ..dense import capture_image_cache
        with capture_image_cache(), torch.cuda.graph(graph):      # (limb images of the weights: split once per captured step)
            metrics = self.train_step(batch, device_step_count=True)
        return CapturedTrainStep(self, graph, {k: v for k, v in metrics.items() if torch.is_tensor(v)}, batch)

    _backward_seed: Dict[Any, torch.Tensor] = {}

    def train_step(self, batch: DeviceBatch, grad_hook=None, device_step_count: bool = False,
                   pre_backward=None) -> Dict[str, torch.Tensor]:
        """forward + backward + per-variable clip + optimizer update == one sess.run with train_step (:287-293).
        pre_backward(): called between the forward and the backward (a bucketed gradient all-reduce arms its hooks there);
        grad_hook(params): between the backward and the clipping (the data-parallel reduction)."""
        self.optimizer.zero_grad()
        metrics = self.forward_batch(batch, training=True)
        if pre_backward is not None:
            pre_backward()
        # one process drives one GPU: running the backward on the calling thread instead of the autograd engine's
        # device thread saves the hand-off per node (host enqueue 1.56 -> 1.29 ms per C2 step) and a busy CPU thread
        from .. import ops
        from contextlib import nullcontext
        # (the weight gradients of the aggregate-first layers run on a side stream; nothing reads them before the reduction / the
        #  optimizer, so their joins move behind the whole backward — not with a reducer that packs gradients from hooks DURING it)
        # (measured on the C2 step, alternated three times: 1.812 / 1.812 / 1.814 ms with the joins per layer, 1.804 / 1.806 / 1.806 deferred)
        defer = ops.deferred_weight_gradient_join() if pre_backward is None else nullcontext()
        with torch.autograd.set_multithreading_enabled(False), defer:
            loss = metrics['loss']
            one = self._backward_seed.get((loss.device, loss.dtype, loss.shape))
            if one is None:       # (loss.backward() would fill a fresh ones_like every step)
                one = self._backward_seed[(loss.device, loss.dtype, loss.shape)] = torch.ones_like(loss)
            loss.backward(gradient=one)
        ops.join_deferred()
        if grad_hook is not None:  # data-parallel gradient all-reduce goes here (before clipping)
            grad_hook(self.optimizer.params)
        lr_scale = 1.0
        lr_n = self.params.get('lr_for_num_graphs_per_batch')
        if lr_n is not None:
            lr_scale = float(batch.num_graphs) / float(lr_n)
        self.optimizer.clip_and_step(lr_scale, device_step_count=device_step_count)
        return metrics

    def _batches(self, data, data_fold: DataFold):
        """Batches of one epoch.  On the GPU the data fold is flattened once and batches come from the C++ builder
        (tasks/batcher.py: packed in pinned memory on a background thread, one async upload each), the stand-in for
        the reference's ThreadedIterator (:272); `native_batching: false` keeps the numpy iterator."""
        native = self.params.get('native_batching', True) and torch.device(self.device).type == "cuda" \
            and hasattr(self.task, "make_graph_store")
        if not native:
            return self.task.make_minibatch_iterator(data, data_fold, self.params['max_nodes_in_batch'])
        from ..tasks.batcher import NativeBatcher
        key = (id(data), len(data))
        cached = self._native_batchers.get(key)
        if cached is not None and cached[0] is data:
            self._native_batchers[key] = self._native_batchers.pop(key)      # most recently used last
        if cached is None or cached[0] is not data:
            store = self.task.make_graph_store(data)
            pipeline = None
            # small enough folds simply live in HBM, bucketed once (tasks/resident.py); `resident_dataset: false`
            # keeps the host packer, `true` insists
            want = self.params.get('resident_dataset', 'auto')
            if want is True or (want == 'auto' and self._fold_fits_hbm(store)):
                from ..tasks.resident import ResidentDataset
                try:
                    pipeline = ResidentDataset(store, self.device)
                except ValueError:
                    if want is True:
                        raise
            if pipeline is None:
                pipeline = NativeBatcher(store, self.device)
            cached = (data, pipeline)
            self._native_batchers[key] = cached
            # train + validation folds (and one more) stay; anything older — e.g. the fresh list every test() call
            # passes — is dropped so that its resident copy of the fold and its pinned arenas are released
            while len(self._native_batchers) > 3:
                self._native_batchers.pop(next(iter(self._native_batchers)))
        return self.task.make_native_minibatch_iterator(cached[1], data_fold, self.params['max_nodes_in_batch'])

    @staticmethod
    def _fold_fits_hbm(store, budget_bytes: int = 32 << 30) -> bool:
        """Flat arrays + fold-level bucketing (~9 int32 per message) well inside the 288 GB of an MI355X."""
        nbytes = sum(a.nbytes for a in store.payload) + sum(a.nbytes for a in store.adj) + sum(a.nbytes for a in store.deg)
        messages = sum(int(o[-1]) for o in store.edge_off)
        nodes = int(store.node_off[-1])
        if nodes * max(store.num_edge_types, 1) >= 2 ** 31 - 1 or messages >= 2 ** 31 - 1:
            return False
        return nbytes + 40 * messages + 8 * nodes * max(store.num_edge_types, 1) < budget_bytes

    def _run_epoch(self, epoch_name: str, data: Iterable[Any], data_fold: DataFold, quiet: bool = False):
        """__run_epoch, :263-311: returns (avg loss, task metric results, graphs, graphs/s, nodes/s, edges/s)."""
        batch_iterator = self._batches(data, data_fold)
        start_time = time.time()
        processed_graphs = processed_nodes = processed_edges = 0
        epoch_loss = 0.0
        task_metric_results = []
        batch_iterator = iter(batch_iterator)
        upcoming = next(batch_iterator, None)
        state = {"graphs": 0, "nodes": 0, "edges": 0, "loss": 0.0, "step": 0}

        def fetch(pending):
            """Read one step's metrics back (the host sync of sess.run's fetch, :293)."""
            m, mb = pending
            m = m.get()
            check_pending_graph_errors()
            state["graphs"] += mb.num_graphs
            state["nodes"] += mb.num_nodes
            state["edges"] += mb.num_edges
            state["loss"] += m['loss'] * mb.num_graphs
            task_metric_results.append(m)
            if not quiet:
                print("Running %s, batch %i (has %i graphs). Loss so far: %.4f"
                      % (epoch_name, state["step"], mb.num_graphs, state["loss"] / state["graphs"]), end='\r')
            state["step"] += 1

        pending = None
        while upcoming is not None:
            mb = upcoming
            batch = mb if isinstance(mb, DeviceBatch) else DeviceBatch(mb, self.device)
            if data_fold == DataFold.TRAIN:
                m = self.train_step(batch)
            else:
                with torch.no_grad():
                    m = self.forward_batch(batch, training=False)
            # Pipeline: the step's metrics start their way to pinned host memory right behind the step (MetricsReadback);
            # ask for the next batch (its upload / assembly run on a side stream under this step's kernels), THEN read
            # the metrics of the PREVIOUS step: that wait is for the previous step's copy only, which the GPU has passed
            # long ago, so the device never idles behind a host round trip.  Every step's metrics are still fetched, one
            # step late.
            readback = MetricsReadback(m)          # (also drops the autograd graph: values are detached)
            upcoming = next(batch_iterator, None)
            if pending is not None:
                fetch(pending)
            pending = (readback, mb)
        if pending is not None:
            fetch(pending)
        processed_graphs, processed_nodes, processed_edges = state["graphs"], state["nodes"], state["edges"]
        epoch_loss = state["loss"]
        epoch_time = time.time() - start_time
        per_graph_loss = epoch_loss / max(processed_graphs, 1)
        return (per_graph_loss, task_metric_results, processed_graphs, processed_graphs / epoch_time,
                processed_nodes / epoch_time, processed_edges / epoch_time)

    def train(self, quiet: bool = False, max_epochs: Optional[int] = None):
        """:318-371 without TensorBoard: early stopping on the task's validation metric."""
        total_time_start = time.time()
        train_data = self.task._loaded_data[DataFold.TRAIN]
        valid_data = self.task._loaded_data[DataFold.VALIDATION]
        # The loaded folds are millions of long-lived Python objects: frozen for the duration of the loop they cost the cyclic
        # collector nothing, and a step's own cyclic garbage (autograd graphs hold device tensors) is collected by cheap, frequent
        # full collections instead of waiting for a quarter as many survivors as there are tracked objects.
        import gc
        gc.collect()
        gc.freeze()
        try:
            return self._train_loop(train_data, valid_data, total_time_start, quiet, max_epochs)
        finally:
            gc.unfreeze()

    def _train_loop(self, train_data, valid_data, total_time_start, quiet, max_epochs):
        best_valid_metric, best_epoch = float("+inf"), 0
        for epoch in range(1, (max_epochs or self.params['max_epochs']) + 1):
            self.log_line("== Epoch %i" % epoch)
            loss, res, n, gs, ns, es = self._run_epoch("epoch %i (training)" % epoch, train_data, DataFold.TRAIN, quiet)
            self.log_line(" Train: loss: %.5f || %s || graphs/sec: %.2f | nodes/sec: %.0f | edges/sec: %.0f"
                          % (loss, self.task.pretty_print_epoch_task_metrics(res, n), gs, ns, es))
            loss, res, n, gs, ns, es = self._run_epoch("epoch %i (validation)" % epoch, valid_data, DataFold.VALIDATION, quiet)
            metric = self.task.early_stopping_metric(res, n)
            self.log_line(" Valid: loss: %.5f || %s || graphs/sec: %.2f | nodes/sec: %.0f | edges/sec: %.0f"
                          % (loss, self.task.pretty_print_epoch_task_metrics(res, n), gs, ns, es))
            if metric < best_valid_metric:
                self.save_model(self.best_model_file)
                best_valid_metric, best_epoch = metric, epoch
            elif epoch - best_epoch >= self.params['patience']:
                self.log_line("Stopping training after %i epochs without improvement." % self.params['patience'])
                break
        self.log_line("Training took %is." % (time.time() - total_time_start))

    def test(self, data, quiet: bool = False):
        loss, res, n, gs, ns, es = self._run_epoch("Test", list(data), DataFold.TEST, quiet)
        self.log_line("Loss %.5f on %i graphs" % (loss, n))
        self.log_line("Metrics: %s" % self.task.pretty_print_epoch_task_metrics(res, n))
