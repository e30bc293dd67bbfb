"""200 training steps on ONE C2 batch under each arithmetic of the Dense products: does the two-fp16-limb route train like fp32?

VERDICT r03, next 3b.  Adam divides every weight's gradient by its own running magnitude, so a route whose per-element RELATIVE
gradient error were large would walk away from the exact-fp32 trajectory faster than rounding noise does.  The control for
"rounding noise" is the bf16 triple (an exact split: fp32-class products in another summation order) and a second exact-fp32 run
with the reference's transform-first order (same arithmetic, other association).

    python scripts/exp_trajectory_routes.py [steps] > gpurun_out/trajectory_routes.json

Per route: the loss curve (every 10th step), and against the exact-fp32 library run at steps 1, 10, 50, 100, 200
  max_rel_weight_divergence = max over variables of max|w - w_lib| / max|w_lib|
  rel_frobenius              = ||w - w_lib|| / ||w_lib|| over all variables
Same seed, same batch, same initial weights; a fresh model per route in one process (config.override).
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ROUTES = {
    "fp32_library": dict(gemm="lib"),
    "fp32_library_transform_first": dict(gemm="lib", rgcn_order="transform_first"),
    "bf16_triple": dict(gemm="limb", limb="triple"),
    "fp16_pair": dict(gemm="limb", limb="pair"),
}
MARKS = (1, 10, 50, 100, 200)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    from tf_gnn_samples_amd import config
    from tf_gnn_samples_amd.models import RGCN_Model
    from tf_gnn_samples_amd.tasks import DataFold, DeviceBatch, PPI_Task
    dev = torch.device("cuda:0")
    task = PPI_Task(PPI_Task.default_params())
    task.load_synthetic(16, 1, seed=0)
    mb = next(task.make_minibatch_iterator(task._loaded_data[DataFold.TRAIN], DataFold.VALIDATION, 10 ** 9))
    p = RGCN_Model.default_params()
    p.update(hidden_size=256, graph_num_layers=3, graph_activation_function="ReLU", message_aggregation_function="sum",
             graph_layer_input_dropout_keep_prob=1.0, random_seed=0)
    runs = {}
    for name, switches in ROUTES.items():
        with config.override(**switches):
            keep, sys.stdout = sys.stdout, sys.stderr           # (the constructor prints the parameter count)
            try:
                model = RGCN_Model(p, task, device=str(dev))
            finally:
                sys.stdout = keep
            batch = DeviceBatch(mb, dev)
            losses, snaps = [], {}
            for step in range(1, steps + 1):
                m = model.train_step(batch)
                losses.append(float(m['loss'].detach()))
                if step in MARKS:
                    snaps[step] = {n: model.variables[n].detach().double().cpu().numpy().copy() for n in model.variables.names()}
            runs[name] = (losses, snaps)
            del model
    ref_losses, ref_snaps = runs["fp32_library"]
    out = {"what": "C2 batch (%d nodes, %d edges), 3-layer RGCN h=256 + PPI head, Adam, per-variable clip: %d steps on the same batch "
                   "per route of the Dense products; divergence from the exact-fp32 library run" % (mb.num_nodes, mb.num_edges, steps),
           "switches": ROUTES, "routes": {}}
    for name, (losses, snaps) in runs.items():
        row = {"loss_every_10th_step": [round(x, 6) for x in losses[9::10]], "final_loss": losses[-1],
               "max_abs_loss_difference_from_fp32_library": float(np.abs(np.array(losses) - np.array(ref_losses)).max())}
        if name != "fp32_library":
            div = {}
            for step, snap in snaps.items():
                rel = {n: float(np.abs(snap[n] - ref_snaps[step][n]).max() / max(np.abs(ref_snaps[step][n]).max(), 1e-300)) for n in snap}
                num = np.sqrt(sum(float(((snap[n] - ref_snaps[step][n]) ** 2).sum()) for n in snap))
                den = np.sqrt(sum(float((ref_snaps[step][n] ** 2).sum()) for n in snap))
                worst = max(rel, key=rel.get)
                div[str(step)] = {"max_rel_weight_divergence": rel[worst], "in_variable": worst, "rel_frobenius": num / den}
            row["divergence_from_fp32_library_at_step"] = div
        out["routes"][name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
