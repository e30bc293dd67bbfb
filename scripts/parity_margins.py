#!/usr/bin/env python
"""Measured forward deviations of the HIP path from the reference-run fixtures, case by case (tests/test_gpu_reference_run.py holds
the asserts; this prints what they are set from): max abs error, the fixture's largest magnitude, error / magnitude.
    python scripts/parity_margins.py > gpurun_out/parity_margins.json"""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "golden"))


def main():
    from test_reference_run_cpu import CASES, MODEL_CASES, MODEL_Z, build_product_model, build_product_task, call_layer
    from tf_gnn_samples_amd import gnns
    from tf_gnn_samples_amd.tasks import DeviceBatch, MinibatchData
    dev = torch.device("cuda:0")
    out = {"layers": [], "models": []}
    for n, (case, h, adj, deg, weights, want) in enumerate(CASES):
        got = call_layer(gnns, case, h, adj, deg, weights, convert=lambda x: torch.as_tensor(np.asarray(x), device=dev))
        err = float(np.abs(got.cpu().numpy() - want).max())
        out["layers"].append({"id": "%s-%d" % (case["function"], n), "err": err, "max_ref": float(np.abs(want).max()),
                              "kwargs": {k: str(v) for k, v in case["kwargs"].items()}})
    z = MODEL_Z
    for n, entry in enumerate(MODEL_CASES):
        k = entry["key"]
        with tempfile.TemporaryDirectory() as d:
            task = build_product_task(entry, Path(d))
            model = build_product_model(entry, task, str(dev))
            with torch.no_grad():
                for v in entry["variables"]:
                    model.variables[v].copy_(torch.as_tensor(z["%s/var/%s" % (k, v)], device=dev))
            payload = entry["payload"]
            feed = {'initial_node_features': z[k + "/features"], 'type_to_num_incoming_edges': z[k + "/deg"],
                    'graph_nodes_list': z[k + "/graph_nodes_list"], payload: z[k + "/" + payload], 'out_layer_dropout_keep_prob': 1.0,
                    'adjacency_lists': [z["%s/adj%d" % (k, l)] for l in range(entry["num_edge_types"])]}
            mb = MinibatchData(feed_dict=feed, num_graphs=entry["num_graphs"], num_nodes=entry["num_nodes"], num_edges=entry["num_edges"])
            batch = DeviceBatch(mb, dev)
            with torch.no_grad():
                final = model.compute_final_node_representations(batch.initial_node_features, batch.adjacency_lists,
                                                                 batch.type_to_num_incoming_edges)
            want = z[k + "/final_node_representations"]
            out["models"].append({"id": "%s-%s-%d" % (entry["model"], entry["task"], n),
                                  "err": float(np.abs(final.cpu().numpy() - want).max()), "max_ref": float(np.abs(want).max())})
    # backward: d sum(out * cotangent) / d (node states, every variable) against the float64 gradients of the reference's own layer code
    from test_reference_run_cpu import AUTOGRAD_Z, NON_SMOOTH
    out["backward"] = []
    for n, (case, h, adj, deg, weights, _) in enumerate(CASES):
        k, z = case["key"], AUTOGRAD_Z
        hd = torch.tensor(h, device=dev, requires_grad=True)
        wd = {v: torch.tensor(a, device=dev, requires_grad=True) for v, a in weights.items()}
        kw = dict(case["kwargs"])
        fn = getattr(gnns, case["function"])
        adj_d = [torch.as_tensor(a, device=dev) for a in adj]
        deg_d = torch.as_tensor(deg, device=dev)
        if case["function"] == "sparse_rgdcn_layer":
            o = fn(hd, adj_d, deg_d, weights=wd, **kw)
        else:
            state_dim = kw.pop("state_dim")
            o = fn(hd, adj_d, deg_d, state_dim, weights=wd, **kw) if case["takes_degrees"] else fn(hd, adj_d, state_dim, weights=wd, **kw)
        cot = torch.as_tensor(z[k + "/cotangent"].astype(np.float32), device=dev)
        (o * cot).sum().backward()
        worst_err = worst_fro = 0.0
        for name in ["h"] + case["variables"]:
            want = z["%s/grad/%s" % (k, "h" if name == "h" else "var/" + name)]
            g = hd.grad if name == "h" else wd[name].grad
            got = np.zeros_like(want) if g is None else g.cpu().numpy().astype(np.float64)
            scale = max(1e-6, float(np.abs(want).max()))
            worst_err = max(worst_err, float(np.abs(got - want).max()) / scale)
            worst_fro = max(worst_fro, float(np.linalg.norm(got - want) / max(1e-12, np.linalg.norm(want))))
        smooth = str(case["kwargs"].get("activation_function")).lower() not in NON_SMOOTH and case["function"] != "sparse_rgat_layer"
        out["backward"].append({"id": "%s-%d" % (case["function"], n), "smooth": smooth, "err_over_scale": worst_err, "fro": worst_fro})
    for key in ("layers", "models"):
        grp = out[key]
        for r in grp:
            r["err_over_max_ref"] = r["err"] / max(r["max_ref"], 1e-30)
            r["strict_1e-5_abs"] = r["err"] <= 1e-5
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
