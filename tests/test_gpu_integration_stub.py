"""The raw-ctypes binding shown in INTEGRATION.md section 2 (no package code involved) against the oracle."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from helpers import random_relational_graph

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_integration_md_stub_runs_and_matches_oracle(gpu_device):
    md = (ROOT / "INTEGRATION.md").read_text()
    code = re.search(r"```python\nimport ctypes, torch\n(.*?)```", md, flags=re.S).group(0)
    code = code[len("```python\n"):-3].replace('ctypes.CDLL("tf_gnn_samples_amd/librelgnn.so")',
                                               'ctypes.CDLL(%r)' % str(ROOT / "tf_gnn_samples_amd" / "librelgnn.so"))
    ns = {}
    exec(code, ns)
    rng = np.random.default_rng(3)
    V, L, D = 300, 3, 256
    adj = random_relational_graph(rng, V, L, [2000, 0, 700])
    Tt = rng.standard_normal((V * L, D)).astype(np.float32)
    out = ns["unsorted_segment_sum_of_gathered_rows"](torch.as_tensor(Tt, device=gpu_device),
                                                      [torch.as_tensor(a, device=gpu_device) for a in adj], V)
    rows = np.concatenate([a[:, 0].astype(np.int64) * L + l for l, a in enumerate(adj)])
    tg = np.concatenate([a[:, 1] for a in adj]).astype(np.int32)
    np.testing.assert_array_equal(out.cpu().numpy(), T.unsorted_segment_sum(Tt[rows], tg, V))


def test_integration_md_dense_stub_runs_and_matches_float64(gpu_device):
    """The second stub of INTEGRATION.md section 2: the layer's Dense product and its weight gradient through raw ctypes."""
    md = (ROOT / "INTEGRATION.md").read_text()
    first = re.search(r"```python\nimport ctypes, torch\n(.*?)```", md, flags=re.S).group(0)
    second = re.search(r"```python\nlib\.relgnn_panel_gemm_zeros_floats\.restype(.*?)```", md, flags=re.S).group(0)
    head = first[len("```python\n"):-3].split("\ndef ")[0].replace('ctypes.CDLL("tf_gnn_samples_amd/librelgnn.so")',
                                                                   'ctypes.CDLL(%r)' % str(ROOT / "tf_gnn_samples_amd" / "librelgnn.so"))
    ns = {}
    exec(head + "\n" + second[len("```python\n"):-3], ns)
    g = torch.Generator(device="cpu").manual_seed(2)
    A = (torch.rand((5003, 768), generator=g) * 2 - 1).to(gpu_device)
    W = ((torch.rand((768, 256), generator=g) * 2 - 1) * 0.08).to(gpu_device)
    G = ((torch.rand((5003, 256), generator=g) * 2 - 1) * 0.05).to(gpu_device)
    out = ns["dense_relu"](A, W)
    assert float((out.double() - torch.relu(A.double() @ W.double())).abs().max()) <= 5e-6
    dW = ns["dense_weight_gradient"](A, G)
    assert float((dW.double() - A.double().t() @ G.double()).abs().max()) <= 5e-6
