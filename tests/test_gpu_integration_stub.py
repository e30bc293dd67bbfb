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
