"""The HIP path against fixtures written by the REFERENCE'S OWN layer code (tests/golden/make_reference_run.py: the unmodified
gnns/*.py executed over a NumPy shim of the TF ops; tests/golden/tf_numpy_shim.py says what that pins).  Same inputs, the variables the
reference created under their TF names, its outputs; the layer functions of the package take the reference's arguments."""
import numpy as np
import pytest
import torch

from test_reference_run_cpu import CASES, call_layer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(len(CASES)), ids=["%s-%d" % (c[0]["function"], n) for n, c in enumerate(CASES)])
def test_hip_layers_reproduce_the_reference_run(gpu_device, i):
    from tf_gnn_samples_amd import gnns
    case, h, adj, deg, weights, want = CASES[i]

    def to_device(x):
        x = np.asarray(x)
        return torch.as_tensor(x, device=gpu_device)

    got = call_layer(gnns, case, h, adj, deg, weights, convert=to_device)
    assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
    err = float(np.abs(got.cpu().numpy() - want).max())
    assert err <= 1e-5 * max(1.0, float(np.abs(want).max())), (case["function"], case["kwargs"], err)
