import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's C kernels are test infrastructure: build them if gcc is around (seconds)
    so = ROOT / "oracle" / "_build" / "liboracle_segment.so"
    if not so.exists():
        try:
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
        except Exception:
            pass  # NumPy fallback inside oracle/tf_ops.py


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
