import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch bundles its own HIP runtime; when both torch and libvbx_hip.so live in one process the
# runtime must be initialised through torch FIRST (the other order leaves torch without a GPU).
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # pragma: no cover - torch is optional for the CPU suite
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py
