import os
import sys

import pytest

# torch bundles its own HIP/HSA runtime (same SONAME as /opt/rocm's).  Whichever copy is mapped first
# serves the whole process; when libhyphy_hip.so pulls in /opt/rocm's copy first, a later
# torch.cuda initialisation reports "No HIP GPUs are available".  Tests that hand torch device
# buffers to the C-ABI therefore need torch loaded first — do it once, here.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU-only tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
