import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from ns2vc_amd import engine
        return engine.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs the whole suite on a CPU box they are skipped, not failed
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def diag():
    """Append-only diagnostics file that travels back from the GPU box (gpurun_out/)."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "test_diag.txt")
    f = open(path, "a")

    def log(*a):
        line = " ".join(str(x) for x in a)
        f.write(line + "\n")
        f.flush()
        print(line)

    yield log
    f.close()
