import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no libdrqv2_hip.so (build artefacts are not tracked): build it once, in-tree, when the
    # compiler is there.  An existing library is never rebuilt from here.
    from drqv2_amd import build as _build
    if not os.path.exists(_build.LIB):
        import shutil
        if os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc"):
            _build.build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
