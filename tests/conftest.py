import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-bayesiancnn_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _reference_dir():
    """The unmodified upstream files: /root/reference in the build container, oracle/_ref/upstream_snapshot.zip (packed by
    __graft_entry__.build(), oracle/ref_snapshot.py) unpacked to a temporary directory on the GPU box; None if neither."""
    import ref_snapshot
    return ref_snapshot.checkout()[1]


REFERENCE = _reference_dir()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the unmodified upstream files (checkout or oracle/_ref snapshot)")


def pytest_collection_modifyitems(config, items):
    have_ref = REFERENCE is not None
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="neither /root/reference nor oracle/_ref/upstream_snapshot.zip present"))


@pytest.fixture(scope="session")
def reference_dir():
    return REFERENCE


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return {n: np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("layers_small", "functions", "models", "uncertainty")}


@pytest.fixture(scope="session")
def golden_driver():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "driver.npz"))
