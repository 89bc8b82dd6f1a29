import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The HIP library + oracle, built in-tree (no-op when the .so files are current)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def kats():
    import json
    return json.load(open(os.path.join(HERE, "golden", "survey_kats.json")))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


@pytest.fixture(scope="session")
def evp():
    return [l.rstrip("\n") for l in open(os.path.join(HERE, "golden", "evp_peparray_probe_sequence.txt"))]
