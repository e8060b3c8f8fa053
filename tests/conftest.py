import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def g_search():
    return np.load(os.path.join(GOLDEN, "newref_search.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def g_pipe():
    return np.load(os.path.join(GOLDEN, "pipeline.npz"), allow_pickle=False)


def ref_dict_from_golden(g):
    """Reference .npz content stored under the 'ref__' prefix in pipeline.npz."""
    return {k[5:]: g[k] for k in g.files if k.startswith("ref__")}


def sample_from_counts(counts, bpc):
    off = np.concatenate(([0], np.cumsum(bpc)))
    return {str(c + 1): counts[off[c]:off[c + 1]].astype(np.int32) for c in range(24)}
