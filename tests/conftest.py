import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """CPU-side artefacts (oracle, generator, and the HIP library, which cross-compiles without a
    GPU) are built on demand so that a fresh checkout can run the CPU tier directly."""
    import __graft_entry__ as ge
    ge.build(only_missing=True)


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def params_from(arr, **over):
    from minimap2_chaindp_amd.params import ChainParams
    p = ChainParams(*[int(x) for x in arr])
    for k, v in over.items():
        setattr(p, k, v)
    return p
