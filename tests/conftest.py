import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz -> dict of numpy arrays."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_sd(g, prefix="sd/"):
    """state_dict (torch tensors) stored under `prefix` in a golden dict."""
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def golden_cfg(g):
    return json.loads(str(g["cfg"]))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is a build artefact (git-ignored): compile it once if this checkout does not have it yet."""
    from studiosr_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()

