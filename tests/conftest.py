import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def matrices():
    return dict(np.load(os.path.join(GOLDEN, "matrices.npz")))


@pytest.fixture(scope="session")
def sw_vectors():
    d = dict(np.load(os.path.join(GOLDEN, "sw_vectors.npz")))
    d["bt"] = bytes(d["bt"]).decode().split("\n")
    return d


@pytest.fixture(scope="session")
def reflib():
    """The real reference, only where oracle/_ref was built and /root/reference/data is readable."""
    from oracle import pyoracle
    if not (pyoracle.ref_available() and pyoracle.ref_matrix_available()):
        pytest.skip("real reference (oracle/_ref + /root/reference/data) not available here")
    return pyoracle.RefLib()


@pytest.fixture(scope="session")
def gpu():
    # torch first: it ships its own HIP runtime, and the process must end up with ONE libamdhip64 (the first one loaded
    # wins the soname); loading libmmgpu.so first makes torch.cuda fail later ("No HIP GPUs are available")
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    import mmseqs2_amd
    g = mmseqs2_amd.MMGpu(0)   # raises if libmmgpu.so is missing or no GPU: no fallback
    yield g
    g.close()
