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


# (round 5 ran the cyclic collector only between tests to mask an intermittent crash; the cause is gone from the product -- profiles/r06_suite_crash_rootcause.txt --
#  and so are the fixtures: Python's own collector, as in any user process)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "ntt_golden.npz"))


@pytest.fixture(scope="session")
def kat_msgs():
    d = np.load(os.path.join(GOLDEN, "kat_msgs.npz"))
    buf = d["msg"].tobytes()
    out, off = [], 0
    for n in d["mlen"]:
        out.append(buf[off:off + int(n)])
        off += int(n)
    return out


def load_kat(level):
    return np.load(os.path.join(GOLDEN, f"kat_{level}.npz"))


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library on cuda:0; GPU tests must not pass on a fallback."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    from dilithium_amd import api
    api.init(0)
    return torch
