import os
import sys

import numpy as np
import pytest

# the CPU tier imports the product module on hosts without a GPU (ABI, host logic, gloo drivers): edt's import-time probe is
# told so; on the GPU box the variable changes nothing (the device node is there)
os.environ.setdefault("EDT_HIP_ALLOW_NO_DEVICE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "euclidean-distance-transform-3d_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_port():
    """Our plain-C restatement of the reference (oracle/edt_oracle.c), built on demand."""
    from oracle import harness
    if not harness.have_port():
        harness.build("port")
    return harness.port()


@pytest.fixture(scope="session")
def oracle_ref():
    """The real reference compiled from /root/reference (only where that tree or a prebuilt
    oracle/_ref exists)."""
    from oracle import harness
    if not harness.have_ref():
        if os.path.isdir("/root/reference/src"):
            harness.build("ref")
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return harness.ref()


@pytest.fixture(scope="session")
def edt_gpu():
    """The product module (HIP path).  Fails loudly when the library or the GPU is missing."""
    import edt
    from edt import _lib
    _lib.load()
    assert _lib.device_count() > 0, "no HIP device visible: GPU tests need an MI355X"
    return edt


def load_golden(name):
    path = os.path.join(ROOT, "tests", "golden", name)
    blob = np.load(path, allow_pickle=False)
    cases = {}
    for key in blob.files:
        idx, field = key.split("/")
        cases.setdefault(int(idx), {})[field] = blob[key]
    return [cases[i] for i in sorted(cases)]
