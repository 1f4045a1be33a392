import os
import sys

import pytest

# torch first: it bundles its own HIP runtime, which must be the one the process loads (importing it after
# libpirgpu.so has pulled in /opt/rocm's copy leaves torch.cuda unable to initialise)
try:
    import torch  # noqa: F401
except ImportError:
    pass

# the library honours its PIRGPU_* tuning / flavour variables only behind this gate (pir_amd/csrc/env_gate.h); the tests
# that sweep flavours and geometries through the environment need it open
os.environ.setdefault("PIRGPU_ALLOW_ENV", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def parse_poly(s, N):
    """SEAL Plaintext hex-string form ("4x^4 + 33x^3 + 42") -> coefficient list."""
    import numpy as np
    out = np.zeros(N, dtype=np.uint64)
    for term in s.split("+"):
        term = term.strip()
        if not term:
            continue
        if "x^" in term:
            c, e = term.split("x^")
            out[int(e)] = int(c, 16)
        else:
            out[0] = int(term, 16)
    return out


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.load()
    return oracle
