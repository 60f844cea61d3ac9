import json
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


def pytest_sessionstart(session):
    """A clean checkout has no built library (it is git-ignored): build it once so that the ABI /
    symbol tests have something to load.  hipcc cross-compiles without a GPU (a few minutes)."""
    lib = os.path.join(ROOT, "plonk-prototype_amd", "lib", "libplonk_mi355x.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "plonk-prototype_amd", "csrc"), "-j3"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """C CPU restatement (test infrastructure; builds with gcc on first use)."""
    from oracle.cpu_oracle import CpuOracle
    return CpuOracle()


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("ntt", "msm", "constants"):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            out[name] = json.load(f)
    return out


@pytest.fixture(scope="session")
def ctx():
    """The product: a context on cuda:0 through the C ABI.  No fallback -- fails without a GPU."""
    import plonk_prototype_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


# ---- helpers shared by the test modules -----------------------------------------------
def hex_to_fr_mont(oracle, hexes):
    from oracle.cpu_oracle import ints_to_limbs
    if not hexes:
        return np.zeros((0, 4), np.uint64)
    return oracle.fr_to_mont(ints_to_limbs([int(h, 16) for h in hexes], 4))


def points_to_mont(oracle, pts):
    """[[xhex, yhex] | None] -> [n, 12] Montgomery, identity = zeros"""
    from oracle.cpu_oracle import ints_to_limbs
    out = np.zeros((len(pts), 12), np.uint64)
    for i, p in enumerate(pts):
        if p is not None:
            out[i] = oracle.fp_to_mont(ints_to_limbs([int(p[0], 16), int(p[1], 16)], 6)).reshape(12)
    return out
