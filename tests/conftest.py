import os
import sys
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "icp-variants_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bunny():
    """The reference's bunny pair as point clouds (tests/golden/bunny_pair.npz, see make_golden.py)."""
    d = np.load(os.path.join(GOLDEN, "bunny_pair.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def bunny_oracle():
    d = np.load(os.path.join(GOLDEN, "bunny_oracle.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates contexts on cuda:0 through the C ABI; fails loudly if the HIP library or GPU is missing."""
    from icp_amd import binding
    binding.load_library()
    made = []

    def make():
        c = binding.Context(0)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()


def pose_error(A, B):
    """(rotation angle error [rad], translation error [m]) between two 4x4 poses."""
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    R = A[:3, :3] @ B[:3, :3].T
    # atan2(|vee(R - R^T)/2|, (tr - 1)/2): well conditioned near 0, unlike arccos of the trace
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    ang = np.arctan2(s, (np.trace(R) - 1) / 2)
    return float(ang), float(np.linalg.norm(A[:3, 3] - B[:3, 3]))
