import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_layer(M, N, K, O=128, seed=0, outlier_gain=20.0):
    """Synthetic MixQ linear in the shape SURVEY.md §8d prescribes: A ~ N(0,1) with the outlier columns scaled up,
    W ~ N(0, 0.02^2), activation scales |N(0,1)| (the real Llama vectors are used where K matches, see test_golden)."""
    rng = np.random.default_rng(seed)
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    A = rng.standard_normal((M, K)).astype(np.float32)
    ind = np.argsort(act, kind="stable")[-O:].astype(np.int32)
    A[:, ind] *= outlier_gain
    return A.astype(np.float16), W, act


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o
