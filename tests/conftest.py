import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def shell12():
    from rigid_body_light_amd import load_structure
    return load_structure(12)[1]


def random_positions(N, wall=False, seed=0, min_dist=2.0):
    """Seeded variant of the reference's tests/utils.py:38-52 rejection sampler."""
    rng = np.random.default_rng(seed)
    X = np.zeros((N, 3))
    n = 0
    lo = 1.0 if wall else -10.0
    while n < N:
        x = rng.uniform(lo, 10.0, 3)
        if n == 0 or np.all(np.linalg.norm(X[:n] - x, axis=1) > min_dist):
            X[n] = x
            n += 1
    Q = rng.standard_normal((N, 4))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    return X, Q


def create_solver(X, Q, rigid_config=None, wall_PC=False, block_PC=False, a=1.0, eta=1.0, dt=1.0):
    """Mirror of the reference's tests/utils.py:22-35."""
    from rigid_body_light_amd import RigidBody, load_structure
    if rigid_config is None:
        rigid_config = load_structure(12)[1]
    return RigidBody(rigid_config, X, Q, a=a, eta=eta, dt=dt, wall_PC=wall_PC, block_PC=block_PC)
