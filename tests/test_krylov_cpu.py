"""CPU tests of the torch comparator loops (tests/torch_krylov.py: what the GPU tests compare librbl's own GMRES / Lanczos
with) with plain torch operators standing in for the HIP ones."""
import numpy as np
import torch

from torch_krylov import gmres_right_pc, lanczos_mhalf, lanczos_mhalf_multi


def test_gmres_right_preconditioned_solves_spd_and_saddle_like_systems():
    rng = np.random.default_rng(0)
    n = 60
    A0 = rng.standard_normal((n, n)); M = A0 @ A0.T + n * np.eye(n)
    K = rng.standard_normal((n, 6))
    S = np.block([[M, -K], [K.T, np.zeros((6, 6))]])
    b = np.concatenate([np.zeros(n), rng.standard_normal(6)])
    St = torch.from_numpy(S)
    Pinv = torch.from_numpy(np.linalg.inv(np.block([[np.diag(np.diag(M)), -K], [K.T, np.zeros((6, 6))]])))
    x, m, resid = gmres_right_pc(lambda v: St @ v, lambda v: Pinv @ v, torch.from_numpy(b), iters=66, rtol=1e-12)
    assert resid < 1e-12 and m <= 66
    np.testing.assert_allclose(x.numpy(), np.linalg.solve(S, b), rtol=1e-8, atol=1e-10)
    # fixed-work form: no host check inside the loop, residual reported at the end
    x2, m2, r2 = gmres_right_pc(lambda v: St @ v, lambda v: Pinv @ v, torch.from_numpy(b), iters=10)
    assert m2 == 10 and r2 < 1.0
    # initial guess: the correction is solved for, the tolerance stays relative to |b|
    xs = np.linalg.solve(S, b)
    x3, m3, r3 = gmres_right_pc(lambda v: St @ v, lambda v: Pinv @ v, torch.from_numpy(b), iters=66, rtol=1e-12,
                                x0=torch.from_numpy(xs * (1.0 + 1e-6)))
    assert r3 < 1e-12 and m3 < m
    np.testing.assert_allclose(x3.numpy(), xs, rtol=1e-8, atol=1e-10)
    x4, m4, r4 = gmres_right_pc(lambda v: St @ v, lambda v: Pinv @ v, torch.from_numpy(b), iters=66, rtol=1e-9,
                                x0=torch.from_numpy(xs))
    assert m4 == 0 and r4 < 1e-9 and np.array_equal(x4.numpy(), xs)


def test_lanczos_square_root():
    rng = np.random.default_rng(1)
    n = 80
    A0 = rng.standard_normal((n, n)); M = A0 @ A0.T / n + np.eye(n)
    lam, V = np.linalg.eigh(M)
    W = rng.standard_normal(n)
    ref = V @ (np.sqrt(lam) * (V.T @ W))
    Mt = torch.from_numpy(M)
    y, m, ch = lanczos_mhalf(lambda v: Mt @ v, torch.from_numpy(W), max_iter=80, tol=1e-12)
    np.testing.assert_allclose(y.numpy(), ref, rtol=1e-8, atol=1e-10)
    assert m <= 80


def test_lanczos_multi_vector_lockstep():
    rng = np.random.default_rng(2)
    n, k = 70, 5
    A0 = rng.standard_normal((n, n)); M = A0 @ A0.T / n + np.eye(n)
    lam, V = np.linalg.eigh(M)
    W = rng.standard_normal((k, n))
    ref = (V @ (np.sqrt(lam)[:, None] * (V.T @ W.T))).T
    Mt = torch.from_numpy(M)
    Y, m, ch = lanczos_mhalf_multi(lambda X: X @ Mt, torch.from_numpy(W), max_iter=70, tol=1e-12)
    np.testing.assert_allclose(Y.numpy(), ref, rtol=1e-8, atol=1e-10)
