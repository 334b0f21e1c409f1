"""The scenarios of the reference's own test-suite (tests/test_interface.py, test_precision.py,
test_wall.py, test_import.py), replayed through the drop-in package name `Rigid` on the GPU build.
Same scenario names; where the reference only asserts "norm > 0" these also check the value against
the oracle or numpy, so a drop-in user's suite and ours fail for the same reasons."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import random_positions

pytestmark = pytest.mark.gpu
BLOBS = 12        # shell_N_12, the structure the reference tests use


def _solver(shell12, n, wall=False, block=False, dtype=np.float64, seed=0, dt=1.0):
    from Rigid import RigidBody                          # the reference's import line
    X, Q = random_positions(n, wall=wall, seed=seed)
    X, Q = X.astype(dtype), Q.astype(dtype)
    return RigidBody(shell12, X, Q, a=1.0, eta=1.0, dt=dt, wall_PC=wall, block_PC=block), X, Q


def _rng(seed):
    return np.random.default_rng(1000 + seed)


def test_import():
    import Rigid
    assert hasattr(Rigid, "RigidBody") and Rigid.c_rigid.CManyBodies.precision == "double"


def test_create(shell12):
    from Rigid import RigidBody
    X, Q = _rng(0).standard_normal((10, 3)), _rng(1).standard_normal((10, 4))
    for kw in ({}, {"wall_PC": True}, {"block_PC": True}):
        assert RigidBody(shell12, X, Q, 1.0, 1.0, dt=0.01, **kw).total_blobs == 10 * BLOBS
    with pytest.raises(RuntimeError):                     # a configuration that is not (N_blobs, 3)
        RigidBody(shell12.flatten()[:-1], X, Q, 1.0, 1.0, dt=0.01)


def test_config(shell12):
    cb, X, Q = _solver(shell12, 10, seed=2)
    Q = _rng(2).uniform(size=(10, 4))                     # un-normalised on purpose
    cb.set_config(X, Q)
    Xg, Qg = cb.get_config()
    assert np.allclose(Xg, X)
    assert np.allclose(Qg, Q / np.linalg.norm(Q, axis=1, keepdims=True))     # stored normalised, scalar first


def test_bad_config(shell12):
    cb, X, Q = _solver(shell12, 10, seed=3)
    for bad in ((X, Q[:-1]), (X[:-1], Q)):
        with pytest.raises(RuntimeError):
            cb.set_config(*bad)


def test_blob_positions(shell12):
    cb, X, Q = _solver(shell12, 5, seed=4)
    pos = cb.get_blob_positions()
    assert pos.shape == (5 * BLOBS, 3)
    cfg = shell12 - shell12.mean(axis=0)
    want = np.concatenate([Rotation.from_quat(q, scalar_first=True).apply(cfg) + x for x, q in zip(X, Q)])
    assert np.allclose(pos, want, atol=1e-12)             # the reference asks for 1e-5


@pytest.mark.parametrize("which", ["K_dot", "KT_dot"])
def test_K_dot_and_KT_dot(shell12, which):
    from oracle import oracle as onp
    cb, X, Q = _solver(shell12, 3, seed=5)
    K = onp.K_matrix(X, onp.normalize_quats(Q), onp.remove_mean(shell12))
    n_in, shape, op = ((18, (3 * BLOBS, 3), K) if which == "K_dot" else (9 * BLOBS, (6, 3), K.T))
    with pytest.raises(RuntimeError):
        getattr(cb, which)(_rng(5).standard_normal(n_in - 3))
    v = _rng(6).standard_normal(n_in)
    out = getattr(cb, which)(v)
    assert out.shape == shape
    assert np.allclose(out.reshape(-1), op @ v, atol=1e-12)


def test_get_K_Kinv(shell12):
    cb, X, Q = _solver(shell12, 3, seed=7)
    K, Kinv = cb.get_K(), cb.get_Kinv()
    assert K.shape == (9 * BLOBS, 18) and Kinv.shape == (18, 9 * BLOBS)
    assert np.allclose((Kinv @ K).toarray(), np.eye(18), atol=1e-10)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("block_PC,wall_PC", [(False, False), (True, False), (False, True), (True, True)])
def test_apply_PC(shell12, block_PC, wall_PC, dtype):     # test_apply_PC + test_pc_precision
    cb, X, Q = _solver(shell12, 3, wall=wall_PC, block=block_PC, dtype=dtype, seed=8)
    size = 9 * BLOBS + 18
    out = cb.apply_PC(_rng(8).standard_normal(size).astype(dtype))
    assert out.shape == (size,) and out.dtype == np.float64 and np.linalg.norm(out) > 0.0
    with pytest.raises(RuntimeError):
        cb.apply_PC(_rng(9).standard_normal(size - 4))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_precision(shell12, dtype):
    cb, X, Q = _solver(shell12, 5, dtype=dtype, seed=10)
    cb.set_config(X, Q)
    ku = cb.K_dot(_rng(10).standard_normal(30).astype(dtype))
    ktl = cb.KT_dot(_rng(11).standard_normal(15 * BLOBS).astype(dtype))
    assert np.linalg.norm(ku) > 0.0 and np.linalg.norm(ktl) > 0.0 and cb.precision == "double"


def test_apply_M(orc, shell12):
    cb, X, Q = _solver(shell12, 2, seed=12)
    n3 = 6 * BLOBS
    F, pos = _rng(12).standard_normal(n3), cb.get_blob_positions()
    for bad in ((F[:-4], pos), (F, pos[:-3]), (F[:-1], pos.reshape(-1)[:-1])):
        with pytest.raises(RuntimeError):
            cb.apply_M(*bad)
    U = cb.apply_M(F, pos)
    assert U.shape == (n3,)
    assert np.linalg.norm(U - orc.apply_M(F, pos, 1.0, 1.0, False)) < 1e-12 * np.linalg.norm(U)
    # one blob more than the object owns (reference tests/test_interface.py:171-177)
    F1 = np.concatenate([F, _rng(13).standard_normal(3)])
    pos1 = np.concatenate([pos, _rng(14).uniform(1.0, 5.0, (1, 3))])
    U1 = cb.apply_M(F1, pos1)
    assert U1.shape == (n3 + 3,)
    assert np.linalg.norm(U1 - orc.apply_M(F1, pos1, 1.0, 1.0, False)) < 1e-12 * np.linalg.norm(U1)


def test_apply_saddle(shell12):
    cb, X, Q = _solver(shell12, 2, seed=15)
    size = 6 * BLOBS + 12
    x = _rng(15).standard_normal(size)
    out = cb.apply_saddle(x)
    lam, U = x[:6 * BLOBS], x[6 * BLOBS:]
    want = np.concatenate([cb.apply_M(lam, cb.get_blob_positions()) - cb.K_dot(U).reshape(-1), cb.KT_dot(lam).reshape(-1)])
    assert out.shape == (size,) and np.allclose(out, want, atol=1e-13)
    with pytest.raises(RuntimeError):
        cb.apply_saddle(x[:-2])


def test_evolve_rigid_bodies(shell12):
    cb, X, Q = _solver(shell12, 3, seed=16)
    U = _rng(16).standard_normal(18)
    U0 = U.copy()
    cb.evolve_rigid_bodies(U)
    Xn, Qn = cb.get_config()
    assert np.allclose(Xn, X + U.reshape(3, 6)[:, :3])    # dt = 1
    assert np.linalg.norm(Qn - Q) > 0.0 and np.allclose(np.linalg.norm(Qn, axis=1), 1.0)
    assert np.array_equal(U, U0)                           # the caller's array is not scaled by dt


def _one_body_at(shell12, z):
    from Rigid import RigidBody
    cb = RigidBody(shell12, np.array([[0.0, 0.0, z]]), np.array([[1.0, 0.0, 0.0, 0.0]]), a=1.0, eta=1.0, dt=1.0,
                   wall_PC=True)
    return cb, _rng(17).standard_normal(3 * BLOBS + 6)


def test_above_wall(shell12):
    cb, vec = _one_body_at(shell12, 1.0)
    for out in (cb.apply_PC(vec), cb.apply_saddle(vec), cb.apply_M(vec[:3 * BLOBS], cb.get_blob_positions())):
        assert np.all(np.isfinite(out)) and np.linalg.norm(out) > 0.0


def test_under_wall(shell12):
    cb, vec = _one_body_at(shell12, 0.0)
    for call in (lambda: cb.apply_saddle(vec), lambda: cb.apply_PC(vec),
                 lambda: cb.apply_M(vec[:3 * BLOBS], cb.get_blob_positions())):
        with pytest.raises(RuntimeError, match="below the wall"):
            call()


def test_one_call_steps_through_the_dropin_class(shell12):
    """beyond the reference's surface: RigidBody.step_deterministic / step_brownian move the bodies consistently
    with solving the saddle system by hand and calling evolve_rigid_bodies."""
    cb, X, Q = _solver(shell12, 4, wall=True, seed=20, dt=0.01)
    F = np.tile([0.0, 0.0, 1.0, 0.0, 0.0, 0.0], 4)
    # by hand: dense solve of the saddle operator columns, then evolve
    n3, nb6 = 3 * 4 * BLOBS, 24
    A = np.column_stack([cb.apply_saddle(e) for e in np.eye(n3 + nb6)])
    U = np.linalg.solve(A, np.concatenate([np.zeros(n3), -F]))[n3:]
    ref, _, _ = _solver(shell12, 4, wall=True, seed=20, dt=0.01)
    ref.evolve_rigid_bodies(U)
    it, res = cb.step_deterministic(F, max_iter=150, rtol=1e-11)
    assert res < 1e-11 and it > 0
    np.testing.assert_allclose(cb.get_config()[0], ref.get_config()[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(cb.get_config()[1], ref.get_config()[1], rtol=0, atol=1e-9)
    X1 = cb.get_config()[0].copy()
    it, res = cb.step_brownian(F, seed=3, max_iter=80, rtol=1e-8)
    assert res < 1e-8 and np.linalg.norm(cb.get_config()[0] - X1) > 0
