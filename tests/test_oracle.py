"""CPU tests of the ORACLE itself: pinned against the reference's compiled pair
kernels (golden fixture + live oracle/_ref when present) and analytic answers."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def unhex(v):
    return np.array([float.fromhex(x) for x in v])


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "pair_kernels.json")) as f:
        return json.load(f)


def test_rpy_bit_exact_vs_reference_golden(orc, golden):
    for c in golden["rpy"]:
        r = unhex(c["r"])
        out = orc.rpy(r[0], r[1], r[2], c["i"], c["j"], float.fromhex(c["inv_a"]))
        assert np.array_equal(out, unhex(c["out6"]))   # bit-exact


def test_wall_bit_exact_vs_reference_golden(orc, golden):
    for c in golden["wall"]:
        a = unhex(c["args"])
        out = orc.wall(a[0], a[1], a[2], unhex(c["M_in"]), c["i"], c["j"], a[3])
        assert np.array_equal(out, unhex(c["M_out"]))  # bit-exact
    assert golden["below_wall_throws"]


def test_assembly_blocks_bit_exact_vs_reference_golden(orc):
    """whole blocks as the reference's assembly loop forms them (:432-447) in the regimes of the BASELINE geometries
    (shell radii, h/a from 1e-6 to 1e3, |r| within 1e-9 of 2a with the wall term, image contact, equal heights):
    the oracle's block equals the reference's compiled kernels fed the same way, bit for bit"""
    with open(os.path.join(HERE, "golden", "pair_blocks_assembly.json")) as f:
        g = json.load(f)
    assert len(g["blocks"]) >= 600
    eta = 0.9
    for c in g["blocks"]:
        a = float.fromhex(c["a"])
        nf = 1.0 / (8.0 * np.pi * eta * a)
        out = orc.pair_block(unhex(c["ri"]), unhex(c["rj"]), c["i"], c["j"], a, eta, c["wall"])
        assert np.array_equal(out.reshape(-1), unhex(c["out9"]) * nf)


def test_survey_known_answers(orc):
    # SURVEY.md section 8c vectors, produced from the verbatim-compiled reference kernels
    np.testing.assert_allclose(
        orc.rpy(3, 0.5, -1, 0, 1, 1.0),
        [0.55340573566317486, 0.036790487631105304, -0.073580975262210607,
         0.33879455781506052, -0.012263495877035102, 0.3571898016306132], rtol=0, atol=0)
    s = orc.rpy(0.7, 0.2, 0.4, 0, 1, 1.0)
    np.testing.assert_allclose(
        s, [1.0955712734889391, 0.021067524290009614, 0.042135048580019228,
            1.0278542311281937, 0.012038585308576922, 1.0459121090910592], rtol=0, atol=0)
    M = np.array([s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]])
    w = orc.wall(0.7, 0.2, 0.4 + 5.0, M, 0, 1, 2.5)
    np.testing.assert_allclose(
        w, [0.83057310683815733, 0.021261300613358222, 0.011983439270544221,
            0.021261300613358222, 0.76223321200950567, 0.0034238397915840602,
            0.06882509098002651, 0.019664311708579001, 0.53788744924214216], rtol=0, atol=0)
    self_ = orc.wall(0, 0, 5.0, np.diag([4 / 3] * 3).ravel(), 2, 2, 2.5)
    np.testing.assert_allclose(self_[[0, 4, 8]], [1.0431466666666667, 1.0431466666666667, 0.77429333333333328],
                               rtol=1e-15)
    with pytest.raises(RuntimeError, match="below the wall"):
        orc.wall(0.1, 0.2, 0.3, np.zeros(9), 0, 1, -0.1)


def test_live_reference_pair_kernels_if_built(orc):
    from oracle import RefPair, ref_lib_path
    if not os.path.exists(ref_lib_path()) and not os.path.exists("/root/reference"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    ref = RefPair()
    rng = np.random.default_rng(7)
    for _ in range(20000):
        v = rng.uniform(-5, 5, 3)
        a = float(rng.uniform(0.05, 2.0))
        assert np.array_equal(orc.rpy(*v, 0, 1, 1 / a), ref.rpy(*v, 0, 1, 1 / a))
        M = rng.standard_normal(9)
        h = float(rng.uniform(0.01, 8))
        args = (v[0], v[1], abs(v[2]) + h)
        assert np.array_equal(orc.wall(*args, M, 0, 1, h), ref.wall(*args, M, 0, 1, h))


def test_analytic_isolated_blob_and_wall(orc):
    a, eta = 0.7, 1.3
    F = np.array([0.3, -1.0, 2.0])
    U = orc.apply_M(F, np.array([0.0, 0.0, 5.0]), a, eta, False)
    np.testing.assert_allclose(U, F / (6 * np.pi * eta * a), rtol=1e-15)
    # single blob above a wall: mu_par, mu_perp expansions
    for h in (1.2, 2.0, 7.5):
        Uw = orc.apply_M(F, np.array([0.0, 0.0, h * a]), a, eta, True)
        par = 1 - 9 / 16 / h + 1 / 8 / h ** 3 - 1 / 16 / h ** 5
        per = 1 - 9 / 8 / h + 1 / 2 / h ** 3 - 1 / 8 / h ** 5
        np.testing.assert_allclose(Uw, F * np.array([par, par, per]) / (6 * np.pi * eta * a), rtol=1e-13)


def test_rpy_continuous_at_2a_and_spd(orc):
    a = 0.5
    lo = orc.rpy(2 * a * (1 - 1e-13), 0, 0, 0, 1, 1 / a)
    hi = orc.rpy(2 * a * (1 + 1e-13), 0, 0, 0, 1, 1 / a)
    np.testing.assert_allclose(lo, hi, atol=1e-12)
    rng = np.random.default_rng(3)
    r = rng.uniform(0, 4, (40, 3)); r[:, 2] += 0.6
    for wall in (False, True):
        M = orc.rotne_prager_tensor(r, a, 1.0, wall)
        assert np.array_equal(M, M.T)                      # mirrored by construction (:451)
        assert np.linalg.eigvalsh(M).min() > 0


def test_dense_matfree_rows_agree(orc):
    rng = np.random.default_rng(5)
    r = rng.uniform(0, 6, (90, 3)); r[:, 2] += 0.3
    F = rng.standard_normal(270)
    for wall in (False, True):
        Ud = orc.apply_M(F, r, 0.4, 1.1, wall, mode="dense")
        Um = orc.apply_M(F, r, 0.4, 1.1, wall, mode="matfree")
        Ur = orc.apply_M_rows(F, r, 0, 90, 0.4, 1.1, wall, nthreads=2)
        np.testing.assert_allclose(Um, Ud, rtol=0, atol=1e-14 * np.abs(Ud).max())
        np.testing.assert_allclose(Ur, Ud, rtol=0, atol=1e-14 * np.abs(Ud).max())
        Us = orc.apply_M_rows(F, r, 17, 55, 0.4, 1.1, wall)
        np.testing.assert_allclose(Us, Ud[51:165], rtol=0, atol=1e-14 * np.abs(Ud).max())


def test_damp_and_wall_flag_roles(orc):
    a = 1.0
    r = np.array([[0, 0, 0.4], [3, 0, 2.0], [0, 3, 1.0]])
    np.testing.assert_array_equal(orc.damp(r, a), [0.4] * 3 + [1.0] * 6)
    with pytest.raises(RuntimeError, match="below the wall"):
        orc.apply_M(np.ones(9), np.array([[0, 0, -0.1], [3, 0, 2.0], [0, 3, 1.0]]), a, 1.0, True)
    with pytest.raises(RuntimeError, match="overlap"):
        orc.apply_M(np.ones(6), np.zeros(6), a, 1.0, False)


def test_cholesky_and_M_half_W_vs_numpy(orc):
    rng = np.random.default_rng(11)
    a, eta = 0.5, 1.0
    g = np.arange(60)
    W = rng.standard_normal(180)
    for wall in (False, True):
        # free-space: blobs in the damp zone z < a exercise B (always applied, :668);
        # with the wall term the Swan-Brady self mobility turns negative for z < a
        # (B M B is then not SPD and Eigen::LLT would return garbage), so keep z >= 1.2 a
        zlo = 0.6 if wall else 0.25
        r = np.stack([1.2 * (g % 8), 1.2 * (g // 8), rng.uniform(zlo, 3.0, 60)], axis=1)  # no overlaps
        assert wall or (r[:, 2] < a).sum() >= 3
        out, L = orc.M_half_W(r, a, eta, wall, W, return_L=True)
        B = orc.damp(r, a)
        M = (B[:, None] * orc.rotne_prager_tensor(r, a, eta, wall)) * B[None, :]
        Lnp = np.linalg.cholesky(M)
        np.testing.assert_allclose(L, Lnp, rtol=0, atol=1e-13)
        np.testing.assert_allclose(out, Lnp @ W, rtol=0, atol=1e-12)
        np.testing.assert_allclose(L @ L.T, M, rtol=0, atol=1e-14)


def test_blob_positions_vs_scipy(orc, shell12):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(2)
    X = rng.uniform(-5, 5, (6, 3)); Q = rng.standard_normal((6, 4))
    cfg = shell12 - shell12.mean(axis=0)
    pos = orc.multi_body_pos(X, Q, cfg).reshape(6, 12, 3)
    for b in range(6):
        ref = Rotation.from_quat(Q[b], scalar_first=True).apply(cfg) + X[b]
        np.testing.assert_allclose(pos[b], ref, atol=1e-13)
