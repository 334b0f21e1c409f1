"""A pin of the wall term's INDEX ROLES that does not come from the reference's text or from my reading of it: for blob
radius a -> 0 the wall-corrected Rotne-Prager-Yamakawa pair mobility must become Blake's image-system Green's function
(J. R. Blake, Proc. Camb. Phil. Soc. 70 (1971) 303: Stokeslet + image Stokeslet + source doublet + Stokeslet doublet),
which vanishes on the wall and in which the height h that multiplies the doublets is the SOURCE's (the blob the force
acts on: j of M_ij, c_rigid_obj.cpp:425-443 `h = r_vectors[j][2]`).  A restatement with the roles of i and j exchanged
in the wall term, or without the transposed mirror of :447-452, differs from Blake's tensor by O(1) for blobs at different
heights; the finite-size terms of Swan & Brady (Phys. Fluids 19 (2007) 113306) are O(a^2 / r^2).

CPU: the oracle's dense assembly (and its matrix-free rows).  GPU: the HIP product on the same blobs."""
import numpy as np
import pytest


def blake(x, y, eta):
    """G[i, j]: velocity component i at x per unit force component j at y, no-slip wall z = 0 (Blake 1971, eq. 2.5)"""
    x = np.asarray(x, float); y = np.asarray(y, float)
    h = y[2]
    r = x - y
    R = x - np.array([y[0], y[1], -h])
    rn, Rn = np.linalg.norm(r), np.linalg.norm(R)
    I = np.eye(3)
    G = (I / rn + np.outer(r, r) / rn ** 3) - (I / Rn + np.outer(R, R) / Rn ** 3)
    e3 = np.array([0.0, 0.0, 1.0])
    for j in range(3):
        s = 1.0 if j < 2 else -1.0
        # d/dR_j of  h R_i / R^3 - (delta_i3 / R + R_i R_3 / R^3)
        d = (h * (I[:, j] / Rn ** 3 - 3.0 * R * R[j] / Rn ** 5) + e3 * R[j] / Rn ** 3
             - (I[:, j] * R[2] + R * e3[j]) / Rn ** 3 + 3.0 * R * R[2] * R[j] / Rn ** 5)
        G[:, j] += 2.0 * h * s * d
    return G / (8.0 * np.pi * eta)


def test_blake_tensor_restatement_is_sound():
    """the comparator itself: no slip on the wall, Lorentz reciprocity G(x, y) = G(y, x)^T"""
    rng = np.random.default_rng(0)
    for _ in range(20):
        y = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.3, 3.0)])
        xw = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), 0.0])
        assert np.abs(blake(xw, y, 1.3)).max() < 1e-14
        x = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.3, 3.0)])
        assert np.abs(blake(x, y, 1.3) - blake(y, x, 1.3).T).max() < 1e-13 * np.abs(blake(x, y, 1.3)).max() + 1e-15


def _pairs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        x = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.2, 3.0)])
        y = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.2, 3.0)])
        if np.linalg.norm(x - y) > 0.5 and abs(x[2] - y[2]) > 0.3:        # clearly different heights: the roles matter
            out.append((x, y))
    return out


def test_oracle_wall_mobility_tends_to_blake_tensor():
    from oracle import Oracle
    orc = Oracle()
    eta = 0.8
    for a, tol in ((1e-2, 3e-3), (1e-3, 3e-5)):                 # the finite-size terms fall like a^2
        worst = 0.0
        for x, y in _pairs(25, 1):
            r = np.concatenate([x, y])
            M = orc.rotne_prager_tensor(r, a, eta, True)
            G = blake(x, y, eta)
            Mij, Mji = M[0:3, 3:6], M[3:6, 0:3]                 # velocity of blob 0 (at x) per force on blob 1 (at y), and back
            scale = np.abs(G).max()
            worst = max(worst, np.abs(Mij - G).max() / scale, np.abs(Mji - G.T).max() / scale)
            # the check discriminates: Blake's tensor with source and field point exchanged is a different matrix
            assert np.abs(blake(y, x, eta) - G).max() / scale > 1e-2
            # the matrix-free rows say the same as the dense assembly
            F = np.zeros(6); F[3:] = [0.3, -0.7, 0.5]
            U = orc.apply_M_rows(F, r, 0, 1, a, eta, True)
            # (apply_M is B M B; both blobs sit higher than a, B = 1)
            assert np.abs(U - G @ F[3:]).max() < tol * scale * np.abs(F).max() * 3
        assert worst < tol, (a, worst)


@pytest.mark.gpu
def test_hip_wall_product_tends_to_blake_tensor(shell12):
    """the same limit through the C ABI: U_i = M_ij F_j of the HIP product for two blobs at different heights"""
    from rigid_body_light_amd import RigidBody
    eta, a = 0.8, 1e-3
    # (the object needs a rigid body to exist -- any will do: apply_M takes the blob set it is given)
    rb = RigidBody(shell12, np.array([[0.0, 0.0, 3.0]]), np.array([[1.0, 0.0, 0.0, 0.0]]), a, eta, 0.01, wall_PC=True)
    worst = 0.0
    for x, y in _pairs(25, 2):
        r = np.concatenate([x, y])
        G = blake(x, y, eta)
        scale = np.abs(G).max()
        for j in range(3):
            F = np.zeros(6); F[3 + j] = 1.0
            U = rb.apply_M(F, r)                               # any blob set may be passed (tests/test_interface.py:171-177)
            worst = max(worst, np.abs(U[0:3] - G[:, j]).max() / scale)
            F = np.zeros(6); F[j] = 1.0                          # force on the blob at x, velocity of the blob at y
            U = rb.apply_M(F, r)
            worst = max(worst, np.abs(U[3:6] - G.T[:, j]).max() / scale)
    assert worst < 3e-5, worst


def test_single_blob_above_a_wall_has_the_published_self_mobility():
    """Swan & Brady (2007), translational self mobility of a sphere of radius a at height h (Rotne-Prager level), eps = a / h:
    parallel  mu0 [1 - 9/16 eps + 1/8 eps^3 - 1/16 eps^5],  perpendicular  mu0 [1 - 9/8 eps + 1/2 eps^3 - 1/8 eps^5],
    mu0 = 1 / (6 pi eta a).  The oracle's one-blob matrix must be exactly that (and diagonal)."""
    from oracle import Oracle
    orc = Oracle()
    eta = 1.7
    for a, h in ((0.5, 0.6), (0.5, 3.0), (1.0, 1.0), (0.1, 25.0)):
        M = orc.rotne_prager_tensor(np.array([0.3, -0.2, h]), a, eta, True)
        e = a / h
        mu0 = 1.0 / (6.0 * np.pi * eta * a)
        par = mu0 * (1.0 - 9.0 / 16.0 * e + e ** 3 / 8.0 - e ** 5 / 16.0)
        perp = mu0 * (1.0 - 9.0 / 8.0 * e + e ** 3 / 2.0 - e ** 5 / 8.0)
        assert np.abs(M - np.diag([par, par, perp])).max() < 1e-14 * mu0
