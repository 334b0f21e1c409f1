"""Experiment: a TWO-LEVEL factor for the preconditioned Lanczos root.  The block-Jacobi factor G = B L leaves the
body-body far field to the Krylov iteration; in the Euclidean norm of the increment the slowly converging part is the
collective (rigid-translation) modes of the bodies.  Monopole model of the far field between bodies b != b':
M_bb' ~ K_t,b T(X_b - X_b') K_t,b'^T with the 3 x 3 pair tensor T of spheres of the bodies' hydrodynamic radius, i.e.
    M~ = D + K_t C K_t^T = L (I + Q E Q^T) L^T,   Q_b = L_b^-1 K_t,b R_b^-1/2  (orthonormal, 3 columns per body),
    R_b = K_t,b^T M_b^-1 K_t,b,   E = R^1/2 C R^1/2   (3 N_bod square),
and with the Cholesky factor I + E = L_E L_E^T:   G = B L (I + Q (L_E - I) Q^T),   G^-1 = (I + Q (L_E^-1 - I) Q^T) L^-1 B^-1.
Any invertible G keeps x = G (G^-1 B M B G^-T)^{1/2} W an exact root.  Counts Lanczos iterations (full re-orthogonalisation)
until the Euclidean error of the increment (against the same recurrence run to 90 iterations) is below 1e-3 / 1e-6.
Dense numpy on the CPU oracle's mobility.   python tests/experiments/two_level_root.py [blobs_per_body]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import Oracle                                      # noqa: E402
from oracle import oracle as onp                               # noqa: E402
from rigid_body_light_amd.synth import load_structure          # noqa: E402

nblb = int(sys.argv[1]) if len(sys.argv) > 1 else 162
nb = 27
orc = Oracle()
params, cfg0 = load_structure(nblb)
a = params["sep"] / 2.0
cfg = onp.remove_mean(cfg0)
rng = np.random.default_rng(0)
m = 3 * nblb


def lanczos_iterates(S, W, mmax):
    n = W.size
    V = np.zeros((mmax + 1, n))
    H = np.zeros((mmax + 1, mmax))
    wn = np.linalg.norm(W); V[0] = W / wn
    Z = []
    for it in range(mmax):
        u = S(V[it])
        for _ in range(2):
            h = V[: it + 1] @ u; u = u - h @ V[: it + 1]; H[: it + 1, it] += h
        H[it + 1, it] = np.linalg.norm(u); V[it + 1] = u / H[it + 1, it]
        T = 0.5 * (H[: it + 1, : it + 1] + H[: it + 1, : it + 1].T)
        lam, Y = np.linalg.eigh(T)
        y = wn * (Y @ (np.sqrt(np.clip(lam, 0, None)) * Y[0]))
        Z.append(y @ V[: it + 1])
    return Z


print("| lattice gap | factor | iterations to 1e-3 (Euclidean error of the increment) | to 1e-6 | energy-norm error at the 1e-3 iteration |")
print("|---|---|---|---|---|")
for gap_a in (3.8, 7.4):
    spacing = 2.0 * (1.0 + a) + gap_a * a
    idx = np.arange(nb)
    X = np.stack([idx % 3, (idx // 3) % 3, idx // 9], axis=1).astype(float) * spacing + rng.uniform(-0.1, 0.1, (nb, 3)) * min(1.0, gap_a / 4.0)
    X[:, 2] += 1.0 + a + 0.3
    Q4 = rng.standard_normal((nb, 4)); Q4 /= np.linalg.norm(Q4, axis=1)[:, None]
    r = orc.multi_body_pos(X, Q4, cfg)
    B = orc.damp(r, a)
    M = orc.rotne_prager_tensor(r, a, 1.0, True)                   # undamped, wall-corrected
    n3 = m * nb
    Ls = [np.linalg.cholesky(M[m * b:m * (b + 1), m * b:m * (b + 1)]) for b in range(nb)]
    Linv = [np.linalg.inv(L) for L in Ls]

    def bd(mats, v, T=False):
        return np.concatenate([(mats[b].T if T else mats[b]) @ v[m * b:m * (b + 1)] for b in range(nb)])

    # coupling of the bodies as spheres of their hydrodynamic radius R_h (structure file header): the pair tensor itself,
    # evaluated at the body centres -- R_b^-1 + C is then (close to) the RPY matrix of those spheres, which is SPD
    Mc = orc.rotne_prager_tensor(X.reshape(-1), params["Rh"], 1.0, True)
    C = Mc.copy()
    for b in range(nb):
        C[3 * b:3 * b + 3, 3 * b:3 * b + 3] = 0.0
    Kt = np.zeros((n3, 3 * nb))
    for b in range(nb):
        for d in range(3):
            Kt[m * b + d:m * (b + 1):3, 3 * b + d] = 1.0
    Zm = np.stack([bd(Linv, Kt[:, j]) for j in range(3 * nb)], axis=1)      # L^-1 K_t
    R = Zm.T @ Zm                                                          # block diagonal 3 x 3
    Rh = np.zeros_like(R); Rih = np.zeros_like(R)
    for b in range(nb):
        w, U = np.linalg.eigh(R[3 * b:3 * b + 3, 3 * b:3 * b + 3])
        Rh[3 * b:3 * b + 3, 3 * b:3 * b + 3] = U @ np.diag(np.sqrt(w)) @ U.T
        Rih[3 * b:3 * b + 3, 3 * b:3 * b + 3] = U @ np.diag(1.0 / np.sqrt(w)) @ U.T
    Qm = Zm @ Rih
    E = Rh @ C @ Rh
    lamE = np.linalg.eigvalsh(E)
    LE = np.linalg.cholesky(np.eye(3 * nb) + E)
    LEi = np.linalg.inv(LE)
    FE, FEi = LE - np.eye(3 * nb), LEi - np.eye(3 * nb)

    variants = {
        "block-Jacobi  G = B L": (lambda v: bd(Linv, v), lambda v: bd(Linv, v, T=True), lambda z: B * bd(Ls, z)),
        "two-level  G = B L (I + Q (L_E - I) Q^T), eigenvalues of E in [%.2f, %.2f]" % (lamE.min(), lamE.max()): (
            lambda v: (lambda w: w + Qm @ (FEi @ (Qm.T @ w)))(bd(Linv, v)),                  # G^-1 (without B)
            lambda v: bd(Linv, v + Qm @ (FEi.T @ (Qm.T @ v)), T=True),                       # G^-T
            lambda z: B * bd(Ls, z + Qm @ (FE @ (Qm.T @ z)))),                                # G z
    }
    W = rng.standard_normal(n3)
    for name, (Gi, GiT, G) in variants.items():
        S = lambda v: Gi(M @ GiT(v))
        Z = lanczos_iterates(S, W, 90)
        xs = [G(z) for z in Z]
        xref, zref = xs[-1], Z[-1]
        err = [np.linalg.norm(x - xref) / np.linalg.norm(xref) for x in xs]
        een = [np.linalg.norm(z - zref) / np.linalg.norm(zref) for z in Z]
        i3 = next(i for i, e in enumerate(err) if e < 1e-3) + 1
        i6 = next(i for i, e in enumerate(err) if e < 1e-6) + 1
        print("| %.1f a | %s | %d | %d | %.1e |" % (gap_a, name, i3, i6, een[i3 - 1]), flush=True)
