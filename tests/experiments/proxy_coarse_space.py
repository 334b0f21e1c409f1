"""Experiment: a DIPOLE-level body space for both Krylov solves.  two_level_saddle_pc.py showed that the monopole model
cannot help the saddle solve (its correction lies in the range of K); what block-Jacobi misses there is the non-rigid far
field.  Here every body carries FOUR proxy points (a regular tetrahedron of circumradius s R around the centre): a blob's
velocity is the affine interpolation of the proxies' velocities (barycentric weights, exact for linear fields), forces are
anterpolated with the transpose -- total force and first moment are kept, i.e. monopole + full dipole (rotlet and stresslet):
    M~ = D + W C W^T,   W: blobs x 12 N_bod (block diagonal),  C: pair tensor of the 4 N_bod proxies, same-body blocks dropped.
Counts, for the block-diagonal, the monopole and the proxy model:  GMRES iterations to 1e-8 of the saddle solve with
P^-1 [s; f]: y = M~^-1 s, U = N~ (f - K^T y), lambda = y + M~^-1 K U;  Lanczos iterations (full re-orthogonalisation) of the
root preconditioned with a factor of M~ until the Euclidean error of the increment is below 1e-3.
Dense numpy on the CPU oracle's mobility; 27 bodies above a wall at the centre spacing of BASELINE cfg 3 in shell radii (2.635)
and at make_config's spacing for this resolution.   python tests/experiments/proxy_coarse_space.py [blobs_per_body]"""
import os, sys
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import Oracle                                      # noqa: E402
from oracle import oracle as onp                               # noqa: E402
from rigid_body_light_amd.synth import load_structure          # noqa: E402

nblb = int(sys.argv[1]) if len(sys.argv) > 1 else 42
orc = Oracle()
params, cfg0 = load_structure(nblb)
a = params["sep"] / 2.0
cfg = onp.remove_mean(cfg0)
Rb = np.linalg.norm(cfg.reshape(-1, 3), axis=1).max()
nb = 27
rng = np.random.default_rng(0)
m = 3 * nblb
tet = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float) / np.sqrt(3.0)


def gmres(A, Pinv, b, tol=1e-8, maxit=200):
    n = b.size
    V = np.zeros((maxit + 1, n)); H = np.zeros((maxit + 1, maxit))
    beta = np.linalg.norm(b); V[0] = b / beta
    for j in range(maxit):
        w = A(Pinv(V[j]))
        for _ in range(2):
            h = V[: j + 1] @ w; w = w - h @ V[: j + 1]; H[: j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w); V[j + 1] = w / H[j + 1, j]
        e1 = np.zeros(j + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e1, rcond=None)
        if np.linalg.norm(H[: j + 2, : j + 1] @ y - e1) / beta < tol:
            return j + 1
    return maxit


def lanczos_count(S, G, W, tol, mmax=40):
    n = W.size
    V = np.zeros((mmax + 1, n)); H = np.zeros((mmax + 1, mmax))
    wn = np.linalg.norm(W); V[0] = W / wn
    xs = []
    for it in range(mmax):
        u = S(V[it])
        for _ in range(2):
            h = V[: it + 1] @ u; u = u - h @ V[: it + 1]; H[: it + 1, it] += h
        H[it + 1, it] = np.linalg.norm(u); V[it + 1] = u / H[it + 1, it]
        T = 0.5 * (H[: it + 1, : it + 1] + H[: it + 1, : it + 1].T)
        lam, Y = np.linalg.eigh(T)
        xs.append(G((wn * (Y @ (np.sqrt(np.clip(lam, 0, None)) * Y[0]))) @ V[: it + 1]))
    err = [np.linalg.norm(x - xs[-1]) / np.linalg.norm(xs[-1]) for x in xs]
    return next(i for i, e in enumerate(err) if e < tol) + 1


print("| centre spacing / shell radius | model in both preconditioners | GMRES iterations to 1e-8 | Lanczos to 1e-3 | Lanczos to 1e-6 | smallest eigenvalue of the model |")
print("|---|---|---|---|---|---|")
for spacing in (2.635, 2.0 * (1.0 + a) + 0.5):
    idx = np.arange(nb)
    X = np.stack([idx % 3, (idx // 3) % 3, idx // 9], axis=1).astype(float) * spacing + rng.uniform(-0.1, 0.1, (nb, 3)) * min(1.0, (spacing - 2 - 2 * a) / 0.5)
    X[:, 2] += 1.0 + a + 0.3
    Q4 = rng.standard_normal((nb, 4)); Q4 /= np.linalg.norm(Q4, axis=1)[:, None]
    r = orc.multi_body_pos(X, Q4, cfg)
    B = orc.damp(r, a)
    M = (B[:, None] * orc.rotne_prager_tensor(r, a, 1.0, True)) * B[None, :]
    K = onp.K_matrix(X, Q4, cfg)
    n3 = m * nb
    A = lambda x: np.concatenate([M @ x[:n3] - K @ x[n3:], K.T @ x[:n3]])
    rhs = np.concatenate([rng.standard_normal(n3), -np.tile([0, 0, -1.0, 0, 0, 0], nb)])
    Wn = rng.standard_normal(n3)
    D = np.zeros_like(M)
    for b in range(nb):
        s = slice(m * b, m * (b + 1))
        D[s, s] = M[s, s]
    pos = r.reshape(nb, nblb, 3)
    models = {"block diagonal (reference)": D}
    Kt = np.zeros((n3, 3 * nb))
    for b in range(nb):
        for d in range(3):
            Kt[m * b + d:m * (b + 1):3, 3 * b + d] = 1.0
    Cs = orc.rotne_prager_tensor(X.reshape(-1), params["Rh"], 1.0, True)
    for b in range(nb):
        Cs[3 * b:3 * b + 3, 3 * b:3 * b + 3] = 0.0
    models["monopole: spheres at the body centres"] = D + Kt @ Cs @ Kt.T
    for sc, ap in ((0.4, a), (0.4, 0.5 * Rb), (0.25, 0.5 * Rb)):
        P = X[:, None, :] + sc * Rb * tet[None, :, :]
        W = np.zeros((n3, 12 * nb))
        for b in range(nb):
            wts = 0.25 + 0.75 * ((pos[b] - X[b]) @ tet.T) / (sc * Rb)          # nblb x 4, barycentric weights in closed form
            for p in range(4):
                for d in range(3):
                    W[m * b + d:m * (b + 1):3, 12 * b + 3 * p + d] = wts[:, p]
        C = orc.rotne_prager_tensor(P.reshape(-1), ap, 1.0, True)
        for b in range(nb):
            C[12 * b:12 * b + 12, 12 * b:12 * b + 12] = 0.0
        models["4 proxies per body, tetrahedron %.2f R, proxy radius %.2f" % (sc, ap)] = D + W @ C @ W.T
    for name, Mt in models.items():
        emin = np.linalg.eigvalsh(Mt).min()
        Mi = np.linalg.inv(Mt)
        MiK = Mi @ K
        Nn = np.linalg.inv(K.T @ MiK)

        def Pinv(x):
            s, f = x[:n3], x[n3:]
            y = Mi @ s
            U = Nn @ (f - K.T @ y)
            return np.concatenate([y + MiK @ U, U])

        it = gmres(A, Pinv, rhs)
        if emin > 0:
            G = np.linalg.cholesky(Mt)
            Gi = sla.solve_triangular(G, np.eye(n3), lower=True)
            l3 = lanczos_count(lambda v: Gi @ (M @ (Gi.T @ v)), lambda z: G @ z, Wn, 1e-3)
            l6 = lanczos_count(lambda v: Gi @ (M @ (Gi.T @ v)), lambda z: G @ z, Wn, 1e-6)
        else:
            l3 = l6 = -1
        print("| %.3f | %s | %d | %d | %d | %.1e |" % (spacing / Rb if False else spacing, name, it, l3, l6, emin), flush=True)
