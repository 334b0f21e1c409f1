"""Experiment: does the TWO-LEVEL model of the mobility that shortened the Lanczos root (two_level_root.py) also shorten
the GMRES saddle solve when it replaces the block-diagonal M~ of the reference's preconditioner (c_rigid_obj.cpp:589-616)?
    M~ = D + K_t C K_t^T   (D: the bodies' own blocks; C: pair tensor of spheres at the body centres, off-diagonal),
    N~ = (K^T M~^-1 K)^-1  (dense 6 N_bod square: D_N + C_N^T ((I + E)^-1 - I) C_N, both factors block diagonal),
    P^-1 [s; f]:  y = M~^-1 s,  U = N~ (f - K^T y),  lambda = y + M~^-1 K U.
Dense numpy on the CPU oracle's mobility; 3 x 3 x 3 bodies above a wall, lattice gaps 3.8 a / 7.4 a, right-preconditioned
GMRES to 1e-8 on [M -K; K^T 0] with a Brownian-type right-hand side.   python tests/experiments/two_level_saddle_pc.py [blobs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import Oracle                                      # noqa: E402
from oracle import oracle as onp                               # noqa: E402
from rigid_body_light_amd.synth import load_structure          # noqa: E402

nblb = int(sys.argv[1]) if len(sys.argv) > 1 else 162
orc = Oracle()
params, cfg0 = load_structure(nblb)
a = params["sep"] / 2.0
cfg = onp.remove_mean(cfg0)
nb = 27
rng = np.random.default_rng(0)
m = 3 * nblb


def gmres(A, Pinv, b, tol=1e-8, maxit=200):
    n = b.size
    V = np.zeros((maxit + 1, n)); H = np.zeros((maxit + 1, maxit))
    beta = np.linalg.norm(b); V[0] = b / beta
    hist = []
    for j in range(maxit):
        w = A(Pinv(V[j]))
        for _ in range(2):
            h = V[: j + 1] @ w; w = w - h @ V[: j + 1]; H[: j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w); V[j + 1] = w / H[j + 1, j]
        e1 = np.zeros(j + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e1, rcond=None)
        res = np.linalg.norm(H[: j + 2, : j + 1] @ y - e1) / beta
        hist.append(res)
        if res < tol:
            break
    return len(hist), hist


print("| lattice gap | model in the preconditioner | GMRES iterations to 1e-8 | to 1e-4 |")
print("|---|---|---|---|")
for gap_a in (3.8, 7.4):
    spacing = 2.0 * (1.0 + a) + gap_a * a
    idx = np.arange(nb)
    X = np.stack([idx % 3, (idx // 3) % 3, idx // 9], axis=1).astype(float) * spacing + rng.uniform(-0.1, 0.1, (nb, 3)) * min(1.0, gap_a / 4.0)
    X[:, 2] += 1.0 + a + 0.3
    Q4 = rng.standard_normal((nb, 4)); Q4 /= np.linalg.norm(Q4, axis=1)[:, None]
    r = orc.multi_body_pos(X, Q4, cfg)
    B = orc.damp(r, a)
    M = (B[:, None] * orc.rotne_prager_tensor(r, a, 1.0, True)) * B[None, :]
    K = onp.K_matrix(X, Q4, cfg)
    n3 = m * nb
    A = lambda x: np.concatenate([M @ x[:n3] - K @ x[n3:], K.T @ x[:n3]])
    rhs = np.concatenate([rng.standard_normal(n3), -np.tile([0, 0, -1.0, 0, 0, 0], nb)])
    Dinv = np.zeros_like(M)
    for b in range(nb):
        s = slice(m * b, m * (b + 1))
        Dinv[s, s] = np.linalg.inv(M[s, s])
    Kt = np.zeros((n3, 3 * nb))
    for b in range(nb):
        for d in range(3):
            Kt[m * b + d:m * (b + 1):3, 3 * b + d] = 1.0
    Mc = orc.rotne_prager_tensor(X.reshape(-1), params["Rh"], 1.0, True)
    C = Mc.copy()
    for b in range(nb):
        C[3 * b:3 * b + 3, 3 * b:3 * b + 3] = 0.0
    models = {"block diagonal (reference)": Dinv}
    # Woodbury:  (D + Kt C Kt^T)^-1 = D^-1 - D^-1 Kt (C^-1 + Kt^T D^-1 Kt)^-1 Kt^T D^-1 = D^-1 - Z (I + C R)^-1 C Z^T,  Z = D^-1 Kt
    Z = Dinv @ Kt
    R = Kt.T @ Z
    models["two-level: monopole coupling of the body centres"] = Dinv - Z @ np.linalg.solve(np.eye(3 * nb) + C @ R, C @ Z.T)
    models["exact M^-1 (bound)"] = np.linalg.inv(M)
    for name, Mi in models.items():
        MiK = Mi @ K
        N = np.linalg.inv(K.T @ MiK)

        def Pinv(x):
            s, f = x[:n3], x[n3:]
            y = Mi @ s
            U = N @ (f - K.T @ y)
            return np.concatenate([y + MiK @ U, U])

        it, hist = gmres(A, Pinv, rhs)
        i4 = next(i for i, e in enumerate(hist) if e < 1e-4) + 1
        print("| %.1f a | %s | %d | %d |" % (gap_a, name, it, i4), flush=True)
