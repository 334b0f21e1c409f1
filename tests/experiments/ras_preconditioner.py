"""Experiment: would a NEAR-FIELD-COUPLED block preconditioner -- restricted additive Schwarz: a
body's block extended by the blobs of neighbouring bodies within a distance delta, solve on the extended set, keep the
body's own entries -- lower the GMRES iteration count of the saddle solve below block-Jacobi's?

Dense numpy on the CPU oracle's mobility (test infrastructure: the product never runs this).  3 x 3 x 3 bodies of
shell_N_162 above a wall, lattice gaps of 3.8 a (make_config's) and 7.4 a (the gap of BASELINE cfg 3 in blob radii);
right-preconditioned GMRES to 1e-8 on [M -K; K^T 0] with the Brownian-type right-hand side [random slip; -F], the
preconditioner P^-1 of c_rigid_obj.cpp:589-616 with its M^-1 replaced by the Schwarz operator (force-block sign restored
as in librbl's solver).  Prints one markdown table.   python tests/experiments/ras_preconditioner.py [blobs_per_body]"""
import os, sys, time
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
    for j in range(maxit):
        w = A(Pinv(V[j]))
        for _ in range(2):
            h = V[: j + 1] @ w; w = w - h @ V[: j + 1]; H[: j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w); V[j + 1] = w / H[j + 1, j]
        e1 = np.zeros(j + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e1, rcond=None)
        res = np.linalg.norm(H[: j + 2, : j + 1] @ y - e1) / beta
        if res < tol:
            return j + 1, res
    return maxit, res


print("| lattice gap | overlap delta | largest extended block | GMRES iterations to 1e-8 | flops of the factorisations vs block-Jacobi |")
print("|---|---|---|---|---|")
for gap_a in (3.8, 7.4):
    spacing = 2.0 * (1.0 + a) + gap_a * a
    idx = np.arange(nb)
    X = np.stack([idx % 3, (idx // 3) % 3, idx // 9], axis=1).astype(float) * spacing + rng.uniform(-0.1, 0.1, (nb, 3)) * min(1.0, gap_a / 4.0)
    X[:, 2] += 1.0 + a + 0.3
    Q = rng.standard_normal((nb, 4)); Q /= np.linalg.norm(Q, axis=1)[:, None]
    r = orc.multi_body_pos(X, Q, cfg)
    B = orc.damp(r, a)
    M = (B[:, None] * orc.rotne_prager_tensor(r, a, 1.0, True)) * B[None, :]
    K = onp.K_matrix(X, Q, cfg)
    n3 = m * nb
    A = lambda x: np.concatenate([M @ x[:n3] - K @ x[n3:], K.T @ x[:n3]])
    rhs = np.concatenate([rng.standard_normal(n3), -np.tile([0, 0, -1.0, 0, 0, 0], nb)])
    pos = r.reshape(-1, 3)
    base_flops = nb * m ** 3 / 3.0
    for delta_a in (0.0, 3.0, 6.0, 10.0, 16.0):
        t0 = time.time()
        blocks = []
        for b in range(nb):
            own = np.arange(nblb * b, nblb * (b + 1))
            ext = own
            if delta_a > 0.0:
                d2 = ((pos[:, None, :] - pos[None, own, :]) ** 2).sum(-1).min(axis=1)
                near = np.where((d2 < (delta_a * a) ** 2))[0]
                ext = np.union1d(own, near)
            dof = (3 * ext[:, None] + np.arange(3)[None, :]).reshape(-1)
            keep = np.where(np.isin(ext, own))[0]
            kdof = (3 * keep[:, None] + np.arange(3)[None, :]).reshape(-1)
            Minv = np.linalg.inv(M[np.ix_(dof, dof)])
            blocks.append((dof, kdof, Minv))

        def Rop(v):                                  # restricted additive Schwarz: solve on the extended set, keep the own rows
            out = np.zeros_like(v)
            for dof, kdof, Minv in blocks:
                out[dof[kdof]] = (Minv @ v[dof])[kdof]
            return out

        RK = np.stack([Rop(K[:, j]) for j in range(6 * nb)], axis=1)
        Ninv = np.linalg.inv(K.T @ RK)

        def Pinv(x):                                 # :601-610 with M^-1 -> R, force block sign restored
            s, f = x[:n3], x[n3:]
            y = Rop(s)
            U = Ninv @ (f - K.T @ y)
            return np.concatenate([y + RK @ U, U])

        it, res = gmres(A, Pinv, rhs)
        fl = sum(len(d) ** 3 / 3.0 for d, _, _ in blocks)
        print("| %.1f a | %s | %d blobs | %d (%.1e) | %.2f |" % (gap_a, "none (block-Jacobi)" if delta_a == 0 else "%.0f a" % delta_a,
                                                              max(len(d) for d, _, _ in blocks) // 3, it, res, fl / base_flops), flush=True)
