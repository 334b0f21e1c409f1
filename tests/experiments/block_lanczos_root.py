"""Experiment: would a BLOCK Lanczos recurrence (block size 2) for the two Brownian increments of a step, M^{1/2} W1 and M^{1/2} W2, need
fewer two-vector products than the two lock-step recurrences the library runs (mhalf_lanczos_dev, nvec = 2)?  Both cost ONE two-vector
product per iteration; the block Krylov space of dimension 2 m contains both single-vector spaces of dimension m.  Same setting as
two_level_root.py (27 bodies above a wall, two-level factor G, Euclidean error of the increment x = G (G^-1 M G^-T)^{1/2} W against a
long run), iterations until BOTH increments are below 1e-3 / 1e-6.  Dense numpy on the CPU oracle's mobility.
    python tests/experiments/block_lanczos_root.py [blobs_per_body]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import Oracle                                      # noqa: E402
from oracle import oracle as onp                               # noqa: E402
from rigid_body_light_amd.synth import load_structure          # noqa: E402

nblb = int(sys.argv[1]) if len(sys.argv) > 1 else 42
nb = 27
orc = Oracle()
params, cfg0 = load_structure(nblb)
a = params["sep"] / 2.0
cfg = onp.remove_mean(cfg0)
rng = np.random.default_rng(0)
m = 3 * nblb


def sqrt_sym(T):
    lam, Y = np.linalg.eigh(0.5 * (T + T.T))
    return (Y * np.sqrt(np.clip(lam, 0, None))) @ Y.T


def single(S, W, mmax):
    n = W.size
    V = np.zeros((mmax + 1, n)); H = np.zeros((mmax + 1, mmax))
    wn = np.linalg.norm(W); V[0] = W / wn
    Z = []
    for it in range(mmax):
        u = S(V[it])
        for _ in range(2):
            h = V[: it + 1] @ u; u = u - h @ V[: it + 1]; H[: it + 1, it] += h
        H[it + 1, it] = np.linalg.norm(u); V[it + 1] = u / H[it + 1, it]
        Z.append(wn * (sqrt_sym(H[: it + 1, : it + 1])[:, 0] @ V[: it + 1]))
    return Z


def block(S, W2, mmax):
    """W2: n x 2.  Returns per iteration the n x 2 estimate of S^{1/2} W2."""
    n = W2.shape[0]
    Q0, R0 = np.linalg.qr(W2)
    V = [Q0]                                                      # list of n x 2 blocks
    T = np.zeros((2 * (mmax + 1), 2 * (mmax + 1)))
    Z = []
    for it in range(mmax):
        U = np.stack([S(V[it][:, 0]), S(V[it][:, 1])], axis=1)
        Vall = np.concatenate(V, axis=1)                          # n x 2 (it + 1)
        for _ in range(2):                                        # block classical Gram-Schmidt, twice
            Hc = Vall.T @ U; U = U - Vall @ Hc; T[: 2 * (it + 1), 2 * it: 2 * it + 2] += Hc
        Qn, Rn = np.linalg.qr(U)
        T[2 * (it + 1): 2 * (it + 2), 2 * it: 2 * it + 2] = Rn
        V.append(Qn)
        k = 2 * (it + 1)
        Z.append(Vall @ (sqrt_sym(T[:k, :k])[:, :2] @ R0))
    return Z


print("| lattice gap | blobs per body | two lock-step recurrences: iterations until BOTH increments < 1e-3 / 1e-6 | block recurrence (block size 2) |")
print("|---|---|---|---|")
for gap_a in (3.8, 7.4):
    spacing = 2.0 * (1.0 + a) + gap_a * a
    idx = np.arange(nb)
    X = np.stack([idx % 3, (idx // 3) % 3, idx // 9], axis=1).astype(float) * spacing + rng.uniform(-0.1, 0.1, (nb, 3)) * min(1.0, gap_a / 4.0)
    X[:, 2] += 1.0 + a + 0.3
    Q4 = rng.standard_normal((nb, 4)); Q4 /= np.linalg.norm(Q4, axis=1)[:, None]
    r = orc.multi_body_pos(X, Q4, cfg)
    B = orc.damp(r, a)
    M = orc.rotne_prager_tensor(r, a, 1.0, True)
    n3 = m * nb
    Ls = [np.linalg.cholesky(M[m * b:m * (b + 1), m * b:m * (b + 1)]) for b in range(nb)]
    Linv = [np.linalg.inv(L) for L in Ls]

    def bd(mats, v, T=False):
        return np.concatenate([(mats[b].T if T else mats[b]) @ v[m * b:m * (b + 1)] for b in range(nb)])

    Mc = orc.rotne_prager_tensor(X.reshape(-1), params["Rh"], 1.0, True)
    C = Mc.copy()
    for b in range(nb):
        C[3 * b:3 * b + 3, 3 * b:3 * b + 3] = 0.0
    Kt = np.zeros((n3, 3 * nb))
    for b in range(nb):
        for d in range(3):
            Kt[m * b + d:m * (b + 1):3, 3 * b + d] = 1.0
    Zm = np.stack([bd(Linv, Kt[:, j]) for j in range(3 * nb)], axis=1)
    R = Zm.T @ Zm
    Rh = np.zeros_like(R); Rih = np.zeros_like(R)
    for b in range(nb):
        w, U = np.linalg.eigh(R[3 * b:3 * b + 3, 3 * b:3 * b + 3])
        Rh[3 * b:3 * b + 3, 3 * b:3 * b + 3] = U @ np.diag(np.sqrt(w)) @ U.T
        Rih[3 * b:3 * b + 3, 3 * b:3 * b + 3] = U @ np.diag(1.0 / np.sqrt(w)) @ U.T
    Qm = Zm @ Rih
    E = Rh @ C @ Rh
    LE = np.linalg.cholesky(np.eye(3 * nb) + E); LEi = np.linalg.inv(LE)
    FE, FEi = LE - np.eye(3 * nb), LEi - np.eye(3 * nb)
    Gi = lambda v: (lambda w: w + Qm @ (FEi @ (Qm.T @ w)))(bd(Linv, v))
    GiT = lambda v: bd(Linv, v + Qm @ (FEi.T @ (Qm.T @ v)), T=True)
    G = lambda z: B * bd(Ls, z + Qm @ (FE @ (Qm.T @ z)))
    S = lambda v: Gi(M @ GiT(v))
    for trial in range(3):
        W2 = rng.standard_normal((n3, 2))
        Zs = [single(S, W2[:, v], 40) for v in range(2)]
        Zb = block(S, W2, 40)
        ref = [G(Zs[v][-1]) for v in range(2)]
        es = [max(np.linalg.norm(G(Zs[v][i]) - ref[v]) / np.linalg.norm(ref[v]) for v in range(2)) for i in range(40)]
        eb = [max(np.linalg.norm(G(Zb[i][:, v]) - ref[v]) / np.linalg.norm(ref[v]) for v in range(2)) for i in range(40)]
        it = lambda e, tol: next(i for i, x in enumerate(e) if x < tol) + 1
        print("| %.1f a | %d | %d / %d | %d / %d |" % (gap_a, nblb, it(es, 1e-3), it(es, 1e-6), it(eb, 1e-3), it(eb, 1e-6)), flush=True)
