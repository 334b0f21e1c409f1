"""Synthetic multibody configurations of SURVEY.md section 8(d) and the structure
reader of the reference tests (reference tests/utils.py:9-19).  Pure numpy; used by
bench.py, tests and examples -- it only PRODUCES inputs, no mobility arithmetic."""
import os

import numpy as np

STRUCT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "structures")


def load_structure(n_blobs):
    """-> (params, cfg[N,3]).  File format: 2 comment lines `# sep,N,rg,rh` / values."""
    path = os.path.join(STRUCT_DIR, "shell_N_%d.csv" % n_blobs)
    with open(path, "r") as f:
        f.readline()
        vals = f.readline().lstrip("#").strip().split(",")
        cfg = np.loadtxt(f)
    params = {"sep": float(vals[0]), "N": int(vals[1]), "Rg": float(vals[2]), "Rh": float(vals[3])}
    return params, cfg.reshape(-1, 3)


def make_config(n_bodies, n_blobs, wall, seed=0):
    """Body centres on a jittered simple-cubic lattice (spacing 2(1+a)+0.5, x fastest),
    random unit quaternions, a = sep/2; with a wall the lowest layer sits at
    z = 1 + a + 0.2 so every blob is above the wall.  Returns dict(cfg, X, Q, a)."""
    params, cfg = load_structure(n_blobs)
    a = params["sep"] / 2.0
    side = int(np.ceil(n_bodies ** (1.0 / 3.0) - 1e-9))
    spacing = 2.0 * (1.0 + a) + 0.5
    idx = np.arange(n_bodies)
    X = np.stack([idx % side, (idx // side) % side, idx // (side * side)], axis=1).astype(np.float64) * spacing
    X += np.random.default_rng(seed).uniform(-0.1, 0.1, X.shape)
    if wall:
        X[:, 2] += 1.0 + a + 0.2 + 0.1
    Q = np.random.default_rng(seed + 1).standard_normal((n_bodies, 4))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    return {"cfg": cfg, "X": X, "Q": Q, "a": a, "eta": 1.0, "dt": 0.01}


def forces_for(n_blobs_total, seed=2):
    return np.random.default_rng(seed).standard_normal(3 * n_blobs_total)


def noise_for(n_blobs_total, seed=3):
    return np.random.default_rng(seed).standard_normal(3 * n_blobs_total)
