#!/usr/bin/env python3
"""Drop-in use: code written for the reference (`from Rigid import RigidBody`) runs unchanged; the
mobility products run on the MI355X.  Needs a GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from Rigid import RigidBody                                  # alias of rigid_body_light_amd.RigidBody
from rigid_body_light_amd import load_structure

params, cfg = load_structure(162)                            # reference geometry file shell_N_162.csv
a = params["sep"] / 2.0
rng = np.random.default_rng(0)
X = np.array([[0, 0, 3.0], [3, 0, 3.0], [0, 3, 3.5]])        # three shells above a wall
Q = rng.standard_normal((3, 4))
cb = RigidBody(cfg, X, Q, a=a, eta=1.0, dt=0.01, wall_PC=True, block_PC=True)

r = cb.get_blob_positions()                                  # (486, 3), computed on the GPU
lam = rng.standard_normal(r.size)
U = cb.apply_M(lam, r)                                       # wall-corrected RPY mobility x forces
x = rng.standard_normal(r.size + 18)
print("apply_M   :", U.shape, np.linalg.norm(U))
print("saddle    :", np.linalg.norm(cb.apply_saddle(x)))     # what a GMRES iteration calls
print("apply_PC  :", np.linalg.norm(cb.apply_PC(x)))         # block-diagonal PC, batched Cholesky on the GPU
print("M^(1/2) W :", np.linalg.norm(cb.M_half_W(seed=7)))    # Brownian increment (C++-only in the reference)
print("M_RFD     :", np.linalg.norm(cb.M_RFD(seed=8)))       # thermal drift by random finite differences
cb.evolve_rigid_bodies(np.tile([0, 0, -1.0, 0, 0, 0], 3))
print("new X     :", cb.get_config()[0])
# the reference's usage model: an external Krylov solver over apply_saddle / apply_PC (src/Rigid.py:69-80) ...
import scipy.sparse.linalg as spla
rhs = np.concatenate([np.zeros(r.size), -np.tile([0, 0, 1.0, 0, 0, 0], 3)])
A = spla.LinearOperator((rhs.size, rhs.size), matvec=lambda y: cb.apply_saddle(cb.apply_PC(y)), dtype=np.float64)
y, info = spla.gmres(A, rhs, rtol=1e-8, atol=0.0, restart=60, maxiter=3)
xs = cb.apply_PC(y)                                          # right preconditioning: x = P^-1 y
# ... or the library's own solver on the same system (every iteration on the GPU)
xn, its, res = cb.solve_saddle(rhs, rtol=1e-8)
print("solve     : SciPy info %d, native %d iterations (%.1e); |x_scipy - x_native| / |x| = %.1e" % (
    info, its, res, np.linalg.norm(xs - xn) / np.linalg.norm(xn)))
# beyond the reference's surface: whole time steps inside the library (GMRES on the saddle system + evolve)
its, res = cb.step_deterministic(np.tile([0, 0, 1.0, 0, 0, 0], 3), rtol=1e-8)
print("det. step :", its, "GMRES iterations, residual %.1e" % res, "-> z =", cb.get_config()[0][:, 2])
its, res = cb.step_brownian(np.tile([0, 0, 1.0, 0, 0, 0], 3), seed=1, rtol=1e-8)   # stochastic midpoint step (kBT = 1)
print("Brownian  :", its, "GMRES iterations, residual %.1e" % res, "-> z =", cb.get_config()[0][:, 2])
