"""Host-side operator surface of the MI355X-native blob-mobility library.

`RigidBody` presents the constructor, attributes, methods, output shapes and RuntimeError
behaviour of the reference's Python wrapper (reference src/Rigid.py:5-135) so that user code written
against `from Rigid import RigidBody` runs unchanged, but every call lands in librbl.so through the
pybind11 class `c_rigid.CManyBodies`.  Size rules are kept in one table (`_expected_size`) instead
of per-method checks; shapes follow the shape X was given in (2-D in -> (-1, 3) out, flat in -> flat out,
reference src/Rigid.py:54,60,66).

Beyond the reference surface (its C++ has these, its Python does not): `solve_saddle`, `solve_saddle_multi`,
`body_mobility_matrix`, `M_half_W`, `M_RFD`,
`KTinv_RFD`, `M_RFD_cfgs`, `M_RFD_from_U`, `KT_RFD_from_U`, `evolve_rigid_bodies_RFD`, `apply_M_multi`,
`dense_mobility`.
"""
import numpy as np

from . import c_rigid as _ext

_KBT_IN_WRAPPER = 1.0   # the reference wrapper passes kBT = 1 whatever the caller wants (src/Rigid.py:23)


def _fail(message):
    raise RuntimeError(message)


class RigidBody:
    X_shape = None
    Q_shape = None

    # ------------------------------------------------------------------ construction
    def __init__(self, rigid_config, X, Q, a, eta, dt, wall_PC=False, block_PC=False):
        template = np.asarray(rigid_config)
        if template.size % 3:
            _fail(f"Rigid config must have length 3N. Rigid config shape: {template.shape}")
        self.cb = _ext.CManyBodies()
        self.precision = self.cb.precision
        self.blobs_per_body = template.size // 3
        self.cb.setParameters(a, dt, _KBT_IN_WRAPPER, eta, template.reshape(self.blobs_per_body, 3))
        self.cb.setWallPC(bool(wall_PC))
        self.cb.setBlkPC(bool(block_PC))
        self.set_config(X, Q)

    # ------------------------------------------------------------------ size / shape rules
    def _expected_size(self, kind):
        n_blob3 = 3 * self.total_blobs
        return {"blob": n_blob3, "body": 6 * self.N_bodies, "system": n_blob3 + 6 * self.N_bodies}[kind]

    def _require(self, vec, kind):
        vec = np.asarray(vec)
        want = self._expected_size(kind)
        if vec.size != want:
            label = {"blob": "lambda must have total size 3*N_blobs",
                     "body": "U must have total size 6*N_bodies",
                     "system": "Rigid system input vector must have total size 3*N_blobs + 6*N_bodies"}[kind]
            _fail(f"{label} = {want}. Got shape: {vec.shape}")
        return vec.reshape(-1)

    def _like_X(self, flat):
        return np.asarray(flat).reshape((-1, 3) if len(self.X_shape) == 2 else (-1,))

    # ------------------------------------------------------------------ configuration
    def set_config(self, X, Q):
        X, Q = np.asarray(X), np.asarray(Q)
        if X.size % 3:
            _fail("X must have total length 3N")
        if Q.size % 4:
            _fail("Q must have total length 4N")
        if X.size // 3 != Q.size // 4:
            _fail("X and Q must have the same number of bodies")
        self.N_bodies = X.size // 3
        self.total_blobs = self.N_bodies * self.blobs_per_body
        self.X_shape, self.Q_shape = X.shape, Q.shape
        self.cb.setConfig(X.reshape(-1), Q.reshape(-1))       # Q is normalised inside (c_rigid_obj.cpp:216)
        self.cb.set_K_mats()

    def get_config(self):
        X, Q = self.cb.getConfig()
        return X.reshape(self.X_shape), Q.reshape(self.Q_shape)

    def get_blob_positions(self):
        return self._like_X(self.cb.multi_body_pos())          # GPU: r_k = R(Q_b) c_k + X_b (:257-300)

    def evolve_rigid_bodies(self, U):
        self.cb.evolve_X_Q(self._require(U, "body"))           # multiplies by dt inside (:869)

    # ------------------------------------------------------------------ geometric operators
    def K_dot(self, U):
        return self._like_X(self.cb.K_x_U(self._require(U, "body")))

    def KT_dot(self, lambda_vec):
        return self._like_X(self.cb.KT_x_Lam(self._require(lambda_vec, "blob")))

    def get_K(self):
        return self.cb.get_K()

    def get_Kinv(self):
        return self.cb.get_Kinv()

    # ------------------------------------------------------------------ mobility
    def apply_M(self, forces, positions):
        """U = M F (or B M B F with the wall term when wall_PC), GPU, matrix-free.  The blob count is
        whatever `positions` holds, not necessarily the object's own (reference tests/test_interface.py:171)."""
        forces, positions = np.asarray(forces), np.asarray(positions)
        if forces.size != positions.size:
            _fail("Positions and forces must be of the same size")
        if forces.size % 3:
            _fail("Positions and forces must have total length 3N, where N is the number of blobs")
        return self.cb.apply_M(forces.reshape(-1), positions.reshape(-1))

    def apply_saddle(self, x):
        """[M lambda - K U ; K^T lambda]  (reference src/Rigid.py:73-80)."""
        return self.cb.apply_saddle(self._require(x, "system"))    # one boundary crossing (rbl_apply_saddle); the reference
                                                                    # composes it from four calls through its extension

    def apply_PC(self, b):
        return self.cb.apply_PC(self._require(b, "system"))

    def solve_saddle(self, rhs, max_iter=100, rtol=1.0e-8, x0=None):
        """Solve [M -K; K^T 0] x = rhs by the library's own right-preconditioned GMRES (apply_PC as preconditioner), all
        iterations on the GPU: what a user of the reference loops over apply_saddle / apply_PC for from outside
        (src/Rigid.py:69-80).  -> (x, iterations, residual estimate)"""
        return self.cb.solve_saddle(self._require(rhs, "system"), int(max_iter), float(rtol),
                                    None if x0 is None else self._require(x0, "system"))

    def solve_saddle_multi(self, rhs, max_iter=100, rtol=1.0e-8):
        """The same solve for SEVERAL right-hand sides of this configuration at once, rhs (k, 3 N_blobs + 6 N_bodies): k GMRES
        recurrences in lock step, each column the iterates `solve_saddle` would give it, the k mobility products of an iteration
        ONE launch on the fp64 matrix cores (16 columns per pass).  -> (x (k, size), iterations (k,), residual estimates (k,))"""
        rhs = np.ascontiguousarray(np.atleast_2d(np.asarray(rhs, dtype=np.float64)))
        if rhs.shape[1] != 3 * self.total_blobs + 6 * self.N_bodies:
            _fail("solve_saddle_multi: rhs must have shape (k, 3*total_blobs + 6*N_bodies)")
        return self.cb.solve_saddle_multi(rhs, int(max_iter), float(rtol))

    def body_mobility_matrix(self, max_iter=100, rtol=1.0e-8, columns=None):
        """The (6 N_bodies) x (6 N_bodies) body mobility matrix N = (K^T M^-1 K)^-1 of the current configuration, U = N F: one
        saddle solve [M -K; K^T 0][lambda; U] = [0; e_c] per unit load e_c (force / torque component c of one body), the 6 N_bodies
        solves through `solve_saddle_multi`.  columns: only these unit loads (default all).  -> (N[:, columns], iterations)"""
        nb6, n3 = 6 * self.N_bodies, 3 * self.total_blobs
        cols = np.arange(nb6) if columns is None else np.asarray(columns, dtype=int).reshape(-1)
        rhs = np.zeros((cols.size, n3 + nb6))
        rhs[np.arange(cols.size), n3 + cols] = 1.0          # K^T lambda = e_c, M lambda = K U  =>  U = (K^T M^-1 K)^-1 e_c
        x, its, _ = self.solve_saddle_multi(rhs, max_iter, rtol)
        return np.ascontiguousarray(x[:, n3:].T), its

    # ------------------------------------------------------------------ beyond the reference's Python surface
    def M_half_W(self, W=None, seed=0, method="cholesky"):
        """Brownian increment M^{1/2} W (reference c_rigid_obj.cpp:661-675, C++ only).
        W=None draws reproducible N(0,1) noise from `seed` on the device."""
        return self.cb.M_half_W(None if W is None else self._require(W, "blob"), seed, method)

    def M_RFD(self, W=None, seed=0, delta=1.0e-4):
        """(1/delta)[M(q + delta/2 Kinv W) - M(q - delta/2 Kinv W)] W  (reference :769-796, C++ only)."""
        return self.cb.M_RFD(None if W is None else self._require(W, "blob"), seed, delta)

    def KTinv_RFD(self, W, delta=1.0e-4):
        """reference c_rigid_obj.cpp:743-767 (C++ only); W has length 6*N_bodies."""
        return self.cb.KTinv_RFD(self._require(W, "body"), delta)

    def M_RFD_cfgs(self, U, delta=1.0e-4):
        """blob positions of the configurations displaced by +-(delta/2) U (reference :798-818, C++ only) -> (r_plus, r_minus)."""
        return self.cb.M_RFD_cfgs(self._require(U, "body"), delta)

    def M_RFD_from_U(self, U, W, delta=1.0e-3):
        """(1/delta)[M(q + delta/2 U) - M(q - delta/2 U)] W for the caller's displacement U (reference :820-842, C++ only)."""
        return self.cb.M_RFD_from_U(self._require(U, "body"), self._require(W, "blob"), delta)

    def KT_RFD_from_U(self, U, W, delta=1.0e-3):
        """(1/delta)[K(q + delta/2 U)^T - K(q - delta/2 U)^T] W (reference :844-863, C++ only); W has length 3*N_blobs."""
        return self.cb.KT_RFD_from_U(self._require(U, "body"), self._require(W, "blob"), delta)

    def evolve_rigid_bodies_RFD(self, U):
        """commit the configuration displaced by U (displacement units, no dt) and keep the preconditioner
        (reference evolve_X_Q_RFD :880-893, C++ only)."""
        self.cb.evolve_X_Q_RFD(self._require(U, "body"))

    def update_X_Q(self, U):
        """configuration displaced by U (translation + rotation vector per body), NOT committed
        (reference c_rigid_obj.cpp:798-863, C++ only) -> (X, Q) flat arrays."""
        return self.cb.update_X_Q(self._require(U, "body"))

    def RHS_and_Midpoint(self, slip, force, W=None, seed=0, method="cholesky", split_rand=True, delta=1.0e-4):
        """Right-hand side and predictor configuration of the stochastic midpoint step (reference
        c_rigid_obj.cpp:917-976, C++ only): ([slip - (kBT M_RFD + BI) ; -force], X_half, Q_half).
        W = [W1 | W2 | W_rfd] (9*N_blobs numbers) or None for seeded device noise."""
        W = None if W is None else np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(-1))
        return self.cb.RHS_and_Midpoint(self._require(slip, "blob"), self._require(force, "body"), W, seed,
                                        method, split_rand, delta)

    def step_deterministic(self, F_body, slip=None, max_iter=50, rtol=1.0e-8, warm_start=False):
        """One deterministic time step inside the library: GMRES on the saddle system with rhs [slip ; -F_body]
        (apply_PC as preconditioner), then evolve_rigid_bodies(U).  warm_start: False/0 cold, True/1 start from the previous
        step's solution, 2 / 3 from the linear / quadratic extrapolation of the last solutions (smooth forcing: far fewer
        iterations).  Returns (iterations, residual estimate)."""
        return self.cb.step_deterministic(self._require(F_body, "body"), None if slip is None else self._require(slip, "blob"),
                                          max_iter, rtol, int(warm_start))

    def step_brownian(self, F_body, slip=None, W=None, seed=0, method="lanczos_pc", split_rand=True, delta=1.0e-4,
                      max_iter=50, rtol=1.0e-8):
        """One stochastic midpoint step inside the library (RHS_and_Midpoint at q^n, solve at q^{n+1/2}, update from q^n).
        Note the reference wrapper fixes kBT = 1 (src/Rigid.py:23); scale dt / forces accordingly."""
        W = None if W is None else np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(-1))
        return self.cb.step_brownian(self._require(F_body, "body"), None if slip is None else self._require(slip, "blob"),
                                     W, seed, method, split_rand, delta, max_iter, rtol)

    def apply_M_multi(self, forces, positions):
        """k right-hand sides at once, forces (k, 3N); k >= 4 runs on the fp64 matrix cores."""
        return self.cb.apply_M_multi(np.atleast_2d(np.asarray(forces)), np.asarray(positions).reshape(-1))

    def dense_mobility(self, positions, scale_damp=False):
        """rotne_prager_tensor (reference :413-459) as a dense (3N, 3N) array."""
        return self.cb.rotne_prager_tensor(np.asarray(positions).reshape(-1), scale_damp)
