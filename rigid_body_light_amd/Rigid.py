"""RigidBody -- host-side mirror of the reference's Python wrapper
(reference src/Rigid.py:5-135): same constructor, attributes, method names,
argument meaning, output shapes and RuntimeError behaviour, on top of the
MI355X-native `c_rigid.CManyBodies`.

Extensions beyond the reference surface (kept out of the way of drop-in use):
`M_half_W`, `apply_M_multi`, `dense_mobility`, `cholesky_lower`.
"""
import numpy as np

from . import c_rigid as crigid


class RigidBody:
    X_shape = None
    Q_shape = None

    def __init__(self, rigid_config, X, Q, a, eta, dt, wall_PC=False, block_PC=False):
        # reference src/Rigid.py:9-35
        self.cb = crigid.CManyBodies()
        self.precision = self.cb.precision
        kbt = 1.0  # the reference hard-codes kBT = 1 in the wrapper (src/Rigid.py:23)
        rigid_config = np.asarray(rigid_config)
        if rigid_config.size % 3 != 0:
            raise RuntimeError(
                f"Rigid config must have length 3N. Rigid config shape: {rigid_config.shape}")
        self.blobs_per_body = rigid_config.size // 3
        self.cb.setParameters(a, dt, kbt, eta, rigid_config.reshape(-1, 3))
        self.cb.setBlkPC(block_PC)
        self.cb.setWallPC(wall_PC)
        self.set_config(X, Q)

    # -- configuration ------------------------------------------------------
    def get_config(self):
        # src/Rigid.py:37-42
        X, Q = self.cb.getConfig()
        return X.reshape(self.X_shape), Q.reshape(self.Q_shape)

    def set_config(self, X, Q):
        # src/Rigid.py:44-51
        X = np.asarray(X)
        Q = np.asarray(Q)
        self._check_and_set_configs(X, Q)
        self.cb.setConfig(X.flatten(), Q.flatten())
        self.cb.set_K_mats()
        self.total_blobs = self.N_bodies * self.blobs_per_body

    def _out_shape(self):
        return (-1, 3) if len(self.X_shape) == 2 else (-1)

    def get_blob_positions(self):
        # src/Rigid.py:53-55
        return np.array(self.cb.multi_body_pos()).reshape(self._out_shape())

    # -- geometric operators --------------------------------------------------
    def KT_dot(self, lambda_vec):
        # src/Rigid.py:57-61
        lambda_vec = np.asarray(lambda_vec)
        self._check_input_size(lambda_vec=lambda_vec)
        return np.array(self.cb.KT_x_Lam(lambda_vec.flatten())).reshape(self._out_shape())

    def K_dot(self, U):
        # src/Rigid.py:63-67
        U = np.asarray(U)
        self._check_input_size(U_vec=U)
        return np.array(self.cb.K_x_U(U.flatten())).reshape(self._out_shape())

    def apply_PC(self, b):
        # src/Rigid.py:69-71
        b = np.asarray(b)
        self._check_input_size(system_input=b)
        return self.cb.apply_PC(b.flatten())

    def apply_saddle(self, x):
        # src/Rigid.py:73-80:  [M lambda - K U ; K^T lambda]
        x = np.asarray(x)
        self._check_input_size(system_input=x)
        lambda_vec = x[: 3 * self.total_blobs]
        U = x[3 * self.total_blobs:]
        r_vecs = self.get_blob_positions().flatten()
        slip = self.apply_M(forces=lambda_vec, positions=r_vecs) - self.K_dot(U).flatten()
        F = self.KT_dot(lambda_vec).flatten()
        return np.concatenate((slip, F))

    def apply_M(self, forces, positions):
        # src/Rigid.py:82-87  (sizes need NOT match the object's own blob count)
        forces = np.asarray(forces)
        positions = np.asarray(positions)
        if np.size(positions) != np.size(forces):
            raise RuntimeError("Positions and forces must be of the same size")
        if np.size(positions) % 3 != 0 or np.size(forces) % 3 != 0:
            raise RuntimeError(
                "Positions and forces must have total length 3N, where N is the number of blobs")
        return self.cb.apply_M(forces.flatten(), positions.flatten())

    def get_K(self):
        return self.cb.get_K()

    def get_Kinv(self):
        return self.cb.get_Kinv()

    def evolve_rigid_bodies(self, U):
        # src/Rigid.py:95-97
        U = np.asarray(U)
        self._check_input_size(U_vec=U)
        self.cb.evolve_X_Q(U.flatten())

    # -- extensions (reference members that are not bound to Python) ----------
    def M_half_W(self, W=None, seed=0, method="cholesky"):
        """Brownian increment M^{1/2} W (reference c_rigid_obj.cpp:661-675, C++ only).
        W=None draws reproducible N(0,1) noise from `seed` on the device."""
        if W is not None:
            W = np.asarray(W)
            self._check_input_size(lambda_vec=W)
            W = W.flatten()
        return self.cb.M_half_W(W, seed, method)

    def M_RFD(self, W=None, seed=0, delta=1.0e-4):
        """Random finite difference of the mobility (reference c_rigid_obj.cpp:769-796, C++ only):
        (1/delta)[M(q + delta/2 Kinv W) - M(q - delta/2 Kinv W)] W -- the thermal-drift term."""
        if W is not None:
            W = np.asarray(W)
            self._check_input_size(lambda_vec=W)
            W = W.flatten()
        return self.cb.M_RFD(W, seed, delta)

    def KTinv_RFD(self, W, delta=1.0e-4):
        """reference c_rigid_obj.cpp:743-767 (C++ only); W has length 6*N_bodies."""
        W = np.asarray(W)
        self._check_input_size(U_vec=W)
        return self.cb.KTinv_RFD(W.flatten(), delta)

    def apply_M_multi(self, forces, positions):
        forces = np.atleast_2d(np.asarray(forces))
        return self.cb.apply_M_multi(forces, np.asarray(positions).flatten())

    def dense_mobility(self, positions, scale_damp=False):
        """rotne_prager_tensor (c_rigid_obj.cpp:413-459) as a dense (3N,3N) array."""
        return self.cb.rotne_prager_tensor(np.asarray(positions).flatten(), scale_damp)

    # -- validation (src/Rigid.py:99-135) -------------------------------------
    def _check_and_set_configs(self, X, Q):
        x_size = int(np.prod(np.shape(X)))
        q_size = int(np.prod(np.shape(Q)))
        if x_size % 3 != 0:
            raise RuntimeError("X must have total length 3N")
        if q_size % 4 != 0:
            raise RuntimeError("Q must have total length 4N")
        nx = x_size // 3
        nq = q_size // 4
        if nx != nq:
            raise RuntimeError("X and Q must have the same number of bodies")
        self.N_bodies = nx
        self.X_shape = X.shape
        self.Q_shape = Q.shape

    def _check_input_size(self, lambda_vec=None, U_vec=None, system_input=None):
        if lambda_vec is not None and lambda_vec.size != 3 * self.total_blobs:
            raise RuntimeError(
                f"lambda must have total size 3*N_blobs = {3 * self.total_blobs}. "
                f"lambda_vec shape: {lambda_vec.shape}")
        if U_vec is not None and U_vec.size != 6 * self.N_bodies:
            raise RuntimeError(
                f"U must have total size 6*N_bodies = {6 * self.N_bodies}. U shape: {U_vec.shape}")
        if system_input is not None:
            expected = 3 * self.total_blobs + 6 * self.N_bodies
            if system_input.size != expected:
                raise RuntimeError(
                    "Rigid system input vector must have total size 3*N_blobs + 6*N_bodies = "
                    f"{expected}. system_input shape: {system_input.shape}")
