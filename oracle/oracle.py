"""ctypes front-end of the CPU ORACLE + numpy restatement of the O(N) rigid-body
bookkeeping.  TEST INFRASTRUCTURE, not the product (see oracle/rbl_oracle.h).

C part  (oracle/rbl_oracle.c): pair kernels, dense assembly, damping, apply_M,
          Cholesky, M_half_W, blob positions  -> reference src/c_rigid_obj.cpp
          :31-142, :413-459, :618-675, :257-300.
numpy part (this file): K / K^T / K^-1, preconditioner, quaternion update
          -> reference src/c_rigid_obj.cpp :302-410, :461-616, :679-710, :865-878.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def oracle_lib_path():
    return os.path.join(_HERE, "liboracle.so")


def ref_lib_path():
    return os.path.join(_HERE, "_ref", "libref_pair.so")


def build(force=False):
    """Compile liboracle.so (and _ref/libref_pair.so when /root/reference exists)."""
    if force or not os.path.exists(oracle_lib_path()):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    if force or not os.path.exists(ref_lib_path()):
        subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")])


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        msg = {1: "two blobs overlap (r < 1e-12 a); the reference exit()s here",
               2: "A blob has its center below the wall (z<0). Cannot compute mobility- check your configuration.",
               3: "matrix is not SPD"}.get(code, "oracle error %d" % code)
        super().__init__(msg)


class Oracle:
    """Plain-C CPU restatement of the reference hot path."""

    def __init__(self):
        if not os.path.exists(oracle_lib_path()):
            build()
        L = C.CDLL(oracle_lib_path())
        L.orc_mobilityUFRPY.argtypes = [C.c_double] * 3 + [_dp, C.c_int, C.c_int, C.c_double]
        L.orc_mobilityUFSingleWallCorrection.argtypes = [C.c_double] * 3 + [_dp, C.c_int, C.c_int, C.c_double]
        L.orc_pair_block.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_rotne_prager_tensor.argtypes = [_dp, C.c_long, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_make_damp.argtypes = [_dp, C.c_long, C.c_double, _dp]
        L.orc_make_damp.restype = None
        L.orc_apply_M_dense.argtypes = [_dp, _dp, C.c_long, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_apply_M_matfree.argtypes = [_dp, _dp, C.c_long, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_apply_M_rows.argtypes = [_dp, _dp, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double,
                                       C.c_int, C.c_int, _dp]
        L.orc_cholesky_lower.argtypes = [_dp, C.c_long]
        L.orc_M_half_W.argtypes = [_dp, C.c_long, C.c_double, C.c_double, C.c_int, _dp, _dp, _dp]
        L.orc_multi_body_pos.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, _dp]
        L.orc_multi_body_pos.restype = None
        self.L = L

    # -- a1 / a2 ------------------------------------------------------------
    def rpy(self, rx, ry, rz, i, j, inv_a):
        out = np.zeros(6)
        rc = self.L.orc_mobilityUFRPY(rx, ry, rz, _p(out), i, j, inv_a)
        if rc:
            raise OracleError(rc)
        return out

    def wall(self, rx, ry, rz, M9, i, j, hj):
        M = _f64(M9).copy()
        rc = self.L.orc_mobilityUFSingleWallCorrection(rx, ry, rz, _p(M), i, j, hj)
        if rc:
            raise OracleError(rc)
        return M

    def pair_block(self, ri, rj, i, j, a, eta, wall):
        ri = _f64(ri); rj = _f64(rj)
        out = np.zeros(9)
        rc = self.L.orc_pair_block(_p(ri), _p(rj), i, j, a, eta, int(wall), _p(out))
        if rc:
            raise OracleError(rc)
        return out.reshape(3, 3)

    # -- a3 / a4 / a5 / a6 ---------------------------------------------------
    def rotne_prager_tensor(self, r, a, eta, wall):
        r = _f64(r); n = r.size
        M = np.zeros((n, n), order="F")
        rc = self.L.orc_rotne_prager_tensor(_p(r), n, a, eta, int(wall), _p(M))
        if rc:
            raise OracleError(rc)
        return M

    def damp(self, r, a):
        r = _f64(r)
        B = np.zeros(r.size)
        self.L.orc_make_damp(_p(r), r.size, a, _p(B))
        return B

    def apply_M(self, F, r, a, eta, wall, mode="dense"):
        F = _f64(F); r = _f64(r)
        U = np.zeros(r.size)
        fn = self.L.orc_apply_M_dense if mode == "dense" else self.L.orc_apply_M_matfree
        rc = fn(_p(F), _p(r), r.size, a, eta, int(wall), _p(U))
        if rc:
            raise OracleError(rc)
        return U

    def apply_M_rows(self, F, r, row_begin, row_end, a, eta, wall, nthreads=1):
        F = _f64(F); r = _f64(r)
        U = np.zeros(3 * (row_end - row_begin))
        rc = self.L.orc_apply_M_rows(_p(F), _p(r), r.size, row_begin, row_end, a, eta,
                                     int(wall), int(nthreads), _p(U))
        if rc:
            raise OracleError(rc)
        return U

    def cholesky_lower(self, M):
        A = np.array(M, dtype=np.float64, order="F", copy=True)
        rc = self.L.orc_cholesky_lower(_p(A), A.shape[0])
        if rc:
            raise OracleError(rc)
        return A

    def M_half_W(self, r, a, eta, wall, W, return_L=False):
        r = _f64(r); W = _f64(W); n = r.size
        out = np.zeros(n)
        Lm = np.zeros((n, n), order="F") if return_L else None
        rc = self.L.orc_M_half_W(_p(r), n, a, eta, int(wall), _p(W), _p(out),
                                 _p(Lm) if return_L else None)
        if rc:
            raise OracleError(rc)
        return (out, Lm) if return_L else out

    # -- a8 ------------------------------------------------------------------
    def multi_body_pos(self, X, Q, ref_cfg):
        X = _f64(X); Q = _f64(Q); cfg = _f64(ref_cfg)
        nb = X.size // 3; nblb = cfg.size // 3
        out = np.zeros(3 * nb * nblb)
        self.L.orc_multi_body_pos(_p(X), _p(Q), _p(cfg), nb, nblb, _p(out))
        return out


class RefPair:
    """The REFERENCE's own compiled pair kernels (oracle/_ref/libref_pair.so)."""

    def __init__(self):
        if not os.path.exists(ref_lib_path()):
            build()
        if not os.path.exists(ref_lib_path()):
            raise FileNotFoundError(ref_lib_path())
        L = C.CDLL(ref_lib_path())
        L.ref_mobilityUFRPY.argtypes = [C.c_double] * 3 + [_dp, C.c_int, C.c_int, C.c_double]
        L.ref_mobilityUFSingleWallCorrection.argtypes = [C.c_double] * 3 + [_dp, C.c_int, C.c_int, C.c_double]
        self.L = L

    def rpy(self, rx, ry, rz, i, j, inv_a):
        out = np.zeros(6)
        self.L.ref_mobilityUFRPY(rx, ry, rz, _p(out), i, j, inv_a)
        return out

    def wall(self, rx, ry, rz, M9, i, j, hj):
        M = _f64(M9).copy()
        rc = self.L.ref_mobilityUFSingleWallCorrection(rx, ry, rz, _p(M), i, j, hj)
        if rc:
            raise OracleError(rc)
        return M


# =========================================================================
# numpy restatement of the O(N_bod) bookkeeping around the hot path
# =========================================================================
def remove_mean(cfg):
    """c_rigid_obj.cpp:176-181."""
    cfg = np.asarray(cfg, dtype=np.float64).reshape(-1, 3)
    return cfg - cfg.mean(axis=0)


def normalize_quats(Q):
    """setConfig, c_rigid_obj.cpp:212-216 (scalar-first, normalised)."""
    Q = np.asarray(Q, dtype=np.float64).reshape(-1, 4)
    return Q / np.linalg.norm(Q, axis=1, keepdims=True)


def rot_matrix(q):
    """Unit quaternion (w,x,y,z) -> rotation matrix (Eigen toRotationMatrix)."""
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    """Hamilton product a*b, scalar-first."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx])


def K_matrix(X, Qn, ref_cfg):
    """Dense K (3N x 6Nb), c_rigid_obj.cpp:368-383: u_blob = U + Omega x r."""
    X = np.asarray(X).reshape(-1, 3); Qn = np.asarray(Qn).reshape(-1, 4)
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    nb, nl = X.shape[0], cfg.shape[0]
    K = np.zeros((3 * nb * nl, 6 * nb))
    for j in range(nb):
        rr = cfg @ rot_matrix(Qn[j]).T
        for k in range(nl):
            row = 3 * (j * nl + k)
            rx, ry, rz = rr[k]
            K[row:row + 3, 6 * j:6 * j + 3] = np.eye(3)
            K[row + 0, 6 * j + 4] = rz;  K[row + 0, 6 * j + 5] = -ry
            K[row + 1, 6 * j + 5] = rx;  K[row + 1, 6 * j + 3] = -rz
            K[row + 2, 6 * j + 3] = ry;  K[row + 2, 6 * j + 4] = -rx
    return K


def Kinv_matrix(X, Qn, ref_cfg):
    """(K^T K)^-1 K^T with the reference's block formula, c_rigid_obj.cpp:302-326,390."""
    X = np.asarray(X).reshape(-1, 3); Qn = np.asarray(Qn).reshape(-1, 4)
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    nb, nl = X.shape[0], cfg.shape[0]
    K = K_matrix(X, Qn, cfg)
    sumr2 = float(np.sum(cfg * cfg))
    MOI = cfg.T @ cfg
    KTKi = np.zeros((6 * nb, 6 * nb))
    for j in range(nb):
        R = rot_matrix(Qn[j])
        D = sumr2 * np.eye(3) - R @ MOI @ R.T
        KTKi[6 * j:6 * j + 3, 6 * j:6 * j + 3] = np.eye(3) / nl
        KTKi[6 * j + 3:6 * j + 6, 6 * j + 3:6 * j + 6] = np.linalg.inv(D)
    return KTKi @ K.T


def apply_PC(orc, IN, X, Qn, ref_cfg, a, eta, wall, block):
    """apply_PC, c_rigid_obj.cpp:589-616 with invM from :461-552 (dense numpy)."""
    X = np.asarray(X).reshape(-1, 3); Qn = np.asarray(Qn).reshape(-1, 4)
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    nb, nl = X.shape[0], cfg.shape[0]
    n = 3 * nb * nl
    r = orc.multi_body_pos(X, Qn, cfg)
    invM = np.zeros((n, n))
    if block:      # Block_diag_invM :461-487
        for b in range(nb):
            sl = slice(3 * nl * b, 3 * nl * (b + 1))
            invM[sl, sl] = np.linalg.inv(orc.rotne_prager_tensor(r[sl], a, eta, wall))
    else:          # diag_invM :489-543 (self blocks only)
        for i in range(nb * nl):
            blk = np.eye(3) * (4.0 / 3.0)
            if wall:
                blk = orc.wall(0.0, 0.0, (0.0 + 2 * r[3 * i + 2]) / a, blk, i, i,
                               r[3 * i + 2] / a).reshape(3, 3)
            invM[3 * i:3 * i + 3, 3 * i:3 * i + 3] = np.linalg.inv(blk) * (8.0 * np.pi * eta * a)
    K = K_matrix(X, Qn, cfg)
    IN = np.asarray(IN, dtype=np.float64).reshape(-1)
    slip, F = IN[:n], IN[n:]
    Ninv = K.T @ invM @ K
    RHS = -F - K.T @ (invM @ slip)
    U = np.zeros(6 * nb)
    for b in range(nb):
        U[6 * b:6 * b + 6] = np.linalg.solve(Ninv[6 * b:6 * b + 6, 6 * b:6 * b + 6], RHS[6 * b:6 * b + 6])
    lam = invM @ (slip + K @ U)          # M_scale = 1 (:194)
    return np.concatenate([lam, U])


def evolve(X, Qn, U, dt):
    """evolve_X_Q, c_rigid_obj.cpp:865-878 with update_X_Q :691-710, Q_from_Om :679-689."""
    X = np.array(X, dtype=np.float64).reshape(-1, 3)
    Qn = np.array(Qn, dtype=np.float64).reshape(-1, 4)
    U = np.asarray(U, dtype=np.float64).reshape(-1, 6) * dt
    for j in range(X.shape[0]):
        om = U[j, 3:]
        nrm = np.linalg.norm(om)
        q = np.array([np.cos(nrm / 2.0), 0.0, 0.0, 0.0])
        if nrm > 1.0e-10:
            q[1:] = (np.sin(nrm / 2.0) / nrm) * om
        q /= np.linalg.norm(q)
        qq = quat_mul(q, Qn[j])
        Qn[j] = qq / np.linalg.norm(qq)
        X[j] += U[j, :3]
    return X, Qn


def update_X_Q(X, Qn, U):
    """update_X_Q, c_rigid_obj.cpp:691-710: U has displacement units (no dt)."""
    return evolve(X, Qn, U, 1.0)


def M_RFD(orc, W, X, Qn, ref_cfg, a, eta, wall, delta):
    """c_rigid_obj.cpp:769-796 with the noise W injected."""
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    uom = Kinv_matrix(X, Qn, cfg) @ W
    Xp, Qp = update_X_Q(X, Qn, 0.5 * delta * uom)
    Xm, Qm = update_X_Q(X, Qn, -0.5 * delta * uom)
    Mp = orc.apply_M(W, orc.multi_body_pos(Xp, Qp, cfg), a, eta, wall)
    Mm = orc.apply_M(W, orc.multi_body_pos(Xm, Qm, cfg), a, eta, wall)
    return (Mp - Mm) / delta


def M_RFD_cfgs(orc, U, X, Qn, ref_cfg, delta):
    """c_rigid_obj.cpp:798-818: blob positions at q +- (delta/2) U."""
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    Xp, Qp = update_X_Q(X, Qn, 0.5 * delta * np.asarray(U))
    Xm, Qm = update_X_Q(X, Qn, -0.5 * delta * np.asarray(U))
    return orc.multi_body_pos(Xp, Qp, cfg), orc.multi_body_pos(Xm, Qm, cfg)


def M_RFD_from_U(orc, U, W, X, Qn, ref_cfg, a, eta, wall, delta=1.0e-3):
    """c_rigid_obj.cpp:820-842 (delta = 1e-3 hard-coded there, :822)."""
    rp, rm = M_RFD_cfgs(orc, U, X, Qn, ref_cfg, delta)
    return (orc.apply_M(W, rp, a, eta, wall) - orc.apply_M(W, rm, a, eta, wall)) / delta


def KT_RFD_from_U(U, W, X, Qn, ref_cfg, delta=1.0e-3):
    """c_rigid_obj.cpp:844-863."""
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    Xp, Qp = update_X_Q(X, Qn, 0.5 * delta * np.asarray(U))
    Xm, Qm = update_X_Q(X, Qn, -0.5 * delta * np.asarray(U))
    return (K_matrix(Xp, Qp, cfg).T @ W - K_matrix(Xm, Qm, cfg).T @ W) / delta


def KTinv_RFD(W, X, Qn, ref_cfg, delta):
    """c_rigid_obj.cpp:743-767."""
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    Xp, Qp = update_X_Q(X, Qn, 0.5 * delta * W)
    Xm, Qm = update_X_Q(X, Qn, -0.5 * delta * W)
    out = (Kinv_matrix(Xp, Qp, cfg).T @ W - Kinv_matrix(Xm, Qm, cfg).T @ W) / delta
    return K_matrix(X, Qn, cfg).T @ out


def RHS_and_Midpoint(orc, Slip, Force, W1, W2, W_rfd, X, Qn, ref_cfg, a, eta, wall, dt, kBT,
                     split_rand=True, delta=1.0e-4):
    """RHS_and_Midpoint, c_rigid_obj.cpp:917-976, with the three noise vectors injected (the reference
    draws them clock-seeded, :730-741).  Returns (RHS, X_half, Q_half); the inputs are not modified."""
    cfg = np.asarray(ref_cfg).reshape(-1, 3)
    Slip = np.array(Slip, dtype=np.float64).reshape(-1)
    Force = np.array(Force, dtype=np.float64).reshape(-1)
    Xh, Qh = np.array(X, dtype=np.float64).reshape(-1, 3), np.array(Qn, dtype=np.float64).reshape(-1, 4)
    if kBT > 1e-10:                                                     # :922
        r = orc.multi_body_pos(X, Qn, cfg)
        mw1 = orc.M_half_W(r, a, eta, wall, W1)                         # :927
        mw2 = orc.M_half_W(r, a, eta, wall, W2) if split_rand else None  # :934-936
        rfd = M_RFD(orc, W_rfd, X, Qn, cfg, a, eta, wall, delta)        # :940
        if split_rand:                                                  # :945-948
            c1, c2 = 2.0 * np.sqrt(kBT / dt), np.sqrt(kBT / dt)
            BI = c2 * (mw1 - mw2)
        else:                                                           # :950-953
            c1 = c2 = np.sqrt(2.0 * kBT / dt)
            BI = c2 * mw1
        uom_half = (dt / 2.0) * (Kinv_matrix(X, Qn, cfg) @ (c1 * mw1))  # :955-956
        Xh, Qh = update_X_Q(X, Qn, uom_half)                            # :958
        Slip = Slip - (kBT * rfd + BI)                                  # :963
    return np.concatenate([Slip, -Force]), Xh, Qh                       # :972-975
