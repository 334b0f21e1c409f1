"""Torch restatements of the Krylov loops, kept as COMPARATORS for the tests (and a few tools): the package's steppers
(rigid_body_light_amd/krylov.py) call librbl's own loops only.  Here the same algorithms are written on torch vectors --
right-preconditioned GMRES with a host least-squares per test, the Lanczos square roots with a host eigen-solve, the
fixed-work Arnoldi loop replayed as one hipGraph, and the multi-GPU step composed from sharded products + torch
collectives -- around the SAME HIP operators (rbl_apply_saddle_dev, rbl_apply_PC_dev, rbl_apply_M_sym_dev, ...)."""
import numpy as np
import torch

from rigid_body_light_amd.krylov import BrownianStepper, DeterministicStepper, ShardedBrownianStepper


def gmres_right_pc(apply_A, apply_Pinv, b, iters, rtol=None, x0=None):
    """Right-preconditioned GMRES(iters), no restart.  apply_A / apply_Pinv: tensor -> tensor.
    Arnoldi with classical Gram-Schmidt applied twice (two GEMVs each, no host sync inside the
    loop unless rtol is given).  x0: initial guess -- the correction is solved for from r0 = b - A x0, the
    residual stays relative to |b|.  Returns (x, number of iterations, relative residual estimate)."""
    n = b.numel()
    dev, dt = b.device, b.dtype
    bnorm = float(torch.linalg.norm(b))
    if x0 is not None:
        b = b - apply_A(x0)
        if rtol is not None and float(torch.linalg.norm(b)) <= rtol * bnorm:
            return x0.clone(), 0, float(torch.linalg.norm(b)) / bnorm
    V = torch.zeros(iters + 1, n, dtype=dt, device=dev)
    H = torch.zeros(iters + 1, iters, dtype=dt, device=dev)
    beta = torch.linalg.norm(b)
    V[0] = b / beta
    m = iters
    for j in range(iters):
        w = apply_A(apply_Pinv(V[j]))
        for _ in range(2):                       # CGS2
            h = V[: j + 1] @ w
            w = w - h @ V[: j + 1]
            H[: j + 1, j] += h
        hn = torch.linalg.norm(w)
        H[j + 1, j] = hn
        V[j + 1] = w / hn
        if rtol is not None:                     # host check costs one sync per iteration
            Hh = H[: j + 2, : j + 1].cpu().numpy()
            e1 = np.zeros(j + 2); e1[0] = float(beta)
            y, res, *_ = np.linalg.lstsq(Hh, e1, rcond=None)
            r = np.linalg.norm(Hh @ y - e1) / bnorm
            if r < rtol:
                m = j + 1
                break
    Hh = H[: m + 1, :m].cpu().numpy()
    e1 = np.zeros(m + 1); e1[0] = float(beta)
    y, *_ = np.linalg.lstsq(Hh, e1, rcond=None)
    resid = float(np.linalg.norm(Hh @ y - e1) / bnorm)
    z = torch.from_numpy(y).to(dev) @ V[:m]
    x = apply_Pinv(z)
    return (x if x0 is None else x0 + x), m, resid



class TorchDeterministicStepper(DeterministicStepper):
    """DeterministicStepper with the torch Arnoldi loop around the HIP operators instead of rbl_gmres_saddle_dev.
    use_graph=True captures the whole fixed-work solve (every HIP kernel of the operators and every torch vector op of
    the Arnoldi process) in ONE hipGraph and replays it each step."""

    def __init__(self, ctx, n_bodies, blobs_per_body, device, use_graph=False):
        super().__init__(ctx, n_bodies, blobs_per_body, device)
        self.use_graph = use_graph
        self._graph = None

    def _A(self, x):
        out = torch.empty_like(x)
        self.ctx.apply_saddle(x.data_ptr(), out.data_ptr())
        return out

    def _Pinv(self, x):
        out = torch.empty_like(x)
        self.ctx.apply_PC(x.contiguous().data_ptr(), out.data_ptr())
        return out

    def _arnoldi(self, b, iters):
        """sync-free part of GMRES: returns (V, H, beta) as device tensors"""
        n = b.numel()
        V = torch.zeros(iters + 1, n, dtype=b.dtype, device=b.device)
        H = torch.zeros(iters + 1, iters, dtype=b.dtype, device=b.device)
        beta = torch.linalg.norm(b)
        V[0] = b / beta
        for j in range(iters):
            w = self._A(self._Pinv(V[j]))
            for _ in range(2):
                h = V[: j + 1] @ w
                w = w - h @ V[: j + 1]
                H[: j + 1, j] += h
            hn = torch.linalg.norm(w)
            H[j + 1, j] = hn
            V[j + 1] = w / hn
        return V, H, beta

    def _finish(self, V, H, beta, iters):
        Hh = H.cpu().numpy()
        e1 = np.zeros(iters + 1); e1[0] = float(beta)
        y, *_ = np.linalg.lstsq(Hh, e1, rcond=None)
        resid = float(np.linalg.norm(Hh @ y - e1) / float(beta))
        z = torch.from_numpy(y).to(self.dev) @ V[:iters]
        return self._Pinv(z), resid

    def solve(self, F_body, iters=20, rtol=None):
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1)
        if rtol is not None or not self.use_graph:
            x0 = self.initial_guess(rtol)
            b = torch.zeros(self.size, dtype=torch.float64, device=self.dev)
            b[self.n3:] = -Fb
            x, m, resid = gmres_right_pc(self._A, self._Pinv, b, iters, rtol, x0=x0)
            self.remember(x)
            return x[: self.n3], x[self.n3:], m, resid
        self.ctx.prepare()
        if self._graph is None or self._graph_iters != iters:
            self._b = torch.zeros(self.size, dtype=torch.float64, device=self.dev)
            self._b[self.n3:] = -Fb
            self._arnoldi(self._b, min(iters, 2))               # eager warm-up: allocator + workspaces
            torch.cuda.synchronize()
            cap_stream = torch.cuda.Stream()
            with torch.cuda.stream(cap_stream):
                self.ctx.set_stream(cap_stream.cuda_stream)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap_stream):
                    self._V, self._H, self._beta = self._arnoldi(self._b, iters)
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            self._graph, self._graph_iters = g, iters
        self._b[self.n3:] = -Fb
        self._graph.replay()
        x, resid = self._finish(self._V, self._H, self._beta, iters)
        return x[: self.n3], x[self.n3:], iters, resid


class TorchBrownianStepper(BrownianStepper, TorchDeterministicStepper):
    """BrownianStepper with the saddle solve in the torch GMRES loop (the right-hand side still comes from librbl)"""

    def __init__(self, ctx, n_bodies, blobs_per_body, device):
        TorchDeterministicStepper.__init__(self, ctx, n_bodies, blobs_per_body, device)

    def saddle_solve(self, rhs, iters, rtol):
        return gmres_right_pc(self._A, self._Pinv, rhs, iters, rtol)


def _error_estimate(d_cur, d_prev):
    """librbl's stopping estimate: the last correction d_m extrapolated geometrically to the error, d_m rho / (1 - rho)"""
    rho = min(d_cur / d_prev, 0.95) if d_prev is not None and d_prev > 0.0 else 0.5
    return d_cur * rho / (1.0 - rho)


def lanczos_mhalf(apply_A, W, max_iter=100, tol=1e-3, check_every=1):
    """Matrix-free M^{1/2} W by Lanczos with full re-orthogonalisation (same algorithm and stopping estimate as librbl's
    rbl_M_half_W(..., LANCZOS), written on torch vectors).  apply_A: tensor -> tensor computing (B M B) v.
    Returns (y, iterations, error estimate)."""
    n = W.numel()
    V = torch.empty(max_iter + 1, n, dtype=W.dtype, device=W.device)
    wnorm = float(torch.linalg.norm(W))
    if wnorm == 0.0:
        return torch.zeros_like(W), 0, 0.0
    V[0] = W / wnorm
    alpha, beta = [], []
    y_prev, y_cur, est, d_prev, m = None, None, 1.0, None, 0
    for it in range(max_iter):
        u = apply_A(V[it])
        h = torch.zeros(it + 1, dtype=W.dtype, device=W.device)
        for _ in range(2):                       # classical Gram-Schmidt twice against the whole basis
            hh = V[: it + 1] @ u
            u = u - hh @ V[: it + 1]
            h += hh
        al = float(h[it])
        be = float(torch.linalg.norm(u))
        alpha.append(al)
        m = it + 1
        T = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1)
        lam, Z = np.linalg.eigh(T)
        y_cur = wnorm * (Z @ (np.sqrt(np.clip(lam, 0.0, None)) * Z[0]))
        if y_prev is not None:
            yp = np.zeros(m); yp[: y_prev.size] = y_prev
            d_cur = float(np.linalg.norm(y_cur - yp) / np.linalg.norm(y_cur))
            est = _error_estimate(d_cur, d_prev)
            d_prev = d_cur
        y_prev = y_cur
        if (m % check_every == 0 or it + 1 == max_iter) and est < tol:
            break
        if be < 1e-300 or it + 1 == max_iter:
            break
        beta.append(be)
        V[it + 1] = u / be
    out = torch.from_numpy(y_cur).to(W.device) @ V[:m]
    return out, m, est


def sharded_mhalf_W(ctx, sm, r_full, Wk, a, wall, tol=1e-3, max_iter=100, precondition=True):
    """Brownian increments (B M B)^{1/2} W_k for the nv = 1 or 2 rows of Wk on the tile-pair-sharded product
    (sm: ShardedMobility; all vectors replicated, one all-reduce per iteration).  Two vectors advance in lock step
    through ONE two-vector product per iteration (shared pair coefficients).
      precondition=True : x = B L S^{1/2} W with S = L^-1 M L^-T and the per-body Cholesky factors L: every rank
                          factors and substitutes only ITS bodies (sm.b0 .. sm.b1) and one all-gather per
                          substitution shares the result; covariance B M B exactly, ~7 iterations instead of ~35;
      precondition=False: Lanczos on B M B itself (the symmetric square root).
    Returns (Y (nv, n), iterations)."""
    nv, n3 = Wk.shape
    z = r_full.view(-1, 3)[:, 2]
    B = torch.where(z >= a, torch.ones_like(z), z / a).repeat_interleave(3)            # make_damp_mat :618-639

    def product(X, no_damp):
        X = X.contiguous()
        part = torch.empty_like(X)
        if no_damp:
            ctx.set_no_damp(True)
        try:
            ctx.apply_M_sym_multi(X.data_ptr(), r_full.data_ptr(), n3 // 3, nv, sm.rank, sm.world, part.data_ptr())
        finally:
            if no_damp:
                ctx.set_no_damp(False)
        return sm.all_reduce_sum(part)

    def bsolve(v, mode):
        v = v.contiguous()
        out = torch.empty_like(v)
        ctx.block_solve(v.data_ptr(), out.data_ptr(), mode, sm.b0, sm.b1)
        if sm.world == 1:
            return out
        return sm.all_gather_rows(out[3 * sm.row0:3 * sm.row1])

    if precondition:
        def S_op(Vk):
            out = product(torch.stack([bsolve(Vk[k], 2) for k in range(nv)]), True)
            return torch.stack([bsolve(out[k], 1) for k in range(nv)])
        Y, its, _ = lanczos_mhalf_multi(S_op, Wk, max_iter, tol, agree=sm.agree)
        return torch.stack([B * bsolve(Y[k], 3) for k in range(nv)]), its
    if wall:                 # the wall kernel applies B M B itself (M_half_W always damps, :668-669)
        A_op = lambda Vk: product(Vk, False)
    else:
        A_op = lambda Vk: B * product(B * Vk, False)
    Y, its, _ = lanczos_mhalf_multi(A_op, Wk, max_iter, tol, agree=sm.agree)
    return Y, its



class TorchShardedBrownianStepper(TorchDeterministicStepper):
    """The multi-GPU stochastic step composed in Python (round 1's implementation): the right-hand side of
    c_rigid_obj.cpp:917-976 from device vector operations, every mobility product -- Lanczos iterations, M_RFD, GMRES
    iterations -- the tile-pair-sharded one + a torch.distributed all-reduce.  tests compare ShardedBrownianStepper
    (librbl's loops through rbl_set_comm) with it."""

    def __init__(self, ctx, sharded, n_bodies, blobs_per_body, device, a, wall, kBT, dt,
                 lanczos_tol=1e-3, lanczos_max_iter=100, precondition=True):
        super().__init__(ctx, n_bodies, blobs_per_body, device)
        self.sm = sharded
        self.a, self.wall, self.kBT, self.dt = a, wall, kBT, dt
        self.ltol, self.lmax = lanczos_tol, lanczos_max_iter
        self.precondition = precondition
        self.lanczos_iterations = []

    def refresh_positions(self):
        self.sm.r_full = torch.empty(self.n3, dtype=torch.float64, device=self.dev)
        self.ctx.blob_positions(0, self.nb, self.sm.r_full.data_ptr())

    def _A(self, x):
        n3 = self.n3
        lam = x[:n3].contiguous()
        part = torch.empty(n3, dtype=torch.float64, device=self.dev)
        self.ctx.apply_M_sym(lam.data_ptr(), self.sm.r_full.data_ptr(), n3 // 3, self.sm.rank, self.sm.world, part.data_ptr())
        Ml = self.sm.all_reduce_sum(part)
        out = torch.empty_like(x)
        ku = torch.empty(n3, dtype=torch.float64, device=self.dev)
        U = x[n3:].contiguous()
        self.ctx.K_x_U(U.data_ptr(), ku.data_ptr())
        out[:n3] = Ml - ku
        kt = torch.empty(6 * self.nb, dtype=torch.float64, device=self.dev)
        self.ctx.KT_x_Lam(lam.data_ptr(), kt.data_ptr())
        out[n3:] = kt
        return out

    def _product(self, r_full, v):
        """apply_M (reference :641-659) on the sharded pairs: B M B with the wall term, plain M without"""
        part = torch.empty(self.n3, dtype=torch.float64, device=self.dev)
        self.ctx.apply_M_sym(v.contiguous().data_ptr(), r_full.data_ptr(), self.n3 // 3, self.sm.rank, self.sm.world,
                             part.data_ptr())
        return self.sm.all_reduce_sum(part)

    def _positions_at(self, X, Q):
        Xn, Qn = self.ctx.get_config(self.nb)
        self.ctx.set_config(X, Q)
        r = torch.empty(self.n3, dtype=torch.float64, device=self.dev)
        self.ctx.blob_positions(0, self.nb, r.data_ptr())
        self.ctx.set_config(Xn, Qn)
        return r

    def rhs_and_midpoint(self, slip, Fb, W, split_rand=True, delta=1.0e-4):
        n3 = self.n3
        Xn, Qn = self.ctx.get_config(self.nb)
        r_n = self._positions_at(Xn, Qn)
        W1, W2, Wr = W[:n3], W[n3:2 * n3], W[2 * n3:]
        Wk = torch.stack([W1, W2]) if split_rand else W1[None, :]                      # :927-936
        Y, its = sharded_mhalf_W(self.ctx, self.sm, r_n, Wk, self.a, self.wall, self.ltol, self.lmax, self.precondition)
        mw1 = Y[0]
        mw2 = Y[1] if split_rand else None
        self.lanczos_iterations = [its] * Wk.shape[0]
        uom = self.ctx.Kinv_x_V(Wr.cpu().numpy(), self.nb)                            # M_RFD :776-794
        Mpm = [self._product(self._positions_at(*self.ctx.update_X_Q(sg * 0.5 * delta * uom, self.nb)), Wr)
               for sg in (1.0, -1.0)]
        rfd = (Mpm[0] - Mpm[1]) / delta
        kd = self.kBT / self.dt
        if split_rand:                                                                # :945-953
            c1, c2 = 2.0 * np.sqrt(kd), np.sqrt(kd)
            BI = c2 * (mw1 - mw2)
        else:
            c1 = c2 = np.sqrt(2.0 * kd)
            BI = c2 * mw1
        uom_half = 0.5 * self.dt * c1 * self.ctx.Kinv_x_V(mw1.cpu().numpy(), self.nb)   # :955-956
        Xh, Qh = self.ctx.update_X_Q(uom_half, self.nb)                               # :958
        rhs = torch.cat([slip - (self.kBT * rfd + BI), -Fb])                          # :963-975
        return rhs, Xh, Qh

    def step(self, F_body, slip=None, W=None, seed=0, iters=20, rtol=None, split_rand=True, delta=1.0e-4):
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1)
        sl = (torch.zeros(self.n3, dtype=torch.float64, device=self.dev) if slip is None else
              torch.as_tensor(slip, dtype=torch.float64, device=self.dev).reshape(-1))
        if W is None:                                     # same seed + same device type -> same numbers on every rank
            g = torch.Generator(device=self.dev); g.manual_seed(int(seed))
            W = torch.randn(3 * self.n3, dtype=torch.float64, device=self.dev, generator=g)
        else:
            W = torch.as_tensor(W, dtype=torch.float64, device=self.dev).reshape(-1)
        Xn, Qn = self.ctx.get_config(self.nb)
        rhs, Xh, Qh = self.rhs_and_midpoint(sl, Fb, W, split_rand, delta)
        self.ctx.set_config(Xh, Qh)                       # solve at the predictor configuration
        self.refresh_positions()
        x, m, resid = gmres_right_pc(self._A, self._Pinv, rhs, iters, rtol)
        U = x[self.n3:].cpu().numpy()
        self.ctx.set_config(Xn, Qn)                       # update from q^n
        self.ctx.evolve(U)
        self.ctx.sync_check()
        return m, resid


def lanczos_mhalf_multi(apply_A_multi, W, max_iter=100, tol=1e-3, agree=None):
    """k independent Brownian increments M^{1/2} W_c at once: k Lanczos recurrences (full re-orthogonalisation, librbl's
    stopping estimate) advanced in lockstep, so that every iteration is ONE multi-vector product -- which librbl runs on
    the fp64 matrix cores for k >= 4 (rbl_apply_M_multi_dev, 16 vectors per pass).  W: (k, n) tensor.
    apply_A_multi: (k, n) -> (k, n) computing (B M B) v_c for every row.  agree: bool -> bool, makes the stopping
    decision the same on every rank of a sharded product (ShardedMobility.agree).  Returns (Y (k,n), iterations, estimate)."""
    k, n = W.shape
    dev = W.device
    V = torch.empty(max_iter + 1, k, n, dtype=W.dtype, device=dev)
    wnorm = torch.linalg.norm(W, dim=1)
    V[0] = W / wnorm[:, None]
    alpha = np.zeros((max_iter, k)); beta = np.zeros((max_iter, k))
    wn = wnorm.cpu().numpy()
    y_prev = [None] * k
    d_prev = [None] * k
    coef = None
    est = np.ones(k)
    m = 0
    for it in range(max_iter):
        U = apply_A_multi(V[it].contiguous())
        al = torch.zeros(k, dtype=W.dtype, device=dev)
        for _ in range(2):                       # classical Gram-Schmidt twice, every recurrence against its own basis
            H = torch.einsum("jcn,cn->jc", V[: it + 1], U)
            U = U - torch.einsum("jc,jcn->cn", H, V[: it + 1])
            al += H[it]
        be = torch.linalg.norm(U, dim=1)
        alpha[it] = al.cpu().numpy(); beta[it] = be.cpu().numpy()
        m = it + 1
        coef = np.zeros((k, m))
        for c in range(k):
            T = np.diag(alpha[:m, c]) + np.diag(beta[:m - 1, c], 1) + np.diag(beta[:m - 1, c], -1)
            lam, Z = np.linalg.eigh(T)
            coef[c] = wn[c] * (Z @ (np.sqrt(np.clip(lam, 0.0, None)) * Z[0]))
            if y_prev[c] is not None:
                yp = np.zeros(m); yp[: y_prev[c].size] = y_prev[c]
                d_cur = np.linalg.norm(coef[c] - yp) / np.linalg.norm(coef[c])
                est[c] = _error_estimate(d_cur, d_prev[c])
                d_prev[c] = d_cur
            y_prev[c] = coef[c]
        done = bool(est.max() < tol or it + 1 == max_iter or beta[it].min() < 1e-300)
        if agree(done) if agree is not None else done:
            break
        V[it + 1] = U / be[:, None]
    Y = torch.einsum("ck,kcn->cn", torch.from_numpy(coef).to(dev), V[:m])
    return Y, m, float(est.max())
