"""Time-step DRIVER on top of the operator surface -- the "external user code" of the reference
(Krylov solver + time loop are NOT in the reference repo).  It defines what
this project calls one deterministic time step:

    positions -> right-preconditioned GMRES on  A = apply_saddle  with  P^-1 = apply_PC ,
                 rhs = [0 ; -F_body]  ->  U  ->  evolve_rigid_bodies(U)

All vectors stay on the GPU (torch tensors as plain device buffers); the operators are the HIP
kernels behind include/rbl.h (rbl_apply_saddle_dev, rbl_apply_PC_dev).  The small Hessenberg
least-squares problem is solved on the host once per solve.
"""
import numpy as np
import torch


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


class DeterministicStepper:
    """One deterministic time step per call (fixed-work: `iters` GMRES iterations = iters+1 apply_M).

    use_graph=True captures the whole fixed-work solve (every HIP kernel of the operators and every
    torch vector op of the Arnoldi process) in ONE hipGraph and replays it each step: the small
    configurations are launch-bound (cfg 1: ~14 us of kernels per apply_M), a replay removes the
    per-launch host cost.  The non-launch work (uploads, PC build) happens in ctx.prepare()."""

    def __init__(self, ctx, n_bodies, blobs_per_body, device, use_graph=False, native=False):
        self.ctx, self.nb, self.nblb, self.dev = ctx, n_bodies, blobs_per_body, device
        self.n3 = 3 * n_bodies * blobs_per_body
        self.size = self.n3 + 6 * n_bodies
        self.use_graph = use_graph
        self.native = native          # librbl's own GMRES (rbl_gmres_saddle_dev) instead of the torch Arnoldi loop
        self.warm_start = False       # native solver, converged mode: start from the previous step's solution ...
        self.extrapolate = 0          # ... 1: from 2 x_n - x_{n-1}, 2: from 3 x_n - 3 x_{n-1} + x_{n-2} (the solution moves
        self._x_hist = []             #     smoothly with the configuration); history of the last solutions, newest first
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
        """Solve the saddle system for rhs = [0 ; -F_body]; returns (lambda, U, iterations, residual)."""
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1)
        h = self._x_hist
        warm = self.warm_start and rtol is not None and len(h) > 0 and h[0].numel() == self.size
        order = min(int(self.extrapolate), len(h) - 1) if warm else 0
        x0 = None if not warm else (3.0 * h[0] - 3.0 * h[1] + h[2] if order >= 2 else 2.0 * h[0] - h[1] if order == 1 else h[0].clone())
        if self.native:
            b = torch.zeros(self.size, dtype=torch.float64, device=self.dev)
            b[self.n3:] = -Fb
            x = x0 if warm else torch.empty_like(b)
            m, resid = self.ctx.gmres_saddle(b.data_ptr(), iters, rtol, x.data_ptr(), use_x0=warm)
            if self.warm_start:
                self._x_hist = [x] + h[:2]
            return x[: self.n3], x[self.n3:], m, resid
        if rtol is not None or not self.use_graph:
            b = torch.zeros(self.size, dtype=torch.float64, device=self.dev)
            b[self.n3:] = -Fb
            x, m, resid = gmres_right_pc(self._A, self._Pinv, b, iters, rtol, x0=x0)
            if self.warm_start:
                self._x_hist = [x] + h[:2]
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

    def step(self, F_body, iters=20, rtol=None):
        lam, U, m, resid = self.solve(F_body, iters, rtol)
        self.ctx.evolve(U.cpu().numpy())          # O(N_bod) host update, then K/positions rebuilt on the GPU
        self.ctx.sync_check()
        return m, resid


def lanczos_mhalf(apply_A, W, max_iter=100, tol=1e-3, check_every=1):
    """Matrix-free M^{1/2} W by Lanczos (same algorithm as librbl's rbl_M_half_W(..., LANCZOS), written
    on torch vectors so the operator can be the multi-GPU sharded product: with the symmetric
    sharding every rank holds the full vectors, the recurrences are replicated and only `apply_A`
    communicates).  apply_A: tensor -> tensor computing (B M B) v.  Returns (y, iterations, change)."""
    n = W.numel()
    V = torch.empty(max_iter + 1, n, dtype=W.dtype, device=W.device)
    wnorm = float(torch.linalg.norm(W))
    if wnorm == 0.0:
        return torch.zeros_like(W), 0, 0.0
    V[0] = W / wnorm
    alpha, beta = [], []
    y_prev, y_cur, change, m = None, None, 1.0, 0
    for it in range(max_iter):
        u = apply_A(V[it])
        if it > 0:
            u = u - beta[-1] * V[it - 1]
        al = float(torch.dot(V[it], u))
        u = u - al * V[it]
        be = float(torch.linalg.norm(u))
        alpha.append(al)
        m = it + 1
        if m % check_every == 0 or it + 1 == max_iter:
            T = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1)
            lam, Z = np.linalg.eigh(T)
            y_cur = wnorm * (Z @ (np.sqrt(np.clip(lam, 0.0, None)) * Z[0]))
            if y_prev is not None:
                yp = np.zeros(m); yp[: y_prev.size] = y_prev
                change = float(np.linalg.norm(y_cur - yp) / np.linalg.norm(y_cur))
            y_prev = y_cur
            if change < tol:
                break
        if be < 1e-300 or it + 1 == max_iter:
            break
        beta.append(be)
        V[it + 1] = u / be
    out = torch.from_numpy(y_cur).to(W.device) @ V[:m]
    return out, m, change


class ShardedDeterministicStepper(DeterministicStepper):
    """The same deterministic step on P GPUs (one process each).  native=True (default): librbl's own GMRES
    (rbl_gmres_saddle_dev) with the context switched to multi-GPU products (DeviceContext.set_comm -> rbl_set_comm):
    every mobility product of the Arnoldi loop is this rank's share of the unordered tile pairs + ONE all-reduce of the
    partial U; the recurrences stay on the device, the host looks at the Hessenberg matrix once per convergence test.
    Everything else -- K ops, preconditioner, Krylov vectors -- is O(N), replicated and bitwise identical on every rank.
    native=False keeps the torch Arnoldi loop around the same sharded product (the first implementation; tests
    compare the two)."""

    def __init__(self, ctx, sharded, n_bodies, blobs_per_body, device, native=True):
        super().__init__(ctx, n_bodies, blobs_per_body, device, use_graph=False, native=native)
        self.sm = sharded
        if native:
            ctx.set_comm(sharded)

    def refresh_positions(self):
        p, n = self.ctx.positions_ptr()                 # replicated body state -> full positions on this rank
        self.sm.r_full = torch.empty(3 * n, dtype=torch.float64, device=self.dev)
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

    def step(self, F_body, iters=20, rtol=None):
        if not self.native:
            self.refresh_positions()
        return super().step(F_body, iters, rtol)


class BrownianStepper(DeterministicStepper):
    """One stochastic (midpoint) time step, assembled from the pieces the reference leaves unassembled
    (`RHS_and_Midpoint`, c_rigid_obj.cpp:917-976):

      1. at q^n      : rhs = [slip - (kBT M_RFD + BI) ; -F_body]  and the predictor configuration
                       q^{n+1/2} = q^n displaced by (dt/2) Kinv c1 M^{1/2} W1      (librbl: rbl_RHS_and_Midpoint_dev)
      2. at q^{n+1/2}: right-preconditioned GMRES on the saddle operator -> [lambda ; U]
      3.               q^{n+1} = q^n displaced by dt U                              (evolve_X_Q, :865-878)

    Steps 2-3 are this driver's completion of the scheme (the reference computes q^{n+1/2} but never
    uses it).  The context must have been created with dt > 0 and kBT; kBT <= 1e-10 reduces this to
    the deterministic step.  method: 0 = dense Cholesky (the reference's M_half_W), 1 = Lanczos."""

    def step(self, F_body, slip=None, W=None, seed=0, method=1, iters=20, rtol=None, split_rand=True,
             delta=1.0e-4):
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1).contiguous()
        sl = (torch.zeros(self.n3, dtype=torch.float64, device=self.dev) if slip is None else
              torch.as_tensor(slip, dtype=torch.float64, device=self.dev).reshape(-1).contiguous())
        Wd = None if W is None else torch.as_tensor(W, dtype=torch.float64, device=self.dev).reshape(-1).contiguous()
        if Wd is not None and Wd.numel() != 3 * self.n3:
            raise ValueError("W must hold 3 noise vectors [W1 | W2 | W_rfd] of length 3*N_blobs each")
        Xn, Qn = self.ctx.get_config(self.nb)
        rhs = torch.empty(self.size, dtype=torch.float64, device=self.dev)
        Xh, Qh = self.ctx.RHS_and_Midpoint(sl.data_ptr(), Fb.data_ptr(), None if Wd is None else Wd.data_ptr(),
                                           seed, method, split_rand, delta, rhs.data_ptr(), self.nb)
        self.ctx.set_config(Xh, Qh)                      # operators and preconditioner at the predictor configuration
        if self.native:
            x = torch.empty_like(rhs)
            m, resid = self.ctx.gmres_saddle(rhs.data_ptr(), iters, rtol, x.data_ptr())
        else:
            x, m, resid = gmres_right_pc(self._A, self._Pinv, rhs, iters, rtol)
        U = x[self.n3:].cpu().numpy()
        self.ctx.set_config(Xn, Qn)                      # the update starts from q^n
        self.ctx.evolve(U)
        self.ctx.sync_check()
        return m, resid


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


class ShardedBrownianStepper(ShardedDeterministicStepper):
    """The stochastic midpoint step of BrownianStepper on P GPUs (BASELINE.json configs[3]).  The right-hand
    side of c_rigid_obj.cpp:917-976 is composed here from device vector operations so that every mobility
    product -- the Lanczos iterations of the two M^{1/2} W, the two products of M_RFD, the GMRES iterations --
    is the tile-pair-sharded one (one all-reduce each); all vectors and the O(N_bod) body state are
    replicated and bitwise identical on every rank (the noise comes from a seeded device generator)."""

    def __init__(self, ctx, sharded, n_bodies, blobs_per_body, device, a, wall, kBT, dt,
                 lanczos_tol=1e-3, lanczos_max_iter=100, precondition=True, native=True):
        super().__init__(ctx, sharded, n_bodies, blobs_per_body, device, native=native)
        self.a, self.wall, self.kBT, self.dt = a, wall, kBT, dt
        self.ltol, self.lmax = lanczos_tol, lanczos_max_iter
        self.precondition = precondition      # block-Jacobi preconditioned square root (librbl's RBL_MHALF_LANCZOS_PC)
        self.lanczos_iterations = []

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
        if self.native:
            # librbl's own stochastic midpoint step pieces (rbl_RHS_and_Midpoint_dev: lock-step Lanczos, M_RFD; then
            # rbl_gmres_saddle_dev), every product sharded through the context's communicator (rbl_set_comm)
            self.ctx.set_lanczos(self.lmax, self.ltol)
            out = BrownianStepper.step(self, F_body, slip=slip, W=W, seed=seed, method=2 if self.precondition else 1,
                                       iters=iters, rtol=rtol, split_rand=split_rand, delta=delta)
            self.lanczos_iterations = [self.ctx.lanczos_report()[0]] * (2 if split_rand else 1)
            return out
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
    """k independent Brownian increments M^{1/2} W_c at once: k Lanczos recurrences advanced in lockstep,
    so that every iteration is ONE multi-vector product -- which librbl runs on the fp64 matrix
    cores for k >= 4 (rbl_apply_M_multi_dev, 16 vectors per pass).  W: (k, n) tensor.
    apply_A_multi: (k, n) -> (k, n) computing (B M B) v_c for every row.  agree: bool -> bool, makes the stopping
    decision the same on every rank of a sharded product (ShardedMobility.agree).  Returns (Y (k,n), iterations, change)."""
    k, n = W.shape
    dev = W.device
    V = torch.empty(max_iter + 1, k, n, dtype=W.dtype, device=dev)
    wnorm = torch.linalg.norm(W, dim=1)
    V[0] = W / wnorm[:, None]
    alpha = np.zeros((max_iter, k)); beta = np.zeros((max_iter, k))
    wn = wnorm.cpu().numpy()
    y_prev = [None] * k
    coef = None
    change = np.ones(k)
    m = 0
    for it in range(max_iter):
        U = apply_A_multi(V[it].contiguous())
        if it > 0:
            U = U - torch.from_numpy(beta[it - 1]).to(dev)[:, None] * V[it - 1]
        al = (V[it] * U).sum(dim=1)
        U = U - al[:, None] * V[it]
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
                change[c] = np.linalg.norm(coef[c] - yp) / np.linalg.norm(coef[c])
            y_prev[c] = coef[c]
        done = bool(change.max() < tol or it + 1 == max_iter or beta[it].min() < 1e-300)
        if agree(done) if agree is not None else done:
            break
        V[it + 1] = U / be[:, None]
    Y = torch.einsum("ck,kcn->cn", torch.from_numpy(coef).to(dev), V[:m])
    return Y, m, float(change.max())
