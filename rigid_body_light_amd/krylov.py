"""Time-step DRIVER on top of the operator surface -- the "external user code" of the reference
(Krylov solver + time loop are NOT in the reference repo).  It defines what
this project calls one time step:

    deterministic:  positions -> right-preconditioned GMRES on  A = apply_saddle  with  P^-1 = apply_PC ,
                    rhs = [0 ; -F_body]  ->  U  ->  evolve_rigid_bodies(U)
    stochastic   :  RHS_and_Midpoint at q^n (2 M^{1/2}W + M_RFD), the same solve at q^{n+1/2}, update from q^n

Every loop runs inside librbl (rbl_gmres_saddle_dev, the Lanczos square roots, rbl_RHS_and_Midpoint_dev); the classes
below only hold the device vectors (torch tensors as plain buffers) and the O(N_bod) host bookkeeping between the
calls.  On N GPUs the same loops run with the context switched to multi-GPU products (DeviceContext.set_comm ->
rbl_comm_init_rccl: RCCL inside librbl; rbl_set_comm_ops callbacks in the gloo rehearsals).  The torch Arnoldi / Lanczos loops the tests compare these with live in tests/torch_krylov.py.
"""
import numpy as np
import torch


class DeterministicStepper:
    """One deterministic time step per call: rbl_gmres_saddle_dev (fixed work: `iters` iterations = iters + 1 apply_M;
    or converged to rtol), then evolve."""

    def __init__(self, ctx, n_bodies, blobs_per_body, device):
        self.ctx, self.nb, self.nblb, self.dev = ctx, n_bodies, blobs_per_body, device
        self.n3 = 3 * n_bodies * blobs_per_body
        self.size = self.n3 + 6 * n_bodies
        self.warm_start = False       # converged mode: start from the previous step's solution ...
        self.extrapolate = 0          # ... 1: from 2 x_n - x_{n-1}, 2: from 3 x_n - 3 x_{n-1} + x_{n-2} (the solution moves
        self._x_hist = []             #     smoothly with the configuration); history of the last solutions, newest first

    def initial_guess(self, rtol):
        h = self._x_hist
        warm = self.warm_start and rtol is not None and len(h) > 0 and h[0].numel() == self.size
        order = min(int(self.extrapolate), len(h) - 1) if warm else 0
        if not warm:
            return None
        return 3.0 * h[0] - 3.0 * h[1] + h[2] if order >= 2 else 2.0 * h[0] - h[1] if order == 1 else h[0].clone()

    def remember(self, x):
        if self.warm_start:
            self._x_hist = [x] + self._x_hist[:2]

    def solve(self, F_body, iters=20, rtol=None):
        """Solve the saddle system for rhs = [0 ; -F_body]; returns (lambda, U, iterations, residual)."""
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1)
        x0 = self.initial_guess(rtol)
        b = torch.zeros(self.size, dtype=torch.float64, device=self.dev)
        b[self.n3:] = -Fb
        x = x0 if x0 is not None else torch.empty_like(b)
        m, resid = self.ctx.gmres_saddle(b.data_ptr(), iters, rtol, x.data_ptr(), use_x0=x0 is not None)
        self.remember(x)
        return x[: self.n3], x[self.n3:], m, resid

    def step(self, F_body, iters=20, rtol=None):
        lam, U, m, resid = self.solve(F_body, iters, rtol)
        self.ctx.evolve(U.cpu().numpy())          # O(N_bod) host update, then K/positions rebuilt on the GPU
        self.ctx.sync_check()
        return m, resid


class ShardedDeterministicStepper(DeterministicStepper):
    """The same deterministic step on P GPUs (one process each): librbl's own GMRES with the context switched to
    multi-GPU products (DeviceContext.set_comm -> rbl_set_comm): every mobility product of the Arnoldi loop is this
    rank's share of the unordered tile pairs + ONE all-reduce of the partial U; the recurrences stay on the device, the
    host looks at the Hessenberg matrix once per convergence test.  Everything else -- K ops, preconditioner, Krylov
    vectors -- is O(N), replicated and bitwise identical on every rank."""

    def __init__(self, ctx, sharded, n_bodies, blobs_per_body, device, set_comm=True):
        super().__init__(ctx, n_bodies, blobs_per_body, device)
        self.sm = sharded
        if set_comm:                  # (False: the caller has given the context its communicator already)
            ctx.set_comm(sharded)


class BrownianStepper(DeterministicStepper):
    """One stochastic (midpoint) time step, assembled from the pieces the reference leaves unassembled
    (`RHS_and_Midpoint`, c_rigid_obj.cpp:917-976):

      1. at q^n      : rhs = [slip - (kBT M_RFD + BI) ; -F_body]  and the predictor configuration
                       q^{n+1/2} = q^n displaced by (dt/2) Kinv c1 M^{1/2} W1      (librbl: rbl_RHS_and_Midpoint_dev)
      2. at q^{n+1/2}: right-preconditioned GMRES on the saddle operator -> [lambda ; U]
      3.               q^{n+1} = q^n displaced by dt U                              (evolve_X_Q, :865-878)

    Steps 2-3 are this driver's completion of the scheme (the reference computes q^{n+1/2} but never
    uses it).  The context must have been created with dt > 0 and kBT; kBT <= 1e-10 reduces this to
    the deterministic step.  method: 0 = dense Cholesky (the reference's M_half_W), 1 = Lanczos, 2 = block-Jacobi
    preconditioned Lanczos."""

    def rhs_and_midpoint(self, F_body, slip, W, seed, method, split_rand, delta):
        Fb = torch.as_tensor(F_body, dtype=torch.float64, device=self.dev).reshape(-1).contiguous()
        sl = (torch.zeros(self.n3, dtype=torch.float64, device=self.dev) if slip is None else
              torch.as_tensor(slip, dtype=torch.float64, device=self.dev).reshape(-1).contiguous())
        Wd = None if W is None else torch.as_tensor(W, dtype=torch.float64, device=self.dev).reshape(-1).contiguous()
        if Wd is not None and Wd.numel() != 3 * self.n3:
            raise ValueError("W must hold 3 noise vectors [W1 | W2 | W_rfd] of length 3*N_blobs each")
        rhs = torch.empty(self.size, dtype=torch.float64, device=self.dev)
        Xh, Qh = self.ctx.RHS_and_Midpoint(sl.data_ptr(), Fb.data_ptr(), None if Wd is None else Wd.data_ptr(),
                                           seed, method, split_rand, delta, rhs.data_ptr(), self.nb)
        return rhs, Xh, Qh

    def saddle_solve(self, rhs, iters, rtol):
        x = torch.empty_like(rhs)
        m, resid = self.ctx.gmres_saddle(rhs.data_ptr(), iters, rtol, x.data_ptr())
        return x, m, resid

    def step(self, F_body, slip=None, W=None, seed=0, method=1, iters=20, rtol=None, split_rand=True,
             delta=1.0e-4):
        Xn, Qn = self.ctx.get_config(self.nb)
        rhs, Xh, Qh = self.rhs_and_midpoint(F_body, slip, W, seed, method, split_rand, delta)
        self.ctx.set_config(Xh, Qh)                      # operators and preconditioner at the predictor configuration
        x, m, resid = self.saddle_solve(rhs, iters, rtol)
        U = x[self.n3:].cpu().numpy()
        self.ctx.set_config(Xn, Qn)                      # the update starts from q^n
        self.ctx.evolve(U)
        self.ctx.sync_check()
        return m, resid


class ShardedBrownianStepper(BrownianStepper):
    """The stochastic midpoint step of BrownianStepper on P GPUs (BASELINE.json configs[3]): librbl's own stochastic
    midpoint pieces (rbl_RHS_and_Midpoint_dev: lock-step Lanczos, M_RFD; then rbl_gmres_saddle_dev) with every product
    sharded through the context's communicator (rbl_set_comm); all vectors and the O(N_bod) body state are replicated
    and bitwise identical on every rank (the noise comes from a seeded device generator)."""

    def __init__(self, ctx, sharded, n_bodies, blobs_per_body, device, a, wall, kBT, dt,
                 lanczos_tol=1e-3, lanczos_max_iter=100, precondition=True, set_comm=True):
        super().__init__(ctx, n_bodies, blobs_per_body, device)
        self.sm = sharded
        if set_comm:                  # (False: the caller has given the context its communicator already)
            ctx.set_comm(sharded)
        self.a, self.wall, self.kBT, self.dt = a, wall, kBT, dt
        self.ltol, self.lmax = lanczos_tol, lanczos_max_iter
        self.precondition = precondition      # block-Jacobi preconditioned square root (librbl's RBL_MHALF_LANCZOS_PC)
        self.lanczos_iterations = []

    def step(self, F_body, slip=None, W=None, seed=0, iters=20, rtol=None, split_rand=True, delta=1.0e-4):
        self.ctx.set_lanczos(self.lmax, self.ltol)
        out = BrownianStepper.step(self, F_body, slip=slip, W=W, seed=seed, method=2 if self.precondition else 1,
                                   iters=iters, rtol=rtol, split_rand=split_rand, delta=delta)
        self.lanczos_iterations = [self.ctx.lanczos_report()[0]] * (2 if split_rand else 1)
        return out
