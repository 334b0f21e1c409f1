#!/usr/bin/env python3
"""Brownian dynamics of 50 shells of 162 blobs above a wall: the stochastic midpoint step
(M^{1/2} W by block-Jacobi preconditioned Lanczos, random finite-difference drift, saddle solve at the predictor configuration),
every O(N^2) operation on the GPU.  Prints the mean height and the mean-square displacement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import BrownianStepper

nb, nblb, kBT = 50, 162, 0.004
c = make_config(nb, nblb, wall=True)
dev = torch.device("cuda:0")
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], kBT=kBT,
                    stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)            # block-diagonal preconditioner
ctx.set_lanczos(100, 1e-4)                # tolerance of the Lanczos square root
ctx.set_config(c["X"], c["Q"])
stepper = BrownianStepper(ctx, nb, nblb, dev)
F = np.tile([0.0, 0.0, 0.2, 0.0, 0.0, 0.0], nb)       # weak pull towards the wall (reference sign convention)
X0 = ctx.get_config(nb)[0].copy()
for n in range(10):
    iters, resid = stepper.step(F, seed=n, method=2, iters=60, rtol=1e-6)     # method 2 = RBL_MHALF_LANCZOS_PC
    X, _ = ctx.get_config(nb)
    print("step %2d: %2d GMRES iterations (%.0e), Lanczos %d, mean height %.4f, MSD %.3e"
          % (n, iters, resid, ctx.lanczos_report()[0], X[:, 2].mean(), ((X - X0) ** 2).sum(1).mean()))
