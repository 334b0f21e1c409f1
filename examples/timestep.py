#!/usr/bin/env python3
"""Deterministic time stepping with every operator resident on the GPU: 50 shells of 162 blobs
sedimenting towards a wall, GMRES on the saddle operator with the block-diagonal preconditioner."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import DeterministicStepper

nb, nblb = 50, 162
c = make_config(nb, nblb, wall=True)
dev = torch.device("cuda:0")
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)
ctx.set_config(c["X"], c["Q"])
stepper = DeterministicStepper(ctx, nb, nblb, dev)   # librbl's own GMRES (rbl_gmres_saddle_dev)
stepper.warm_start = True                                         # start each solve from the previous steps' solutions:
stepper.extrapolate = 2                                           # 3 x_n - 3 x_{n-1} + x_{n-2} once three exist
# sign convention of the reference's saddle system (rhs = [slip ; -F], K^T lambda = -F): with this F the
# bodies settle towards the wall
F = np.tile([0.0, 0.0, 1.0, 0.0, 0.0, 0.0], nb)
for n in range(8):
    iters, resid = stepper.step(F, iters=50, rtol=1e-8)
    X, _ = ctx.get_config(nb)
    print("step %d: %2d GMRES iterations, residual %.1e, mean height %.5f" % (n, iters, resid, X[:, 2].mean()))
