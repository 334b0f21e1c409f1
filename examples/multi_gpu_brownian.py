#!/usr/bin/env python3
"""The same Brownian dynamics on N GPUs, one process per GPU:

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/multi_gpu_brownian.py

Every rank holds the same (replicated) body state and calls the same library entry points with the same arguments;
`DeviceContext.set_comm` makes librbl's own Lanczos / GMRES loops multi-GPU: with an `nccl` process group it creates an
RCCL communicator INSIDE librbl (C ABI: rbl_comm_unique_id on rank 0, broadcast through the group, rbl_comm_init_rccl) --
each mobility product is this rank's share of the unordered blob-tile pairs followed by one ncclAllReduce on the context's
stream, per-body factors and substitutions are done for the rank's own bodies only and completed by an all-gather; no Python
runs inside a solve.  With one visible GPU and several ranks it falls back to the `gloo` backend (callbacks, host-staged
collectives) -- a rehearsal, not a speed-up.  The same thing without Python: examples/host_rccl_step.cpp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.dist import ShardedMobility
from rigid_body_light_amd.krylov import ShardedBrownianStepper

world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
ndev = torch.cuda.device_count()
dev = torch.device("cuda", local % max(ndev, 1))
torch.cuda.set_device(dev)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl" if ndev >= world else "gloo")
rank = dist.get_rank() if world > 1 else 0

nb, nblb, kBT = 50, 162, 0.004
c = make_config(nb, nblb, wall=True)
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], kBT=kBT,
                    stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)                       # block-diagonal preconditioner, sharded by bodies
ctx.set_config(c["X"], c["Q"])
sm = ShardedMobility(nb, nblb, device=dev, ctx=ctx)
stepper = ShardedBrownianStepper(ctx, sm, nb, nblb, dev, c["a"], True, kBT, c["dt"], lanczos_tol=1e-4)   # calls ctx.set_comm(sm)
F = np.tile([0.0, 0.0, 0.2, 0.0, 0.0, 0.0], nb)
X0 = ctx.get_config(nb)[0].copy()
for n in range(10):
    iters, resid = stepper.step(F, seed=n, iters=60, rtol=1e-6)      # the same seed on every rank: the same noise
    X, _ = ctx.get_config(nb)
    if rank == 0:
        print("step %2d on %d rank(s): %2d GMRES iterations (%.0e), Lanczos %s, mean height %.4f, MSD %.3e"
              % (n, world, iters, resid, stepper.lanczos_iterations, X[:, 2].mean(), ((X - X0) ** 2).sum(1).mean()))
if world > 1:
    dist.destroy_process_group()
