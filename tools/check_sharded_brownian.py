"""Rehearsal of the multi-rank Brownian step on ONE GPU: launch with
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/check_sharded_brownian.py
(gloo process group, every rank on cuda:0).  Each rank advances the same small system by one stochastic
midpoint step with ShardedBrownianStepper (tile-pair-sharded products, all-reduce through the group) and
compares the new configuration with the single-process BrownianStepper (librbl's RHS_and_Midpoint)."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config                      # noqa: E402
from rigid_body_light_amd._lib import DeviceContext               # noqa: E402
from rigid_body_light_amd.dist import ShardedMobility             # noqa: E402
from rigid_body_light_amd.krylov import BrownianStepper, ShardedBrownianStepper   # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from torch_krylov import TorchBrownianStepper, TorchShardedBrownianStepper        # noqa: E402  (comparators)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    nb, nblb, wall, kBT = int(os.environ.get("RBL_CHECK_BODIES", "6")), int(os.environ.get("RBL_CHECK_BLOBS", "162")), True, 0.05
    block_pc = os.environ.get("RBL_CHECK_BLOCK_PC", "0") == "1"      # block-diagonal preconditioner in the saddle solve
    ltol, gtol = float(os.environ.get("RBL_CHECK_LANCZOS_TOL", "1e-11")), float(os.environ.get("RBL_CHECK_GMRES_TOL", "1e-10"))
    c = make_config(nb, nblb, wall)
    n3 = 3 * nb * nblb
    W = np.random.default_rng(11).standard_normal(3 * n3)
    Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
    native = os.environ.get("RBL_CHECK_NATIVE", "1") == "1"          # librbl's own loops with rbl_set_comm (default) or the torch loops
    out = []
    for sharded in (True, False):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=kBT,
                            stream_ptr=torch.cuda.current_stream().cuda_stream)
        ctx.set_config(c["X"], c["Q"])
        if not native:
            ctx.set_option("lanczos_two_level", 0)      # the torch comparator composes the root with the block-Jacobi factor; librbl's default is two-level
        if block_pc:
            from rigid_body_light_amd._lib import lib
            lib().rbl_set_blk_pc(ctx.h, 1)
        if sharded:
            cls = ShardedBrownianStepper if native else TorchShardedBrownianStepper
            st = cls(ctx, ShardedMobility(nb, nblb, device=dev, ctx=ctx), nb, nblb, dev, c["a"], wall, kBT,
                     c["dt"], lanczos_tol=ltol, lanczos_max_iter=255)
            if native:                # work split of the library's sharded products: 0 tile pairs + all-reduce, 1 rows + all-gather
                ctx.set_option("comm_split", int(os.environ.get("RBL_CHECK_SPLIT", "0")))
                if os.environ.get("RBL_CHECK_ALLREDUCE_ONLY", "0") == "1":      # the round-2/3 callback form: zero-padded sums instead of all-gathers
                    from rigid_body_light_amd._lib import lib as _lib
                    import ctypes as _C
                    _lib().rbl_set_comm(ctx.h, rank, world, _C.cast(ctx._comm_cb, _C.c_void_p), None)
            dump = os.environ.get("RBL_CHECK_DUMP", "")
            if dump and rank == 0:        # for the test's ORACLE check of the sharded solve: system, solution and the predictor configuration it lives at
                solve = st.saddle_solve

                def capture(rhs, iters, rtol, solve=solve, ctx=ctx):
                    x, m_, r_ = solve(rhs, iters, rtol)
                    Xh, Qh = ctx.get_config(nb)
                    np.savez(dump, rhs=rhs.cpu().numpy(), x=x.cpu().numpy(), X=Xh, Q=Qh)
                    return x, m_, r_
                st.saddle_solve = capture
            m, resid = st.step(Fb, W=W, iters=150, rtol=gtol)
        else:
            ctx.set_lanczos(255, ltol)
            m, resid = (BrownianStepper if block_pc else TorchBrownianStepper)(ctx, nb, nblb, dev).step(Fb, W=W, method=2, iters=150, rtol=gtol)   # preconditioned square root, as the sharded driver
        out.append(ctx.get_config(nb))
    dX = float(np.abs(out[0][0] - out[1][0]).max()); dQ = float(np.abs(out[0][1] - out[1][1]).max())
    moved = float(np.abs(out[0][0] - c["X"]).max())
    t = torch.tensor([dX, dQ]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("world %d: max |X_sharded - X_single| = %.3e, max |Q diff| = %.3e (bodies moved by %.3e)" % (world, t[0], t[1], moved))
    ok = t[0] < 1e-8 and t[1] < 1e-8 and moved > 1e-4 and resid < gtol
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
