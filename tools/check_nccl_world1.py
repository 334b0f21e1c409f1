"""A communicator of ONE rank drives, on one GPU, exactly the code N ranks run:
  * RCCL INSIDE librbl (rbl_comm_init_rccl from a unique id; ncclAllReduce / ncclAllGather on the context's stream; the staged
    form of the all-gather that ragged shares take, forced by RBL_OPT_COMM_FORCE_STAGED) under
    rbl_gmres_saddle_dev (block preconditioner sharded by bodies), the preconditioned Lanczos square root and a whole
    stochastic midpoint step -- with both work splits (RBL_OPT_COMM_SPLIT 0: unordered tile pairs + all-reduce, 1: rows by
    body index + all-gather of positions and U), on torch's current stream and on a side stream;
  * the callback form (rbl_set_comm_ops: torch.distributed `nccl` all_reduce / broadcast on DEVICE buffers, the raw
    hipMalloc pointer wrapped as a torch tensor) that the gloo rehearsals use with host staging;
  * the collectives themselves (rbl_comm_allreduce_dev / rbl_comm_allgatherv_dev) on device buffers.
Every result must equal the un-sharded one to rounding (the sharded path composes K^T lambda in another kernel; the row
split uses the ordered-pair kernel).  Run as its own process:  python tools/check_nccl_world1.py"""
import ctypes as C
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config                      # noqa: E402
from rigid_body_light_amd._lib import DeviceContext, lib          # noqa: E402
from rigid_body_light_amd.dist import ShardedMobility             # noqa: E402
from rigid_body_light_amd.krylov import BrownianStepper, ShardedBrownianStepper   # noqa: E402


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    nb, nblb, wall, kBT = 6, 162, True, 0.05
    c = make_config(nb, nblb, wall)
    N = nb * nblb; n3 = 3 * N; nsys = n3 + 6 * nb
    rng = np.random.default_rng(5)
    ok = True

    def report(name, err, tol):
        nonlocal ok
        good = bool(err <= tol)
        ok = ok and good
        print("%-86s %.3e (<= %g) %s" % (name, err, tol, "ok" if good else "FAILED"), flush=True)

    side = torch.cuda.Stream()
    # staged = 1 (RBL_OPT_COMM_FORCE_STAGED): every in-place all-gather of the native communicator takes the padded staging
    # buffer, pack / ncclAllGather / unpack -- the code a job with N_bod % world != 0 runs, exercised here with one rank
    variants = [("current stream", torch.cuda.current_stream(), True, 0, 0), ("side stream", side, True, 0, 0),
                ("side stream", side, True, 1, 0), ("current stream", torch.cuda.current_stream(), True, 0, 1),
                ("side stream", side, True, 1, 1), ("current stream", torch.cuda.current_stream(), False, 0, 0),
                ("side stream", side, False, 1, 0)]
    for label, stream, native, split, staged in variants:
        label = "%s, %s%s, split %d" % (label, "RCCL in librbl" if native else "callbacks", ", staged all-gathers" if staged else "", split)

        def fresh(sharded):
            ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=kBT, stream_ptr=stream.cuda_stream)
            lib().rbl_set_blk_pc(ctx.h, 1)
            ctx.set_option("block_explicit_large", 1)     # (the same form of the per-body factors on both sides: a multi-GPU context inverts them by default)
            ctx.set_config(c["X"], c["Q"])
            sm = None
            if sharded:
                sm = ShardedMobility(nb, nblb, device=dev, ctx=ctx, force_collectives=True)
                assert sm.collectives and not sm.stage_cpu
                ctx.set_comm(sm, native=native)
                ctx.set_option("comm_split", split)
                ctx.set_option("comm_force_staged", staged)
                assert ctx.comm_info() == (0, 1, 2 if native else 1)
                ctx.set_timing(True)
            return ctx, sm

        ctx, sm = fresh(True)
        # -- the context's collectives on a device buffer
        x = torch.from_numpy(rng.standard_normal(n3)).to(dev)
        y = x.clone()
        ctx._chk(ctx.L.rbl_comm_allreduce_dev(ctx.h, y.data_ptr(), n3))
        offs, cnts = (C.c_int64 * 1)(5), (C.c_int64 * 1)(n3 - 9)
        ctx._chk(ctx.L.rbl_comm_allgatherv_dev(ctx.h, y.data_ptr(), offs, cnts))
        ctx.sync_check()
        report("[%s] all-reduce + all-gather of a device buffer (world 1)" % label, float((y - x).abs().max()), 0.0)
        # -- GMRES on the saddle operator: products + block preconditioner through the communicator
        rhs = torch.from_numpy(rng.standard_normal(nsys)).to(dev)
        torch.cuda.synchronize()
        sols = []
        for sharded in (False, True):
            if sharded:
                cx = ctx
            else:
                cx, _ = fresh(False)
            xs = torch.empty_like(rhs)
            if sharded:
                cx.reset_timings()
            m, res = cx.gmres_saddle(rhs.data_ptr(), 80, 1e-10, xs.data_ptr())
            cx.sync_check()
            sols.append((xs.clone(), m, res))
            if not sharded:
                plain = cx
        ncoll = ctx.timings()["collective"][1]
        err = float(torch.linalg.norm(sols[1][0] - sols[0][0]) / torch.linalg.norm(sols[0][0]))
        report("[%s] rbl_gmres_saddle_dev sharded vs plain (%d its, %d collectives)" % (label, sols[1][1], ncoll), err,
               1e-10)
        assert ncoll >= 2 * sols[1][1] and sols[1][2] < 1e-10 and abs(sols[1][1] - sols[0][1]) <= 1
        # -- preconditioned Lanczos square root
        rpos = torch.empty(n3, dtype=torch.float64, device=dev)
        ctx.blob_positions(0, nb, rpos.data_ptr())
        W = torch.from_numpy(rng.standard_normal(n3)).to(dev)
        torch.cuda.synchronize()
        outs = []
        ctx.reset_timings()
        for cx in (plain, ctx):
            cx.set_lanczos(100, 1e-10)
            o = torch.empty_like(W)
            cx.M_half_W(rpos.data_ptr(), N, W.data_ptr(), "lanczos_pc", o.data_ptr()); cx.sync_check()
            outs.append(o)
        report("[%s] M_half_W lanczos_pc sharded vs plain (%d collectives)" % (label, ctx.timings()["collective"][1]),
               float(torch.linalg.norm(outs[1] - outs[0]) / torch.linalg.norm(outs[0])), 1e-10)
        # -- one whole stochastic midpoint step
        Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
        Wn = rng.standard_normal(3 * n3)
        with torch.cuda.stream(stream):
            plain.set_lanczos(100, 1e-10)
            BrownianStepper(plain, nb, nblb, dev).step(Fb, W=Wn, method=2, iters=100, rtol=1e-10)
            st = ShardedBrownianStepper(ctx, sm, nb, nblb, dev, c["a"], wall, kBT, c["dt"], lanczos_tol=1e-10, set_comm=False)
            st.step(Fb, W=Wn, iters=100, rtol=1e-10)
        Xa, Qa = plain.get_config(nb); Xb, Qb = ctx.get_config(nb)
        report("[%s] stochastic midpoint step sharded vs plain (max |dX|)" % label, float(np.abs(Xa - Xb).max()),
               1e-10)
        assert float(np.abs(Xa - c["X"]).max()) > 1e-5
        plain.close(); ctx.close()
    dist.destroy_process_group()
    print("nccl world-1: %s" % ("ALL OK" if ok else "FAILED"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
