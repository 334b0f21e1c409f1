"""A process group of ONE rank on the `nccl` backend (= RCCL) drives, on one GPU, exactly the code N ranks run:
  * ShardedMobility.all_gather_rows / all_reduce_sum on DEVICE buffers (no host staging: that is the gloo rehearsal),
  * DeviceContext.set_comm's callback -- librbl hands a raw hipMalloc pointer to Python, which wraps it as a torch tensor
    and all-reduces it on the context's stream -- inside rbl_gmres_saddle_dev (block preconditioner sharded by bodies),
    inside the preconditioned Lanczos square root and inside a whole stochastic midpoint step,
with a context bound to torch's current stream and with one bound to a side stream (the callback must then make that
stream current).  Every result must equal the un-sharded one.  Run as its own process:  python tools/check_nccl_world1.py"""
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
        print("%-58s %.3e (<= %g) %s" % (name, err, tol, "ok" if good else "FAILED"), flush=True)

    side = torch.cuda.Stream()
    for label, stream in (("current stream", torch.cuda.current_stream()), ("side stream", side)):
        def fresh(sharded):
            ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=kBT, stream_ptr=stream.cuda_stream)
            lib().rbl_set_blk_pc(ctx.h, 1)
            ctx.set_config(c["X"], c["Q"])
            sm = None
            if sharded:
                sm = ShardedMobility(nb, nblb, device=dev, ctx=ctx, force_collectives=True)
                assert sm.collectives and not sm.stage_cpu
                ctx.set_comm(sm)
            return ctx, sm

        # -- the two collectives of the sharded product on device buffers
        ctx, sm = fresh(True)
        x = torch.from_numpy(rng.standard_normal(n3)).to(dev)
        g = sm.all_gather_rows(x)
        r = sm.all_reduce_sum(x.clone())
        torch.cuda.synchronize()
        report("[%s] all_gather_rows (RCCL, world 1)" % label, float((g - x).abs().max()), 0.0)
        report("[%s] all_reduce_sum  (RCCL, world 1)" % label, float((r - x).abs().max()), 0.0)
        assert sm.n_all_gather == 1 and sm.n_all_reduce == 1
        # -- GMRES on the saddle operator: products + block preconditioner through the callback
        rhs = torch.from_numpy(rng.standard_normal(nsys)).to(dev)
        torch.cuda.synchronize()
        sols = []
        for sharded in (False, True):
            if sharded:
                cx, smx = ctx, sm
            else:
                cx, smx = fresh(False)
            xs = torch.empty_like(rhs)
            m, res = cx.gmres_saddle(rhs.data_ptr(), 80, 1e-10, xs.data_ptr())
            cx.sync_check()
            sols.append((xs.clone(), m, res))
            if not sharded:
                plain = cx
        calls_gmres = sm.n_all_reduce - 1
        report("[%s] rbl_gmres_saddle_dev sharded vs plain (%d its, %d all-reduces)" % (label, sols[1][1], calls_gmres),
               float(torch.linalg.norm(sols[1][0] - sols[0][0]) / torch.linalg.norm(sols[0][0])), 1e-9)
        assert calls_gmres >= 2 * sols[1][1] and sols[1][2] < 1e-10
        # -- preconditioned Lanczos square root
        rpos = torch.empty(n3, dtype=torch.float64, device=dev)
        ctx.blob_positions(0, nb, rpos.data_ptr())
        W = torch.from_numpy(rng.standard_normal(n3)).to(dev)
        torch.cuda.synchronize()
        outs = []
        for cx in (plain, ctx):
            cx.set_lanczos(100, 1e-10)
            o = torch.empty_like(W)
            cx.M_half_W(rpos.data_ptr(), N, W.data_ptr(), "lanczos_pc", o.data_ptr()); cx.sync_check()
            outs.append(o)
        report("[%s] M_half_W lanczos_pc sharded vs plain (%d all-reduces)" % (label, sm.n_all_reduce - 1 - calls_gmres),
               float(torch.linalg.norm(outs[1] - outs[0]) / torch.linalg.norm(outs[0])), 1e-9)
        # -- one whole stochastic midpoint step
        Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
        Wn = rng.standard_normal(3 * n3)
        with torch.cuda.stream(stream):
            plain.set_lanczos(100, 1e-10)
            BrownianStepper(plain, nb, nblb, dev).step(Fb, W=Wn, method=2, iters=100, rtol=1e-10)
            ShardedBrownianStepper(ctx, sm, nb, nblb, dev, c["a"], wall, kBT, c["dt"], lanczos_tol=1e-10).step(Fb, W=Wn, iters=100, rtol=1e-10)
        Xa, Qa = plain.get_config(nb); Xb, Qb = ctx.get_config(nb)
        report("[%s] stochastic midpoint step sharded vs plain (max |dX|)" % label, float(np.abs(Xa - Xb).max()), 1e-9)
        assert float(np.abs(Xa - c["X"]).max()) > 1e-5
        plain.close(); ctx.close()
    dist.destroy_process_group()
    print("nccl world-1: %s" % ("ALL OK" if ok else "FAILED"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
