"""Rehearsal of the lock-step multi-RHS GMRES (rbl_gmres_saddle_multi_dev) on a multi-rank context, several ranks on ONE GPU:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/check_sharded_multi_rhs.py
(gloo process group, every rank on cuda:0).  On a sharded context every product of every column is this rank's share + one
all-reduce and every preconditioner application the owners' bodies + one all-gather; the columns must equal the single-process
solves of the same right-hand sides."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config                      # noqa: E402
from rigid_body_light_amd._lib import DeviceContext, lib          # noqa: E402
from rigid_body_light_amd.dist import ShardedMobility             # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    nb, nblb, wall = int(os.environ.get("RBL_CHECK_BODIES", "7")), int(os.environ.get("RBL_CHECK_BLOBS", "162")), True
    c = make_config(nb, nblb, wall)
    n3 = 3 * nb * nblb; nsys = n3 + 6 * nb
    k = 5
    rhs = np.random.default_rng(17).standard_normal((k, nsys)); rhs[:, :n3] *= 0.1
    rhs_d = torch.from_numpy(rhs).to(dev)
    sols = []
    for sharded in (True, False):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
        lib().rbl_set_blk_pc(ctx.h, 1)
        ctx.set_option("block_explicit_large", 1)
        ctx.set_config(c["X"], c["Q"])
        x = torch.zeros_like(rhs_d)
        if sharded:
            ctx.set_comm(ShardedMobility(nb, nblb, device=dev, ctx=ctx))
            ctx.set_option("comm_split", int(os.environ.get("RBL_CHECK_SPLIT", "0")))
            its, res = ctx.gmres_saddle_multi(rhs_d.data_ptr(), k, 100, 1e-10, x.data_ptr())
        else:
            its, res = [], []
            for j in range(k):
                m, r = ctx.gmres_saddle(rhs_d[j].data_ptr(), 100, 1e-10, x[j].data_ptr()); its.append(m); res.append(r)
        ctx.sync_check()
        sols.append((x.clone(), its, res))
        ctx.close()
    err = max(float(torch.linalg.norm(sols[0][0][j] - sols[1][0][j]) / torch.linalg.norm(sols[1][0][j])) for j in range(k))
    t = torch.tensor([err]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = float(t[0]) < 1e-9 and max(sols[0][2]) < 1e-10 and all(abs(a - b) <= 1 for a, b in zip(sols[0][1], sols[1][1]))
    if rank == 0:
        print("world %d: max column difference sharded lock-step vs single-process solves = %.3e, iterations %s vs %s" % (world, float(t[0]), sols[0][1], sols[1][1]))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
