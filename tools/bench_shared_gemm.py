"""Free space: the ONE body-frame inverse / preconditioner table applied to all bodies -- matrix-matrix product on the fp64 matrix
cores (RBL_OPT_SHARED_GEMM = 1) against the batched matrix-vector form (= 0), per application, on one box.
    python tools/bench_shared_gemm.py [n_bodies blobs_per_body]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib

nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 162)
c = make_config(nb, nblb, False)
dev = torch.device("cuda:0")
n3 = 3 * nb * nblb; nsys = n3 + 6 * nb
v = torch.randn(n3, dtype=torch.float64, device=dev); o = torch.empty_like(v)
x = torch.randn(nsys, dtype=torch.float64, device=dev); y = torch.empty_like(x)
ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)
ctx.set_config(c["X"], c["Q"])


def t_us(fn, reps=300):
    for _ in range(20):
        fn()
    ctx.sync_check(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync_check()
    return (time.perf_counter() - t0) / reps * 1e6


print("%d x shell_N_%d, free space (n = %d per body): microseconds per application, batched matrix-vector / MFMA product, twice" % (nb, nblb, 3 * nblb))
for name, fn in (("(G G^T)^-1 v (two sweeps)", lambda: ctx.block_solve(v.data_ptr(), o.data_ptr(), 0)),
                 ("G^-1 v", lambda: ctx.block_solve(v.data_ptr(), o.data_ptr(), 1)),
                 ("G^-T v", lambda: ctx.block_solve(v.data_ptr(), o.data_ptr(), 2)),
                 ("apply_PC (block, body frame)", lambda: ctx.apply_PC(x.data_ptr(), y.data_ptr()))):
    r = []
    for g in (0, 1, 0, 1):
        ctx.set_option("shared_gemm", g)
        r.append(t_us(fn))
    print("%-30s %6.1f / %6.1f   %6.1f / %6.1f" % (name, r[0], r[1], r[2], r[3]), flush=True)
ctx.close()
