#!/usr/bin/env python3
"""Timing of the dense path (build B M B, in-place Cholesky, L W) and Lanczos at a given size.
usage: bench_dense.py n_bodies blobs_per_body [wall] [--lanczos]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

nb, nblb = int(sys.argv[1]), int(sys.argv[2])
wall = len(sys.argv) > 3 and sys.argv[3] == "wall"
c = make_config(nb, nblb, wall)
N = nb * nblb; n = 3 * N
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream)
ctx.set_config(c["X"], c["Q"])
r = torch.empty(n, dtype=torch.float64, device=dev)
ctx.blob_positions(0, nb, r.data_ptr())
W = torch.from_numpy(np.random.default_rng(3).standard_normal(n)).to(dev)
out = torch.empty_like(W)


def timed(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync_check(); return (time.perf_counter() - t0) / reps


if "--lanczos" in sys.argv:
    ctx.set_lanczos(200, 1e-8)
    t = timed(lambda: ctx.M_half_W(r.data_ptr(), N, W.data_ptr(), "lanczos", out.data_ptr()))
    t = timed(lambda: ctx.M_half_W(r.data_ptr(), N, W.data_ptr(), "lanczos", out.data_ptr()))
    print("lanczos N=%d: %.2f ms, iters/resid %s" % (N, t * 1e3, ctx.lanczos_report()))
    sys.exit(0)
M = torch.empty(n * n, dtype=torch.float64, device=dev)
tb = timed(lambda: ctx.build_M(r.data_ptr(), N, True, M.data_ptr()))
tb = timed(lambda: ctx.build_M(r.data_ptr(), N, True, M.data_ptr()))
print("build  n=%d: %.2f ms  %.1f GB/s write" % (n, tb * 1e3, 8.0 * n * n / tb / 1e9))
tc = timed(lambda: ctx.cholesky(M.data_ptr(), n, False))
print("chol   n=%d: %.2f ms  %.2f TFLOP/s" % (n, tc * 1e3, n ** 3 / 3.0 / tc / 1e12))
tt = timed(lambda: ctx.trmv_lower(M.data_ptr(), n, W.data_ptr(), out.data_ptr()), 3)
print("trmv   n=%d: %.2f ms  %.1f GB/s read" % (n, tt * 1e3, 4.0 * n * n / tt / 1e9))
