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
if "--build-only" in sys.argv:
    sys.exit(0)
verify = "--verify" in sys.argv
if verify:
    d0 = M[:: n + 1].clone()                       # diag(B M B) before the factorisation
tc = timed(lambda: ctx.cholesky(M.data_ptr(), n, verify))
print("chol   n=%d: %.2f ms  %.2f TFLOP/s" % (n, tc * 1e3, n ** 3 / 3.0 / tc / 1e12))
if verify:                                         # diag(L L^T) == diag(M): n independent row checks
    V = M.view(n, n)                               # V[a][b] = L[b][a]
    acc = torch.zeros(n, dtype=torch.float64, device=dev)
    for a0 in range(0, n, 2048):
        acc += (V[a0:a0 + 2048] ** 2).sum(0)
    err = float(((acc - d0).abs() / d0.abs()).max())
    print("verify n=%d: max_i |sum_j L_ij^2 - M_ii| / M_ii = %.3e" % (n, err))
    # and L W against a row-sampled reference  (L W)_i = sum_j L_ij W_j
    tt0 = torch.empty_like(W); ctx.trmv_lower(M.data_ptr(), n, W.data_ptr(), tt0.data_ptr()); ctx.sync_check()
    rows = torch.tensor([0, 1, n // 3, n // 2, n - 2, n - 1], device=dev)
    ref = (V[:, rows] * W[:, None]).sum(0)
    print("verify L*W on 6 rows: max rel err %.3e" % float(((tt0[rows] - ref).abs() / ref.abs()).max()))
tt = timed(lambda: ctx.trmv_lower(M.data_ptr(), n, W.data_ptr(), out.data_ptr()), 3)
print("trmv   n=%d: %.2f ms  %.1f GB/s read" % (n, tt * 1e3, 4.0 * n * n / tt / 1e9))
