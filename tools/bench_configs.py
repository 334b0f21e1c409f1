#!/usr/bin/env python3
"""apply_M timings (device-resident) for every BASELINE config and both kernel variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
print("| config | N | variant | ms / apply_M | M.F GFLOP/s (18 N^2) | pair-flop TFLOP/s |")
print("|---|---|---|---|---|---|")
for name, nb, nblb, wall in (("cfg1 10x12 free", 10, 12, False), ("cfg1 10x12 wall", 10, 12, True),
                             ("cfg2 50x162 free", 50, 162, False), ("50x162 wall", 50, 162, True),
                             ("cfg3 200x642 wall", 200, 642, True), ("200x642 free", 200, 642, False),
                             ("cfg5 20x2562 free", 20, 2562, False)):
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev)
    U = torch.empty_like(F)
    for vname, v in (("ordered", 1), ("symmetric", 2)):
        ctx.set_option("matvec_kernel", v)
        reps = 200 if N < 10000 else 5
        for _ in range(3):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t = (time.perf_counter() - t0) / reps
        fl = (204.0 if wall else 59.0) * N * N
        print("| %s | %d | %s | %.4f | %.1f | %.2f |" % (name, N, vname, t * 1e3, 18.0 * N * N / t / 1e9, fl / t / 1e12), flush=True)
    if N >= 8000:   # 16 right-hand sides on the fp64 matrix cores
        F16 = torch.from_numpy(np.random.default_rng(5).standard_normal((16, 3 * N))).to(dev)
        U16 = torch.empty_like(F16)
        ctx.set_option("matvec_kernel", 0)
        for _ in range(2):
            ctx.apply_M_multi(F16.data_ptr(), r.data_ptr(), N, 16, U16.data_ptr())
        ctx.sync_check()
        reps = 20 if N < 10000 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply_M_multi(F16.data_ptr(), r.data_ptr(), N, 16, U16.data_ptr())
        ctx.sync_check()
        t = (time.perf_counter() - t0) / reps
        print("| %s | %d | 16-RHS MFMA | %.4f (= %.4f per vector) | %.1f | MFMA %.2f TFLOP/s |" % (
            name, N, t * 1e3, t * 1e3 / 16, 16 * 18.0 * N * N / t / 1e9, 16 * 18.0 * N * N / t / 1e12), flush=True)
    ctx.close()
