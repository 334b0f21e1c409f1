#!/usr/bin/env python3
"""One configuration, 16 right-hand sides through the fp64-MFMA matvec (for rocprof)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb = int(sys.argv[1]), int(sys.argv[2]); wall = sys.argv[3] == "wall"; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = make_config(nb, nblb, wall); N = nb * nblb
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
F = torch.from_numpy(np.random.default_rng(5).standard_normal((16, 3 * N))).to(dev); U = torch.empty_like(F)
ctx.apply_M_multi(F.data_ptr(), r.data_ptr(), N, 16, U.data_ptr()); ctx.sync_check()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.apply_M_multi(F.data_ptr(), r.data_ptr(), N, 16, U.data_ptr())
ctx.sync_check(); t = (time.perf_counter() - t0) / reps
print("N=%d wall=%s 16 RHS: %.3f ms, MFMA %.2f TFLOP/s (16 x 18 N^2)" % (N, wall, t * 1e3, 16 * 18.0 * N * N / t / 1e12))
