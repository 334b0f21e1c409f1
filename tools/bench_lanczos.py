#!/usr/bin/env python3
"""Lanczos M^{1/2} W: iterations / time vs tolerance (matrix-free, any size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
print("| config | N | tol | iterations | last relative change | ms |"); print("|---|---|---|---|---|---|")
for name, nb, nblb, wall in (("cfg2 50x162 free", 50, 162, False), ("cfg3/4 200x642 wall", 200, 642, True)):
    c = make_config(nb, nblb, wall); N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
    W = torch.from_numpy(np.random.default_rng(3).standard_normal(3 * N)).to(dev); out = torch.empty_like(W)
    ref = None
    for method in ("lanczos", "lanczos_pc"):
        for tol in (1e-2, 1e-3, 1e-4, 1e-6):
            ctx.set_lanczos(300, tol)
            ctx.M_half_W(r.data_ptr(), N, W.data_ptr(), method, out.data_ptr()); ctx.sync_check()
            t0 = time.perf_counter(); ctx.M_half_W(r.data_ptr(), N, W.data_ptr(), method, out.data_ptr()); ctx.sync_check()
            t = time.perf_counter() - t0
            it, res = ctx.lanczos_report()
            print("| %s, %s | %d | %g | %d | %.2e | %.1f |" % (name, method, N, tol, it, res, t * 1e3), flush=True)
    ctx.close()
