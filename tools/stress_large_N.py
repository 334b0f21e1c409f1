#!/usr/bin/env python3
"""Large-N robustness check of apply_M: sizes beyond the BASELINE configs, including the one where the symmetric
kernel's slab workspace exceeds its budget and the ordered kernel takes over; a few rows against the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
from oracle import Oracle

orc = Oracle()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
for nb, nblb, wall in ((400, 642, True), (800, 642, True), (100, 2562, False)):
    c = make_config(nb, nblb, wall); N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
    x = torch.from_numpy(np.random.default_rng(9).standard_normal(3 * N)).to(dev); out = torch.empty_like(x)
    ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, out.data_ptr()); ctx.sync_check()
    t0 = time.perf_counter(); ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, out.data_ptr()); ctx.sync_check()
    t = time.perf_counter() - t0
    rh, xh, oh = r.cpu().numpy(), x.cpu().numpy(), out.cpu().numpy()
    worst = 0.0
    for b in (0, N // 2 + 7, N - 5):
        Uo = orc.apply_M_rows(xh, rh, b, b + 4, c["a"], c["eta"], wall, nthreads=16)
        worst = max(worst, float(np.linalg.norm(oh[3 * b:3 * b + 12] - Uo) / np.linalg.norm(Uo)))
    print("N=%7d wall=%-5s apply_M %9.2f ms  (%.2f ps per ordered pair)  max rel err vs oracle on 12 rows %.2e"
          % (N, wall, t * 1e3, t / (float(N) ** 2) * 1e12, worst), flush=True)
    assert worst < 1e-11
    ctx.close(); del r, x, out
    torch.cuda.empty_cache()
