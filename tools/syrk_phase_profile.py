#!/usr/bin/env python3
"""Phase breakdown of the trailing-update kernel k_syrk_mfma from in-kernel s_memtime stamps.

Needs a DIAGNOSTIC library (the normal build has no stamps):
    hipcc ... -DRBL_SYRK_PROF -c rigid_body_light_amd/csrc/rbl_dense.hip   (or RBL_EXTRA_FLAGS=-DRBL_SYRK_PROF build.py
    into a scratch copy), linked as a second librbl, and selected with RBL_LIBRARY=/path/to/librbl_prof.so.
usage: RBL_LIBRARY=... syrk_phase_profile.py n_bodies blobs_per_body"""
import os, sys, ctypes, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
nb, nblb = int(sys.argv[1]), int(sys.argv[2])
c = make_config(nb, nblb, False)
N = nb * nblb; n = 3 * N
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], stream_ptr=st.cuda_stream)
ctx.set_config(c["X"], c["Q"])
r = torch.empty(n, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
M = torch.empty(n * n, dtype=torch.float64, device=dev)
L = lib()
buf = (ctypes.c_ulonglong * 16)()
for rep in range(2):
    ctx.build_M(r.data_ptr(), N, True, M.data_ptr()); ctx.sync_check()
    L.rbl_debug_syrk_prof(buf, 1)
    t0 = time.perf_counter(); ctx.cholesky(M.data_ptr(), n, False); ctx.sync_check(); dt = time.perf_counter() - t0
    L.rbl_debug_syrk_prof(buf, 1)
v = np.array(list(buf), dtype=np.float64)
print("chol n=%d %.1f ms %.2f TF" % (n, dt * 1e3, n ** 3 / 3 / dt / 1e12))
names = ["lwrite(+vmcnt)", "gload issue", "compute", "barrier", "prologue", "peeled tail", "epilogue"]
tiles = v[8]
print("interior wave-0 tiles: %d ; avg shader cycles per tile %.0f ; in-kernel clock %.3f GHz"
      % (tiles, v[7] / tiles, v[7] / max(v[9], 1.0) * 0.1))
print("MFMA pipe share of a tile: 2 waves x 32 stages x 64 MFMA x 64 cycles = 262144 -> %.1f %% busy" % (100 * 262144.0 / (v[7] / tiles)))
for i, nm in enumerate(names):
    print("  %-16s %6.2f %%   %.0f cycles/tile" % (nm, 100 * v[i] / v[7], v[i] / tiles))
