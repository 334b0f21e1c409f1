#!/usr/bin/env python3
"""Phase breakdown of the trailing-update kernel k_syrk_mfma from in-kernel s_memtime stamps.

Runs against a DIAGNOSTIC library (the normal build has no stamps): rbl_dense.hip compiled with -DRBL_SYRK_PROF and
linked with the other objects of the normal build into rigid_body_light_amd/build/librbl_prof.so.  `--build-only`
makes that library (hipcc cross-compiles, no GPU needed); a run builds it when missing and loads it via RBL_LIBRARY.
usage: syrk_phase_profile.py n_bodies blobs_per_body | --build-only"""
import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "rigid_body_light_amd", "build", "librbl_prof.so")


def build_prof_lib():
    from rigid_body_light_amd import build as b
    b.build()
    obj = os.path.join(b.OBJ, "rbl_dense_prof.o")
    subprocess.check_call([b.HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + b.ARCH, "-x", "hip", "-DRBL_SYRK_PROF",
                           "-c", os.path.join(b.CSRC, "rbl_dense.hip"), "-o", obj])
    others = [os.path.join(b.OBJ, s.rsplit(".", 1)[0] + ".o") for s in b.HIP_SOURCES if s != "rbl_dense.hip"]
    subprocess.check_call([b.HIPCC, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", PROF_LIB, obj] + others)


if "--build-only" in sys.argv:
    build_prof_lib(); print(PROF_LIB); sys.exit(0)
if "RBL_LIBRARY" not in os.environ:
    if not os.path.exists(PROF_LIB):
        build_prof_lib()
    os.environ["RBL_LIBRARY"] = PROF_LIB
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
