#!/usr/bin/env python3
"""Where a workgroup of the dataflow tile factorisation (k_tile_chol, csrc/rbl_tilechol.hip) spends its time: shader-clock stamps
around the phases of every task, summed over all workgroups.  Runs against a DIAGNOSTIC library (rbl_tilechol.hip compiled with
-DRBL_TILE_PROF, linked with the normal build's other objects into build/librbl_tileprof.so; the normal build has no stamps).
usage: tile_phase_profile.py [bodies blobs [inverse 0|1]] | --build-only"""
import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "rigid_body_light_amd", "build", "librbl_tileprof.so")


def build_prof_lib():
    from rigid_body_light_amd import build as b
    b.build()
    obj = os.path.join(b.OBJ, "rbl_tilechol_prof.o")
    subprocess.check_call([b.HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + b.ARCH, "-x", "hip", "-DRBL_TILE_PROF",
                           "-c", os.path.join(b.CSRC, "rbl_tilechol.hip"), "-o", obj])
    others = [os.path.join(b.OBJ, s.rsplit(".", 1)[0] + ".o") for s in b.HIP_SOURCES if s != "rbl_tilechol.hip"]
    subprocess.check_call([b.HIPCC, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", PROF_LIB, obj] + others)


if "--build-only" in sys.argv:
    build_prof_lib(); print(PROF_LIB); sys.exit(0)
if "RBL_LIBRARY" not in os.environ:
    if not os.path.exists(PROF_LIB):
        build_prof_lib()
    os.environ["RBL_LIBRARY"] = PROF_LIB
import numpy as np, torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
inv = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = make_config(nb, nblb, True)
m = 3 * nblb
dev = torch.device("cuda:0")
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
ctx.set_config(c["X"], c["Q"])
ctx.set_option("bodyframe_factor", 0); ctx.set_option("block_explicit_large", inv)
v = torch.randn(m * nb, dtype=torch.float64, device=dev); o = torch.empty_like(v)
L = lib(); buf = (ctypes.c_ulonglong * 16)()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    ctx.set_option("block_tile_factor", 1)
    L.rbl_debug_tile_prof(buf, 1)
    e0.record(); ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); e1.record(); torch.cuda.synchronize(); ctx.sync_check()
    L.rbl_debug_tile_prof(buf, 1)
x = np.array(list(buf), dtype=np.float64)
names = {0: "claim a task", 1: "product of a factor tile (waits incl.)", 2: "C update + barrier", 3: "diagonal tile (potrf)", 4: "wait for a diagonal tile",
         5: "triangular solve", 6: "publish (drain, barrier, release, add)", 8: "128 x 128 inverse of a diagonal tile", 7: "product + store of an inverse tile", 9: "end-of-task barrier", 10: "reading the counters", 11: "the acquire of a task (INV: wait for its row of L)"}
tot = x.sum()
NT = (m + 127) // 128
print("%d x shell_N_%d, inverse %d: build + one application %.2f ms; %d tasks; summed workgroup time %.1f ms at 2.4 GHz over 512 resident workgroups = %.2f ms each"
      % (nb, nblb, inv, e0.elapsed_time(e1), nb * NT * (NT + 1) // (1 if inv else 2), tot / 2.4e6, tot / 2.4e6 / 512))
for i, nm in names.items():
    print("  %-42s %6.2f %%" % (nm, 100 * x[i] / tot))

# timeline of body 0 (microseconds after its first claim): the factor front diag(s), and the inverse tiles of rows 0 and s - 1 in column s
tl = (ctypes.c_ulonglong * (2 * 32 * 32 * 2))()
if hasattr(L, "rbl_debug_tile_line") and L.rbl_debug_tile_line(tl) == 0 and NT <= 32:
    T = np.array(list(tl), dtype=np.float64).reshape(2, 32, 32, 2) / 100.0
    t0 = T[0, 0, 0, 0]
    print("body 0, us after its first claim: diagonal tile (s, s) claimed / published | chain tile (s + 1, s) published | INV(0, s) claimed / published | INV(s, s) published")
    for s_ in range(NT):
        ch = T[0, s_ + 1, s_, 1] - t0 if s_ + 1 < NT else float("nan")
        iv = (T[1, 0, s_, 0] - t0, T[1, 0, s_, 1] - t0, T[1, s_, s_, 1] - t0) if inv else (float("nan"),) * 3
        print("  s = %2d   %8.1f %8.1f | %8.1f | %8.1f %8.1f | %8.1f" % (s_, T[0, s_, s_, 0] - t0, T[0, s_, s_, 1] - t0, ch, iv[0], iv[1], iv[2]))
