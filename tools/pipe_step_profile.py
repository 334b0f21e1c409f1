#!/usr/bin/env python3
"""Step by step through the pipelined substitution (k_block_solve_pipe, csrc/rbl_dense.hip): shader-clock stamps of one streaming
thread after every step's barrier, for every body.  Runs against a DIAGNOSTIC library (rbl_dense.hip compiled with -DRBL_PIPE_PROF,
linked with the normal build's other objects into build/librbl_pipeprof.so; the normal build has no stamps).
usage: pipe_step_profile.py [bodies blobs] | --build-only"""
import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "rigid_body_light_amd", "build", "librbl_pipeprof.so")


def build_prof_lib():
    from rigid_body_light_amd import build as b
    b.build()
    obj = os.path.join(b.OBJ, "rbl_dense_prof.o")
    subprocess.check_call([b.HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + b.ARCH, "-x", "hip", "-DRBL_PIPE_PROF",
                           "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(b.CSRC, "rbl_dense.hip"), "-o", obj])
    others = [os.path.join(b.OBJ, s.rsplit(".", 1)[0] + ".o") for s in b.HIP_SOURCES if s != "rbl_dense.hip"]
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
c = make_config(nb, nblb, True)
m = 3 * nblb
dev = torch.device("cuda:0")
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
ctx.set_config(c["X"], c["Q"])
ctx.set_option("bodyframe_factor", 0); ctx.set_option("block_explicit_large", 0)
v = torch.randn(m * nb, dtype=torch.float64, device=dev); o = torch.empty_like(v)
L = lib(); buf = (ctypes.c_ulonglong * (256 * 512))()
ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); ctx.sync_check()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L.rbl_debug_pipe_prof(buf, 1)
e0.record(); ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); e1.record(); torch.cuda.synchronize(); ctx.sync_check()
ms = e0.elapsed_time(e1)
L.rbl_debug_pipe_prof(buf, 0)
x = np.array(list(buf), dtype=np.float64).reshape(256, 512)[:min(nb, 256)]
nsteps = (m + 31) // 32
nst = int((x[0] > 0).sum())
d = np.diff(x[:, :nst], axis=1)                     # per body, per step (shader clocks at 100 MHz?  printed as a share of the whole)
tot = (x[:, nst - 1] - x[:, 0])
print("%d x shell_N_%d: (L L^T)^-1 v %.3f ms; %d stamps a body; stamped span: median %.0f, min %.0f, max %.0f clocks"
      % (nb, nblb, ms, nst, np.median(tot), tot.min(), tot.max()))
scale = ms * 1e3 / np.median(tot)                   # us per clock, if the stamped span were the whole launch
md = np.median(d, axis=0)
print("step: streamed KB, median clocks, ~us, ~GB/s a body   (forward steps first, then the backward ones)")
k = 0
for sweep in ("forward", "backward"):
    cnt = nsteps - 1 if sweep == "forward" else nst - 1 - (nsteps - 1) - 1
    for j in range(cnt):
        if sweep == "forward":
            kb = max(0, m - 32 * j - 64) * 32 * 8 / 1e3
        else:
            s = cnt - j
            kb = max(0, 32 * (s - 1)) * 32 * 8 / 1e3
        idx = k + j
        if j % 4 == 0 or j > cnt - 4:
            us = md[idx] * scale
            print("  %-8s %3d  %7.1f KB  %8.0f  %6.2f us  %6.1f" % (sweep, j, kb, md[idx], us, kb / us if us > 0 else 0))
    k += cnt + (1 if sweep == "forward" else 0)
ctx.close()
# the diag wave's own stamps (forward sweep): after the barrier | head sums done | next diagonal solve done | next loads issued
y = np.array(list(buf), dtype=np.float64).reshape(256, 512)[128:128 + min(nb, 128)]
ns1 = nsteps - 1
if (y[0, :4 * ns1] > 0).all():
    z = y[:, :4 * ns1].reshape(-1, ns1, 4)
    wait = np.zeros(z.shape[:2]); wait[:, :-1] = z[:, 1:, 0] - z[:, :-1, 3]
    seg = np.stack([z[:, :, 1] - z[:, :, 0], z[:, :, 2] - z[:, :, 1], z[:, :, 3] - z[:, :, 2], wait], axis=2)
    med = np.median(seg, axis=0) * scale
    print("diag wave, forward step: head sums (incl. the wait for its loads) | diagonal solve | issue of the next loads | barrier, us")
    for j in list(range(0, ns1 - 6, 8)) + list(range(ns1 - 6, ns1)):
        print("  step %3d   %5.2f  %5.2f  %5.2f  %5.2f" % (j, med[j, 0], med[j, 1], med[j, 2], med[j, 3]))
