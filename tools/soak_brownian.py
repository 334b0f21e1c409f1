"""Soak test: 3 200 stochastic midpoint steps of 50 x shell_N_162 above a wall through rbl_step_brownian; free device memory
and host RSS before / after the last 3 000 (no growth: 293 704 MiB and 1 401 MiB on both sides, 6.4 ms per step, round 3; 293 706 / 1 180 MiB, 6.0 ms per step at the end
of round 4).
usage: soak_brownian.py [bodies blobs steps]   (e.g. 12 642 600: bodies on the tile factorisation and the pipelined substitution, round 5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psutil
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
nb, nblb, nsteps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 162, 3000)
warm = max(nsteps // 15, 2)
c = make_config(nb, nblb, True)
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], dt=c["dt"], kBT=0.004, stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1); ctx.set_lanczos(100, 1e-3); ctx.set_config(c["X"], c["Q"])
F = np.zeros(6 * nb)
p = psutil.Process()
def snap(): return torch.cuda.mem_get_info()[0] / 2**20, p.memory_info().rss / 2**20
for n in range(warm): ctx.step_brownian(F, 60, 1e-6, seed=n, method=2)
f0, r0 = snap(); t0 = time.time()
for n in range(warm, warm + nsteps): ctx.step_brownian(F, 60, 1e-6, seed=n, method=2)
f1, r1 = snap()
X, Q = ctx.get_config(nb)
print("%d x shell_N_%d: %d steps in %.1f s; free device memory %.0f -> %.0f MiB; host RSS %.0f -> %.0f MiB; finite %s; |Q|-1 max %.1e" % (nb, nblb, nsteps, time.time() - t0, f0, f1, r0, r1, np.all(np.isfinite(X)), abs(np.linalg.norm(Q, axis=1) - 1).max()))
