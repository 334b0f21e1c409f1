"""k right-hand sides of one configuration: the lock-step GMRES on the fp64-MFMA product (rbl_gmres_saddle_multi_dev) against k
sequential rbl_gmres_saddle_dev solves, block PC, rtol 1e-8; every column compared with its sequential solve.
usage: bench_multi_rhs.py [bodies blobs [wall|free] [k]]"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
wall = not (len(sys.argv) > 3 and sys.argv[3] == "free")
k = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
N = nb * nblb; n3 = 3 * N; nsys = n3 + 6 * nb
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)
ctx.set_config(c["X"], c["Q"])
rng = np.random.default_rng(3)
rhs = np.zeros((k, nsys))
rhs[:, n3:] = rng.standard_normal((k, 6 * nb))          # k different body loads (the body mobility matrix's columns are such)
rhs_d = torch.from_numpy(rhs).to(dev)
xs = torch.empty_like(rhs_d); xm = torch.empty_like(rhs_d)
ctx.gmres_saddle(rhs_d[0].data_ptr(), 200, 1e-8, xs[0].data_ptr()); ctx.sync_check()       # builds the preconditioner
torch.cuda.synchronize(); t0 = time.perf_counter()
its_s = []
nseq = 1 if os.environ.get("ONLY_MULTI") else k           # ONLY_MULTI=1 (counter passes): no sequential solves beyond the one that compares
for j in range(nseq):
    m, r = ctx.gmres_saddle(rhs_d[j].data_ptr(), 200, 1e-8, xs[j].data_ptr()); its_s.append(m)
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) * k / nseq
ctx.gmres_saddle_multi(rhs_d.data_ptr(), k, 200, 1e-8, xm.data_ptr())                      # (workspace growth outside the timing)
torch.cuda.synchronize(); t0 = time.perf_counter()
its_m, res_m = ctx.gmres_saddle_multi(rhs_d.data_ptr(), k, 200, 1e-8, xm.data_ptr())
torch.cuda.synchronize(); tm = time.perf_counter() - t0
ctx.set_timing(True); ctx.reset_timings()
ctx.gmres_saddle_multi(rhs_d.data_ptr(), k, 200, 1e-8, xm.data_ptr())
tmg = ctx.timings(); ctx.set_timing(False)
err = max(float(torch.linalg.norm(xm[j] - xs[j]) / torch.linalg.norm(xs[j])) for j in range(nseq))
print("%d x shell_N_%d %s, %d right-hand sides, block PC, rtol 1e-8" % (nb, nblb, "wall" if wall else "free", k))
print("  sequential: %.1f ms (%.1f ms a solve), iterations %s" % (ts * 1e3, ts * 1e3 / k, its_s))
print("  lock step : %.1f ms (%.1f ms a solve), iterations %s, max residual %.2e" % (tm * 1e3, tm * 1e3 / k, its_m, max(res_m)))
print("  ratio %.3f; largest column difference to its sequential solve %.2e" % (tm / ts, err))
print("  phases of one lock-step solve (ms): " + ", ".join("%s %.1f" % (kk, v[0]) for kk, v in tmg.items()))
