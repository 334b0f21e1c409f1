"""Substitution through the per-body factors of large bodies: the one-barrier pipeline (RBL_OPT_BLOCK_SOLVE_PIPE = 1, k_block_solve_pipe)
against the two-barrier kernel of rounds 1-4 (0), per mode, for all bodies and for a rank's share at P = 8; bytes = the factor's
lower triangle once per sweep.  usage: [ONLY_MODE=0|1|2] bench_block_pipe.py [bodies blobs [wall|free]]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
wall = not (len(sys.argv) > 3 and sys.argv[3] == "free")
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
m = 3 * nblb
torch.manual_seed(1)
v = torch.randn(m * nb, dtype=torch.float64, device=dev)
print("%d x shell_N_%d, %s: n = %d" % (nb, nblb, "wall" if wall else "free", m))
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
ctx.set_config(c["X"], c["Q"])
ctx.set_option("bodyframe_factor", 0)
ctx.set_option("block_explicit_large", 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for b0, b1, what in ((0, nb, "all %d bodies" % nb), (0, max(nb // 8, 1), "%d bodies" % max(nb // 8, 1))):
    for mode, name, sweeps in ((0, "(L L^T)^-1 v", 2), (1, "L^-1 v", 1), (2, "L^-T v", 1)):
        if os.environ.get("ONLY_MODE", str(mode)) != str(mode):      # (counter passes: one mode a run, the kernels share a name)
            continue
        line = "%-12s %-14s" % (what, name)
        for pipe in (0, 1):
            ctx.set_option("block_solve_pipe", pipe)
            o = torch.zeros_like(v)
            ctx.block_solve(v.data_ptr(), o.data_ptr(), mode, b0, b1); ctx.sync_check()
            reps = 20
            e0.record()
            for _ in range(reps):
                ctx.block_solve(v.data_ptr(), o.data_ptr(), mode, b0, b1)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            gb = (b1 - b0) * sweeps * m * (m + 1) / 2 * 8 / 1e9
            res[(b1 - b0, mode, pipe)] = o[: (b1 - b0) * m].clone()
            line += "  %s %.3f ms = %.2f TB/s" % ("pipeline" if pipe else "two-barrier", t, gb / t)
        a, b = res[(b1 - b0, mode, 1)], res[(b1 - b0, mode, 0)]
        print(line + "  max rel. diff %.1e" % float((a - b).abs().max() / b.abs().max()), flush=True)
ctx.close()
