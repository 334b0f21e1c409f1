"""Per-body factor (and explicit-inverse) BUILD for large bodies: the dataflow tile kernel (RBL_OPT_BLOCK_TILE_FACTOR = 1,
rbl_tilechol.hip) against the batched panel kernels of rounds 1-4 (0), factor only and factor + inverse, for all bodies and for a
rank's share at P = 8; the two paths' factor applications are compared entry by entry.
usage: bench_block_factor.py [bodies blobs [wall|free]]"""
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
flops = nb * m ** 3 / 3.0
print("%d x shell_N_%d, %s: n = %d, factor flops %.3e (inverse: the same again)" % (nb, nblb, "wall" if wall else "free", m, flops))
ref = {}
for tile in (0, 1):
    for inv in (0, 1):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
        ctx.set_config(c["X"], c["Q"])
        ctx.set_option("bodyframe_factor", 0)
        ctx.set_option("block_tile_factor", tile)
        ctx.set_option("block_explicit_large", inv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for b0, b1, what in ((0, nb, "all %d bodies" % nb), (0, max(nb // 8, 1), "%d bodies (a rank's share at P = 8)" % max(nb // 8, 1))):
            o = torch.zeros_like(v)
            ctx.block_solve(v.data_ptr(), o.data_ptr(), 0, b0, b1); ctx.sync_check()
            reps = 5
            e0.record()
            for _ in range(reps):
                ctx.block_solve(v.data_ptr(), o.data_ptr(), 0, b0, b1)
            e1.record(); torch.cuda.synchronize()
            t_apply = e0.elapsed_time(e1) / reps
            e0.record()
            for _ in range(reps):
                ctx.set_option("block_tile_factor", tile)        # invalidates the factors: the next call rebuilds them
                ctx.block_solve(v.data_ptr(), o.data_ptr(), 0, b0, b1)
            e1.record(); torch.cuda.synchronize()
            ctx.sync_check()
            t_build = e0.elapsed_time(e1) / reps - t_apply
            fl = (b1 - b0) * m ** 3 / 3.0 * (2 if inv else 1)
            outs = []
            for mode in (0, 1, 2, 3):
                oo = torch.zeros_like(v)
                ctx.block_solve(v.data_ptr(), oo.data_ptr(), mode, b0, b1)
                outs.append(oo[: (b1 - b0) * m].clone())
            ctx.sync_check()
            key = (inv, b1 - b0)
            diff = ""
            if tile == 0:
                ref[key] = outs
            else:
                diff = "  max rel. diff to the panel kernels, modes 0-3: " + " ".join(
                    "%.1e" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs, ref[key]))
            print("tile=%d inverse=%d  %-36s build (incl. the %.2f ms mobility assembly) %7.2f ms = %5.1f TFLOP/s; application %.3f ms%s"
                  % (tile, inv, what, 0.0, t_build, fl / t_build / 1e9, t_apply, diff), flush=True)
        ctx.close()
