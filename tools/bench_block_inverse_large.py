"""Per-body factor application for LARGE bodies (shell_N_642 / 2562): substitution chains (RBL_OPT_BLOCK_EXPLICIT_LARGE = 0) vs explicit
inverses (1) vs their single-precision copy (RBL_OPT_BLOCK_INVERSE_F32 = 1), for all bodies and for one rank's share at P = 8, with the achieved
HBM rate (bytes = the triangle(s) of the factor / inverse one application reads) and the cost of the build.
usage: bench_block_inverse_large.py [bodies blobs [wall|free] [option=value ...]]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
wall = (len(sys.argv) > 3 and sys.argv[3] == "wall") or len(sys.argv) <= 2
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
m = 3 * nblb
v = torch.randn(m * nb, dtype=torch.float64, device=dev)
o = torch.empty_like(v)
print("%d x shell_N_%d, %s: n = %d, factor %.2f GB (fp64, lower triangles %.2f GB)" % (nb, nblb, "wall" if wall else "free", m, 8e-9 * m * m * nb, 4e-9 * m * m * nb))
for opts, name, bpe in (({"block_explicit_large": 0}, "substitution", 8.0), ({"block_explicit_large": 1}, "explicit inverse fp64", 8.0),
                        ({"block_explicit_large": 1, "block_inverse_f32": 1}, "explicit inverse, fp32 copy", 4.0)):
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    ctx.set_option("bodyframe_factor", 0)
    for kv in sys.argv[4:]:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for k_, v_ in opts.items():
        ctx.set_option(k_, v_)
    ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); ctx.sync_check()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ctx.set_option("block_explicit_large", opts["block_explicit_large"])   # invalidates the factors: the next call rebuilds them
        ctx.block_solve(v.data_ptr(), o.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    print("%-36s build + one application %.2f ms" % (name, e0.elapsed_time(e1) / 3), flush=True)
    for b0, b1, what in ((0, nb, "all %d bodies" % nb), (0, max(nb // 8, 1), "a rank's share at P = 8 (%d bodies)" % max(nb // 8, 1))):
        line = "    %-34s" % what
        for mode in (0, 1, 2, 3):
            reps = 20
            ctx.block_solve(v.data_ptr(), o.data_ptr(), mode, b0, b1)
            e0.record()
            for _ in range(reps):
                ctx.block_solve(v.data_ptr(), o.data_ptr(), mode, b0, b1)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tri = 0.5 * m * m * (b1 - b0) * (2 if mode == 0 else 1) * (8.0 if mode == 3 else bpe)
            line += "  mode %d %.3f ms (%.2f TB/s)" % (mode, ms, tri / ms / 1e9)
        print(line, flush=True)
    ctx.close()
