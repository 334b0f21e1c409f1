"""Per-body factor application (block-diagonal preconditioner / block-Jacobi square root) at a given body size:
substitution kernels vs explicit inverses, and the cost of building the factors.  usage: bench_block_solve.py [bodies blobs]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 162)
dev = torch.device("cuda:0")
c = make_config(nb, nblb, False)
v = torch.randn(3 * nb * nblb, dtype=torch.float64, device=dev)
o = torch.empty_like(v)
for variant, name in ((0, "substitution"), (1, "explicit inverse")):
    ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    ctx.set_option("block_explicit_small", variant)
    ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.set_option("block_explicit_small", variant)   # invalidates the factors: the next call rebuilds them
        ctx.block_solve(v.data_ptr(), o.data_ptr(), 0)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 20
    line = "%-17s build+solve %.1f us;" % (name, tb * 1e6)
    for mode in (0, 1, 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ctx.block_solve(v.data_ptr(), o.data_ptr(), mode)
        e1.record(); torch.cuda.synchronize()
        line += " mode %d %.1f us" % (mode, e0.elapsed_time(e1) * 1e3 / 200)
    print(line, flush=True)
    ctx.close()
# per-configuration factors (what a wall-corrected system needs) for comparison
ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
ctx.set_config(c["X"], c["Q"]); ctx.set_option("bodyframe_factor", 0)
ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); torch.cuda.synchronize()
line = "per-configuration Cholesky (tuning 71):"
for mode in (0, 1, 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.block_solve(v.data_ptr(), o.data_ptr(), mode)
    e1.record(); torch.cuda.synchronize()
    line += " mode %d %.1f us" % (mode, e0.elapsed_time(e1) * 1e3 / 50)
print(line, flush=True)
