"""Mid-size products (one row per lane): the wave-unit kernel (RBL_OPT_SYM_WAVE_UNITS = 1, default) against the round-3 kernel
(= 0) on the same box, with the difference of the two results.  RBL_LIBRARY selects an A/B build (tools/build_variant.sh).
    python tools/bench_midsize.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
print("library:", os.environ.get("RBL_LIBRARY", "default"), " chunk:", os.environ.get("CHUNK", "heuristic"), " queue:", os.environ.get("QUEUE", "default"))
for name, nb, nblb, wall in (("cfg2 50x162 free", 50, 162, False), ("50x162 wall", 50, 162, True), ("37x162 free", 37, 162, False),
                             ("25x162 free", 25, 162, False), ("12x642 wall", 12, 642, True), ("cfg1 10x12 free", 10, 12, False)):
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    if os.environ.get("QUEUE"):
        ctx.set_option("sym_work_queue", int(os.environ["QUEUE"]))
    if os.environ.get("CHUNK"):
        ctx.set_option("sym_chunk", int(os.environ["CHUNK"]))
    if os.environ.get("ROWS"):                               # rows per lane of the one-vector kernels (default: the library's rule)
        ctx.set_option("sym_rows_per_lane", int(os.environ["ROWS"]))
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev)
    res = {}
    for mode in (0, 1, 0, 1):
        ctx.set_option("sym_wave_units", mode)
        U = torch.empty_like(F)
        for _ in range(20):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t0 = time.perf_counter()
        for _ in range(300):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t = (time.perf_counter() - t0) / 300
        res.setdefault(mode, []).append((t, U))
    d = float(torch.linalg.norm(res[1][0][1] - res[0][0][1]) / torch.linalg.norm(res[0][0][1]))
    print("%-18s N = %6d   round-3 kernel %.1f / %.1f us   wave units %.1f / %.1f us   |difference| %.1e" % (
        name, N, res[0][0][0] * 1e6, res[0][1][0] * 1e6, res[1][0][0] * 1e6, res[1][1][0] * 1e6, d), flush=True)
    ctx.close()
