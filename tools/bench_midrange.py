"""Products between cfg 2 and cfg 3 (12 000 - 51 000 blobs, wall): time, share of the fp64 issue slots (75 VALU instructions per
unordered pair x 4 cycles / (1024 SIMDs x 2.4 GHz)), for the default geometry and for forced ones (RBL_OPT_SYM_WAVES, RBL_OPT_SYM_CHUNK).
    python tools/bench_midrange.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
variants = [("default", {}), ("symw", {"sym_waves": 1, "sym_rows_per_lane": 1}), ("sw=1", {"sym_waves": 1}), ("sw=4", {"sym_waves": 4}),
            ("C=2", {"sym_chunk": 2})]
for nb in (13, 16, 19, 25, 37, 51, 80):
    nblb, wall = 642, True
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev)
    U = torch.empty_like(F); ref = None
    F2 = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 3 * N))).to(dev).contiguous(); U2 = torch.empty_like(F2)
    out = []
    for _ in range(200 if N < 30000 else 40):            # clocks and caches settled before the first variant is timed
        ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
    ctx.sync_check()
    for name, opts in variants:
        for k in ("sym_waves", "sym_chunk", "sym_rows_per_lane"):
            ctx.set_option(k, 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        reps = 50 if N < 60000 else 10
        for _ in range(5):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check()
        t = (time.perf_counter() - t0) / reps
        if ref is None:
            ref = U.clone()
        err = float(torch.linalg.norm(U - ref) / torch.linalg.norm(ref))
        # the two-vector product (the Lanczos pair's) under the same options
        for _ in range(3):
            ctx.apply_M_multi(F2.data_ptr(), r.data_ptr(), N, 2, U2.data_ptr())
        ctx.sync_check()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply_M_multi(F2.data_ptr(), r.data_ptr(), N, 2, U2.data_ptr())
        ctx.sync_check()
        t2 = (time.perf_counter() - t0) / reps
        ni, ch, _ = ctx.apply_M_sym_info(N, 1, 1)
        issue = 75.0 * 0.5 * N * N / 64.0 * 4.0 / (1024 * 2.4e9) / t
        out.append("%s: %.3f ms (issue %.2f, NI %d C %d, diff %.0e; two vectors %.3f)" % (name, t * 1e3, issue, ni, ch, err, t2 * 1e3))
    print("%6d blobs  " % N + "  |  ".join(out), flush=True)
    ctx.close()
