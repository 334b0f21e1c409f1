"""Products around and between cfg 2 and cfg 3 (4 860 - 51 360 blobs): time and share of the fp64 issue slots (75 VALU instructions per
unordered wall pair, 39 in free space, x 4 cycles / (1024 SIMDs x 2.4 GHz)) of the one-vector product under the default launch
geometry and under forced ones (RBL_OPT_SYM_WAVES, RBL_OPT_SYM_ROWS_PER_LANE, RBL_OPT_SYM_WAVE_UNITS), the kernel each one launches
(rbl_apply_M_sym_kernel), the difference from the default's result, and the two-vector product under the same options checked
against two one-vector products.
    python tools/bench_midrange.py            SIZES=50x162xf,19x642xw python tools/bench_midrange.py   (bodies x blobs x f|w)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
variants = [("default", {}), ("wave units, 1 row/lane", {"sym_waves": 1, "sym_rows_per_lane": 1}),
            ("wave units, 2 rows/lane", {"sym_waves": 1, "sym_rows_per_lane": 2}),
            ("round-3 kernel, 1 wave", {"sym_waves": 1, "sym_rows_per_lane": 2, "sym_wave_units": 0}), ("4 waves", {"sym_waves": 4, "sym_rows_per_lane": 2}),
            ("pair: 2 rows/lane", {"sym2_rows_per_lane": 2}), ("pair: 1 row/lane", {"sym2_rows_per_lane": 1})]
sizes = [(50, 162, False), (80, 162, False)] + [(nb, 642, True) for nb in (13, 16, 19, 25, 37, 51)]
if os.environ.get("SIZES"):
    sizes = [tuple(int(x) for x in t.split("x")[:2]) + (t.split("x")[2] == "w",) for t in os.environ["SIZES"].split(",")]
for nb, nblb, wall in sizes:
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev)
    U = torch.empty_like(F); ref = None; ref2 = None
    F2 = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 3 * N))).to(dev).contiguous(); U2 = torch.empty_like(F2)
    out = []
    for _ in range(200 if N < 30000 else 40):            # clocks and caches settled before the first variant is timed
        ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
    ctx.sync_check()
    for name, opts in variants:
        for k in ("sym_waves", "sym_chunk", "sym_rows_per_lane", "sym2_rows_per_lane"):
            ctx.set_option(k, 0)
        ctx.set_option("sym_wave_units", 1)
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
        if ref2 is None:                                      # the pair against two one-vector products of the default kernel
            ref2 = torch.empty_like(F2)
            for v in range(2):
                ctx.apply_M(F2[v].data_ptr(), r.data_ptr(), N, 0, N, ref2[v].data_ptr())
            ctx.sync_check()
        err2 = float(torch.linalg.norm(U2 - ref2) / torch.linalg.norm(ref2))
        ni, ch, _ = ctx.apply_M_sym_info(N, 1, 1)
        issue = (75.0 if wall else 39.0) * 0.5 * N * N / 64.0 * 4.0 / (1024 * 2.4e9) / t
        out.append("%s [%s]: %.3f ms (issue %.2f, C %d, diff %.0e; two vectors [%s] %.3f, diff %.0e)" % (name, ctx.apply_M_sym_kernel(N, wall), t * 1e3, issue, ch, err, ctx.apply_M_sym_kernel(N, wall, nrhs=2), t2 * 1e3, err2))
    print("%6d blobs %s\n    " % (N, "wall" if wall else "free") + "\n    ".join(out), flush=True)
    ctx.close()
