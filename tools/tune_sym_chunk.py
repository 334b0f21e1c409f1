"""apply_M time against the chunk length C of the symmetric kernel (RBL_OPT_SYM_CHUNK) at cfg 2 (free / wall), cfg 5 and cfg 3:
    python tools/tune_sym_chunk.py   (the measurements behind sym_geometry's chunk rule)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
for name, nb, nblb, wall in (("cfg2", 50, 162, False), ("cfg2w", 50, 162, True), ("cfg5", 20, 2562, False), ("cfg3", 200, 642, True)):
    c = make_config(nb, nblb, wall); N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
    F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev); U = torch.empty_like(F)
    for C in (0, 1, 2, 4, 8, 16, 32, 64):
        ctx.set_option("matvec_kernel", 2 if C else 0); ctx.set_option("sym_chunk", C)
        reps = 100 if N < 60000 else 5
        for _ in range(3): ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check(); t0 = time.perf_counter()
        for _ in range(reps): ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())
        ctx.sync_check(); print(name, "C=%d" % C, "%.4f ms" % ((time.perf_counter() - t0) / reps * 1e3), flush=True)
    ctx.close()
