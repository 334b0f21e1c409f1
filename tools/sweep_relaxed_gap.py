"""Relaxed (packed single-precision far field) product against RBL_OPT_RELAXED_GAP_RATIO: product error in the wide suspension of
tests/test_gpu_parity.py::test_relaxed_product_in_a_wide_suspension and at cfg 3, and the cfg 3 kernel time.
    python tools/sweep_relaxed_gap.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

dev = torch.device("cuda:0")


def case(name, nb, nblb, wall, X=None):
    c = make_config(nb, nblb, wall)
    if X is not None:
        c["X"] = X(c)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    x = torch.from_numpy(np.random.default_rng(12).standard_normal(3 * N)).to(dev)
    ref = torch.empty_like(x); rlx = torch.empty_like(x)
    ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, ref.data_ptr())
    ctx.set_option("relaxed_always", 1)
    for ratio in (3, 4, 6, 8, 10, 15, 25):
        ctx.set_option("relaxed_gap_ratio", ratio)
        ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, rlx.data_ptr())
        ctx.sync_check()
        err = float(torch.linalg.norm(rlx - ref) / torch.linalg.norm(ref))
        rows = float(((rlx - ref).view(-1, 3).norm(dim=1) / ref.view(-1, 3).norm(dim=1).mean()).max())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, rlx.data_ptr())
        torch.cuda.synchronize()
        print("%-28s ratio %2d: error %.2e  worst row %.2e  %.3f ms" % (name, ratio, err, rows, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    ctx.close()


def wide(c):
    nb, a = 60, c["a"]
    h = 1.0 + a + 0.3
    X = np.zeros((nb, 3))
    for k in range(nb):
        X[k] = [1000.0 * (k // 2) + (2.0 * (1.0 + a) + 0.5) * (k % 2), 0.37 * (k % 2), h + 0.2 * (k % 3)]
    return X


case("wide suspension, wall", 60, 162, True, wide)
case("wide suspension, free", 60, 162, False, wide)
case("60 x 162 lattice, wall", 60, 162, True)
case("cfg 3 (200 x 642, wall)", 200, 642, True)
