"""A/B of launch-level switches on the cfg 2 Brownian step (50 x shell_N_162, free space; bench.py's `configs.cfg2` entry):
    python tools/bench_cfg2_step.py [options_a options_b] [steps]    options: comma-separated name=value (include/rbl.h RBL_OPT_*)
    default: "fused_krylov=0,sym_wave_units=0" against "fused_krylov=1,sym_wave_units=1" (the round-3 kernels against round 4's)
Both settings run on the same context, interleaved in blocks of `steps` steps, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import BrownianStepper

ta = sys.argv[1] if len(sys.argv) > 2 else "fused_krylov=0,sym_wave_units=0"
tb = sys.argv[2] if len(sys.argv) > 2 else "fused_krylov=1,sym_wave_units=1"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10


def apply(spec):
    for kv in spec.split(","):
        k, _, v = kv.partition("=")
        ctx.set_option(k.strip(), int(v))
nb, nblb = 50, 162
c = make_config(nb, nblb, False)
dev = torch.device("cuda:0")
ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], dt=c["dt"], kBT=1.0, stream_ptr=torch.cuda.current_stream().cuda_stream)
lib().rbl_set_blk_pc(ctx.h, 1)
ctx.set_lanczos(200, 1e-3)
ctx.set_config(c["X"], c["Q"])
Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
bst = BrownianStepper(ctx, nb, nblb, dev)
seed = 0
for _ in range(3):
    seed += 1; bst.step(Fb, seed=seed, method=2, iters=200, rtol=1e-8)
for rnd in range(3):
    for t in (ta, tb):
        apply(t)
        X, Q = ctx.get_config(nb)
        seed0 = 100 * rnd                                   # the same noise and the same start for both settings
        ctx.set_config(c["X"], c["Q"])
        its = []
        bst.step(Fb, seed=seed0, method=2, iters=200, rtol=1e-8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(steps):
            m, r = bst.step(Fb, seed=seed0 + 1 + k, method=2, iters=200, rtol=1e-8); its.append(m)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print("round %d [%s]: %.3f ms per step, GMRES iterations %s" % (rnd, t, dt * 1e3, its), flush=True)
    # the same steps through the one-call C entry point (rbl_step_brownian): no Python between the pieces of a step
    ctx.set_config(c["X"], c["Q"])
    its = []
    ctx.step_brownian(Fb, 200, 1e-8, seed=100 * rnd, method=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        m, r = ctx.step_brownian(Fb, 200, 1e-8, seed=100 * rnd + 1 + k, method=2); its.append(m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("round %d one-call entry [%s]: %.3f ms per step, GMRES iterations %s" % (rnd, tb, dt * 1e3, its), flush=True)
