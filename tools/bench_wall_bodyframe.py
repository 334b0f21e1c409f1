"""Wall-corrected system: exact per-configuration block factors vs the free-space body-frame factor used as an approximate
block factor (RBL_OPT_BODYFRAME_WALL_APPROX = 1): Brownian step time and iteration counts at cfg 3."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import BrownianStepper
dev = torch.device("cuda:0")
nb, nblb, wall = 200, 642, True
c = make_config(nb, nblb, wall)
Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
stream = torch.cuda.current_stream()
for variant in (73, 74):
    for relaxed in (False, True):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=1.0, stream_ptr=stream.cuda_stream)
        lib().rbl_set_blk_pc(ctx.h, 1)
        ctx.set_config(c["X"], c["Q"]); ctx.set_lanczos(200, 1e-3)
        ctx.set_block_refresh(2)
        ctx.set_option("bodyframe_wall_approx", variant - 73)
        if relaxed: ctx.set_option("relaxed_krylov", 1)
        st = BrownianStepper(ctx, nb, nblb, dev)
        st.step(Fb, seed=0, method=2, iters=200, rtol=1e-8)
        torch.cuda.synchronize(); t0 = time.perf_counter(); its = []; lz = []
        K = 6
        for k in range(K):
            m, r = st.step(Fb, seed=k + 1, method=2, iters=200, rtol=1e-8); its.append(m); lz.append(ctx.lanczos_report()[0])
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K
        print("variant %d relaxed %d: %.1f ms/step gmres %s lanczos %s resid %.2e" % (variant, relaxed, t * 1e3, its, lz, r), flush=True)
        del st, ctx
