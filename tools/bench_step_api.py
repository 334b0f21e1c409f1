"""Converged Brownian step through the Python stepper (krylov.BrownianStepper, native loops) vs the single C call
rbl_step_brownian.  usage: bench_step_api.py bodies blobs wall [tuning variant, e.g. 71]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import BrownianStepper
nb, nblb, wall = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
for api in ("python stepper", "rbl_step_brownian"):
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=1.0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    lib().rbl_set_blk_pc(ctx.h, 1)
    ctx.set_config(c["X"], c["Q"]); ctx.set_lanczos(200, 1e-3); ctx.set_block_refresh(2)
    if len(sys.argv) > 4: ctx.set_option(sys.argv[4].split("=")[0], int(sys.argv[4].split("=")[1]))
    st = BrownianStepper(ctx, nb, nblb, dev)
    one = (lambda k: st.step(Fb, seed=k, method=2, iters=200, rtol=1e-8)) if api == "python stepper" else \
          (lambda k: ctx.step_brownian(Fb, max_iter=200, rtol=1e-8, seed=k, method=2))
    one(0); torch.cuda.synchronize()
    t0 = time.perf_counter(); its = []
    for k in range(20):
        m, r = one(k + 1); its.append(m)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print("%-18s %.3f ms/step  gmres %s resid %.1e  min blob z %.3f" % (api, t * 1e3, its[:8], r, float(np.min(ctx.get_config(nb)[0].reshape(-1, 3)[:, 2]))), flush=True)
    del st; ctx.close()
