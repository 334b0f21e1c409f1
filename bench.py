#!/usr/bin/env python3
"""bench.py -- headline benchmark of the blob-mobility hot path on MI355X.

Workload (BASELINE.json configs[2]): 200 bodies x shell_N_642 = 128 400 blobs,
wall-corrected RPY mobility with wall damping, fp64, synthetic configuration of
SURVEY.md section 8(d).  One "step" = one pass of the hot path:

    blob positions from (X, Q) on the device  ->  [all-gather positions, forces]  ->
    matrix-free  U = B M B F  for this rank's rows

N = 1:  everything on one GPU.   N > 1: bodies sharded contiguously over ranks (one
process per GPU, torch.distributed / RCCL all-gather), total work fixed -> "strong".

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline`
(dominant kernel k_apply_M timed with events on its own stream) and `cpu_baseline`
(the CPU oracle -- a port of the reference algorithm -- on a bounded row sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (bodies, blobs/body, wall)
    "cfg1": (10, 12, False),
    "cfg2": (50, 162, False),
    "cfg3": (200, 642, True),
    "cfg5": (20, 2562, False),
}
FLOPS_PER_PAIR = {False: 59.0, True: 204.0}   # reference arithmetic per ordered pair
PEAK_FP64_TFLOPS = 78.6                       # MI355X fp64 vector == fp64 matrix peak


def cpu_baseline(c, nb, nblb, wall, budget_s):
    """Time the CPU oracle (port of reference :413-459,:641-659, matrix-free restatement)
    on a bounded sample of rows of the SAME workload; scale to one full apply_M."""
    from oracle import Oracle
    orc = Oracle()
    cfg = c["cfg"] - c["cfg"].mean(axis=0)
    r = orc.multi_body_pos(c["X"], c["Q"], cfg)
    N = nb * nblb
    F = np.random.default_rng(2).standard_normal(3 * N)
    out = {}
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, int(os.environ.get("RBL_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores
    for label, nthreads in (("1core", 1), ("allcores", ncores)):
        rows = 4 * nthreads
        t0 = time.perf_counter(); orc.apply_M_rows(F, r, 0, rows, c["a"], c["eta"], wall, nthreads); t1 = time.perf_counter()
        per_row = (t1 - t0) / rows
        rows = int(max(rows, min(N, budget_s / max(per_row, 1e-9))))
        b = (N // 2 // max(nblb, 1)) * nblb
        b = min(b, N - rows)
        t0 = time.perf_counter(); orc.apply_M_rows(F, r, b, b + rows, c["a"], c["eta"], wall, nthreads); t1 = time.perf_counter()
        t_full = (t1 - t0) * N / rows
        out[label] = {"value": 1.0 / t_full, "unit": "steps/s", "cores": nthreads, "kind": "port",
                      "sample": "%d of %d rows x all %d columns of the same workload, %.1f s measured, "
                                "scaled by N/rows; oracle/rbl_oracle.c orc_apply_M_rows (gcc -O3 -march=x86-64-v3, "
                                "reference pair arithmetic, matrix-free because the reference's dense 3Nx3N matrix "
                                "would need %.2f TB)" % (rows, N, N, t1 - t0, 8.0 * (3 * N) ** 2 / 1e12),
                      "seconds_per_step": t_full}
    return out


def timestep_mode(args, dev, world=1, rank=0):
    """1 step = one deterministic time step, all operators on the GPU(s); with N > 1 the mobility
    product of every GMRES iteration is tile-pair sharded (one all-reduce per iteration)."""
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    from rigid_body_light_amd.dist import ShardedMobility
    from rigid_body_light_amd.krylov import DeterministicStepper, ShardedDeterministicStepper
    nb, nblb, wall = CONFIGS[args.config]
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    brownian = args.kBT > 1e-10
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=args.kBT,
                        stream_ptr=torch.cuda.current_stream().cuda_stream)
    if args.pc == "block":
        from rigid_body_light_amd._lib import lib
        lib().rbl_set_blk_pc(ctx.h, 1)
    ctx.set_config(c["X"], c["Q"])
    iters = 20 if args.rtol <= 0 else 200
    rtol = args.rtol if args.rtol > 0 else None
    Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
    lanczos_its = None
    if brownian:   # stochastic midpoint step: 2 M^{1/2}W + M_RFD + Kinv, then the saddle solve at q^{n+1/2}
        from rigid_body_light_amd.krylov import BrownianStepper, ShardedBrownianStepper
        method = {"cholesky": 0, "lanczos": 1, "lanczos_pc": 2}[args.mhalf]
        if world > 1 or args.sharded_driver:
            if method == 0:
                raise SystemExit("the sharded Brownian step uses the Lanczos square root")
            bst = ShardedBrownianStepper(ctx, ShardedMobility(nb, nblb, device=dev, ctx=ctx), nb, nblb, dev, c["a"], wall,
                                         args.kBT, c["dt"], lanczos_tol=1e-3)
            stp_step = lambda k: bst.step(Fb, seed=k, iters=iters, rtol=rtol)
            lanczos_its = lambda: list(bst.lanczos_iterations)
        else:
            ctx.set_lanczos(100, 1e-3)
            bst = BrownianStepper(ctx, nb, nblb, dev, native=(not args.graph and (args.native or args.solver == "native")))
            stp_step = lambda k: bst.step(Fb, seed=k, method=method, iters=iters, rtol=rtol)
            lanczos_its = (lambda: [ctx.lanczos_report()[0]]) if method != 0 else None
    else:
        if world > 1:
            stp = ShardedDeterministicStepper(ctx, ShardedMobility(nb, nblb, device=dev, ctx=ctx), nb, nblb, dev)
        else:
            solver = "graph" if args.graph else ("native" if args.native else args.solver)
        stp = DeterministicStepper(ctx, nb, nblb, dev, use_graph=(solver == "graph"), native=(solver == "native"))
        stp.warm_start = args.warm_start or args.extrapolate > 0
        if args.block_refresh > 1:
            ctx.set_block_refresh(args.block_refresh)
        stp.extrapolate = args.extrapolate
        stp_step = lambda k: stp.step(Fb, iters, rtol)
    res, used = [], []
    for k in range(args.warmup):
        stp_step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        m_used, r_last = stp_step(args.warmup + k)
        res.append(r_last); used.append(m_used)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    sec_t = torch.tensor([(t1 - t0) / args.steps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(sec_t, op=dist.ReduceOp.MAX)
    sec = float(sec_t.item())
    iters = int(round(sum(used) / len(used)))
    if world > 1 and rank != 0:
        dist.destroy_process_group()
        return
    kind = "deterministic step"
    extra = {}
    if brownian:
        kind = ("stochastic midpoint step, kBT=%g: 2 M^{1/2}W (%s) + M_RFD (2 apply_M) + Kinv, then" % (args.kBT, args.mhalf))
        n_prod = iters + 1 + 2
        if lanczos_its is not None:
            extra["lanczos_iterations_last_step"] = lanczos_its()
            n_prod += sum(extra["lanczos_iterations_last_step"]) * (1 if len(extra["lanczos_iterations_last_step"]) > 1 else 2)
        extra["apply_M_per_step"] = n_prod
    print(json.dumps({
        "metric": "timesteps/sec (%s: %d GMRES iterations (%s, %s PC) = %d apply_M + PC + K ops + evolve), "
                  "%d x shell_N_%d, %s, fp64" % (kind, iters, "fixed work" if rtol is None else "converged to %g" % rtol, args.pc,
                                                 iters + 1, nb, nblb, "wall-corrected" if wall else "free-space"),
        "value": 1.0 / sec, "unit": "timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "config": {"workload": args.config, "bodies": nb, "blobs_per_body": nblb, "n_blobs": N, "wall": wall},
        "mf_gflops": extra.get("apply_M_per_step", iters + 1) * 18.0 * float(N) ** 2 / sec / 1e9, "gmres_residual": res[-1], "gmres_iterations": used,
        "block_refresh": args.block_refresh, "initial_guess": (["previous solution", "2 x_n - x_{n-1}", "3 x_n - 3 x_{n-1} + x_{n-2}"][args.extrapolate]
                          if (args.warm_start or args.extrapolate) and not brownian else "zero"), **extra}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def brownian_mode(args, dev, world, rank):
    """BASELINE cfg 4: wall-corrected + Brownian on N GPUs.  1 step = one Brownian increment M^{1/2} W by
    Lanczos (tol 1e-3) on the tile-pair-sharded product (all-gather once, one all-reduce per iteration)."""
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    from rigid_body_light_amd.dist import ShardedMobility
    from rigid_body_light_amd.krylov import lanczos_mhalf, lanczos_mhalf_multi
    nb, nblb, wall = CONFIGS[args.config]
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    sm = ShardedMobility(nb, nblb, device=dev, ctx=ctx)
    r_local = torch.empty(3 * (sm.row1 - sm.row0), dtype=torch.float64, device=dev)
    ctx.blob_positions(sm.b0, sm.b1, r_local.data_ptr())
    sm.set_positions_local(r_local)
    W = torch.from_numpy(np.random.default_rng(3).standard_normal(3 * N)).to(dev)   # identical on every rank

    def A(v):    # wall=True: the kernel applies B M B; vectors are replicated, only this product communicates
        part = torch.empty(3 * N, dtype=torch.float64, device=dev)
        ctx.apply_M_sym(v.contiguous().data_ptr(), sm.r_full.data_ptr(), N, rank, world, part.data_ptr())
        return sm.all_reduce_sum(part)

    if args.nvec > 1:   # k increments at once: every Lanczos iteration is one multi-vector product (MFMA for k >= 4)
        if world != 1:
            raise SystemExit("--nvec > 1 is a single-GPU mode")
        Wk = torch.from_numpy(np.random.default_rng(3).standard_normal((args.nvec, 3 * N))).to(dev)

        def Ak(Vk):
            out = torch.empty_like(Vk)
            ctx.apply_M_multi(Vk.data_ptr(), sm.r_full.data_ptr(), N, args.nvec, out.data_ptr())
            return out

        for _ in range(args.warmup):
            lanczos_mhalf_multi(Ak, Wk, 100, 1e-3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            Y, its, ch = lanczos_mhalf_multi(Ak, Wk, 100, 1e-3)
        torch.cuda.synchronize(); sec = (time.perf_counter() - t0) / args.steps
        ctx.sync_check()
        print(json.dumps({
            "metric": "Brownian increments/sec (%d independent M^{1/2} W by lock-step Lanczos to 1e-3, %d iterations, multi-RHS "
                      "product on the fp64 matrix cores), %d x shell_N_%d, %s, fp64" % (args.nvec, its, nb, nblb, "wall-corrected, B M B" if wall else "free-space M without damping"),
            "value": args.nvec / sec, "unit": "increments/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": args.config, "n_blobs": N, "wall": wall, "nvec": args.nvec},
            "lanczos_iterations": its, "mf_gflops": args.nvec * its * 18.0 * float(N) ** 2 / sec / 1e9}), flush=True)
        return
    its = 0
    from rigid_body_light_amd.krylov import sharded_mhalf_W
    pc = args.mhalf == "lanczos_pc"       # block-Jacobi preconditioned square root (default) or plain Lanczos
    if args.mhalf == "cholesky":
        raise SystemExit("--mode brownian measures the matrix-free square roots (--mhalf lanczos_pc | lanczos)")
    one = lambda: sharded_mhalf_W(ctx, sm, sm.r_full, W[None, :], c["a"], wall, 1e-3, 100, pc)
    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y, its = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sec = torch.tensor([(time.perf_counter() - t0) / args.steps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    ctx.sync_check()
    if rank == 0:
        sec = float(sec.item())
        print(json.dumps({
            "metric": "Brownian increments/sec (M^{1/2} W by %s to 1e-3, %d iterations), %d x shell_N_%d, %s, fp64"
                      % ("block-Jacobi preconditioned Lanczos" if pc else "Lanczos", its, nb, nblb, "wall-corrected" if wall else "free-space"),
            "value": 1.0 / sec, "unit": "increments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": "BASELINE.json configs[3]" if args.config == "cfg3" else args.config,
                                            "bodies": nb, "blobs_per_body": nblb, "n_blobs": N, "wall": wall,
                                            "parallelism": "tile-pair-sharded x%d, all-reduce(U) per Lanczos iteration" % world},
            "lanczos_iterations": its, "mf_gflops": its * 18.0 * float(N) ** 2 / sec / 1e9}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work per cpu_baseline leg (0 = skip)")
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="0 heuristic, 1 ordered-rows kernel, 2 symmetric kernel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path with several ranks on ONE GPU)")
    ap.add_argument("--mode", default="apply_M", choices=["apply_M", "timestep", "brownian"],
                    help="apply_M: 1 step = one M.F pass (default).  timestep: 1 step = one deterministic time step "
                         "(SURVEY.md 8d fixed-work: 20 GMRES iterations = 21 apply_M + PC + K ops + evolve), 1 GPU")
    ap.add_argument("--timestep-steps", type=int, default=2, help="also time this many deterministic time steps (0 = skip)")
    ap.add_argument("--nvec", type=int, default=1, help="--mode brownian: independent noise vectors advanced in lockstep "
                    "(>= 4 uses the fp64-MFMA multi-RHS product; 1 GPU)")
    ap.add_argument("--kBT", type=float, default=0.0, help="--mode timestep: > 0 runs the stochastic midpoint (Brownian) step")
    ap.add_argument("--mhalf", default="lanczos_pc", choices=["lanczos_pc", "lanczos", "cholesky"],
                    help="square root used by the Brownian step (lanczos_pc: block-Jacobi preconditioned Lanczos)")
    ap.add_argument("--sharded-driver", action="store_true", help="use the multi-GPU Brownian driver also at N = 1")
    ap.add_argument("--pc", default="diag", choices=["diag", "block"], help="preconditioner of --mode timestep")
    ap.add_argument("--solver", default="native", choices=["native", "torch", "graph"],
                    help="--mode timestep, N = 1: librbl's own GMRES (rbl_gmres_saddle_dev, default), the torch Arnoldi "
                         "loop, or that loop replayed as one hipGraph")
    ap.add_argument("--native", action="store_true", help="alias of --solver native")
    ap.add_argument("--warm-start", action="store_true", help="--mode timestep --rtol ...: native GMRES starts from the previous step's solution")
    ap.add_argument("--extrapolate", type=int, default=0, choices=[0, 1, 2], help="--mode timestep --rtol ...: start from the linear (1) "
                    "or quadratic (2) extrapolation of the last solutions (implies --warm-start)")
    ap.add_argument("--block-refresh", type=int, default=1, help="--mode timestep --pc block: rebuild the per-body Cholesky factors only "
                    "every k-th configuration (rbl_set_block_refresh)")
    ap.add_argument("--graph", action="store_true", help="--mode timestep: replay the fixed-work solve as one hipGraph")
    ap.add_argument("--rtol", type=float, default=0.0, help="--mode timestep: converge GMRES to this relative residual "
                    "instead of the fixed 20 iterations")
    ap.add_argument("--dump-check", default="", help="write a row sample of the result to PATH.rank<r>.npz (tests compare it with the CPU oracle)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)     # several ranks may share a GPU only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    from rigid_body_light_amd.dist import ShardedMobility

    if args.mode == "timestep":
        return timestep_mode(args, dev, world, rank)
    if args.mode == "brownian":
        return brownian_mode(args, dev, world, rank)
    nb, nblb, wall = CONFIGS[args.config]
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    stream = torch.cuda.current_stream()
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=stream.cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    if args.jsplit or args.variant:
        ctx.set_tuning(args.jsplit, args.variant)
    sm = ShardedMobility(nb, nblb, device=dev, ctx=ctx)
    nrows = sm.row1 - sm.row0

    F_full_host = np.random.default_rng(2).standard_normal(3 * N)
    F_local = torch.from_numpy(F_full_host[3 * sm.row0:3 * sm.row1].copy()).to(dev)
    r_local = torch.empty(3 * nrows, dtype=torch.float64, device=dev)
    U_local = torch.empty(3 * nrows, dtype=torch.float64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    use_sym = args.variant != 1     # symmetric kernel (each unordered pair once) unless the ordered kernel is forced
    U_part = torch.empty(3 * N, dtype=torch.float64, device=dev) if use_sym else None

    r_all = torch.empty(3 * N, dtype=torch.float64, device=dev) if use_sym else None

    def step(k=None):
        # a8: blob positions from (X,Q); one exchange; then this rank's share of U = B M B F.
        # Symmetric sharding needs all positions on every rank: the O(N_bod) body state is replicated, so each
        # rank evaluates them itself (a ~5 us kernel) instead of gathering them; the force vector arrives
        # sharded (one all-gather) and the partial U is completed by one all-reduce.
        if use_sym:
            ctx.blob_positions(0, nb, r_all.data_ptr())
            r_full = r_all
        else:
            ctx.blob_positions(sm.b0, sm.b1, r_local.data_ptr())
            r_full = sm.set_positions_local(r_local) if world > 1 else r_local
        F_full = sm.all_gather_rows(F_local) if world > 1 else F_local
        if k is not None:
            ev[k][0].record(stream)
        if use_sym:   # this rank's share of the unordered tile pairs -> partial full-length U -> all-reduce
            ctx.apply_M_sym(F_full.data_ptr(), r_full.data_ptr(), N, rank, world, U_part.data_ptr())
        else:         # ordered pairs, this rank's rows, no reduction
            ctx.apply_M(F_full.data_ptr(), r_full.data_ptr(), N, sm.row0, sm.row1, U_local.data_ptr())
        if k is not None:
            ev[k][1].record(stream)
        if use_sym:
            sm.all_reduce_sum(U_part)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.sync_check()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    t1 = time.perf_counter()
    ctx.sync_check()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    kern_ms = torch.tensor([sum(a.elapsed_time(b) for a, b in ev) / args.steps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item()); kern_ms = float(kern_ms.item())

    if args.dump_check:   # for tests/: a row sample of what was just timed (the oracle comparison happens in the test)
        b0 = sm.row0 + (nrows // 2)
        got = (U_part[3 * b0:3 * b0 + 24] if use_sym else U_local[3 * (b0 - sm.row0):3 * (b0 - sm.row0) + 24]).cpu().numpy()
        np.savez("%s.rank%d.npz" % (args.dump_check, rank), row0=b0, values=got, world=world, rank=rank)

    tstep = None
    if args.timestep_steps > 0:
        # the reference has no time-step driver; ours (SURVEY.md 8d "fixed-work" step, krylov.py): 20 right-
        # preconditioned GMRES iterations on apply_saddle (= 21 apply_M + diagonal PC + K ops) + evolve.  With N > 1
        # the mobility product of every iteration is tile-pair sharded (one all-reduce per iteration).
        from rigid_body_light_amd.krylov import DeterministicStepper, ShardedDeterministicStepper
        stp = (ShardedDeterministicStepper(ctx, sm, nb, nblb, dev) if world > 1 else DeterministicStepper(ctx, nb, nblb, dev, native=True))
        Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
        stp.step(Fb, 20)
        barrier(); ts0 = time.perf_counter()
        for _ in range(args.timestep_steps):
            m_it, res_it = stp.step(Fb, 20)
        barrier()
        tsv = torch.tensor([(time.perf_counter() - ts0) / args.timestep_steps], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tsv, op=dist.ReduceOp.MAX)
        ts = float(tsv.item())
        tstep = {"timesteps_per_sec": 1.0 / ts, "ms_per_timestep": ts * 1e3, "apply_M_per_timestep": 21,
                 "definition": "deterministic fixed-work step: 20 GMRES iterations on the saddle operator (diagonal PC) "
                               "+ evolve, all operators on the GPU(s)",
                 "gmres_residual": res_it, "steps_timed": args.timestep_steps}
        if world == 1:   # SURVEY 8d's second variant: converged to 1e-8 (block-diagonal PC, extrapolated warm start)
            from rigid_body_light_amd._lib import lib
            lib().rbl_set_blk_pc(ctx.h, 1)
            stp.warm_start = True; stp.extrapolate = 2     # initial guess 3 x_n - 3 x_{n-1} + x_{n-2}
            ctx.set_block_refresh(4)                       # per-body factors rebuilt every 4th configuration
            for _ in range(3):                             # fill the history (18, 12, 6 iterations), then 2-3 per step
                stp.step(Fb, 200, 1e-8)
            barrier(); ts0 = time.perf_counter()
            its = []
            for _ in range(args.timestep_steps):
                m_it, res_it = stp.step(Fb, 200, 1e-8)
                its.append(m_it)
            barrier()
            tc = (time.perf_counter() - ts0) / args.timestep_steps
            lib().rbl_set_blk_pc(ctx.h, 0); ctx.set_block_refresh(1)
            tstep["converged"] = {"rtol": 1e-8, "preconditioner": "block-diagonal, per-body factors rebuilt every 4th step",
                                  "warm_start": "quadratic extrapolation of the last three solutions (12 iterations from the "
                                                "previous solution alone, 18 cold)", "gmres_iterations": its,
                                  "gmres_residual": res_it, "timesteps_per_sec": 1.0 / tc, "ms_per_timestep": tc * 1e3}
            # SURVEY 8d's Brownian step (BASELINE configs[3] on one GPU): 2 M^{1/2} W + M_RFD + Kinv before the fixed-work
            # solve at the predictor configuration.  With N > 1: `--mode timestep --kBT 1 --gpus N` (sharded driver).
            from rigid_body_light_amd.krylov import BrownianStepper
            bctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=1.0, stream_ptr=stream.cuda_stream)
            bctx.set_config(c["X"], c["Q"]); bctx.set_lanczos(100, 1e-3)
            bst = BrownianStepper(bctx, nb, nblb, dev, native=True)
            bst.step(Fb, seed=0, method=2, iters=20, rtol=None)
            barrier(); ts0 = time.perf_counter()
            for k in range(args.timestep_steps):
                bst.step(Fb, seed=1 + k, method=2, iters=20, rtol=None)
            barrier()
            tb = (time.perf_counter() - ts0) / args.timestep_steps
            tstep["brownian"] = {"kBT": 1.0, "definition": "stochastic midpoint step: 2 M^{1/2}W (block-Jacobi preconditioned Lanczos to 1e-3, "
                                 "two vectors in lock step) + M_RFD (2 apply_M) + Kinv, then 20 GMRES iterations (diag PC) at the "
                                 "predictor configuration + evolve", "lanczos_iterations": bctx.lanczos_report()[0],
                                 "timesteps_per_sec": 1.0 / tb, "ms_per_timestep": tb * 1e3}
            del bst, bctx

    if rank == 0:
        sec_per_step = elapsed / args.steps
        # ordered-pair equivalents one launch (this rank) covers: its rows x all columns, or its
        # 1/world share of the unordered tile pairs (each standing for two ordered pairs)
        pairs_per_launch = float(N) * float(N) / world if use_sym else float(nrows) * float(N)
        flops_alg = FLOPS_PER_PAIR[wall] * pairs_per_launch
        achieved = flops_alg / (kern_ms * 1e-3) / 1e12
        line = {
            "metric": "timesteps/sec + M.F GFLOP/s (BASELINE.json metric): value = M.F passes/sec, 1 step = one matrix-free "
                      "apply_M pass of the hot path (blob positions -> [all-gather] -> U = B M B F); M.F GFLOP/s in `mf_gflops`, "
                      "timesteps/sec in `timestep`; %d x shell_N_%d, %s, fp64" % (nb, nblb, "wall-corrected" if wall else "free-space"),
            "value": 1.0 / sec_per_step,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: 200 bodies x shell_N_642 blobs, wall-corrected mobility"
                                   if args.config == "cfg3" else args.config,
                       "bodies": nb, "blobs_per_body": nblb, "n_blobs": N, "wall": wall,
                       "parallelism": ("tile-pair-sharded x%d, positions replicated, all-gather(F) + all-reduce(U)" if use_sym else
                                       "body-row-sharded x%d, all-gather(pos,F)") % world},
            "mf_gflops": 18.0 * float(N) ** 2 / sec_per_step / 1e9,
            "roofline": {"bound": "fp64-valu", "kernel": "%s<%s>" % ("k_apply_M_sym" if use_sym else "k_apply_M", "true" if wall else "false"),
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS,
                         # HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 per the
                         # gfx950 correction + WRITE_SIZE, KB -> B): profiles/r01_bench_cfg3_sym_pmc_summary.txt
                         "traffic": (2 * 164947e3 + 1.70253e9) if (args.config == "cfg3" and world == 1 and use_sym) else None,
                         "kernel_ms": kern_ms,
                         "algorithmic": "%.0f flop/ordered pair (SURVEY.md 8d) x %.4g pairs/launch" % (FLOPS_PER_PAIR[wall], pairs_per_launch),
                         # executed-instruction view of the same launch (PMC, same profile file): what the VALU actually issued
                         "executed": ({"valu_insts_per_launch": 1.1108e10, "valu_insts_per_ordered_pair": 43.1,
                                       "valu_issue_cycles_frac": 0.90, "sustained_clock_ghz": 2.13,
                                       "source": "profiles/r01_bench_cfg3_sym_pmc_summary.txt (SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"}
                                      if (args.config == "cfg3" and world == 1 and use_sym) else None),
                         "note": "fp64 VALU-issue bound; on gfx950 fp64 VALU and fp64 MFMA share one pipe (SQ_VALU_MFMA_COEXEC_CYCLES = 0), "
                                 "so `peak` is the fp64 vector = fp64 MFMA peak.  frac > 1 because `achieved` prices the launch at the "
                                 "reference's 204 flop per ordered pair while the kernel issues 43 VALU instructions per ordered pair "
                                 "(each unordered pair once, division-free Horner-form algebra): see `executed` -- the VALU issues "
                                 "90 % of the kernel's cycles, the two quarter-rate v_rsq_f64 per pair account for most of the rest"},
        }
        if tstep is not None:
            line["timestep"] = tstep
        if world == 1 and args.cpu_budget > 0:
            cb = cpu_baseline(c, nb, nblb, wall, args.cpu_budget)
            line["cpu_baseline"] = cb["1core"]
            line["cpu_baseline_allcores"] = cb["allcores"]
            line["speedup_vs_cpu_1core"] = line["value"] / cb["1core"]["value"]
            line["speedup_vs_cpu_allcores"] = line["value"] / cb["allcores"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
