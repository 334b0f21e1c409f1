#!/usr/bin/env python3
"""Per-rank kernel time of the tile-pair-sharded symmetric product, rehearsed on ONE GPU: for world = 1, 2, 4, 8
every rank's launch (i_first = rank, i_step = world) is timed in turn.  max over ranks x world / t(world = 1)
is the load-balance bound on strong scaling (communication not included).
usage: bench_shard_kernel.py [n_bodies blobs_per_body wall|free]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext

nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
wall = (sys.argv[3] == "wall") if len(sys.argv) > 3 else True
c = make_config(nb, nblb, wall); N = nb * nblb
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
F = torch.from_numpy(np.random.default_rng(2).standard_normal(3 * N)).to(dev)
U = torch.empty_like(F)
full = None
REPS = 10
WORLDS = [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]
if os.environ.get("CHUNK"):                      # force the chunk length C of the symmetric kernel (tuning experiments)
    ctx.set_option("matvec_kernel", 2); ctx.set_option("sym_chunk", int(os.environ["CHUNK"]))
for kv in os.environ.get("TUNE", "").split(","):   # named options (e.g. TUNE=sym_work_queue=0: pair kernels without the work queue)
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for world in WORLDS:
    ts = []; acc = torch.zeros_like(F)
    order = list(range(world))
    if os.environ.get("REVERSE"):                # rank order of the measurement (is the first one slow because it is first?)
        order.reverse()
    for rank in order:
        ctx.apply_M_sym(F.data_ptr(), r.data_ptr(), N, rank, world, U.data_ptr()); ctx.sync_check()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(REPS):
            ctx.apply_M_sym(F.data_ptr(), r.data_ptr(), N, rank, world, U.data_ptr())
        b.record(st); ctx.sync_check()
        ts.append(a.elapsed_time(b) / REPS); acc += U
    if world == 1:
        full = acc.clone(); t1 = ts[0]
    err = float((acc - full).norm() / full.norm())
    print("world %d: per-rank ms min %.3f max %.3f ; t1/(world*max) = %.3f ; sum of shards vs world 1: %.1e ; in rank order %s: %s"
          % (world, min(ts), max(ts), t1 / (world * max(ts)), err, order, " ".join("%.3f" % t for t in ts)))
