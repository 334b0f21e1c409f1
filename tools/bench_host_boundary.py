#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in host-pointer boundary: RigidBody.apply_M with numpy arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rigid_body_light_amd import RigidBody, make_config
for name, nb, nblb, wall in (("cfg1", 10, 12, False), ("cfg2", 50, 162, False), ("cfg3", 200, 642, True)):
    c = make_config(nb, nblb, wall)
    rb = RigidBody(c["cfg"], c["X"], c["Q"], c["a"], c["eta"], c["dt"], wall_PC=wall)
    r = rb.get_blob_positions(); F = np.random.default_rng(2).standard_normal(r.size)
    rb.apply_M(F, r)
    reps = 50 if nb < 100 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rb.apply_M(F, r)
    t = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        rb.get_blob_positions()
    tp = (time.perf_counter() - t0) / reps
    x = np.random.default_rng(4).standard_normal(r.size + 6 * nb)
    rb.apply_saddle(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        rb.apply_saddle(x)
    ts = (time.perf_counter() - t0) / reps
    print("| %s | RigidBody.apply_M (numpy in/out, H2D+kernel+D2H) %.3f ms | get_blob_positions %.3f ms | apply_saddle %.3f ms |" % (name, t * 1e3, tp * 1e3, ts * 1e3))
