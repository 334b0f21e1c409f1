#!/usr/bin/env python3
"""CPU baselines of BASELINE.md section 3 (C1-C4), measured with the CPU oracle (a port of
the reference algorithm; the reference itself needs Eigen/nanobind and cannot be built here).
Runs on the host cores of whatever box executes it; writes a markdown table.
usage: cpu_baselines.py out.md [--quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import Oracle
from rigid_body_light_amd.synth import make_config

quick = "--quick" in sys.argv
orc = Oracle()
try:
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("RBL_CPU_THREADS", "16")))
except AttributeError:
    ncores = os.cpu_count()
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
lines = ["# CPU baselines (oracle = port of the reference algorithm, gcc -O3 -march=x86-64-v3 -ffp-contract=off)", "",
         "host: %s, threads used for the multi-core legs: %d" % (cpu[0] if cpu else "?", ncores), ""]


def positions(nb, nblb, wall):
    c = make_config(nb, nblb, wall)
    cfg = c["cfg"] - c["cfg"].mean(axis=0)
    return c, orc.multi_body_pos(c["X"], c["Q"], cfg)


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


# C1: reference-faithful apply_M = dense build + GEMV, 1 thread
lines += ["## C1  reference-faithful `apply_M` (dense build + GEMV, reference c_rigid_obj.cpp:413-459,641-659), 1 thread", "",
          "| config | N | n=3N | seconds / apply_M | ns / unordered pair |", "|---|---|---|---|---|"]
c1 = []
for name, nb, nblb, wall, reps in (("cfg1 10x12 free", 10, 12, False, 5), ("30x42 free", 30, 42, False, 3),
                                   ("20x162 free", 20, 162, False, 3), ("cfg2 50x162 free", 50, 162, False, 1 if quick else 3),
                                   ("20x162 wall", 20, 162, True, 3)):
    c, r = positions(nb, nblb, wall)
    F = np.random.default_rng(2).standard_normal(r.size)
    t = med(lambda: orc.apply_M(F, r, c["a"], c["eta"], wall, mode="dense"), reps)
    N = nb * nblb
    c1.append((N, t, wall))
    lines.append("| %s | %d | %d | %.4g | %.1f |" % (name, N, 3 * N, t, t / (N * (N + 1) / 2) * 1e9))
fit = [(N, t) for N, t, w in c1 if not w and N >= 1000]
coef = np.mean([t / N ** 2 for N, t in fit])
lines += ["", "C2 (extrapolation t = c N^2, c = %.3e s from the free-space rows with N >= 1000): cfg5 20x2562 -> %.0f s; "
          "cfg3 200x642 would need a 1.19 TB matrix (cannot exist); arithmetic alone (wall) -> see C4." % (coef, coef * 51240 ** 2), ""]

# C3: reference-faithful M_half_W (dense B M B + Cholesky + L W), 1 thread
lines += ["## C3  reference-faithful `M_half_W` (reference :661-675), 1 thread", "",
          "| config | n | seconds | GFLOP/s (n^3/3) |", "|---|---|---|---|"]
c3 = []
for name, nb, nblb in (("cfg1 10x12", 10, 12), ("12x42", 12, 42), ("8x162", 8, 162)) + (() if quick else (("16x162", 16, 162),)):
    c, r = positions(nb, nblb, False)
    W = np.random.default_rng(3).standard_normal(r.size)
    t = med(lambda: orc.M_half_W(r, c["a"], c["eta"], False, W), 1)
    n = r.size
    c3.append((n, t))
    lines.append("| %s | %d | %.4g | %.2f |" % (name, n, t, n ** 3 / 3 / t / 1e9))
k3 = c3[-1][1] / c3[-1][0] ** 3
lines += ["", "cubic extrapolation from the largest row: cfg2 (n=24300) -> %.0f s, cfg5 (n=153720) -> %.2e s" % (k3 * 24300 ** 3, k3 * 153720 ** 3), ""]

# C4: matrix-free apply_M, 1 thread and all cores, every config (row sample scaled by N/rows)
lines += ["## C4  matrix-free `apply_M` (same pair arithmetic, no matrix), row sample scaled to the full product", "",
          "| config | N | 1 thread s/apply_M | %d threads s/apply_M |" % ncores, "|---|---|---|---|"]
budget = 3.0 if quick else 8.0
for name, nb, nblb, wall in (("cfg1 10x12 free", 10, 12, False), ("cfg2 50x162 free", 50, 162, False),
                             ("cfg3/4 200x642 wall", 200, 642, True), ("cfg5 20x2562 free", 20, 2562, False)):
    c, r = positions(nb, nblb, wall)
    N = nb * nblb
    F = np.random.default_rng(2).standard_normal(r.size)
    res = []
    for nt in (1, ncores):
        rows = min(N, 4 * nt)
        t0 = time.perf_counter(); orc.apply_M_rows(F, r, 0, rows, c["a"], c["eta"], wall, nt); per = (time.perf_counter() - t0) / rows
        rows = int(max(rows, min(N, budget / max(per, 1e-9))))
        b = max(0, N // 2 - rows // 2)
        t0 = time.perf_counter(); orc.apply_M_rows(F, r, b, b + rows, c["a"], c["eta"], wall, nt); t = time.perf_counter() - t0
        res.append(t * N / rows)
    lines.append("| %s | %d | %.4g | %.4g |" % (name, N, res[0], res[1]))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
