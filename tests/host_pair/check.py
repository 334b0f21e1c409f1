#!/usr/bin/env python3
"""CPU check of the device pair arithmetic (host build of rbl_pair.hpp, see pair_host.cpp) against the reference
fixtures in tests/golden/: prints the worst error of the fast ordered form and of the symmetric form (M_ij and its
transpose M_ji) relative to max(|block|, free-space scale).  Development aid for algebra changes in rbl_pair.hpp (lives under
tests/ because it uses the oracle as its checker).   python tests/host_pair/check.py"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = "/tmp/libpair_host.so"
subprocess.check_call(["g++", "-O2", "-ffp-contract=fast", "-mfma", "-std=c++17", "-shared", "-fPIC",
                       "-I" + os.path.join(ROOT, "tests", "host_pair"), "-o", SO,
                       os.path.join(ROOT, "tests", "host_pair", "pair_host.cpp")])
L = C.CDLL(SO)
dp = C.POINTER(C.c_double)
L.fast_block.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_int, dp]
L.sym_blocks.argtypes = [dp, dp, C.c_double, C.c_int, C.c_int, dp, dp]
p = lambda a: a.ctypes.data_as(dp)
unhex = lambda v: np.array([float.fromhex(x) for x in v])

g = json.load(open(os.path.join(ROOT, "tests", "golden", "pair_blocks_assembly.json")))
sys.path.insert(0, ROOT)
from oracle import Oracle
orc = Oracle()
worst = {"fast": 0.0, "sym_ij": 0.0, "sym_ji": 0.0}
rng = np.random.default_rng(5)
cases = [(unhex(c["ri"]), unhex(c["rj"]), c["i"], c["j"], float.fromhex(c["a"]), c["wall"], unhex(c["out9"]).reshape(3, 3)) for c in g["blocks"]]
# plus random wall pairs at the BASELINE radii, both orders, checked against the C oracle (itself bit-exact vs the reference)
for a in (0.06752768, 0.13100878, 1.0):
    for _ in range(4000):
        ri = np.append(rng.uniform(-20, 20, 2), 10 ** rng.uniform(-3, 2)) * a
        rj = np.append(rng.uniform(-20, 20, 2), 10 ** rng.uniform(-3, 2)) * a
        if np.linalg.norm(ri - rj) < 2.0 * a and rng.uniform() < 0.7:
            continue
        nf = 1.0 / (8 * np.pi * a)
        cases.append((ri, rj, 0, 1, a, True, orc.pair_block(ri, rj, 0, 1, a, 1.0, True) / nf))
for ri, rj, i, j, a, wall, ref in cases:
    rhat = np.linalg.norm(ri - rj) / a
    scale = max(np.linalg.norm(ref), min(4.0 / 3.0, 1.0 / max(rhat, 1e-30)))
    out = np.zeros(9)
    L.fast_block(p(ri), p(rj), i, j, a, int(wall), p(out))
    worst["fast"] = max(worst["fast"], np.abs(out.reshape(3, 3) - ref).max() / scale)
    if i != j:
        mij, mji = np.zeros(9), np.zeros(9)
        L.sym_blocks(p(ri), p(rj), a, int(wall), 1, p(mij), p(mji))
        worst["sym_ij"] = max(worst["sym_ij"], np.abs(mij.reshape(3, 3) - ref).max() / scale)
        worst["sym_ji"] = max(worst["sym_ji"], np.abs(mji.reshape(3, 3) - ref.T).max() / scale)
print(len(cases), "cases; worst relative errors:", {k: "%.2e" % v for k, v in worst.items()})
