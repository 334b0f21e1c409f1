#!/usr/bin/env python3
"""Generate tests/golden/pair_kernels.json from the REFERENCE's own compiled
pair kernels (oracle/_ref/libref_pair.so, built by oracle/build_ref.sh from
/root/reference/src/c_rigid_obj.cpp:31-142).

Run in the build container (needs /root/reference):
    python tests/golden/gen_pair_golden.py
The fixture holds inputs and the reference's outputs only (bit-exact, as C99
hex floats); it contains no reference source.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import RefPair  # noqa: E402


def hx(v):
    return [float(x).hex() for x in np.asarray(v, dtype=np.float64).reshape(-1)]


def main():
    ref = RefPair()
    rng = np.random.default_rng(20251212)
    cases_rpy, cases_wall = [], []
    # --- free-space RPY: far branch, overlap branch, r ~ 2a boundary, self ---
    vecs = []
    vecs += [rng.uniform(-6, 6, 3) for _ in range(96)]            # mostly far
    vecs += [rng.uniform(-1.1, 1.1, 3) for _ in range(96)]        # mostly overlap
    for _ in range(32):                                           # |r| ~ 2a
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        vecs.append(d * (2.0 + rng.uniform(-1e-9, 1e-9)))
    vecs += [np.array([2.0, 0.0, 0.0]), np.array([0.0, 0.0, 2.0]),
             np.array([3.0, 0.5, -1.0]), np.array([0.7, 0.2, 0.4])]
    for v in vecs:
        for a in (1.0, 0.41642068, 2.5):
            inv_a = 1.0 / a
            out = ref.rpy(v[0] * a, v[1] * a, v[2] * a, 0, 1, inv_a)
            cases_rpy.append({"r": hx(v * a), "i": 0, "j": 1, "inv_a": float(inv_a).hex(), "out6": hx(out)})
    cases_rpy.append({"r": hx([0, 0, 0]), "i": 5, "j": 5, "inv_a": (1.0).hex(),
                      "out6": hx(ref.rpy(0.0, 0.0, 0.0, 5, 5, 1.0))})
    # --- wall correction on top of an RPY block, args as the assembly passes them
    for _ in range(192):
        zi, zj = rng.uniform(0.05, 6.0, 2)
        dx, dy = rng.uniform(-4, 4, 2)
        a = float(rng.choice([1.0, 0.41642068, 0.06752768]))
        rx, ry, rz = dx * a, dy * a, (zi - zj) * a
        s = ref.rpy(rx, ry, rz, 0, 1, 1.0 / a)
        M = np.array([s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]])
        args = (rx / a, ry / a, (rz + 2 * zj * a) / a, zj * a / a)
        out = ref.wall(args[0], args[1], args[2], M, 0, 1, args[3])
        cases_wall.append({"args": hx(args), "i": 0, "j": 1, "M_in": hx(M), "M_out": hx(out)})
    for h in (0.2, 0.5, 1.0, 2.5, 17.0):                          # self term
        M = np.array([4 / 3, 0, 0, 0, 4 / 3, 0, 0, 0, 4 / 3])
        out = ref.wall(0.0, 0.0, 2 * h, M, 3, 3, h)
        cases_wall.append({"args": hx((0.0, 0.0, 2 * h, h)), "i": 3, "j": 3, "M_in": hx(M), "M_out": hx(out)})
    # --- below-wall behaviour
    try:
        ref.wall(0.1, 0.2, 0.3, np.zeros(9), 0, 1, -0.1)
        throws = False
    except RuntimeError:
        throws = True
    doc = {"source": "reference src/c_rigid_obj.cpp:31-142 compiled by oracle/build_ref.sh (g++ -O2 -ffp-contract=off, double)",
           "format": "C99 hex floats; out6 = xx,xy,xz,yy,yz,zz; M row-major 3x3",
           "rpy": cases_rpy, "wall": cases_wall, "below_wall_throws": throws}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_kernels.json"), "w") as f:
        json.dump(doc, f, indent=0)
    print("wrote", len(cases_rpy), "rpy cases,", len(cases_wall), "wall cases; below-wall throws:", throws)


if __name__ == "__main__":
    main()
