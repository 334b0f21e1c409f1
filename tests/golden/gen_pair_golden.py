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


def assembly_cases():
    """Second fixture, pair_blocks_assembly.json: whole 3x3 blocks as the reference's ASSEMBLY loop forms them
    (c_rigid_obj.cpp:432-447: r = r_i - r_j; mobilityUFRPY(r, i, j, 1/a); wall correction with the a-normalised
    image vector (rx/a, ry/a, (rz + 2 z_j)/a) and h = z_j/a), unscaled (before `Mob *= 1/(8 pi eta a)`, :456),
    in the regimes the BASELINE geometries feed the kernels: blob radii of the shell files, heights from 1e-6 a to
    1e3 a, |r| within 1e-9 of the 2a branch switch WITH the wall term, blob touching its neighbour's image, equal
    heights (h_hat = 1/2), self blocks.  The argument preparation below is the same sequence of IEEE double
    operations the reference performs; the two kernels are the reference's own compiled functions."""
    ref = RefPair()
    rng = np.random.default_rng(20261004)
    radii = [0.13100878, 0.03420498, 0.06752768, 0.41642068, 1.0]
    cases = []

    def add(ri, rj, i, j, a, wall):
        ri = np.asarray(ri, dtype=np.float64); rj = np.asarray(rj, dtype=np.float64)
        a = float(a)
        rx, ry, rz = ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]                  # :432-434
        s = ref.rpy(float(rx), float(ry), float(rz), i, j, 1.0 / a)               # :435-436
        M = np.array([s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]])      # :437-439
        if wall:                                                                  # :440-445
            M = ref.wall(float(rx / a), float(ry / a), float((rz + 2 * rj[2]) / a), M, i, j, float(rj[2] / a))
        cases.append({"ri": hx(ri), "rj": hx(rj), "i": i, "j": j, "a": a.hex(), "wall": bool(wall), "out9": hx(M)})

    for a in radii:
        for wall in (True, False):
            for _ in range(24):                                   # generic pairs, heights 0.05 a .. 8 a
                ri = np.append(rng.uniform(-5, 5, 2), rng.uniform(0.05, 8)) * a
                rj = np.append(rng.uniform(-5, 5, 2), rng.uniform(0.05, 8)) * a
                add(ri, rj, 0, 1, a, wall)
        for _ in range(16):                                       # far above the wall: h/a up to 1e3
            h = 10.0 ** rng.uniform(1, 3)
            ri = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), h + rng.uniform(-2, 2)]) * a
            rj = np.array([0.0, 0.0, h]) * a
            add(ri, rj, 2, 7, a, True)
        for _ in range(16):                                       # h -> 0+: both blobs within 1e-6 a .. 1e-2 a of the wall
            zi, zj = 10.0 ** rng.uniform(-6, -2, 2)
            d = rng.uniform(2.0, 6.0); th = rng.uniform(0, 2 * np.pi)
            add(np.array([d * np.cos(th), d * np.sin(th), zi]) * a, np.array([0.0, 0.0, zj]) * a, 0, 1, a, True)
        for _ in range(24):                                       # |r|/a within 1e-9 of 2 (branch switch :62), with the wall term
            d = rng.standard_normal(3); d /= np.linalg.norm(d)
            d *= 2.0 + rng.uniform(-1e-9, 1e-9)
            zj = rng.uniform(1.5, 4.0)
            add((np.array([0.3, -0.2, zj]) + d) * a, np.array([0.3, -0.2, zj]) * a, 0, 1, a, True)
        for _ in range(12):                                       # equal heights: h_hat = h_j / R_z = 1/2
            z = 10.0 ** rng.uniform(-1, 1.5)
            d = rng.uniform(0.5, 5.0); th = rng.uniform(0, 2 * np.pi)
            add(np.array([d * np.cos(th), d * np.sin(th), z]) * a, np.array([0.0, 0.0, z]) * a, 3, 4, a, True)
        for _ in range(12):                                       # blob i touching the IMAGE of j: |r_i - image(r_j)| ~ 2a
            zj = rng.uniform(0.2, 1.8); zi = 2.0 - zj + rng.uniform(-1e-9, 1e-9)
            lat = rng.uniform(0.0, 1e-3, 2)
            add(np.array([lat[0], lat[1], zi]) * a, np.array([0.0, 0.0, zj]) * a, 0, 1, a, True)
        for h in (1e-6, 1e-3, 0.3, 1.0, 2.0, 31.0, 1e3):          # self blocks (:40-46, :98-104)
            add(np.array([0.1, 0.2, h]) * a, np.array([0.1, 0.2, h]) * a, 5, 5, a, True)
        add(np.array([0.1, 0.2, 0.7]) * a, np.array([0.1, 0.2, 0.7]) * a, 5, 5, a, False)
    doc = {"source": "reference src/c_rigid_obj.cpp:31-142 compiled by oracle/build_ref.sh (g++ -O2 -ffp-contract=off, double), "
                     "called with the arguments the assembly loop :432-447 forms",
           "format": "C99 hex floats; out9 = row-major 3x3 block (i <= j roles), NOT scaled by 1/(8 pi eta a)",
           "blocks": cases}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_blocks_assembly.json"), "w") as f:
        json.dump(doc, f, indent=0)
    print("wrote", len(cases), "assembly-level blocks")


if __name__ == "__main__":
    main()
    assembly_cases()
